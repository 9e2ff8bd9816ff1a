"""not-gpu: the N > 1 data-parallel path on gloo, world_size 2 (two CPU processes):
  - FlatGradAllReduce averages gradients (missing grads count as zero) and leaves all ranks equal
  - all_reduce_sum_ of per-rank prototype sums == sums of the concatenated batch (so centroids are
    the global-batch centroids), AllReduceSum back-propagates the identity
  - loader sharding gives disjoint, complete coverage
  - the product Trainer_baseline steps two ranks in lock-step: identical parameters afterwards and
    equal to averaging the two ranks' single-process gradients by hand."""
import os
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _init(rank, world, port):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_utils(rank, world, port, out):
    _init(rank, world, port)
    from uda_clr_amd import parallel
    from uda_clr_amd.train_process._common import shard_loader
    torch.manual_seed(10 + rank)
    a, b = torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5))
    a.grad = torch.full_like(a, float(rank + 1))
    if rank == 0:
        b.grad = torch.ones_like(b) * 4.0           # rank 1 has no grad for b
    red = parallel.FlatGradAllReduce([a, b])
    red.all_reduce_mean()
    assert torch.allclose(a.grad, torch.full_like(a, 1.5)) and torch.allclose(b.grad, torch.full_like(b, 2.0))
    # prototype sums: per-rank partial sums -> global sums
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(2, 40, 6, generator=g).double()       # [rank, pixels, C]
    w = torch.rand(2, 40, 4, generator=g).double()
    part = torch.cat([w[rank].t() @ feat[rank], w[rank].sum(0)[:, None]], 1)
    tot = parallel.all_reduce_sum_(part.clone())
    ref = torch.cat([torch.cat([w[0], w[1]]).t() @ torch.cat([feat[0], feat[1]]), torch.cat([w[0], w[1]]).sum(0)[:, None]], 1)
    assert torch.allclose(tot, ref)
    x = torch.randn(3, requires_grad=True)
    y = parallel.AllReduceSum.apply(x * 2.0)
    y.sum().backward()
    assert torch.allclose(x.grad, torch.full_like(x, 2.0))
    seen = [i for i in shard_loader(list(range(7)), rank, world)]
    assert seen == list(range(rank, 7, world)) and len(shard_loader(list(range(7)), rank, world)) == len(seen)
    open(os.path.join(out, "ok%d" % rank), "w").write("1")
    dist.destroy_process_group()


def _worker_trainer(rank, world, port, out):
    _init(rank, world, port)
    from make_golden_inputs import synth_loader
    from oracle import deeplab_ref
    from oracle_ops import OracleOps
    from uda_clr_amd.networks.deeplabv3 import DeepLab
    from uda_clr_amd.train_process import Trainer_baseline
    torch.manual_seed(1337)
    m = deeplab_ref.OracleDeepLab(DeepLab(num_classes=2, backbone="mobilenet", output_stride=16).state_dict())
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    loader = synth_loader(2, 2, 64, 60)            # 2 batches: rank r trains on batch r
    tr = Trainer_baseline.Trainer(cuda=False, model_gen=m, optimizer_gen=opt, val_loader=loader, domain_loaderS=loader,
                                  domain_loaderT=loader, out=os.path.join(out, "run"), max_epoch=1, stop_epoch=1,
                                  interval_validate=100, batch_size=2, warmup_epoch=-1)
    tr.ops = OracleOps()
    assert tr.world == 2 and len(tr.domain_loaderS) == 1
    tr.epoch = 0
    tr.iteration = 0
    torch.manual_seed(77)                          # same dropout stream on both ranks
    tr.train()
    torch.save({k: v.detach().clone() for k, v in m.named_parameters()}, os.path.join(out, "p%d.pt" % rank))
    dist.destroy_process_group()


def _spawn(fn, out):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(fn, args=(2, port, out), nprocs=2, join=True)


def test_parallel_utilities_gloo_world2():
    with tempfile.TemporaryDirectory() as out:
        _spawn(_worker_utils, out)
        assert os.path.exists(os.path.join(out, "ok0")) and os.path.exists(os.path.join(out, "ok1"))


def test_trainer_baseline_data_parallel_gloo_world2():
    sys.path.insert(0, HERE)
    from make_golden_inputs import synth_loader
    from oracle import deeplab_ref, step_ref
    from uda_clr_amd.networks.deeplabv3 import DeepLab
    with tempfile.TemporaryDirectory() as out:
        _spawn(_worker_trainer, out)
        p0, p1 = torch.load(os.path.join(out, "p0.pt")), torch.load(os.path.join(out, "p1.pt"))
        for k in p0:
            assert torch.equal(p0[k], p1[k]), "ranks diverged on " + k
        # hand-made reference: per-rank gradients on its own batch, averaged, one SGD step
        loader = synth_loader(2, 2, 64, 60)
        grads = []
        for r in range(2):
            torch.manual_seed(1337)
            m = deeplab_ref.OracleDeepLab(DeepLab(num_classes=2, backbone="mobilenet", output_stride=16).state_dict()).train()
            torch.manual_seed(77)
            s = loader[r]
            o = m(s["image"])
            step_ref.seg_loss(o[0], o[1], s["map"], s["boundary"]).backward()
            grads.append({k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in m.named_parameters()})
            init = {k: v.detach().clone() for k, v in m.named_parameters()}
        for k in p0:
            want = init[k] - 0.1 * 0.5 * (grads[0][k] + grads[1][k])
            assert torch.allclose(p0[k], want, rtol=1e-5, atol=1e-6), k
