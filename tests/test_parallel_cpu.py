"""not-gpu: the N > 1 data-parallel path on gloo, world_size 2 (two CPU processes):
  - FlatGradAllReduce averages gradients (missing grads count as zero) and leaves all ranks equal
  - all_reduce_sum_ of per-rank prototype sums == sums of the concatenated batch (so centroids are
    the global-batch centroids), AllReduceSum back-propagates the identity
  - loader sharding gives complete coverage with the SAME number of iterations on every rank (wrap-around padding)
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
    seen = [i for i in shard_loader(list(range(7)), rank, world)]      # 7 batches on 2 ranks: 4 each, the last one wraps around
    assert seen == [(i * world + rank) % 7 for i in range(4)] and len(shard_loader(list(range(7)), rank, world)) == 4
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


# ---------------------------------------------------------------- equal iteration counts on every rank
def test_shard_loader_gives_every_rank_the_same_number_of_batches_and_no_single_image_batch():
    """n = 385, world 8, batch 16 (the example of the round-1 review: rank 0 got 4 batches incl. a 1-image one, the others 3):
    every rank now iterates the same number of batches, none of size 1, and the ranks together cover the dataset."""
    from torch.utils.data import DataLoader, TensorDataset
    from uda_clr_amd.train_process._common import shard_count, shard_loader
    for n, world, bs, shuffle in ((385, 8, 16, True), (385, 8, 16, False), (33, 2, 16, True), (7, 2, 4, False), (5, 4, 2, True),
                                  (64, 8, 8, True)):
        ds = TensorDataset(torch.arange(n))
        base = DataLoader(ds, batch_size=bs, shuffle=shuffle, drop_last=False)
        per_rank = []
        for epoch in range(2):
            seen, counts = [], []
            loaders = per_rank or [shard_loader(base, r, world, seed=123) for r in range(world)]
            per_rank = loaders
            for ld in loaders:
                batches = [b[0].tolist() for b in ld]
                assert len(batches) == len(ld)
                counts.append(len(batches))
                assert all(len(b) >= 2 for b in batches) or shard_count(n, world, bs) == 1, (n, world, bs, [len(b) for b in batches])
                seen.append([i for b in batches for i in b])
            assert len(set(counts)) == 1, (n, world, bs, counts)
            assert len({len(s) for s in seen}) == 1
            union = set(i for s in seen for i in s)
            dropped = n - len(union)
            assert dropped <= world, (n, world, bs, dropped)            # only the trailing single-image batches are given up
            if shuffle and n > 2 * world:
                flat0 = seen[0]
                if epoch == 0:
                    first_epoch = flat0
                else:
                    assert flat0 != first_epoch, "the global order must change from epoch to epoch"
    # plain sequences of batches: wrapped around to a common count
    for n, world in ((7, 2), (3, 4), (8, 4)):
        lens = [len(list(shard_loader(list(range(n)), r, world))) for r in range(world)]
        assert len(set(lens)) == 1 and lens[0] == (n + world - 1) // world
        assert set(i for r in range(world) for i in shard_loader(list(range(n)), r, world)) == set(range(n))


# ---------------------------------------------------------------- Trainer_prototype_full under data parallelism
def _worker_proto(rank, world, port, out):
    _init(rank, world, port)
    import dp_statement
    from kernel_spec import SpecKernels
    from make_golden_inputs import synth_loader
    from uda_clr_amd import ops
    from uda_clr_amd.train_process import Trainer_prototype_full
    ops._K = SpecKernels()          # the product's front-ends (incl. the prototype all-reduce) on the torch statement of the kernels
    c = dp_statement.PF
    m, d1, d2 = dp_statement.proto_setup()
    if rank == 1:                   # a replica that starts apart: the Trainer must bring it back to rank 0's state
        with torch.no_grad():
            for p in list(m.parameters())[:3] + list(d1.parameters())[:1]:
                p.add_(0.5)
            next(iter(m.buffers())).add_(1.0)
    og = torch.optim.SGD(m.parameters(), lr=c["lr"])
    od, od2 = torch.optim.SGD(d1.parameters(), lr=c["lr_d"]), torch.optim.SGD(d2.parameters(), lr=c["lr_d"])
    loaderS = synth_loader(2, c["B"], c["S"], c["loaderS_seed"])     # rank r trains on batch r of each domain
    loaderT = synth_loader(2, c["B"], c["S"], c["loaderT_seed"])
    tr = Trainer_prototype_full.Trainer(cuda=False, model_gen=m, model_dis=d1, model_uncertainty_dis=d2, optimizer_gen=og,
                                        optimizer_dis=od, optimizer_uncertainty_dis=od2, val_loader=loaderT, domain_loaderS=loaderS,
                                        domain_loaderT=loaderT, out=os.path.join(out, "run"), max_epoch=1, use_global=True, use_pid=True,
                                        retrify_pesudo=True, global_pro_weight=0.9, pro_weight=c["pro_weight"], stop_epoch=1,
                                        interval_validate=100, batch_size=c["B"], warmup_epoch=-1)
    assert tr.world == 2 and len(tr.domain_loaderS) == 1 and len(tr.domain_loaderT) == 1
    tr.epoch = 0
    tr.iteration = 0
    m.train(); d1.train(); d2.train()
    torch.manual_seed(c["drop_seed"] + rank)         # each rank its own dropout stream (reproduced by the hand-made reference)
    vals = tr.train_step(next(iter(tr.domain_loaderS)), next(iter(tr.domain_loaderT)))
    torch.save(dp_statement.rank_record(tr, m, d1, d2, vals), os.path.join(out, "pf%d.pt" % rank))
    dist.destroy_process_group()


def test_trainer_prototype_full_data_parallel_gloo_world2():
    """BASELINE.json configs[3] in miniature: two ranks x (B source + B target) through the product Trainer_prototype_full on
    the product's prototype front-ends (sum all-reduce of the [4][C+1] sums before the division, identity backward, alignment
    loss weighted by world before the averaged backward), checked against the single-process hand-made statement of
    tests/dp_statement.py (the same worker runs on real devices in tests/test_parallel_gpu.py)."""
    sys.path.insert(0, HERE)
    import dp_statement
    with tempfile.TemporaryDirectory() as out:
        _spawn(_worker_proto, out)
        r0, r1 = torch.load(os.path.join(out, "pf0.pt")), torch.load(os.path.join(out, "pf1.pt"))
    dp_statement.check_ranks_agree(r0, r1)
    dp_statement.check_global_statement(r0)
