"""-m gpu: the data-parallel Trainer_prototype_full step on REAL devices and the HIP kernels (BASELINE.json configs[3] in
miniature: two ranks x (B source + B target), tests/dp_statement.py), against the single-process statement of the same update
on the concatenated batch (oracle on the host):

  test_..._rccl_world2           two ranks, one GPU each, backend nccl (= RCCL over xGMI).  Skipped on a box with one GPU
                                 (the build pool's boxes have one; the 8-GPU node of the scaling runs has eight).
  test_..._two_ranks_one_gpu     the SAME worker with both ranks on cuda:0 and the gloo backend carrying the device tensors:
                                 everything of the N > 1 path except RCCL itself (replica broadcast, flat gradient all-reduce,
                                 prototype sums all-reduced before the division, x world weighting) on the HIP kernels.

Dropout: each rank replays the torch CPU stream the hand-made statement draws from (MaskFeeder, as tests/test_trainers_gpu.py)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out, backend, share):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0 if share else rank)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    import dp_statement
    import model_cases
    from make_golden_inputs import synth_loader
    from test_trainers_gpu import MaskFeeder
    from uda_clr_amd.networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator
    from uda_clr_amd.train_process import Trainer_prototype_full
    c = dp_statement.PF
    m = MaskFeeder(model_cases.seeded_model().to(dev))              # seed 1337 parameters, as dp_statement.proto_setup()
    torch.manual_seed(1338)
    d1, d2 = BoundaryDiscriminator().to(dev), UncertaintyDiscriminator().to(dev)
    if rank == 1:                   # a replica that starts apart: the Trainer must bring it back to rank 0's state
        with torch.no_grad():
            for p in list(m.parameters())[:3] + list(d1.parameters())[:1]:
                p.add_(0.5)
            next(iter(m.model.buffers())).add_(1.0)
    og = torch.optim.SGD(m.parameters(), lr=c["lr"])
    od, od2 = torch.optim.SGD(d1.parameters(), lr=c["lr_d"]), torch.optim.SGD(d2.parameters(), lr=c["lr_d"])
    loaderS = synth_loader(2, c["B"], c["S"], c["loaderS_seed"])     # rank r trains on batch r of each domain
    loaderT = synth_loader(2, c["B"], c["S"], c["loaderT_seed"])
    tr = Trainer_prototype_full.Trainer(cuda=True, model_gen=m, model_dis=d1, model_uncertainty_dis=d2, optimizer_gen=og,
                                        optimizer_dis=od, optimizer_uncertainty_dis=od2, val_loader=loaderT, domain_loaderS=loaderS,
                                        domain_loaderT=loaderT, out=os.path.join(out, "run"), max_epoch=1, use_global=True, use_pid=True,
                                        retrify_pesudo=True, global_pro_weight=0.9, pro_weight=c["pro_weight"], stop_epoch=1,
                                        interval_validate=100, batch_size=c["B"], warmup_epoch=-1)
    assert tr.world == 2 and len(tr.domain_loaderS) == 1 and len(tr.domain_loaderT) == 1
    tr.epoch = 0
    tr.iteration = 0
    m.train(); d1.train(); d2.train()
    torch.manual_seed(c["drop_seed"] + rank)         # each rank its own dropout stream (reproduced by the hand-made statement)
    vals = tr.train_step(next(iter(tr.domain_loaderS)), next(iter(tr.domain_loaderT)))
    torch.cuda.synchronize()
    torch.save(dp_statement.rank_record(tr, m.model, d1, d2, vals), os.path.join(out, "pf%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run(backend, share):
    import socket
    sys.path.insert(0, HERE)
    import dp_statement
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(2, port, out, backend, share), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(out, "pf0.pt")), torch.load(os.path.join(out, "pf1.pt"))
    dp_statement.check_ranks_agree(r0, r1)
    gmean, top = dp_statement.check_against_fp64_statement(r0)
    print("generator update vs the fp64 statement: noise ratio to the fp32 statement (geometric mean) %.2f; largest distances" % gmean, top)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); the one-GPU variant below covers the rest")
def test_trainer_prototype_full_data_parallel_rccl_world2():
    _run("nccl", share=False)


def test_trainer_prototype_full_data_parallel_two_ranks_one_gpu():
    _run("gloo", share=True)
