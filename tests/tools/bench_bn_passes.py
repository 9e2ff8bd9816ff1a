"""Streaming rates of the BatchNorm passes (uda_bnbwd_reduce / uda_bnbwd_apply / uda_colstats / uda_bn_apply) at a few layer shapes of the
step, beside torch's fill / add on the same bytes (what the chip streams: ~6.8 TB/s written, ~5.8 TB/s at 2 reads + 1 write).
(TEST TOOL, GPU box.)     python tests/tools/bench_bn_passes.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from uda_clr_amd.acts import ACT_RELU6, Act, BNRec, round4
from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N, H, C in ((16, 128, 256), (16, 256, 96), (16, 128, 144), (16, 64, 192), (16, 32, 384), (16, 32, 960)):
    P = N * H * H
    Cp = round4(C)
    y = torch.randn(P, Cp, device=dev)[:, :C]
    dU = torch.randn(P, Cp, device=dev)[:, :C]
    out = torch.empty(P, Cp, device=dev)[:, :C]
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    mean, inv = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    a = Act(y, N, H, H, sc, sh, ACT_RELU6, None, 1.0, BNRec("t", mean, inv, float(P), False))
    sums = torch.zeros(16, 3, C, dtype=torch.float64, device=dev)
    st2 = torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
    c1, c2 = torch.randn(C, device=dev), torch.randn(C, device=dev)
    mb = P * C * 4 / 1e6
    t_red = timeit(lambda: K.bnbwd_reduce(dU, a, sums))
    t_app = timeit(lambda: K.bnbwd_apply(dU, a, c1, c2, out))
    t_cs = timeit(lambda: K.colstats(y, st2))
    t_add = timeit(lambda: torch.add(y, dU, out=out))
    print("[%7d x %3d] %6.1f MB | bnbwd_reduce %6.1f us %.2f TB/s | bnbwd_apply %6.1f us %.2f TB/s | colstats %6.1f us %.2f TB/s | torch add (2r+1w) %6.1f us %.2f TB/s"
          % (P, C, mb, t_red, 2 * mb / t_red, t_app, 3 * mb / t_app, t_cs, mb / t_cs, t_add, 3 * mb / t_add), flush=True)
