"""Streaming rates of the depthwise 3x3 kernels (forward / weight gradient / input gradient) at the backbone's layer shapes.
(TEST TOOL, GPU box.)     python tests/tools/bench_dw.py"""
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


# (H, C, stride, dil) of the inverted-residual blocks at 512 x 512, output stride 16
for H, C, stride, dil in ((256, 32, 1, 1), (256, 96, 2, 1), (128, 144, 1, 1), (128, 144, 2, 1), (64, 192, 1, 1), (64, 192, 2, 1), (32, 384, 1, 1),
                          (32, 576, 1, 1), (32, 960, 1, 2)):
    N = 16
    P = N * H * H
    Ho = (H - 1) // stride + 1
    Po = N * Ho * Ho
    x = torch.randn(P, round4(C), device=dev)[:, :C]
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    a = Act(x, N, H, H, sc, sh, ACT_RELU6, None, 1.0, BNRec("t", sh, sc, float(P), True))
    w9 = K.relayout_dw(torch.randn(C, 1, 3, 3, device=dev))
    y = torch.empty(Po, round4(C), device=dev)[:, :C]
    st = torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
    dy = torch.randn(Po, round4(C), device=dev)[:, :C]
    dx = torch.empty(P, round4(C), device=dev)[:, :C]
    dw = torch.empty(C, 1, 3, 3, device=dev)
    t_f = timeit(lambda: K.dwconv_fwd(a, w9, stride, dil, 1, y, st))
    t_w = timeit(lambda: K.dwconv_wgrad(a, dy, stride, dil, 1, dw))
    t_d = timeit(lambda: K.dwconv_dgrad(dy, w9, stride, dil, N, H, H, dx))
    mi, mo = P * C * 4 / 1e6, Po * C * 4 / 1e6
    print("[%3d^2 x %3d s%d d%d] in %6.1f MB out %6.1f MB | fwd %6.1f us %.2f TB/s | wgrad %6.1f us %.2f TB/s | dgrad %6.1f us %.2f TB/s"
          % (H, C, stride, dil, mi, mo, t_f, (mi + mo) / t_f, t_w, (mi + mo) / t_w, t_d, (mi + mo) / t_d), flush=True)
