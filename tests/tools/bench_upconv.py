"""Timing of uda_upconv_fwd (decoder conv0's interpolation pass, upconv.hip) at the step's shapes.  (TEST TOOL, GPU box.)

    python tests/tools/bench_upconv.py        # UDA_UPCONV_TILE=0 selects the strip kernel"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels()
for N, rows_of in ((16, 16), (32, 16)):          # grad-mode forward; stochastic pass (repeated batch shares the low-level conv y0)
    h = w = 32
    H = W = 128
    C = 256
    g = torch.randn(N * h * w, 9 * C, device=dev)
    y0 = torch.randn(rows_of * H * W, C, device=dev)
    out = torch.empty(N * H * W, C, device=dev)
    st = torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
    for _ in range(3):
        K.upconv_fwd(g, N, h, w, out, H, W, addend=y0, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.upconv_fwd(g, N, h, w, out, H, W, addend=y0, stats=st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = 4.0 * (g.numel() + y0.numel() + out.numel())
    print("upconv_fwd N=%d (y0 of %d images): %.1f us, %.2f TB/s of (g + y0 + out once)" % (N, rows_of, us, by / us / 1e6), flush=True)
