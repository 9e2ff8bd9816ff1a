"""Micro-benchmark of the decoder-conv0 interpolation kernels at the bench shapes: 32x32 -> 128x128, 256 channels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uda_clr_amd.kernels import HipKernels
K = HipKernels(); dev = torch.device('cuda:0')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for N in (16, 32):
    dy = torch.randn(N * 128 * 128, 256, device=dev)
    dg = torch.empty(N * 32 * 32, 2304, device=dev)
    g = torch.randn(N * 32 * 32, 2304, device=dev)
    y0 = torch.randn(16 * 128 * 128, 256, device=dev)
    y = torch.empty(N * 128 * 128, 256, device=dev)
    st = torch.zeros(16, 2, 256, dtype=torch.float64, device=dev)
    print("N=%d upconv_bwd %.3f ms   upconv_fwd(+stats) %.3f ms" % (N, t(lambda: K.upconv_bwd(dy, N, 128, 128, dg, 32, 32)),
          t(lambda: K.upconv_fwd(g, N, 32, 32, y, 128, 128, addend=y0, stats=st))))
