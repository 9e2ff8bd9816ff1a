"""Does a second HIP stream hide small kernels behind a dense one?  A 3x3 conv (256 -> 256, 128^2, B=16: ~2.6 ms of MFMA work) on one
stream, 100 small BN-backward applies (384 ch, 32^2, B=16: ~18 us each, launch-floor bound) on another; sequential vs concurrent."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uda_clr_amd.acts import ACT_RELU6, Act, BNRec, conv_weight_shape
from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels()
B = 16
x = torch.randn(B * 128 * 128, 256, device=dev)
w = torch.randn(conv_weight_shape(256, 3, 256), device=dev)
y = torch.empty(B * 128 * 128, 256, device=dev)
P, C = B * 32 * 32, 384
ys = torch.randn(P, C, device=dev)
dU = torch.randn(P, C, device=dev)
out = torch.empty(P, C, device=dev)
coef = torch.rand(6, C, device=dev) + 0.5
act = Act(ys, B, 32, 32, coef[0].contiguous(), coef[1].contiguous(), ACT_RELU6, None, 1.0, BNRec("t", coef[2].contiguous(), coef[3].contiguous(), float(P)))


def dense(n=4):
    for _ in range(n):
        K.conv(Act(x, B, 128, 128), w, 3, 1, y)


def small(n=400):
    for _ in range(n):
        K.bnbwd_apply(dU, act, coef[4], coef[5], out)


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


side = torch.cuda.Stream()


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dense()
    small()
    torch.cuda.current_stream().wait_stream(side)


for _ in range(2):
    dense(); small(); both()
td, ts, tb = wall(dense), wall(small), wall(both)
print("dense alone %.2f ms, small alone %.2f ms, sum %.2f ms; concurrent on two streams %.2f ms" % (td, ts, td + ts, tb))
