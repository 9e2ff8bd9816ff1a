"""Micro-benchmark of the discriminator layers at the bench shape (B=16, 512x512): per-kernel ms and TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uda_clr_amd.acts import Act, conv_weight_shape, round4
from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


grids = [258, 131, 67, 35, 19]
chans = [2, 64, 128, 256, 512, 1]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for l in range(5):
    Hz, Cc, O = grids[l], chans[l], chans[l + 1]
    P = B * Hz * Hz
    z = torch.randn(P, 4 * Cc, device=dev)
    dy = torch.randn(P, round4(O), device=dev)[:, :O]
    w = torch.randn(conv_weight_shape(O, 2, 4 * Cc), device=dev)
    wd = torch.randn(conv_weight_shape(4 * Cc, 2, O), device=dev)
    y = torch.empty(P, round4(O), device=dev)[:, :O]
    dz = torch.empty(P, 4 * Cc, device=dev)
    dw = torch.empty(O, 4 * Cc, 2, 2, device=dev)
    fl = 2.0 * P * O * 16 * Cc
    for name, fn in (("fwd", lambda: K.conv(Act(z, B, Hz, Hz), w, 2, 1, y, origin=0)),
                     ("dgrad", lambda: K.conv(Act(dy, B, Hz, Hz), wd, 2, 1, dz, origin=1)),
                     ("wgrad", lambda: K.conv_wgrad(Act(z, B, Hz, Hz), dy, 2, 1, dw, origin=0))):
        ms = timeit(fn)
        tot[name] += ms
        print("L%d %-6s P=%7d %4d->%4d  %8.3f ms  %7.2f TFLOP/s" % (l + 1, name, P, 4 * Cc, O, ms, fl / ms / 1e9), flush=True)
print("totals per discriminator pass: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms" % (tot["fwd"], tot["dgrad"], tot["wgrad"]))
