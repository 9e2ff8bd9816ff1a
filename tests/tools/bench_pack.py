"""uda_x3_pack on the decoder's conv4 operand shapes (raw / BN + ReLU / + keep-mask).  (TEST TOOL, GPU box.)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from uda_clr_amd.acts import ACT_RELU, Act
from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels(mfma="bf16x3")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, H, C = 16, 128, 256
P = N * H * H
x = torch.randn(P, C, device=dev)
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
mk = (torch.rand(P, C, device=dev) > 0.5).to(torch.uint8)
for name, a in (("raw", Act(x, N, H, H)), ("bn+relu", Act(x, N, H, H, sc, sh, ACT_RELU)), ("bn+relu+mask", Act(x, N, H, H, sc, sh, ACT_RELU, mk, 2.0))):
    u = K._src(a)
    for _ in range(3):
        K.x3_pack(u, P, C, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.x3_pack(u, P, C, dev)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    by = P * C * (4 + 6 + (1 if a.mask is not None else 0))
    print("x3_pack %-13s [%d x %d]: %6.1f us %.2f TB/s" % (name, P, C, us, by / us / 1e6), flush=True)
