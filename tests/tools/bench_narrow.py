"""Micro-benchmark of the backbone's 1x1 convolutions (MobileNetV2 inverted-residual expand / project convs and their input
gradients, mobilenet.py:43-57) at the B = 16, 512 x 512 shapes - the layers that run on the narrow fp32-MFMA kernel
(igemm_conv_kernel) or the 1x1 form of the wide-tile kernel.  (TEST TOOL, GPU box.)

    python tests/tools/bench_narrow.py [reps=20]      -> table: us, fp32 TFLOP/s, GB/s (in + out once)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from uda_clr_amd.acts import ACT_RELU6, Act, round4
from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
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


def run(name, N, H, Cin, Cout, lazy, stats, addend):
    P = N * H * H
    x = torch.randn(P, round4(Cin), device=dev)[:, :Cin]
    sc = sh = None
    if lazy:
        sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    s = Act(x, N, H, H, sc, sh, ACT_RELU6 if lazy else 0)
    w = K.relayout_ohwi(torch.randn(Cout, Cin, 1, 1, device=dev))
    out = torch.empty(P, round4(Cout), device=dev)[:, :Cout]
    st = torch.zeros(16, 2, Cout, dtype=torch.float64, device=dev) if stats else None
    ad = torch.randn(P, round4(Cout), device=dev)[:, :Cout] if addend else None
    us = timeit(lambda: K.conv(s, w, 1, 1, out, addend=ad, stats=st))
    fl, by = 2.0 * P * Cin * Cout, 4.0 * P * (Cin + Cout * (2 if addend else 1))
    print("%-46s %8.1f us %7.1f TF %8.0f GB/s" % (name, us, fl / us / 1e6, by / us / 1e3), flush=True)
    return us


B = 16
total = 0.0
# (name, H, Cin, Cout, calls per step as forward (lazy input, statistics) / as input gradient (raw input, addend))
layers = [("expand 16->96 @256", 256, 16, 96, 2, 0), ("project 96->24 @128", 128, 96, 24, 2, 0), ("expand 24->144 @128", 128, 24, 144, 4, 2),
          ("project 144->24 @128", 128, 144, 24, 2, 2), ("project 144->32 @64", 64, 144, 32, 2, 0), ("expand 32->192 @64", 64, 32, 192, 6, 4),
          ("project 192->32 @64", 64, 192, 32, 4, 6), ("project 192->64 @32", 32, 192, 64, 2, 0), ("expand 64->384 @32", 32, 64, 384, 8, 6),
          ("project 384->64 @32", 32, 384, 64, 6, 8), ("project 384->96 @32", 32, 384, 96, 2, 0), ("expand 96->576 @32", 32, 96, 576, 6, 4),
          ("project 576->96 @32", 32, 576, 96, 4, 6), ("project 576->160 @32", 32, 576, 160, 2, 0), ("expand 160->960 @32", 32, 160, 960, 6, 4),
          ("project 960->160 @32", 32, 960, 160, 4, 6), ("project 960->320 @32", 32, 960, 320, 2, 0)]
for name, H, Cin, Cout, nf, nb in layers:
    if nf:
        total += nf * run(name + " fwd (lazy, stats) x%d" % nf, B, H, Cin, Cout, True, True, False)
    if nb:
        total += nb * run(name + " as dgrad (raw, addend) x%d" % nb, B, H, Cin, Cout, False, False, True)
print("weighted sum over a prototype_full step: %.2f ms" % (total / 1e3))
