"""One 1x1 conv shape under the prologue / epilogue combinations (which part of a narrow layer costs what).  (TEST TOOL, GPU box.)

    python tests/tools/bench_one_conv.py N H Cin Cout [reps]"""
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
N, H, Cin, Cout = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
P = N * H * H
x = torch.randn(P, round4(Cin), device=dev)[:, :Cin]
sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
w = K.relayout_ohwi(torch.randn(Cout, Cin, 1, 1, device=dev))
out = torch.empty(P, round4(Cout), device=dev)[:, :Cout]
ad = torch.randn(P, round4(Cout), device=dev)[:, :Cout]
for lazy in (False, True):
    for stats in (False, True):
        for addend in (False, True):
            s = Act(x, N, H, H, sc if lazy else None, sh if lazy else None, ACT_RELU6 if lazy else 0)
            st = torch.zeros(16, 2, Cout, dtype=torch.float64, device=dev) if stats else None

            def fn():
                K.conv(s, w, 1, 1, out, addend=ad if addend else None, stats=st)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            by = 4.0 * P * (Cin + Cout * (2 if addend else 1))
            print("%d x %d^2 %d->%d lazy=%d stats=%d addend=%d: %7.1f us %6.0f GB/s" % (N, H, Cin, Cout, lazy, stats, addend, us, by / us / 1e3), flush=True)
