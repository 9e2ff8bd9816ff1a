import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from uda_clr_amd.acts import Act, round4
from uda_clr_amd.kernels import HipKernels
dev = torch.device("cuda:0"); K = HipKernels()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N in (16, 4):
    for Cout in (64, 32):
        for Cin in (32, 64, 128, 256, 384, 768, 1536):
            P = N * 32 * 32
            x = torch.randn(P, round4(Cin), device=dev)[:, :Cin]
            s = Act(x, N, 32, 32)
            w = K.relayout_ohwi(torch.randn(Cout, Cin, 1, 1, device=dev))
            out = torch.empty(P, round4(Cout), device=dev)[:, :Cout]
            us = timeit(lambda: K.conv(s, w, 1, 1, out))
            print("P=%6d Cout=%3d K=%5d chunks=%3d  %7.1f us" % (P, Cout, Cin, (Cin + 31) // 32, us), flush=True)
# python-side overhead of a conv call (no kernel): time 200 calls wall
import time
x = torch.randn(1024, 64, device=dev); s = Act(x, 1, 32, 32); w = K.relayout_ohwi(torch.randn(64, 64, 1, 1, device=dev)); out = torch.empty(1024, 64, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): K.conv(s, w, 1, 1, out)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host time per K.conv call: %.1f us" % ((t1 - t0) / 500 * 1e6))
