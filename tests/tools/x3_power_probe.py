"""Is the bf16x3 conv kernel bound by the matrix pipe's ISSUE rate or by the clock the chip holds under its load?  (TEST TOOL, GPU box.)

The same launch (decoder conv4, 3x3 256 -> 256 at 128^2: 1024 / 2048 tiles of 256 x 256, 144 K-chunks) is timed
  * on random operands and on all-zero operands (MI355X_MICROARCH.md, 'DVFS give-back': identical instruction stream and cycle
    count, the zero run holds a higher clock - the ratio is the share of the time that is clock, not issue slots);
  * sustained (50 back-to-back launches) and as single launches after the chip sat idle for 30 ms.
Printed beside the cycle floor of the kernel: tiles / 256 CUs x 144 chunks x 3072 matrix-pipe cycles per chunk and SIMD.

    python tests/tools/x3_power_probe.py        -> gpurun_out/x3_power_probe.txt"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from uda_clr_amd.acts import Act
import uda_clr_amd.kernels as _kmod
from uda_clr_amd.kernels import HipKernels

if os.environ.get("UDA_PROBE_LIB"):          # A/B of two builds of the library on one box (this tool only)
    _kmod._LIB_PATH = os.path.abspath(os.environ["UDA_PROBE_LIB"])

dev = torch.device("cuda:0")
K = HipKernels(mfma="bf16x3")
out_f = open(os.path.join("gpurun_out", "x3_power_probe.txt"), "a") if os.path.isdir("gpurun_out") else None


def emit(s):
    print(s, flush=True)
    if out_f:
        out_f.write(s + "\n")
        out_f.flush()


def run(N, fill):
    H = W = 128
    Cin = Cout = 256
    P = N * H * W
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(P, Cin, generator=g, device=dev) if fill != "zero-x" and fill != "zero-all" else torch.zeros(P, Cin, device=dev)
    if fill == "sparse75":                  # what conv4 reads inside the step: ReLU then dropout 0.5 leave ~25 % non-zeros
        x = x * (torch.rand(P, Cin, generator=g, device=dev) < 0.25)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, device=dev) / 48.0 if fill != "zero-all" else torch.zeros(Cout, Cin, 3, 3, device=dev)
    wl = K.relayout_ohwi(w)
    out = torch.empty(P, Cout, device=dev)
    src = Act(x, N, H, W)
    K.conv(src, wl, 3, 1, out)            # packs (cached on x / wl)
    torch.cuda.synchronize()
    flops = 2.0 * P * Cout * Cin * 9
    # sustained
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        K.conv(src, wl, 3, 1, out)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        K.conv(src, wl, 3, 1, out)
    e1.record()
    torch.cuda.synchronize()
    sus = e0.elapsed_time(e1) / 50
    # single launches after an idle gap
    singles = []
    for _ in range(8):
        torch.cuda.synchronize()
        time.sleep(0.03)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        K.conv(src, wl, 3, 1, out)
        b.record()
        torch.cuda.synchronize()
        singles.append(a.elapsed_time(b))
    singles.sort()
    tiles = (P // 256) * 1
    cyc = (tiles / 256.0) * 144 * 3072
    emit("conv4 N=%2d %-9s sustained %.3f ms = %5.1f TF-eq (%4.0f bf16 TF; clock implied by the cycle floor %.2f GHz) | "
         "after 30 ms idle: median %.3f ms = %5.1f TF-eq (%.2f GHz), best %.3f ms"
         % (N, fill, sus, flops / sus / 1e9, 6 * flops / sus / 1e9, cyc / (sus * 1e-3) / 1e9,
            singles[len(singles) // 2], flops / singles[len(singles) // 2] / 1e9, cyc / (singles[len(singles) // 2] * 1e-3) / 1e9, singles[0]))


emit("bf16x3 conv4 (3x3 256->256 @128^2, 256x256 tiles): matrix-pipe floor = tiles/256 x 144 chunks x 3072 cycles")
for N in (16, 32):
    for fill in ("random", "sparse75", "zero-x", "zero-all"):
        run(N, fill)
