"""Micro-benchmark of the dominant kernels at the bench shapes (B=16, 512x512 input).
python tests/bench_kernels.py [reps]  -> table of ms, TFLOP/s (algorithmic), GB/s (algorithmic)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uda_clr_amd.acts import ACT_RELU, ACT_RELU6, Act, BNRec, round4
from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = sys.argv[2] if len(sys.argv) > 2 else ""


def timeit(fn):
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


def src(N, H, W, C, lazy=False, mask=False):
    P = N * H * W
    x = torch.randn(P, round4(C), device=dev)[:, :C]
    sc = sh = None
    if lazy:
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    m = (torch.rand(P, round4(C), device=dev) > 0.5).to(torch.uint8)[:, :C] if mask else None
    return Act(x, N, H, W, sc, sh, ACT_RELU if lazy else 0, m, 2.0)


def report(name, ms, flops, bytes_):
    print("%-44s %8.3f ms  %7.2f TFLOP/s  %8.1f GB/s" % (name, ms, flops / ms / 1e9, bytes_ / ms / 1e6), flush=True)


B = 16
cases = []


def conv_case(name, N, H, W, Cin, Cout, k, dil, lazy=False, mask=False, stats=True):
    def run():
        s = src(N, H, W, Cin, lazy, mask)
        w = K.relayout_ohwi(torch.randn(Cout, Cin, k, k, device=dev))
        out = torch.empty(s.P, round4(Cout), device=dev)[:, :Cout]
        st = torch.zeros(16, 2, Cout, dtype=torch.float64, device=dev) if stats else None
        ms = timeit(lambda: K.conv(s, w, k, dil, out, stats=st))
        report(name, ms, 2.0 * s.P * Cout * k * k * Cin, 4.0 * s.P * (Cin + Cout))
    cases.append((name, run))


def wgrad_case(name, N, H, W, Cin, Cout, k, dil, lazy=False, mask=False):
    def run():
        s = src(N, H, W, Cin, lazy, mask)
        dy = torch.randn(s.P, round4(Cout), device=dev)[:, :Cout]
        dw = torch.empty(Cout, Cin, k, k, device=dev)
        ms = timeit(lambda: K.conv_wgrad(s, dy, k, dil, dw))
        report(name, ms, 2.0 * s.P * Cout * k * k * Cin, 4.0 * s.P * (Cin + Cout))
    cases.append((name, run))


conv_case("conv3x3 304->256 128^2 (decoder 0)", B, 128, 128, 304, 256, 3, 1)
conv_case("conv3x3 256->256 128^2 lazy+mask (decoder 4)", B, 128, 128, 256, 256, 3, 1, True, True)
conv_case("dgrad3x3 256->304 128^2", B, 128, 128, 256, 304, 3, 1, stats=False)
conv_case("conv3x3 320->256 32^2 dil6 (aspp)", B, 32, 32, 320, 256, 3, 6)
conv_case("conv1x1 1280->256 32^2 lazy (aspp.conv1)", B, 32, 32, 1280, 256, 1, 1, True)
conv_case("conv1x1 16->96 256^2 (expand)", B, 256, 256, 16, 96, 1, 1)
conv_case("conv1x1 16->96 256^2 (expand) no stats", B, 256, 256, 16, 96, 1, 1, stats=False)
conv_case("conv1x1 96->24 128^2 lazy (project)", B, 128, 128, 96, 24, 1, 1, True)
conv_case("conv1x1 144->24 128^2 lazy (project)", B, 128, 128, 144, 24, 1, 1, True)
conv_case("conv1x1 24->144 128^2 (expand)", B, 128, 128, 24, 144, 1, 1)
conv_case("conv1x1 305->2 128^2 lazy+mask (last_conv)", B, 128, 128, 305, 2, 1, 1, True, True, False)
wgrad_case("wgrad3x3 304->256 128^2", B, 128, 128, 304, 256, 3, 1)
wgrad_case("wgrad3x3 256->256 128^2 lazy+mask", B, 128, 128, 256, 256, 3, 1, True, True)
wgrad_case("wgrad1x1 16->96 256^2", B, 256, 256, 16, 96, 1, 1)
wgrad_case("wgrad1x1 144->24 128^2 lazy", B, 128, 128, 144, 24, 1, 1, True)
wgrad_case("wgrad1x1 1280->256 32^2 lazy", B, 32, 32, 1280, 256, 1, 1, True)


def ew_cases():
    for (Cc, H) in ((96, 256), (144, 128), (24, 128), (384, 32), (960, 32), (576, 32), (305, 128)):      # backbone shapes: no mask, relu6
        P = B * H * H
        y = src(B, H, H, Cc, True, False)
        y.act = ACT_RELU6
        y.bn = BNRec("t", torch.randn(Cc, device=dev), torch.rand(Cc, device=dev) + 0.5, float(P))
        dU = torch.randn(P, round4(Cc), device=dev)[:, :Cc]
        c = torch.randn(4, Cc, device=dev)
        sums = torch.zeros(16, 3, Cc, dtype=torch.float64, device=dev)
        report("bnbwd_reduce %dch %d^2" % (Cc, H), timeit(lambda: K.bnbwd_reduce(dU, y, sums)), 0, P * Cc * 8.0)
        report("bnbwd_apply %dch %d^2" % (Cc, H), timeit(lambda: K.bnbwd_apply(dU, y, c[0], c[1], dU)), 0, P * Cc * 12.0)
    P, C = B * 128 * 128, 256
    y = src(B, 128, 128, C, True, True)
    y.bn = BNRec("t", torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, float(P))
    dU = torch.randn(P, C, device=dev)
    c = torch.randn(4, C, device=dev)
    sums = torch.zeros(16, 3, C, dtype=torch.float64, device=dev)
    report("bnbwd_reduce 256ch 128^2", timeit(lambda: K.bnbwd_reduce(dU, y, sums)), 0, P * C * 9.0)
    report("bnbwd_apply 256ch 128^2", timeit(lambda: K.bnbwd_apply(dU, y, c[0], c[1], dU)), 0, P * C * 13.0)
    for (Cc, H, s, d) in ((96, 256, 2, 1), (144, 128, 1, 1), (32, 256, 1, 1), (960, 32, 1, 2)):
        a = src(B, H, H, Cc, True)
        a.act = ACT_RELU6
        w9 = torch.randn(9, Cc, device=dev)
        Ho = (H - 1) // s + 1
        out = torch.empty(B * Ho * Ho, Cc, device=dev)
        st = torch.zeros(16, 2, Cc, dtype=torch.float64, device=dev)
        report("dw fwd C=%d %d^2 s%d" % (Cc, H, s), timeit(lambda: K.dwconv_fwd(a, w9, s, d, 1, out, st)), 0, 4.0 * Cc * B * (H * H + Ho * Ho))
        dy = torch.randn(B * Ho * Ho, Cc, device=dev)
        dw = torch.empty(Cc, 1, 3, 3, device=dev)
        report("dw wgrad C=%d %d^2 s%d" % (Cc, H, s), timeit(lambda: K.dwconv_wgrad(a, dy, s, d, 1, dw)), 0, 4.0 * Cc * B * (H * H + Ho * Ho))
        dx = torch.empty(B * H * H, Cc, device=dev)
        report("dw dgrad C=%d %d^2 s%d" % (Cc, H, s), timeit(lambda: K.dwconv_dgrad(dy, w9, s, d, B, H, H, dx)), 0, 4.0 * Cc * B * (H * H + Ho * Ho))


cases.append(("elementwise", ew_cases))
for name, run in cases:
    if only in name:
        run()
