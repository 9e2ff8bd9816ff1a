"""Micro-benchmark: the wide-tile conv kernels on the fp32 MFMA (exact fma chain) and on the bf16 pipe with exact 3-way operand
splitting (UDA_MFMA_BF16X3), same operands; error of both against a float64 evaluation.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kernel_cases import make_src, act_to, gen
from kernel_spec import SpecKernels, transform, _nchw, _rows
from uda_clr_amd.acts import ACT_RELU, Act
from uda_clr_amd.kernels import HipKernels
dev = torch.device("cuda:0")
K = HipKernels()
S = SpecKernels()


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


cases = [("decoder conv4 fwd masked B=16", 16, 128, 128, 256, 256, 3, 1, True, True),
         ("decoder conv4 fwd masked B=32 (MC batch)", 32, 128, 128, 256, 256, 3, 1, True, True),
         ("conv4 dgrad raw B=16", 16, 128, 128, 256, 256, 3, 1, False, False),
         ("ASPP atrous d=6 B=16", 16, 32, 32, 320, 256, 3, 6, True, False),
         ("disc L2 fwd 2x2 64->128 (z 256) B=16", 16, 131, 131, 256, 128, 2, 1, False, False),
         ("disc L2 dgrad 2x2 (dy 128 -> z 256) B=16", 16, 131, 131, 128, 256, 2, 1, False, False),
         ("disc L3 2x2 128->256 B=16", 16, 67, 67, 512, 256, 2, 1, False, False),
         ("disc L3 dgrad 2x2 (dy 256 -> z 512) B=16", 16, 67, 67, 256, 512, 2, 1, False, False),
         ("disc L4 fwd 2x2 (z 1024 -> 512) B=16", 16, 35, 35, 1024, 512, 2, 1, False, False),
         ("disc L4 dgrad 2x2 (dy 512 -> z 1024) B=16", 16, 35, 35, 512, 1024, 2, 1, False, False),
         ("1x1 1280->256 B=16", 16, 32, 32, 1280, 256, 1, 1, True, False),
         ("1x1 tap GEMM 256->2304 B=32", 32, 32, 32, 256, 2304, 1, 1, False, False),
         ("1x1 tap GEMM 256->2304 B=16", 16, 32, 32, 256, 2304, 1, 1, False, False)]
if os.environ.get("X3_ONLY"):
    cases = [c for c in cases if os.environ["X3_ONLY"] in c[0]]
for name, N, H, W, Cin, Cout, k, dil, lazy, mask in cases:
    g = gen(1)
    src = make_src(N, H, W, Cin, g, lazy, ACT_RELU, mask)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    P = N * H * W
    sh = act_to(src, dev)
    wl = K.relayout_ohwi(w.to(dev))
    out = torch.empty(P, Cout, device=dev)
    flops = 2.0 * P * Cout * Cin * k * k
    res = {}
    for mode, tag in ((K.MFMA_F32, "f32"), (K.MFMA_BF16X3, "bf16x3")):
        K.mfma = mode
        ms = timeit(lambda: K.conv(sh, wl, k, dil, out))
        res[tag] = (ms, out.clone())
    # float64 reference on the first image only (the whole batch is slow on the host): borders inside one image, so a slice works
    HW = H * W
    s1 = Act(src.x[:HW].double(), 1, H, W, None if src.scale is None else src.scale.double(), None if src.shift is None else src.shift.double(),
             src.act, None if src.mask is None else src.mask[:HW], src.mask_scale)
    ref = torch.empty(HW, Cout, dtype=torch.float64)
    S.conv(s1, S.relayout_ohwi(w.double()), k, dil, ref)
    line = "%-42s %6.1f GFLOP" % (name, flops / 1e9)
    for tag in ("f32", "bf16x3"):
        ms, o = res[tag]
        e = ((o[:H * W].double().cpu() - ref).norm() / ref.norm()).item()
        line += " | %s %7.3f ms %6.1f TF err64 %.2e" % (tag, ms, flops / ms / 1e9, e)
    line += " | speedup %.2fx" % (res["f32"][0] / res["bf16x3"][0])
    print(line, flush=True)

# ---------------------------------------------------------------- weight gradients (packed operands exist already in a training step:
# timed with the packs cached, and once with dy's pack included)
wcases = [("wgrad conv4 3x3 256->256 B=16", 16, 128, 128, 256, 256, 3, 1, True, True),
          ("wgrad conv3 3x3 304->256 B=16", 16, 128, 128, 304, 256, 3, 1, False, False),
          ("wgrad ASPP d=6 320->256 B=16", 16, 32, 32, 320, 256, 3, 6, True, False),
          ("wgrad disc L2 2x2 256->128 B=16", 16, 131, 131, 256, 128, 2, 1, False, False),
          ("wgrad disc L3 2x2 512->256 B=16", 16, 67, 67, 512, 256, 2, 1, False, False),
          ("wgrad disc L4 2x2 1024->512 B=16", 16, 35, 35, 1024, 512, 2, 1, False, False)]
if os.environ.get("X3_ONLY"):
    wcases = [c for c in wcases if os.environ["X3_ONLY"] in c[0]]
for name, N, H, W, Cin, Cout, k, dil, lazy, mask in wcases:
    g = gen(1)
    src = make_src(N, H, W, Cin, g, lazy, ACT_RELU, mask)
    P = N * H * W
    dy = torch.randn(P, Cout, generator=g).to(dev)
    sh = act_to(src, dev)
    dw = torch.empty(Cout, Cin, k, k, device=dev)
    flops = 2.0 * P * Cout * Cin * k * k
    res = {}
    for mode, tag in ((K.MFMA_F32, "f32"), (K.MFMA_BF16X3, "bf16x3")):
        K.mfma = mode
        ms = timeit(lambda: K.conv_wgrad(sh, dy, k, dil, dw, origin=0))
        res[tag] = (ms, dw.clone())

    def with_pack():
        if hasattr(dy, "_x3"):
            del dy._x3
        K.conv_wgrad(sh, dy, k, dil, dw, origin=0)
    ms_pack = timeit(with_pack)
    e = ((res["bf16x3"][1].double() - res["f32"][1].double()).norm() / res["f32"][1].double().norm()).item()
    print("%-36s %6.1f GFLOP | f32 %7.3f ms %6.1f TF | bf16x3 %7.3f ms %6.1f TF (%.3f ms with the dy pack) | speedup %.2fx | rel diff %.1e" % (
        name, flops / 1e9, res["f32"][0], flops / res["f32"][0] / 1e9, res["bf16x3"][0], flops / res["bf16x3"][0] / 1e9, ms_pack,
        res["f32"][0] / res["bf16x3"][0], e), flush=True)
