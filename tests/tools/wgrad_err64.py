"""Weight-gradient error against a float64 evaluation (CPU, tests/kernel_spec.py) in both matrix modes (GPU box)."""
import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
from kernel_cases import make_src, act_to, gen
from kernel_spec import SpecKernels
from uda_clr_amd.acts import ACT_RELU, Act
from uda_clr_amd.kernels import HipKernels
dev = torch.device("cuda:0"); K = HipKernels(); S = SpecKernels()
torch.set_num_threads(16)
for (N, H, W, Cin, Cout, k, lazy, mask) in ((2, 128, 128, 256, 256, 3, True, True), (2, 128, 128, 48, 256, 3, False, False), (2, 131, 131, 256, 128, 2, False, False)):
    g = gen(1)
    src = make_src(N, H, W, Cin, g, lazy, ACT_RELU, mask)
    P = N * H * W
    dy = torch.randn(P, Cout, generator=g)
    s64 = Act(src.x.double(), N, H, W, None if src.scale is None else src.scale.double(), None if src.shift is None else src.shift.double(), src.act, src.mask, src.mask_scale)
    ref = torch.empty(Cout, Cin, k, k, dtype=torch.float64)
    S.conv_wgrad(s64, dy.double(), k, 1, ref, origin=0)
    sh = act_to(src, dev); dyd = dy.to(dev)
    line = "wgrad %dx%d %d->%d P=%d:" % (k, k, Cin, Cout, P)
    for mode, tag in ((K.MFMA_F32, "f32"), (K.MFMA_BF16X3, "bf16x3")):
        K.mfma = mode
        dw = torch.empty(Cout, Cin, k, k, device=dev)
        K.conv_wgrad(sh, dyd, k, 1, dw, origin=0)
        e = ((dw.double().cpu() - ref).norm() / ref.norm()).item()
        line += "  %s err64 %.2e" % (tag, e)
    print(line, flush=True)
