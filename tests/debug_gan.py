"""Layer-by-layer comparison of the discriminator engine on the HIP kernels vs the torch statement (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kernel_spec import SpecKernels
from uda_clr_amd.gan_engine import PatchDiscriminatorEngine
from uda_clr_amd.kernels import HipKernels


class Rec:
    def __init__(self, K):
        self.K, self.log = K, []

    def __getattr__(self, name):
        fn = getattr(self.K, name)

        def wrap(*a, **k):
            r = fn(*a, **k)
            outs = {"conv": 4, "conv_wgrad": 4, "s2d_fwd": 9, "s2d_bwd": 9}
            if name in outs:
                self.log.append((name, a[outs[name]].detach().float().cpu().clone()))
            return r
        return wrap


B, C, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator().manual_seed(0)
ws = [torch.randn(o, i, 4, 4, generator=g) * 0.02 for i, o in zip([C, 64, 128, 256, 512], [64, 128, 256, 512, 1])]
x = torch.rand(B, C, S, S, generator=g)
res = []
for dev, K in (("cpu", SpecKernels()), ("cuda:0", HipKernels())):
    R = Rec(K)
    E = PatchDiscriminatorEngine(R)
    out, ctx = E.forward(x.to(dev), [w.to(dev) for w in ws], True)
    go = torch.ones_like(out)
    dx, dws = E.backward(ctx, go, [w.to(dev) for w in ws], True, True)
    res.append(R.log)
for (n0, a), (n1, b) in zip(*res):
    err = (a.double() - b.double()).abs().max().item() / max(a.double().abs().max().item(), 1e-30)
    print("%-12s %-28s rel err %.3e" % (n0, tuple(a.shape), err))
