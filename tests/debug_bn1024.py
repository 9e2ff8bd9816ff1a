import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kernel_cases import *
dev = torch.device("cuda:0")
for (P, C) in ((300, 1024), (32, 1024), (64, 1024), (300, 960), (300, 512), (300, 768), (1000, 1024)):
    g = gen(5)
    x = padded(P, C, g, scale=2.0)
    cr = torch.empty(4, C)
    st = torch.zeros(2, C, dtype=torch.float64); SPEC.colstats(x, st)
    gamma, beta = 0.5 + torch.rand(C, generator=g), torch.randn(C, generator=g)
    SPEC.bn_finalize(st, float(P), gamma, beta, torch.zeros(C), torch.ones(C), 0.1, 1e-5, cr[0], cr[1], cr[2], cr[3])
    y = Act(x, 1, 1, P, cr[0].clone(), cr[1].clone(), ACT_RELU, None, 1.0, BNRec("t", cr[2].clone(), cr[3].clone(), float(P), False))
    dU = padded(P, C, g)
    s_r = torch.zeros(3, C, dtype=torch.float64); SPEC.bnbwd_reduce(dU, y, s_r)
    s_h = torch.zeros(3, C, dtype=torch.float64, device=dev)
    hip().bnbwd_reduce(to_dev(dU, dev), act_to(y, dev), s_h)
    d = (s_h.cpu() - s_r).abs()
    rows = [float(d[q].max() / s_r[q].abs().max()) for q in range(3)]
    badc = (d.max(0).values > 1e-4 * s_r.abs().max()).nonzero().flatten()
    print("P=%d C=%d row errs %s bad cols %d first %s last %s" % (P, C, ["%.1e" % r for r in rows], badc.numel(), badc[:6].tolist(), badc[-6:].tolist()), flush=True)
    if badc.numel():
        c = int(badc[0]); print("   col %d: hip %s ref %s" % (c, s_h[:, c].tolist(), s_r[:, c].tolist()))
