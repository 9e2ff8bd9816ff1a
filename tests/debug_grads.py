import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import model_cases
from oracle import deeplab_ref, step_ref
from make_golden_inputs import synth_targets
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = 2
m = model_cases.seeded_model().train()
gen = torch.Generator().manual_seed(3)
x = torch.randn(B, 3, S, S, generator=gen)
tmap, tbd = synth_targets(B, S, S, 11)
masks = deeplab_ref.draw_masks(B, S, S, gen)
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
o32 = deeplab_ref.canonical_state(sd0, requires_grad=True)
r32 = deeplab_ref.deeplab_forward(o32, x, True, masks)
step_ref.seg_loss(r32[0], r32[1], tmap, tbd).backward()
o64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone()) for k, v in deeplab_ref.canonical_state(sd0, True).items()}
r64 = deeplab_ref.deeplab_forward(o64, x.double(), True, masks)
step_ref.seg_loss(r64[0], r64[1], tmap.double(), tbd.double()).backward()
m.to(dev); m.set_dropout_masks(masks)
out = m(x.to(dev))
step_ref.seg_loss(out[0], out[1], tmap.to(dev), tbd.to(dev)).backward()
live = m._flat_state()
rows = []
for k in deeplab_ref.parameter_keys(o32):
    g = live[k].grad.double().cpu().reshape(-1); r = o64[k].grad.reshape(-1); f = o32[k].grad.double().reshape(-1)
    nr = r.norm().item()
    e_h = (g - r).norm().item() / max(nr, 1e-30); e_f = (f - r).norm().item() / max(nr, 1e-30)
    rows.append((e_h, e_f, nr, k))
rows.sort(reverse=True)
print("S=%d: L2 relative gradient error vs fp64 oracle: HIP | fp32 oracle | ||g||  (worst 25 by HIP error)" % S)
for e_h, e_f, nr, k in rows[:25]:
    print("  %-48s %.2e | %.2e | %.2e" % (k, e_h, e_f, nr))
import statistics
print("median HIP %.2e  median fp32-oracle %.2e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))
