"""-m gpu: the headline metric's second half - "val Dice vs ref".  The same seeded DeepLabV3+/MobileNetV2 is trained for 48
Adam steps on the same synthetic fundus-like batches (i) by the product on the HIP kernels (``Trainer_baseline`` code path)
and (ii) by the CPU restatement of the reference (oracle/step_ref.baseline_step), each with its own dropout stream, and both
are scored on a held-out set with the reference's Dice (utils/metrics.py:118-132).  north_star: Dice within +-0.2 of the
reference; the two runs are also required to actually learn."""
import pytest
import torch

import model_cases
from make_golden_inputs import synth_loader
from oracle import deeplab_ref, metrics_ref, step_ref
from uda_clr_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _dice(model, loader, dev):
    model.eval()
    cup = disc = 0.0
    with torch.no_grad():
        for s in loader:
            p = model(s["image"].to(dev))[0].float().cpu()
            c, d = metrics_ref.dice_coeff_2label(p, s["map"])
            cup += c
            disc += d
    model.train()
    return cup / len(loader), disc / len(loader)


def test_val_dice_tracks_the_reference_after_training():
    S, B, steps = 128, 8, 48
    train, val = synth_loader(12, B, S, 3000), synth_loader(4, B, S, 4000)
    m = model_cases.seeded_model()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    # ---- reference arithmetic on the host cores
    om = deeplab_ref.OracleDeepLab(sd0).train()
    oo = torch.optim.Adam(om.parameters(), lr=1e-3, betas=(0.9, 0.99))
    torch.manual_seed(5)
    for i in range(steps):
        s = train[i % len(train)]
        step_ref.baseline_step(om, oo, s["image"], s["map"], s["boundary"])
    ref_cup, ref_disc = _dice(om, val, "cpu")
    # ---- product on the MI355X
    m.to(DEV).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.99), fused=True)
    for i in range(steps):
        s = train[i % len(train)]
        opt.zero_grad(set_to_none=True)
        o, b = m(s["image"].to(DEV))[:2]
        loss = ops.seg_loss(o, b, s["map"].to(DEV), s["boundary"].to(DEV))
        loss.backward()
        opt.step()
    cup, disc = _dice(m, val, DEV)
    print("val Dice (cup, disc): reference arithmetic %.3f %.3f, HIP path %.3f %.3f" % (ref_cup, ref_disc, cup, disc))
    assert abs(cup - ref_cup) < 0.2 and abs(disc - ref_disc) < 0.2
    assert disc > 0.5 and ref_disc > 0.5, "neither run learned the disc"
