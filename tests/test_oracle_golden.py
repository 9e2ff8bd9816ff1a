"""not-gpu: the oracle restatement against the fixtures written by the reference itself
(tests/golden/, generator: tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from make_golden_inputs import synth_loader, synth_targets
from oracle import deeplab_ref, metrics_ref, proto_ref, step_ref
from uda_clr_amd.networks.deeplabv3 import DeepLab

NAMES = ("x1", "x2", "feature", "x_bu_feature", "x_feature", "x1_before", "x2_before")


def _seeded_sd():
    torch.manual_seed(1337)
    return DeepLab(num_classes=2, backbone="mobilenet", output_stride=16).state_dict()


def _sample(t, n=97):
    f = t.detach().double().reshape(-1)
    return f[torch.linspace(0, f.numel() - 1, n).long()].numpy()


@pytest.mark.parametrize("tag", ["64", "512"])
def test_forward_fixture(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "forward_%s.npz" % tag))
    B, S = int(z["B"]), int(z["S"])
    torch.manual_seed(int(z["input_seed"]))
    x = torch.randn(B, 3, S, S)
    sd0 = _seeded_sd()
    with torch.no_grad():
        out = deeplab_ref.deeplab_forward(deeplab_ref.canonical_state(sd0), x, training=False)
    for n, t in zip(NAMES, out):
        np.testing.assert_allclose(_sample(t), z["eval.%s.smp" % n], rtol=1e-5, atol=1e-6)
        assert abs(t.double().sum().item() - float(z["eval.%s.sum" % n])) <= 1e-6 * float(z["eval.%s.abs" % n])
    tmap, tbd = synth_targets(B, S, S, int(z["target_seed"]))
    osd = deeplab_ref.canonical_state(sd0, requires_grad=True)
    rec = {}
    torch.manual_seed(int(z["dropout_seed"]))
    out = deeplab_ref.deeplab_forward(osd, x, training=True, record=rec)
    loss = step_ref.seg_loss(out[0], out[1], tmap, tbd)
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-6
    for k, v in rec.items():
        assert int(v.sum()) == int(z["mask.%s.sum" % k])
    gn = np.array([osd[k].grad.double().norm().item() for k in z["train.grad_keys"]])
    np.testing.assert_allclose(gn, z["train.grad_norm"], rtol=1e-4)
    bs = np.array([osd[k].double().sum().item() for k in z["train.bn_keys"]])
    np.testing.assert_allclose(bs, z["train.bn_sum"], rtol=1e-5, atol=1e-6)


def test_prototype_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "proto.npz"))
    g = torch.Generator().manual_seed(int(z["gp.seed"]))
    B, C, h = 2, 305, 32
    feat = torch.randn(B, C, h, h, generator=g)
    hard = (torch.rand(B, 2, h, h, generator=g) > 0.6).float()
    soft = torch.rand(B, 2, h, h, generator=g)
    for tag, pred in (("hard", hard), ("soft", soft)):
        for i, c in enumerate(proto_ref.gen_prototype(pred, feat)):
            np.testing.assert_allclose(c.reshape(-1).numpy(), z["gp.%s.%d" % (tag, i)], rtol=1e-6, atol=1e-7)
    g = torch.Generator().manual_seed(int(z["rt.seed"]))
    B, T = 1, 8
    base = 2.0 * torch.randn(B, 2, 512, 512, generator=g)
    base = torch.nn.functional.avg_pool2d(base, 9, 1, 4) * 6.0
    preds = base.repeat(T, 1, 1, 1) + 0.35 * torch.randn(T * B, 2, 512, 512, generator=g) * \
        (torch.rand(1, 2, 512, 512, generator=g) > 0.5).float()
    oT = torch.nn.functional.interpolate(base, size=(128, 128), mode="bilinear", align_corners=True).clone().requires_grad_(True)
    xt = torch.randn(B, 305, 128, 128, generator=g).requires_grad_(True)
    res = proto_ref.gen_prototype_retrify(oT, xt, preds, T, B)
    for n, r in zip(("c0_obj", "c1_obj", "c0_bck", "c1_bck"), res[:4]):
        np.testing.assert_allclose(r.detach().reshape(-1).numpy(), z["rt." + n], rtol=1e-6, atol=1e-7)
    for n, r in zip(("std_map", "mask_0", "mask_1"), res[4:]):
        assert abs(r.double().sum().item() - float(z["rt.%s.sum" % n])) <= 1e-6 * max(1.0, float(z["rt.%s.abs" % n]))
    sum(r.sum() for r in res[:4]).backward()
    assert abs(xt.grad.double().abs().sum().item() - float(z["rt.grad_xt.abs"])) < 1e-6 * float(z["rt.grad_xt.abs"])
    assert oT.grad is None or float(oT.grad.abs().sum()) == float(z["rt.grad_oT.abs"]) == 0.0      # quirk Q6


def test_metrics_fixture(golden_dir):
    z = json.load(open(os.path.join(golden_dir, "metrics.json")))
    g = torch.Generator().manual_seed(z["logit_seed"])
    tmap, _ = synth_targets(z["B"], z["S"], z["S"], z["target_seed"])
    logits = (tmap * 2 - 1) * 2.0 + 1.5 * torch.randn(z["B"], 2, z["S"], z["S"], generator=g)
    np.testing.assert_allclose(metrics_ref.dice_coeff_2label(logits, tmap), z["dice"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(metrics_ref.pixel_acc(logits, tmap), z["pixel_acc"], rtol=0, atol=1e-12)


def test_baseline_trainer_trajectory(golden_dir):
    """BASELINE.json configs[0]-shaped plumbing case: the reference's own Trainer_baseline rows."""
    z = json.load(open(os.path.join(golden_dir, "trainer_baseline.json")))
    om = deeplab_ref.OracleDeepLab(_seeded_sd())
    opt = torch.optim.Adam(om.parameters(), lr=1e-3, betas=(0.9, 0.99))
    loaderS = synth_loader(z["n_batches_S"], z["B"], z["S"], z["loaderS_seed"])
    torch.manual_seed(z["torch_seed"])
    om.train()
    got = []
    for ep in range(z["epochs"]):
        for s in loaderS:
            got.append(step_ref.baseline_step(om, opt, s["image"], s["map"], s["boundary"]))
        if ep == 0:     # the reference validated here (eval forward, no RNG use)
            pass
    np.testing.assert_allclose(got, z["train_loss"], rtol=2e-4)
