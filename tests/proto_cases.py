"""The product's prototype / metric front-ends (uda_clr_amd.ops, uda_clr_amd.utils.metrics) against the fixtures the
REFERENCE wrote (tests/golden/proto.npz from utils/Utils.py:108-131,159-225; metrics.json from utils/metrics.py:118-168).

One driver, two users: the -m gpu tests run it on the HIP kernels (through the C ABI), the CPU tests run it with
``ops._K`` bound to tests/kernel_spec.py - which pins the spec's proto / mc / seg_counts statements (what the per-kernel
GPU cases compare with) to the reference's outputs and to oracle/proto_ref.py / metrics_ref.py."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sample(t, n=97):
    f = t.detach().double().reshape(-1).cpu()
    return f[torch.linspace(0, f.numel() - 1, n).long()].numpy()


def gp_inputs(z):
    g = torch.Generator().manual_seed(int(z["gp.seed"]))
    B, C, h = 2, 305, 32
    feat = torch.randn(B, C, h, h, generator=g)
    hard = (torch.rand(B, 2, h, h, generator=g) > 0.6).float()
    soft = torch.rand(B, 2, h, h, generator=g)
    v = torch.randn(4, C, generator=g)
    return feat, hard, soft, v


def rt_inputs(z):
    g = torch.Generator().manual_seed(int(z["rt.seed"]))
    B, T = 1, 8
    base = 2.0 * torch.randn(B, 2, 512, 512, generator=g)
    base = torch.nn.functional.avg_pool2d(base, 9, 1, 4) * 6.0
    preds = base.repeat(T, 1, 1, 1) + 0.35 * torch.randn(T * B, 2, 512, 512, generator=g) * \
        (torch.rand(1, 2, 512, 512, generator=g) > 0.5).float()
    oT = torch.nn.functional.interpolate(base, size=(128, 128), mode="bilinear", align_corners=True).clone()
    xt = torch.randn(B, 305, 128, 128, generator=g)
    return oT, xt, preds, T, B


def _close_summary(t, z, key, rtol):
    d = t.detach().double().cpu()
    scale = max(1.0, float(z[key + ".abs"]))
    assert abs(d.sum().item() - float(z[key + ".sum"])) <= rtol * scale, key
    assert abs(d.abs().sum().item() - float(z[key + ".abs"])) <= rtol * scale, key
    np.testing.assert_allclose(sample(t), z[key + ".smp"], rtol=10 * rtol, atol=10 * rtol * float(np.abs(z[key + ".smp"]).max()),
                               err_msg=key)


def check_gen_prototype(ops, dev):
    """ops.gen_prototype on hard labels and on soft predictions (gradient into the prediction AND the feature)."""
    z = np.load(os.path.join(GOLDEN, "proto.npz"))
    feat, hard, soft, v = gp_inputs(z)
    for tag, pred in (("hard", hard), ("soft", soft)):
        with torch.no_grad():
            cents = ops.gen_prototype(pred.to(dev), feat.to(dev))
        for i, c in enumerate(cents):
            assert tuple(c.shape) == (1, 305, 1, 1)
            np.testing.assert_allclose(c.reshape(-1).cpu().numpy(), z["gp.%s.%d" % (tag, i)], rtol=1e-5, atol=1e-7,
                                       err_msg="gen_prototype %s %d" % (tag, i))
    # the label-map variant of the trainers (nearest resize folded into the weights kernel) on the same hard labels
    with torch.no_grad():
        big = hard.repeat_interleave(4, 2).repeat_interleave(4, 3)
        for i, c in enumerate(ops.gen_prototype_from_labels(big.to(dev), feat.to(dev))):
            np.testing.assert_allclose(c.reshape(-1).cpu().numpy(), z["gp.hard.%d" % i], rtol=1e-5, atol=1e-7)
    sp, ft = soft.to(dev).requires_grad_(True), feat.to(dev).requires_grad_(True)
    cents = ops.gen_prototype(sp, ft)
    sum((c.reshape(-1) * v[i].to(dev)).sum() for i, c in enumerate(cents)).backward()
    _close_summary(sp.grad, z, "gp.soft.d_pred", 2e-5)       # quirk Q6: the soft branch back-propagates into the prediction
    _close_summary(ft.grad, z, "gp.soft.d_feat", 2e-5)


def check_gen_prototype_retrify(ops, dev):
    z = np.load(os.path.join(GOLDEN, "proto.npz"))
    oT, xt, preds, T, B = rt_inputs(z)
    oT_d, xt_d = oT.to(dev).requires_grad_(True), xt.to(dev).requires_grad_(True)
    res = ops.gen_prototype_retrify(oT_d, xt_d, preds.to(dev), None, T, B)
    # the two thresholds (sigmoid > 0.75, std < 0.04) are gates: a pixel within rounding of one may legitimately flip between
    # two fp32 evaluations and would move a centroid by ~1/count - the fixture's seeds have none, so the masks must be EXACT
    for n, r in zip(("mask_0", "mask_1"), res[5:]):
        assert tuple(r.shape) == (B, 1, 128, 128)
        assert r.double().sum().item() == float(z["rt.%s.sum" % n]), "%s: a reliability gate flipped" % n
        np.testing.assert_array_equal(sample(r), z["rt.%s.smp" % n])
    _close_summary(res[4], z, "rt.std_map", 1e-5)
    for n, r in zip(("c0_obj", "c1_obj", "c0_bck", "c1_bck"), res[:4]):
        np.testing.assert_allclose(r.detach().reshape(-1).cpu().numpy(), z["rt." + n], rtol=1e-5, atol=1e-7, err_msg=n)
    sum(r.sum() for r in res[:4]).backward()
    got = xt_d.grad.double().abs().sum().item()
    assert abs(got - float(z["rt.grad_xt.abs"])) <= 1e-5 * float(z["rt.grad_xt.abs"])
    assert float(z["rt.grad_oT.abs"]) == 0.0 and (oT_d.grad is None or float(oT_d.grad.abs().sum()) == 0.0)      # quirk Q6


def check_metrics(metrics, dev):
    from make_golden_inputs import synth_targets
    z = json.load(open(os.path.join(GOLDEN, "metrics.json")))
    g = torch.Generator().manual_seed(z["logit_seed"])
    tmap, _ = synth_targets(z["B"], z["S"], z["S"], z["target_seed"])
    logits = (tmap * 2 - 1) * 2.0 + 1.5 * torch.randn(z["B"], 2, z["S"], z["S"], generator=g)
    # integer pixel counts -> closed forms: exact (the fixture has no logit within rounding of the 0.75 gate)
    assert list(metrics.dice_coeff_2label(logits.to(dev), tmap.to(dev))) == z["dice"]
    assert list(metrics.pixel_acc(logits.to(dev), tmap.to(dev))) == z["pixel_acc"]
