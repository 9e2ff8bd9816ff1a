"""Whole-generator parity helpers shared by the GPU tests, the GPU report and smoke()."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import deeplab_ref, step_ref
from uda_clr_amd.networks.deeplabv3 import DeepLab

NAMES = ("x1", "x2", "feature", "x_bu_feature", "x_feature", "x1_before", "x2_before")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if not torch.isfinite(a).all():
        return float("inf")
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def l2rel(a, b, trim=0.01):
    """||a-b|| / ||b|| in fp64 after dropping the `trim` fraction (at least one element) of largest
    deviations: a ReLU gate whose pre-activation sits within rounding of 0 legitimately flips between
    two fp32 evaluation orders and moves ONE element of a BN-affine gradient by O(1)."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    if not torch.isfinite(a).all():
        return float("inf")
    d = (a - b).abs()
    if d.numel() > 8:
        k = max(1, int(trim * d.numel()))
        d = torch.sort(d).values[:-k]
    # a gradient that is analytically zero (e.g. a BN bias feeding another BN) is pure rounding noise in
    # every fp32 evaluation: measure it against an absolute floor instead of against ~0
    return d.norm().item() / max(b.norm().item(), 1e-5)


def grad_ok(err_hip, err_fp32_oracle, factor=10.0):
    """Per-tensor gradient criterion.  The fp64 oracle is the ground truth; the fp32 oracle (= the
    reference's own arithmetic) is itself 1e-4 .. 1e-2 away from it on this network (BN-bias gradients are
    sums with near-total cancellation, and one ReLU gate whose pre-activation sits within rounding of 0 moves
    a whole layer's gradient by ~1e-3), so the HIP path must stay within a multiple of THAT distance:
    10x per tensor (two fp32 evaluation orders of an ill-conditioned sum differ by a heavy-tailed random
    factor) + 2e-3, and ``grads_ok`` bounds the geometric mean over all tensors by 1.5 (measured on MI355X:
    0.70 at 512^2, 0.77 at 64^2 - the HIP gradients are closer to fp64 than the reference's own fp32;
    a wrong kernel moves a tensor by O(1), i.e. 100-1000x its fp32 distance)."""
    return err_hip < factor * err_fp32_oracle + 2e-3


def grads_ok(grads):
    """grads: {key: (err_hip, err_fp32_oracle)} -> (offending keys, geometric-mean ratio)."""
    import math
    bad = {k: v for k, v in grads.items() if not grad_ok(v[0], v[1])}
    # Heavy tail of the per-tensor ratio: a gradient that is an (almost) exactly cancelling sum - measured: only
    # backbone.features.17.conv.7.bias, the BatchNorm bias below the ASPP, whose contributions through the 1x1 branch cancel
    # exactly (sum over pixels of a BatchNorm input gradient) - is 1-8 % from fp64 in the fp32 ORACLE itself, and the ratio of two
    # such draws exceeds 10 for a few percent of the seeds (profiles/r02_grad_noise_seeds.txt: HIP / oracle ratios 1.6-11.5 over
    # four seeds and both matrix modes, every other tensor within 3.5).  At most two tensors whose fp32-oracle distance is itself
    # above 5e-3 may therefore sit between 10x and 30x; everything else keeps the 10x bound.
    tail = {k: v for k, v in bad.items() if v[1] > 5e-3 and grad_ok(v[0], v[1], factor=30.0)}
    if len(tail) <= 2:
        bad = {k: v for k, v in bad.items() if k not in tail}
    ratios = [math.log(max(v[0], 1e-7) / max(v[1], 1e-7)) for v in grads.values()]
    gmean = math.exp(sum(ratios) / max(len(ratios), 1))
    return bad, gmean


def frozen_grads_ok(grads):
    """Criterion for frozen-BatchNorm training passes (use ``train_parity(seed=11)``).  Without batch statistics there are no
    near-cancelling sums: on inputs without a marginal ReLU gate (seed 11) every gradient tensor of the engine sits 2e-5 from the
    fp64 oracle, like the fp32 oracle's own.  A gate within rounding of 0 flips between two fp32 evaluation orders and moves all
    gradients below it by 0.05-2 % (seeds 14, 21-23: the fp32 ORACLE itself is 5e-4 ... 2e-2 from fp64 there), a wrong term (e.g.
    the quirk-Q1 border sum) moves the tensors it feeds by 10-50 %.  Bound: every tensor < 5e-2 (measured on MI355X with a flipped
    gate: 2.4e-2), and the median < 5e-4 or within 10x the fp32 oracle's own median."""
    errs, floor = sorted(v[0] for v in grads.values()), sorted(v[1] for v in grads.values())
    assert errs[-1] < 5e-2, max(grads.items(), key=lambda kv: kv[1][0])
    med, fmed = errs[len(errs) // 2], floor[len(floor) // 2]
    assert med < max(5e-4, 10.0 * fmed), (med, fmed)


def _is_running(k, what=("running_mean", "running_var")):
    return k.rsplit(".", 1)[-1].startswith(what)          # incl. TransNorm's *_source / *_target buffers


def seeded_model(seed=1337, perturb=False, backbone="mobilenet", output_stride=16, transnorm=False):
    """transnorm=True: the --use_TN model, DeepLab(sync_bn=False) (train_use_fix_initial.py:180-181)."""
    from uda_clr_amd.networks.sync_batchnorm.batchnorm import BatchNorm2d as TransNorm2d
    torch.manual_seed(seed)
    m = DeepLab(num_classes=2, backbone=backbone, output_stride=output_stride, sync_bn=not transnorm, freeze_bn=False,
                method="prototype_full")
    if perturb:
        g = torch.Generator().manual_seed(5)
        for k, v in m.state_dict().items():
            if _is_running(k, "running_mean"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif _is_running(k, "running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, TransNorm2d)):
                mod.weight.data.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
                mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    return m


def eval_parity(dev, B=2, S=64, perturb=True, backbone="mobilenet", transnorm=False):
    """HIP eval forward vs the oracle on the same weights; returns {output: rel err}."""
    m = seeded_model(perturb=perturb, backbone=backbone, transnorm=transnorm).eval()
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(0))
    if transnorm:
        xc = torch.randn(6, 3, S, S, generator=torch.Generator().manual_seed(1))
        xc[3:] = 0.6 * xc[3:] - 0.3
        calibrate_running_stats(m, xc)
    sd = deeplab_ref.canonical_state(m.state_dict())
    with torch.no_grad():
        ref = deeplab_ref.deeplab_forward(sd, x, training=False)
        m.to(dev)
        out = m(x.to(dev))
    return {n: rel(a, b) for n, a, b in zip(NAMES, out, ref)}


def calibrate_running_stats(m, x):
    """Running statistics := batch statistics of x (one oracle training forward with momentum 1): a normalised eval network.
    (TransNorm with its initial buffers normalises nothing and doubles every layer's output, alpha = 1; rounding
    differences between two fp32 evaluations then grow ~2.3x per block.)"""
    sd = deeplab_ref.canonical_state(m.state_dict())
    keep, deeplab_ref.BN_MOMENTUM = deeplab_ref.BN_MOMENTUM, 1.0
    try:
        with torch.no_grad():
            deeplab_ref.deeplab_forward(sd, x, training=True)
    finally:
        deeplab_ref.BN_MOMENTUM = keep
    m.load_state_dict(sd, strict=False)


def train_parity(dev, B=2, S=64, extra_heads=True, backbone="mobilenet", output_stride=16, transnorm=False, frozen_bn=False,
                 engine=None, seed=3):
    """HIP training forward + backward (injected dropout masks) vs the fp64 oracle.  Returns
    (forward errs vs fp32 oracle, {param: (err vs fp64, fp32-oracle err vs fp64)}, running-stat err,
    {output: (err vs fp64, fp32-oracle err vs fp64)})."""
    m = seeded_model(perturb=True, backbone=backbone, output_stride=output_stride, transnorm=transnorm).train()
    if engine is not None:           # CPU tests: the engine bound to the torch statement of the kernels
        m._engine_override = engine
    bn_tr = None
    if frozen_bn:                    # DeepLab.freeze_bn() while training (deeplabv3.py:43-50): eval-mode BatchNorm, live dropout
        if transnorm:                # frozen TransNorm statistics must describe the activations (every layer is scaled by 1 + alpha)
            SHc, SWc = (S, S) if isinstance(S, int) else S
            xc = torch.randn(6, 3, SHc, SWc, generator=torch.Generator().manual_seed(1))
            xc[3:] = 0.6 * xc[3:] - 0.3
            calibrate_running_stats(m, xc)
        m.freeze_bn()
        bn_tr = False
    gen = torch.Generator().manual_seed(seed)
    SH, SW = (S, S) if isinstance(S, int) else S
    x = torch.randn(B, 3, SH, SW, generator=gen)
    if transnorm:
        x[B // 2:] = 0.7 * x[B // 2:] + 0.2         # the two domain halves differ in statistics
    tmap = (torch.rand(B, 2, SH, SW, generator=gen) > 0.5).float()
    tbd = torch.rand(B, 1, SH, SW, generator=gen)
    masks = deeplab_ref.draw_masks(B, SH, SW, gen)
    if output_stride == 8:          # the ASPP output (and its dropout mask) lives at 1/8 resolution
        masks["aspp.dropout"] = (torch.rand(B, 256, SH // 8, SW // 8, generator=gen) >= 0.5).to(torch.uint8)
    wf = [torch.randn(t, generator=gen) for t in (256, 304, 305, 2, 1)]
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}

    def total(outs, dt, dv):
        loss = step_ref.seg_loss(outs[0], outs[1], tmap.to(dv, dt), tbd.to(dv, dt))
        if extra_heads:
            for t, w in zip(outs[2:], wf):
                loss = loss + 1e-2 * (t * w.to(dv, dt).view(1, -1, 1, 1)).pow(2).mean()
        return loss

    o32 = deeplab_ref.canonical_state(sd0, requires_grad=True)
    r32 = deeplab_ref.deeplab_forward(o32, x, training=True, masks=masks, output_stride=output_stride, bn_training=bn_tr)
    total(r32, torch.float32, "cpu").backward()
    o64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone())
           for k, v in deeplab_ref.canonical_state(sd0, requires_grad=True).items()}
    r64 = deeplab_ref.deeplab_forward(o64, x.double(), training=True, masks=masks, output_stride=output_stride, bn_training=bn_tr)
    total(r64, torch.float64, "cpu").backward()
    m.to(dev)
    m.set_dropout_masks(masks)
    out = m(x.to(dev))
    total(out, torch.float32, dev).backward()
    fwd = {n: rel(a, b) for n, a, b in zip(NAMES, out, r32)}
    live = m._flat_state()
    grads = {}
    for k in deeplab_ref.parameter_keys(o32):
        g = live[k].grad
        grads[k] = (float("inf") if g is None else l2rel(g, o64[k].grad), l2rel(o32[k].grad, o64[k].grad))
    stats = max(rel(live[k], v) for k, v in o32.items() if _is_running(k))
    fwd64 = {n: (rel(a, c), rel(b, c)) for n, a, b, c in zip(NAMES, out, r32, r64)}
    return fwd, grads, stats, fwd64


def golden_parity(dev, tag):
    """HIP path vs the fixtures written by the reference itself (tests/golden/forward_<tag>.npz):
    seeded init, seeded input, eval outputs (checksums + samples), train loss and grad norms with the
    oracle-recovered dropout masks of the reference's own draw."""
    z = np.load(os.path.join(GOLDEN, "forward_%s.npz" % tag))
    B, S = int(z["B"]), int(z["S"])
    m = seeded_model(backbone="resnet" if tag.startswith("resnet") else "mobilenet", transnorm="tn" in tag.split("_"))
    torch.manual_seed(int(z["input_seed"]))
    x = torch.randn(B, 3, S, S)
    errs = {}
    m.to(dev).eval()
    with torch.no_grad():
        out = m(x.to(dev))
    # TransNorm with its INITIAL buffers (both domains mean 0 / var 1) normalises nothing in eval mode and doubles every
    # layer's output (alpha = 1): rounding differences between two fp32 evaluations grow ~2.3x per block (0.1 at the outputs
    # for ANY other summation order than the reference's own), so the eval half of the tn fixture pins the oracle only;
    # the eval path itself is checked on calibrated statistics (eval_parity(transnorm=True))
    for n, t in zip(NAMES, [] if "tn" in tag.split("_") else out):
        d = t.double().cpu()
        f = d.reshape(-1)
        idx = torch.linspace(0, f.numel() - 1, 97).long()
        errs["eval." + n + ".smp"] = (f[idx] - torch.from_numpy(z["eval.%s.smp" % n])).abs().max().item() / max(
            float(np.abs(z["eval.%s.smp" % n]).max()), 1e-30)
        errs["eval." + n + ".abs"] = abs(d.abs().sum().item() - float(z["eval.%s.abs" % n])) / float(z["eval.%s.abs" % n])
    # training step: the reference drew its dropout masks from torch.manual_seed(dropout_seed); the
    # oracle (pinned bit-exact to the reference) re-draws the same stream and hands the masks over
    from make_golden_inputs import synth_targets
    tmap, tbd = synth_targets(B, S, S, int(z["target_seed"]))
    m.train()
    sd0 = deeplab_ref.canonical_state({k: v.cpu() for k, v in m.state_dict().items()})
    rec = {}
    torch.manual_seed(int(z["dropout_seed"]))
    with torch.no_grad():
        deeplab_ref.deeplab_forward(sd0, x, training=True, record=rec)
    for k, v in rec.items():
        assert int(v.sum()) == int(z["mask.%s.sum" % k]), "dropout stream differs from the reference's draw"
    m.set_dropout_masks(rec)
    out = m(x.to(dev))
    bad = [n for n, t in zip(NAMES, out) if not bool(torch.isfinite(t).all())]      # (torch's BCE kernel asserts on the device otherwise)
    assert not bad, "non-finite training outputs: %s" % bad
    loss = step_ref.seg_loss(out[0], out[1], tmap.to(dev), tbd.to(dev))
    loss.backward()
    errs["train.loss"] = abs(loss.item() - float(z["train.loss"])) / abs(float(z["train.loss"]))
    for n, t in zip(NAMES, out):
        d = t.detach().double().cpu()
        errs["train." + n + ".abs"] = abs(d.abs().sum().item() - float(z["train.%s.abs" % n])) / float(z["train.%s.abs" % n])
    live = m._flat_state()
    keys = [str(k) for k in z["train.grad_keys"]]
    gn = np.array([live[k].grad.double().norm().item() for k in keys])
    rel_gn = np.abs(gn - z["train.grad_norm"]) / np.maximum(z["train.grad_norm"], 1e-12)
    conv = np.array([live[k].dim() == 4 for k in keys])
    # conv-weight gradient norms are well conditioned; BN affine gradients are near-cancelling sums
    # whose fp32 value is noise-dominated in the reference itself (see grad_ok), so only their median counts
    errs["train.grad_norm.conv"] = float(rel_gn[conv].max())
    errs["train.grad_norm.median"] = float(np.median(rel_gn))
    bs = np.array([live[k].double().sum().item() for k in z["train.bn_keys"]])
    errs["train.bn_sum"] = float(np.max(np.abs(bs - z["train.bn_sum"]) / np.maximum(np.abs(z["train.bn_sum"]), 1e-3)))
    return errs
