"""CPU check of the ORCHESTRATION in uda_clr_amd.engine (launch order, buffer windows, the
hand-written backward, quirk-Q1 bookkeeping): the engine is driven with the tests' fp32 torch
statement of the kernel entry points (tests/kernel_spec.py) and compared with the oracle.
The HIP kernels themselves are checked against the same statements in the -m gpu tests."""
import pytest
import torch

import model_cases
from kernel_spec import SpecKernels
from oracle import deeplab_ref, step_ref
from uda_clr_amd.engine import GeneratorEngine
from uda_clr_amd.networks.deeplabv3 import DeepLab

NAMES = ("x1", "x2", "feature", "x_bu_feature", "x_feature", "x1_before", "x2_before")


def _model(seed=1337, backbone="mobilenet"):
    torch.manual_seed(seed)
    m = DeepLab(num_classes=2, backbone=backbone, output_stride=16)
    m._engine_override = GeneratorEngine(SpecKernels(), backbone=backbone)
    # perturb BN affine/running stats so that every term of the BN math is exercised
    g = torch.Generator().manual_seed(5)
    for k, v in m.state_dict().items():
        if k.endswith("running_mean"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
            mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    return m


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize("size,backbone", [(64, "mobilenet"), (96, "mobilenet"), (64, "resnet"), (96, "resnet")])
def test_eval_forward_matches_oracle(size, backbone):
    m = _model(backbone=backbone).eval()
    x = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        mine = m(x)
        ref = deeplab_ref.deeplab_forward(deeplab_ref.canonical_state(m.state_dict()), x, training=False)
    for n, a, b in zip(NAMES, mine, ref):
        assert a.shape == b.shape, n
        assert _rel(a, b) < 2e-4, (n, _rel(a, b))


@pytest.mark.parametrize("backbone", ["mobilenet", "resnet"])
def test_train_forward_backward_matches_oracle(backbone):
    m = _model(backbone=backbone).train()
    B, S = 2, 64
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, S, S, generator=gen)
    tmap = (torch.rand(B, 2, S, S, generator=gen) > 0.5).float()
    tbd = torch.rand(B, 1, S, S, generator=gen)
    masks = deeplab_ref.draw_masks(B, S, S, gen)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    # extra heads so that every one of the 7 outputs receives a gradient
    wf = [torch.randn(t, generator=gen) for t in (256, 304, 305, 2, 1)]

    def total(outs):
        x1, x2, feat, xbu, xf, x1b, x2b = outs
        loss = step_ref.seg_loss(x1, x2, tmap, tbd)
        for t, w in zip((feat, xbu, xf, x1b, x2b), wf):
            loss = loss + 1e-2 * (t * w.view(1, -1, 1, 1)).pow(2).mean()
        return loss

    m.set_dropout_masks(masks)
    outs = m(x)
    loss = total(outs)
    loss.backward()
    osd = deeplab_ref.canonical_state(sd0, requires_grad=True)
    ref = deeplab_ref.deeplab_forward(osd, x, training=True, masks=masks)
    rloss = total(ref)
    rloss.backward()
    # fp64 run of the same oracle = ground truth; the fp32 oracle's own distance to it is the
    # noise floor of this (tiny, ill-conditioned: BN over 2x4x4 samples) problem
    o64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone())
           for k, v in deeplab_ref.canonical_state(sd0, requires_grad=True).items()}
    wf64 = [w.double() for w in wf]
    r64 = deeplab_ref.deeplab_forward(o64, x.double(), training=True, masks=masks)
    l64 = step_ref.seg_loss(r64[0], r64[1], tmap.double(), tbd.double())
    for t, w in zip(r64[2:], wf64):
        l64 = l64 + 1e-2 * (t * w.view(1, -1, 1, 1)).pow(2).mean()
    l64.backward()
    # outputs: as close to the fp64 truth as the fp32 oracle is (training-mode BN over 2x4x4 samples
    # through 100 layers of the ResNet amplifies fp32 rounding to 1e-3; MobileNetV2 stays at 1e-5)
    for n, a, b, c in zip(NAMES, outs, ref, r64):
        assert _rel(a, c) < 3.0 * _rel(b, c) + 2e-4, (n, _rel(a, c), _rel(b, c))
    assert abs(loss.item() - l64.item()) < 3.0 * abs(rloss.item() - l64.item()) + 1e-5 * abs(l64.item())
    live = m._flat_state()
    grads = {}
    for k in deeplab_ref.parameter_keys(osd):
        g = live[k].grad
        assert g is not None, k
        grads[k] = (model_cases.l2rel(g, o64[k].grad), model_cases.l2rel(osd[k].grad, o64[k].grad))
    bad, gmean = model_cases.grads_ok(grads)
    assert not bad, list(bad.items())[:10]
    assert gmean < 1.5, gmean          # same arithmetic as the fp32 oracle up to summation order
    for k, v in osd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert _rel(live[k], o64[k]) < 3.0 * _rel(v, o64[k]) + 5e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(live[k]) == int(v) == 1


def test_no_grad_train_forward_updates_running_stats_only():
    m = _model().train()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        outs = m(x)
    assert not outs[0].requires_grad
    assert int(m.state_dict()["aspp.bn1.num_batches_tracked"]) == 1


def test_mc_fast_path_equals_plain_stochastic_forwards():
    """GeneratorEngine.mc_forward (reuse of the deterministic pre-dropout activations + running-stat
    replay) against the plain loop of Trainer_prototype_full.py:358-368 on identical dropout masks:
    same logits, same BN running statistics, same num_batches_tracked."""
    B, S, passes = 2, 64, 2
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(B, 3, S, S, generator=gen)
    m0 = deeplab_ref.draw_masks(B, S, S, gen)
    mc_masks = [deeplab_ref.draw_masks(2 * B, S, S, gen) for _ in range(passes)]
    res = []
    for fast in (False, True):
        m = _model().train()
        m.set_dropout_masks(m0)
        outs = m(x)                                   # the grad-mode training forward on x
        if fast:
            preds = m.mc_dropout_logits(x, passes=passes, reps=2, masks=mc_masks)
        else:
            m._recent = []                            # forces the plain path
            preds = m.mc_dropout_logits(x, passes=passes, reps=2, masks=mc_masks)
        res.append((preds, {k: v.clone() for k, v in m.state_dict().items()}, outs[0].detach().clone()))
    (p0, s0, o0), (p1, s1, o1) = res
    assert _rel(p1, p0) < 1e-5
    assert torch.equal(o0, o1)
    for k in s0:
        if k.endswith("num_batches_tracked"):
            assert int(s0[k]) == int(s1[k]) == 1 + passes, k
        elif k.endswith("running_mean") or k.endswith("running_var"):
            assert _rel(s1[k], s0[k]) < 2e-5, (k, _rel(s1[k], s0[k]))


# ------------------------------------------------------------------ edge cases the reference exhibits
def test_single_image_training_batch_raises_like_the_reference():
    """quirk Q8: the ASPP image-pooling branch normalises an [N, 256, 1, 1] tensor, so N = 1 in training mode fails in
    F.batch_norm ("Expected more than 1 value per channel when training"); the product raises the same ValueError."""
    m = _model().train()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        m(torch.randn(1, 3, 64, 64))
    sd = deeplab_ref.canonical_state(m.state_dict())
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        deeplab_ref.deeplab_forward(sd, torch.randn(1, 3, 64, 64), training=True)


def test_single_image_eval_and_non_square_input():
    m = _model().eval()
    for shape in ((1, 3, 64, 64), (2, 3, 64, 128), (2, 3, 96, 48 + 16)):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(9))
        with torch.no_grad():
            mine = m(x)
            ref = deeplab_ref.deeplab_forward(deeplab_ref.canonical_state(m.state_dict()), x, training=False)
        for n, a, b in zip(NAMES, mine, ref):
            assert a.shape == b.shape, (n, shape)
            assert _rel(a, b) < 2e-4, (n, shape, _rel(a, b))


def test_input_size_must_be_a_multiple_of_16():
    m = _model().eval()
    with pytest.raises(ValueError, match="multiples of 16"):
        m(torch.randn(2, 3, 72, 64))
    with pytest.raises(ValueError, match=r"\[N, 3, H, W\]"):
        m(torch.randn(2, 1, 64, 64))


def test_output_stride_8_matches_oracle():
    """--out-stride 8 (train_use_fix_initial.py:86-90): the last two stages trade stride for dilation (mobilenet.py:93-101),
    the ASPP rates double (aspp.py:43-48)."""
    torch.manual_seed(11)
    m = DeepLab(num_classes=2, backbone="mobilenet", output_stride=8)
    m._engine_override = GeneratorEngine(SpecKernels(), output_stride=8)
    m.eval()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        mine = m(x)
        ref = deeplab_ref.deeplab_forward(deeplab_ref.canonical_state(m.state_dict()), x, training=False, output_stride=8)
    for n, a, b in zip(NAMES, mine, ref):
        assert a.shape == b.shape, n
        assert _rel(a, b) < 2e-4, (n, _rel(a, b))


def test_nonfinite_activation_is_flagged_and_the_trainer_raises(tmp_path):
    """A NaN pixel in the input: the engine's per-pass reduction over the BN statistics arena raises the device-side flag
    (forward and backward), DeepLab.pop_nonfinite hands it over once, and Trainer_baseline turns it into the reference's
    ValueError even when the loss itself stays finite."""
    from oracle_ops import OracleOps
    from uda_clr_amd.train_process import Trainer_baseline
    m = _model().train()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    m(x)
    assert m.pop_nonfinite() is not None and not bool(m._engine_override.nonfinite or False)
    xb = x.clone()
    xb[1, 2, 10, 10] = float("nan")
    m(xb)
    assert bool(m.pop_nonfinite()) and m.pop_nonfinite() is None
    tmap = (torch.rand(2, 2, 64, 64) > 0.5).float()
    loader = [{"image": xb, "map": tmap, "boundary": torch.rand(2, 1, 64, 64), "img_name": ["a", "b"]}]
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    tr = Trainer_baseline.Trainer(cuda=False, model_gen=m, optimizer_gen=opt, val_loader=loader, domain_loaderS=loader,
                                  domain_loaderT=loader, out=str(tmp_path), max_epoch=1, stop_epoch=1, interval_validate=100,
                                  batch_size=2, warmup_epoch=-1)
    tr.ops = OracleOps()
    # a loss that stays finite whatever the logits hold (on the HIP path the clamps already return finite values for NaN)
    tr.ops.seg_loss = lambda o, b, mp, bd: torch.nan_to_num(o).mean() + torch.nan_to_num(b).mean()
    with pytest.raises(ValueError, match="nan/inf"):
        tr.train_epoch()


def test_mc_fast_path_refuses_stale_activations():
    """mc_dropout_logits reuses a training forward's activations only for the SAME unmodified input under unchanged parameters:
    an in-place write to the input, note_params_changed() (what the trainers call after optimizer.step) or a mode change send
    it down the plain-forward path instead of returning logits of the old state."""
    m = _model().train()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    calls = []
    eng = m._engine_override
    orig = eng.mc_forward
    eng.mc_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    m(x)
    m.mc_dropout_logits(x, passes=1, reps=2)
    assert len(calls) == 1                                   # fresh: reused
    m(x)
    x.mul_(1.0)                                              # same address, new version
    m.mc_dropout_logits(x, passes=1, reps=2)
    assert len(calls) == 1
    m(x)
    m.note_params_changed()
    m.mc_dropout_logits(x, passes=1, reps=2)
    assert len(calls) == 1 and m._recent == []
    m(x)
    m.eval(); m.train()
    m.mc_dropout_logits(x, passes=1, reps=2)
    assert len(calls) == 1
    m(x)
    m.mc_dropout_logits(x, passes=1, reps=2)
    assert len(calls) == 2


def test_frozen_batchnorm_training_matches_oracle_and_reference_fixture(golden_dir):
    """DeepLab.freeze_bn() while training (deeplabv3.py:43-50; train_use_fix_initial.py:92-96 makes ANY --freeze-bn value True):
    running statistics normalise and stay put, dropout is live, gamma / beta / weights receive gradients - incl. the quirk-Q1
    border term with a frozen depthwise BN behind it.  (1) the oracle's frozen mode reproduces the fixture the reference's own
    DeepLab(freeze_bn=True) wrote (forward_frozen_64.npz); (2) the engine (torch statement of the kernels) against the fp64
    oracle (model_cases.frozen_grads_ok: without batch statistics the fp32 noise floor is ~1e-5, and what remains are ReLU gates
    within rounding of 0 - one flipped gate in the ASPP moves every backbone gradient by 0.05-0.5 %)."""
    import os
    import numpy as np
    from make_golden_inputs import synth_targets
    z = np.load(os.path.join(golden_dir, "forward_frozen_64.npz"))
    B, S = int(z["B"]), int(z["S"])
    m = _model()                      # perturbed exactly like the fixture's model (generator seed 5, same draw order)
    torch.manual_seed(int(z["input_seed"]))
    x = torch.randn(B, 3, S, S)
    tmap, tbd = synth_targets(B, S, S, int(z["target_seed"]))
    osd = deeplab_ref.canonical_state({k: v.clone() for k, v in m.state_dict().items()}, requires_grad=True)
    torch.manual_seed(int(z["dropout_seed"]))
    ref = deeplab_ref.deeplab_forward(osd, x, training=True, bn_training=False)
    loss_ref = step_ref.seg_loss(ref[0], ref[1], tmap, tbd)
    loss_ref.backward()
    assert abs(loss_ref.item() - float(z["train.loss"])) < 1e-6
    keys = [str(k) for k in z["train.grad_keys"]]
    np.testing.assert_allclose([osd[k].grad.double().norm().item() for k in keys], z["train.grad_norm"], rtol=1e-4)
    # (2)
    from kernel_spec import SpecKernels
    from uda_clr_amd.engine import GeneratorEngine
    fwd, grads, stats, _ = model_cases.train_parity(torch.device("cpu"), frozen_bn=True, engine=GeneratorEngine(SpecKernels()), seed=11)
    assert max(fwd.values()) < 2e-4, fwd
    assert stats == 0.0               # frozen statistics: bit-identical to where they started
    model_cases.frozen_grads_ok(grads)


def test_fused_grad_accumulation_equals_autograd_accumulation():
    """Two generator passes under one backward (source + target of a step): inside ``fused_grad_accumulation()`` the second node adds
    its gradients into ``.grad`` itself (one multi-tensor launch); the sums are AccumulateGrad's, bit for bit, and outside the scope
    nothing changes (``torch.autograd.grad`` still receives every gradient)."""
    m = _model()
    m.train()
    g = torch.Generator().manual_seed(3)
    xa, xb = torch.randn(2, 3, 64, 64, generator=g), torch.randn(2, 3, 64, 64, generator=g)
    masks = [model_cases.random_masks(2, 64, 64, seed) for seed in (11, 12)] if hasattr(model_cases, "random_masks") else None
    params = [p for p in m.parameters() if p.requires_grad]

    def loss():
        total = 0.0
        for i, x in enumerate((xa, xb)):
            if masks is not None:
                m.set_dropout_masks(masks[i])
            o = m(x)
            total = total + o[0].square().mean() + o[4].mean() + o[1].sum() * 1e-3
        return total

    m.set_dropout_seed(7)
    m._engine_override.rng_offset = 0
    loss().backward(inputs=params)
    ref = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    m._engine_override.rng_offset = 0
    calls, orig = [], torch._foreach_add_

    def spy(a, b, *args, **kw):
        calls.append(len(a))
        return orig(a, b, *args, **kw)
    torch._foreach_add_ = spy
    try:
        with m.fused_grad_accumulation():
            loss().backward(inputs=params)
    finally:
        torch._foreach_add_ = orig
    assert not m._fuse_accum and m._accum_stash is None
    assert max(calls) == len(params), "the second pass must have added all its gradients with one multi-tensor launch"
    for p, r in zip(params, ref):
        assert torch.equal(p.grad, r)
    # a second run inside a fresh scope accumulates into .grad as autograd always does
    m._engine_override.rng_offset = 0
    with m.fused_grad_accumulation():
        loss().backward(inputs=params)
    for p, r in zip(params, ref):
        assert torch.equal(p.grad, r + r)
    m._engine_override.rng_offset = 0
    got = torch.autograd.grad(loss(), params, allow_unused=True)
    assert all(a is not None for a in got)
