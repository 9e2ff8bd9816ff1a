"""-m gpu: the whole generator (one autograd node on the HIP kernels) against the oracle and
against the fixtures the reference itself produced."""
import pytest
import torch

import model_cases

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("size", [64, 96])
def test_eval_forward_matches_oracle(size):
    errs = model_cases.eval_parity(DEV, 2, size)
    assert max(errs.values()) < 1e-3, errs          # north_star tolerance: 1e-3 relative fp32


@pytest.mark.parametrize("size", [64, 512])
def test_train_forward_backward_matches_oracle(size):
    """B = 2 at 64^2 (deepest BNs over 32 samples) and at the training resolution 512^2."""
    fwd, grads, stats, _ = model_cases.train_parity(DEV, S=size)
    assert max(fwd.values()) < 1e-3, fwd
    assert stats < 1e-3
    # gradients (L2): every tensor within 10x the fp32 oracle's own distance to the fp64 oracle, and on (geometric) average
    # no further from fp64 than 1.5x the reference's own fp32 arithmetic (measured: 0.77 / 0.70)
    bad, gmean = model_cases.grads_ok(grads)
    print("gradient noise vs the fp32 oracle's: geometric mean %.3f over %d tensors" % (gmean, len(grads)))
    assert not bad, list(bad.items())[:10]
    assert gmean < 1.5, gmean


def test_padding_columns_never_leak(monkeypatch):
    """Every fp32 work matrix of the engine starts as NaN / +-Inf / 3e38 (its [P, round4(C)] padding columns are never written):
    outputs and gradients must be the same as on clean buffers.  (A head kernel that relied on zero WEIGHTS to cancel the
    padding lanes produced Inf * 0 = NaN logits whenever the allocator handed it a block with large values in it.)"""
    from uda_clr_amd import engine
    monkeypatch.setattr(engine, "POISON_BUFFERS", True)
    fwd, grads, stats, _ = model_cases.train_parity(DEV, S=96)
    assert max(fwd.values()) < 1e-3, fwd
    assert stats < 1e-3
    bad, gmean = model_cases.grads_ok(grads)
    assert not bad, list(bad.items())[:10]


@pytest.mark.parametrize("tag", ["64", "512"])
def test_matches_reference_fixtures(tag):
    errs = model_cases.golden_parity(DEV, tag)
    tol = {"train.grad_norm.conv": 5e-2, "train.grad_norm.median": 2e-2, "train.bn_sum": 1e-3}
    for k, v in errs.items():
        assert v < tol.get(k, 1e-3), (k, v)


# ---- ResNet-101 variant (BASELINE.json configs[4])
@pytest.mark.parametrize("size", [64, 96])
def test_resnet_eval_forward_matches_oracle(size):
    errs = model_cases.eval_parity(DEV, 2, size, backbone="resnet")
    assert max(errs.values()) < 1e-3, errs


def test_resnet_train_forward_backward_matches_oracle():
    """Training-mode BN over 2x4x4 samples through 100 layers amplifies fp32 rounding: the fp32 oracle
    itself sits 1e-3 from its fp64 run on this case, so outputs are held to 3x THAT distance."""
    fwd, grads, stats, fwd64 = model_cases.train_parity(DEV, backbone="resnet")
    for n, (e, floor) in fwd64.items():
        assert e < 3.0 * floor + 2e-4, (n, e, floor)
    assert stats < 5e-3
    bad, gmean = model_cases.grads_ok(grads)
    print("gradient noise vs the fp32 oracle's: geometric mean %.3f over %d tensors" % (gmean, len(grads)))
    assert not bad, list(bad.items())[:10]
    assert gmean < 4.0, gmean


@pytest.mark.parametrize("tag", ["resnet_128", "resnet_256"])
def test_resnet_matches_reference_fixtures(tag):
    """Fixtures written by the reference's own DeepLab(backbone='resnet').  A plain fp32 torch evaluation of
    the same graph in a different summation order already sits 1e-2 (gradient norms) / 2e-3 (running
    statistics) from them: 101 layers of training-mode BN amplify rounding."""
    errs = model_cases.golden_parity(DEV, tag)
    tol = {"train.grad_norm.conv": 5e-2, "train.grad_norm.median": 2e-2, "train.bn_sum": 5e-3}
    for k, v in errs.items():
        assert v < tol.get(k, 5e-3 if k.startswith("train.") else 1e-3), (k, v)


# ---- TransNorm variant (--use_TN): per-domain-half launches of the same kernels
def test_transnorm_eval_forward_matches_oracle():
    errs = model_cases.eval_parity(DEV, 3, 64, transnorm=True)
    assert max(errs.values()) < 1e-3, errs


@pytest.mark.parametrize("B", [8, 7])
def test_transnorm_train_forward_backward_matches_oracle(B):
    """Halves of 4 + 4 and of 3 + 4 images.  Outputs are held to 3x the fp32 oracle's own distance from its fp64 run (the
    image-pooling TransNorm normalises 3-4 values per channel and half)."""
    fwd, grads, stats, fwd64 = model_cases.train_parity(DEV, B=B, transnorm=True)
    for n, (e, floor) in fwd64.items():
        assert e < 3.0 * floor + 2e-4, (n, e, floor)
    assert stats < 2e-3
    bad, gmean = model_cases.grads_ok(grads)
    print("gradient noise vs the fp32 oracle's: geometric mean %.3f over %d tensors" % (gmean, len(grads)))
    assert not bad, list(bad.items())[:10]
    assert gmean < 4.0, gmean


def test_transnorm_matches_reference_fixture():
    """Fixture written by the reference's own DeepLab(sync_bn=False) at B = 4 (two images per domain half: the reference's
    fp32 arithmetic is itself 6e-2 from an fp64 evaluation on the BN-affine gradients there, see tests/test_transnorm_cpu.py),
    so the eval outputs and the training loss / outputs / running statistics carry the comparison."""
    errs = model_cases.golden_parity(DEV, "tn_64")
    tol = {"train.grad_norm.conv": 0.2, "train.grad_norm.median": 0.2, "train.bn_sum": 2e-3}
    for k, v in errs.items():
        assert v < tol.get(k, 2e-3 if k.startswith("train.") else 1e-3), (k, v)


def test_transnorm_mc_fast_path_equals_plain_stochastic_forwards():
    """HIP: the repeated batch's deterministic part once on x (repeat_prefix) + the stochastic tail per pass, against the oracle's
    plain training-mode forwards on x.repeat(2): logits, every running buffer, num_batches_tracked."""
    from oracle import deeplab_ref
    B, S, passes = 4, 64, 2
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(B, 3, S, S, generator=gen)
    m = model_cases.seeded_model(perturb=True, transnorm=True).train()
    m0 = deeplab_ref.draw_masks(B, S, S, gen)
    mk = [deeplab_ref.draw_masks(2 * B, S, S, gen) for _ in range(passes)]
    sd1 = deeplab_ref.canonical_state(m.state_dict())
    with torch.no_grad():
        deeplab_ref.deeplab_forward(sd1, x, training=True, masks=m0)            # the grad-mode forward's effect on the buffers
        ref = torch.cat([deeplab_ref.deeplab_forward(sd1, x.repeat(2, 1, 1, 1), training=True, masks=mk[ps])[0] for ps in range(passes)], 0)
    m.to(DEV)
    m.set_dropout_masks(m0)
    m(x.to(DEV))
    preds = m.mc_dropout_logits(x.to(DEV), passes=passes, reps=2, masks=mk)
    assert model_cases.rel(preds, ref) < 1e-3
    live = m.state_dict()
    for k, v in sd1.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf.startswith(("running_mean", "running_var")):
            assert model_cases.rel(live[k], v) < 2e-3, k
        elif leaf == "num_batches_tracked":
            assert int(live[k]) == int(v) == 1 + passes, k


def test_no_grad_and_determinism():
    m = model_cases.seeded_model().to(DEV).train()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    from oracle import deeplab_ref
    masks = deeplab_ref.draw_masks(2, 64, 64, torch.Generator().manual_seed(2))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    outs = []
    for _ in range(2):
        m.load_state_dict(sd)
        m.set_dropout_masks(masks)
        with torch.no_grad():
            outs.append([t.clone() for t in m(x)])
    for a, b in zip(*outs):
        assert torch.equal(a, b), "the forward path must be bitwise reproducible"
    assert not outs[0][0].requires_grad


def test_mc_fast_path_equals_plain_stochastic_forwards():
    """HIP: GeneratorEngine.mc_forward vs the plain loop of Trainer_prototype_full.py:358-368 on identical masks."""
    from oracle import deeplab_ref
    B, S, passes = 2, 64, 2
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(B, 3, S, S, generator=gen).to(DEV)
    m0 = deeplab_ref.draw_masks(B, S, S, gen)
    mc_masks = [deeplab_ref.draw_masks(2 * B, S, S, gen) for _ in range(passes)]
    res = []
    for fast in (False, True):
        m = model_cases.seeded_model(perturb=True).to(DEV).train()
        m.set_dropout_masks(m0)
        m(x)
        if not fast:
            m._recent = []
        preds = m.mc_dropout_logits(x, passes=passes, reps=2, masks=mc_masks)
        res.append((preds, {k: v.clone() for k, v in m.state_dict().items()}))
    (p0, s0), (p1, s1) = res
    assert model_cases.rel(p1, p0) < 1e-4
    for k in s0:
        if k.endswith("num_batches_tracked"):
            assert int(s0[k]) == int(s1[k]) == 1 + passes
        elif k.endswith("running_mean") or k.endswith("running_var"):
            assert model_cases.rel(s1[k], s0[k]) < 1e-4, k


# ---- edge cases
def test_non_square_and_single_image_eval():
    from oracle import deeplab_ref
    m = model_cases.seeded_model(perturb=True).eval()
    sd = deeplab_ref.canonical_state(m.state_dict())
    m.to(DEV)
    for shape in ((1, 3, 64, 64), (2, 3, 64, 128), (2, 3, 96, 64), (3, 3, 160, 32)):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(9))
        with torch.no_grad():
            ref = deeplab_ref.deeplab_forward(sd, x, training=False)
            out = m(x.to(DEV))
        for n, a, b in zip(model_cases.NAMES, out, ref):
            assert a.shape == b.shape, (n, shape)
            assert model_cases.rel(a, b) < 1e-3, (n, shape)


def test_single_image_training_batch_raises_like_the_reference():
    m = model_cases.seeded_model().to(DEV).train()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        m(torch.randn(1, 3, 64, 64, device=DEV))


def test_odd_training_batch_and_non_square_backward():
    """B = 3, 64 x 96: forward + backward against the fp64 oracle (ragged pixel counts: P is no multiple of any tile)."""
    fwd, grads, stats, fwd64 = model_cases.train_parity(DEV, B=3, S=(64, 96))
    assert max(fwd.values()) < 1e-3, fwd
    bad, gmean = model_cases.grads_ok(grads)
    print("gradient noise vs the fp32 oracle's: geometric mean %.3f over %d tensors" % (gmean, len(grads)))
    assert not bad, list(bad.items())[:10]
    assert gmean < 4.0, gmean


def test_full_size_eval_is_batch_independent():
    """BASELINE size (16 x 3 x 512 x 512) through a size-independent property: in eval mode every image is processed
    independently, so image k of the 16-batch (256-row workgroup tiles, 8 math waves) must equal the same image run in a
    batch of 2 (128-row tiles) - two different kernel configurations of every wide convolution against each other."""
    m = model_cases.seeded_model(perturb=True).to(DEV).eval()
    x = torch.randn(16, 3, 512, 512, generator=torch.Generator().manual_seed(12)).to(DEV)
    with torch.no_grad():
        big = [t.clone() for t in m(x)]
        for k in (0, 14):
            small = m(x[k:k + 2])
            for n, a, b in zip(model_cases.NAMES, big, small):
                assert model_cases.rel(a[k:k + 2], b) < 1e-4, (n, k)


def test_full_size_backward_agrees_with_forward_differences():
    """BASELINE size (B = 16, 512 x 512, training mode, fixed dropout masks): the directional derivative of the
    segmentation loss along the computed gradient must equal the central difference of two more forwards - ties the whole
    hand-written backward (256-row dgrad tiles, large-P weight-gradient splits, depthwise / BN backward) to the forward path."""
    from oracle import deeplab_ref
    import make_golden_inputs
    B, S = 16, 512
    m = model_cases.seeded_model(perturb=True).to(DEV).train()
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(B, 3, S, S, generator=gen).to(DEV)
    tmap, tbd = make_golden_inputs.synth_targets(B, S, S, 22)
    tmap, tbd = tmap.to(DEV), tbd.to(DEV)
    masks = deeplab_ref.draw_masks(B, S, S, gen)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    from uda_clr_amd import ops

    def loss_at(shift=None):
        m.load_state_dict(sd0)
        if shift is not None:
            with torch.no_grad():
                for p, d in zip(params, shift):
                    p.add_(d)
        m.set_dropout_masks(masks)
        out = m(x)
        return ops.seg_loss(out[0], out[1], tmap, tbd)

    params = [p for p in m.parameters() if p.requires_grad]
    loss = loss_at()
    loss.backward()
    grads = [p.grad.detach().clone() for p in params]
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).item()
    res = {}
    for eps in (1e-3, 2.5e-4):            # the loss is strongly curved along its own gradient: the difference quotient
        step = [g * (eps / gnorm) for g in grads]      # must approach the analytic slope as the step shrinks
        with torch.no_grad():
            lp = loss_at(step).item()
            lm = loss_at([-s for s in step]).item()
        res[eps] = (lp - lm) / (2 * eps)
    an = gnorm
    assert abs(res[2.5e-4] - an) < 0.02 * an, (res, an)
    assert abs(res[2.5e-4] - an) <= abs(res[1e-3] - an) + 0.005 * an, (res, an)


def test_output_stride_8_forward_backward():
    """--out-stride 8: dilated last stages (dilation 2 and 4 depthwise convs, ASPP rates 12/24/36)."""
    fwd, grads, stats, _ = model_cases.train_parity(DEV, B=2, S=64, output_stride=8)
    assert max(fwd.values()) < 1e-3, fwd
    assert stats < 1e-3
    bad, gmean = model_cases.grads_ok(grads)
    print("gradient noise vs the fp32 oracle's: geometric mean %.3f over %d tensors" % (gmean, len(grads)))
    assert not bad, list(bad.items())[:10]
    assert gmean < 4.0, gmean


def test_nan_in_the_input_raises_in_the_trainer_although_the_clamps_swallow_it(tmp_path):
    """The fused BN + activation prologues clamp with v_med3_f32, which maps NaN to a finite bound (torch.relu would propagate
    it), so a diverged activation need not reach the loss.  The engine reduces every pass's BN-statistics arena to a
    device-side non-finite flag; Trainer_baseline fetches it with its one host sync and raises the reference's ValueError
    (Trainer_baseline.py:210-212 checks the loss for NaN)."""
    from make_golden_inputs import synth_loader
    from uda_clr_amd.train_process import Trainer_baseline
    m = model_cases.seeded_model().to(DEV).train()
    loader = synth_loader(1, 2, 64, 5)
    m(loader[0]["image"].to(DEV))
    assert not bool(m.pop_nonfinite())
    loader[0]["image"][1, 2, 10, 10] = float("nan")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    tr = Trainer_baseline.Trainer(cuda=True, model_gen=m, optimizer_gen=opt, val_loader=loader, domain_loaderS=loader,
                                  domain_loaderT=loader, out=str(tmp_path), max_epoch=1, stop_epoch=1, interval_validate=100,
                                  batch_size=2, warmup_epoch=-1)
    with pytest.raises(ValueError, match="nan"):
        tr.train_epoch()


def test_frozen_batchnorm_training_matches_oracle_and_reference_fixture():
    """DeepLab.freeze_bn() while training (deeplabv3.py:43-50) on the HIP kernels: eval-mode BatchNorm backward
    (uda_bnbwd_finalize with count = inf, the quirk-Q1 border total handed in), live dropout; against the fp64 oracle and
    against the fixture the reference's own DeepLab(freeze_bn=True) wrote (forward_frozen_64.npz)."""
    import os
    import numpy as np
    from make_golden_inputs import synth_targets
    from oracle import deeplab_ref, step_ref
    fwd, grads, stats, _ = model_cases.train_parity(DEV, frozen_bn=True, seed=11)
    assert max(fwd.values()) < 1e-3, fwd
    assert stats == 0.0
    # the kernels of this path are checked exactly in tests/kernel_cases.py ("bn frozen ..."), the engine's use of them on a seed
    # without marginal ReLU gates in tests/test_engine_cpu.py (2e-5 on every tensor).  Here a gate within rounding of 0 may flip
    # between the HIP and the fp64 evaluation and move every gradient below it by ~1 % (measured: median 0.9 %, worst 2.4 %; the
    # fp32 oracle shows the same on other seeds): sanity bounds only - a wrong term would be 10-50 % on the tensors it feeds
    errs = sorted(v[0] for v in grads.values())
    assert errs[-1] < 8e-2 and errs[len(errs) // 2] < 3e-2, (errs[-1], errs[len(errs) // 2])
    z = np.load(os.path.join(model_cases.GOLDEN, "forward_frozen_64.npz"))
    B, S = int(z["B"]), int(z["S"])
    m = model_cases.seeded_model(perturb=True).train()       # the fixture's seeded perturbation (generator seed 5, same draw order)
    m.freeze_bn()
    torch.manual_seed(int(z["input_seed"]))
    x = torch.randn(B, 3, S, S)
    tmap, tbd = synth_targets(B, S, S, int(z["target_seed"]))
    rec = {}
    torch.manual_seed(int(z["dropout_seed"]))
    with torch.no_grad():
        deeplab_ref.deeplab_forward(deeplab_ref.canonical_state({k: v.clone() for k, v in m.state_dict().items()}), x, training=True,
                                    record=rec, bn_training=False)
    m.to(DEV)
    m.set_dropout_masks(rec)
    out = m(x.to(DEV))
    loss = step_ref.seg_loss(out[0], out[1], tmap.to(DEV), tbd.to(DEV))
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-3 * float(z["train.loss"])
    for n, t in zip(model_cases.NAMES, out):
        a = t.detach().double().abs().sum().item()
        assert abs(a - float(z["train.%s.abs" % n])) < 1e-3 * float(z["train.%s.abs" % n]), n
    live = m._flat_state()
    gn = np.array([live[str(k)].grad.double().norm().item() for k in z["train.grad_keys"]])
    rel_gn = np.abs(gn - z["train.grad_norm"]) / np.maximum(z["train.grad_norm"], 1e-12)
    # gradient NORMS against the reference's fixture: the same marginal-gate effect (measured 0.9 % worst / 7e-4 median in
    # bf16x3 mode, 0.5 % / 3e-4 in f32 mode); a missing term of the frozen-BN backward moves norms by tens of percent
    assert rel_gn.max() < 3e-2 and np.median(rel_gn) < 5e-3, (rel_gn.max(), np.median(rel_gn))


def test_frozen_transnorm_training_matches_oracle_and_reference_fixture():
    """DeepLab(sync_bn=False).freeze_bn() while training on the HIP kernels (deeplabv3.py:47-50 evals the TransNorm layers too): the
    target running statistics normalise, the gain 1 + alpha of both domains' running statistics is a constant of the pass; against
    the fp64 oracle and against the fixture the reference's own DeepLab(sync_bn=False, freeze_bn=True) wrote on calibrated
    statistics (forward_frozen_tn_64.npz)."""
    import os
    import numpy as np
    from make_golden_inputs import synth_targets
    from oracle import deeplab_ref, step_ref
    fwd, grads, stats, _ = model_cases.train_parity(DEV, B=4, transnorm=True, frozen_bn=True, seed=11)
    assert max(fwd.values()) < 1e-3, fwd
    assert stats == 0.0
    errs = sorted(v[0] for v in grads.values())
    assert errs[-1] < 8e-2 and errs[len(errs) // 2] < 3e-2, (errs[-1], errs[len(errs) // 2])      # bounds: the plain-BN frozen test above
    z = np.load(os.path.join(model_cases.GOLDEN, "forward_frozen_tn_64.npz"))
    B, S = int(z["B"]), int(z["S"])
    m = model_cases.seeded_model(perturb=True, transnorm=True).train()
    xc = torch.randn(6, 3, S, S, generator=torch.Generator().manual_seed(int(z["calibration_seed"])))
    xc[3:] = 0.6 * xc[3:] - 0.3
    model_cases.calibrate_running_stats(m, xc)
    m.freeze_bn()
    torch.manual_seed(int(z["input_seed"]))
    x = torch.randn(B, 3, S, S)
    tmap, tbd = synth_targets(B, S, S, int(z["target_seed"]))
    rec = {}
    torch.manual_seed(int(z["dropout_seed"]))
    with torch.no_grad():
        deeplab_ref.deeplab_forward(deeplab_ref.canonical_state({k: v.clone() for k, v in m.state_dict().items()}), x, training=True,
                                    record=rec, bn_training=False)
    m.to(DEV)
    m.set_dropout_masks(rec)
    out = m(x.to(DEV))
    loss = step_ref.seg_loss(out[0], out[1], tmap.to(DEV), tbd.to(DEV))
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-3 * float(z["train.loss"])
    for n, t in zip(model_cases.NAMES, out):
        a = t.detach().double().abs().sum().item()
        assert abs(a - float(z["train.%s.abs" % n])) < 1e-3 * float(z["train.%s.abs" % n]), n
    live = m._flat_state()
    gn = np.array([live[str(k)].grad.double().norm().item() for k in z["train.grad_keys"]])
    rel_gn = np.abs(gn - z["train.grad_norm"]) / np.maximum(z["train.grad_norm"], 1e-12)
    print("frozen TransNorm gradient norms vs the reference fixture: worst %.2e median %.2e" % (rel_gn.max(), np.median(rel_gn)))
    assert rel_gn.max() < 3e-2 and np.median(rel_gn) < 5e-3, (rel_gn.max(), np.median(rel_gn))


def test_resnet_transnorm_on_the_hip_kernels():
    """DeepLab(backbone='resnet', sync_bn=False) (deeplabv3.py:17-23): per-domain-half launches of the ResNet-101 sequence on
    the HIP kernels against the fp64 oracle (halves of 2 + 2 images), and against the fixture the reference's own model wrote
    (forward_resnet_tn_128.npz: training loss, outputs, running statistics; gradient norms in the median)."""
    fwd, grads, stats, fwd64 = model_cases.train_parity(DEV, B=4, S=64, backbone="resnet", transnorm=True)
    for n, (e, floor) in fwd64.items():
        assert e < 3.0 * floor + 2e-4, (n, e, floor)
    assert stats < 5e-3
    bad, gmean = model_cases.grads_ok(grads)
    print("gradient noise vs the fp32 oracle's: geometric mean %.3f over %d tensors" % (gmean, len(grads)))
    assert not bad, list(bad.items())[:10]
    assert gmean < 4.0, gmean
    errs = model_cases.golden_parity(DEV, "resnet_tn_128")
    tol = {"train.grad_norm.conv": 0.2, "train.grad_norm.median": 0.2, "train.bn_sum": 1e-2}
    for k, v in errs.items():
        assert v < tol.get(k, 5e-3), (k, v)


# ---- BASELINE.json configs[3]: the per-GPU batch of 32 at 512 x 512 (the doubled MC batch has P = 2 * 32 * 128^2 = 1,048,576 rows)
def test_per_gpu_batch_32_at_512_properties():
    """Size-independent properties at the full configs[3] shape (no CPU oracle runs at this size):
      * eval forward: batch independence - images 5..8 of the batch of 32 alone give the same outputs;
      * training forward + backward: permutation equivariance - permuting the images (and their injected dropout masks)
        permutes the outputs and leaves the parameter gradients and the running statistics unchanged (batch statistics and
        weight gradients are sums over the batch);
      * the MC fast path on the doubled batch (one stochastic pass on 64 images) equals the plain training-mode forward on
        x.repeat(2) with the same masks."""
    B, S = 32, 512
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B, 3, S, S, generator=g, device=DEV)
    keep = lambda shp, p: (torch.rand(shp, generator=g, device=DEV) >= p).to(torch.uint8)
    sites = {"aspp.dropout": ((256, S // 16, S // 16), 0.5), "decoder.last_conv_boundary.3": ((256, S // 4, S // 4), 0.5),
             "decoder.last_conv_boundary.7": ((256, S // 4, S // 4), 0.1), "decoder.last_conv.2": ((305, S // 4, S // 4), 0.1)}
    masks = {k: keep((B,) + shp, p) for k, (shp, p) in sites.items()}
    m = model_cases.seeded_model(perturb=True).to(DEV)
    # --- eval: batch independence
    m.eval()
    with torch.no_grad():
        full = m(x)
        part = m(x[4:8].contiguous())
    for n, a, b in zip(model_cases.NAMES, full, part):
        assert model_cases.rel(a[4:8], b) < 1e-5, n
    del full, part
    # --- training forward + backward: permutation equivariance
    tmap = (torch.rand(B, 2, S, S, generator=g, device=DEV) > 0.5).float()
    tbd = torch.rand(B, 1, S, S, generator=g, device=DEV)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    runs = []
    from uda_clr_amd import ops
    for p in (None, perm):
        m.load_state_dict(sd0)
        m.train()
        for q in m.parameters():
            q.grad = None
        sel = (lambda t: t) if p is None else (lambda t: t[p].contiguous())
        m.set_dropout_masks({k: sel(v) for k, v in masks.items()})
        out = m(sel(x))
        ops.seg_loss(out[0], out[1], sel(tmap), sel(tbd)).backward()
        runs.append((out[0].detach(), {k: q.grad.clone() for k, q in m.named_parameters()},
                     {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
        del out
    (o0, g0, r0), (o1, g1, r1) = runs
    assert model_cases.rel(o1, o0[perm]) < 1e-4
    for k in r0:
        assert model_cases.rel(r1[k], r0[k]) < 1e-5, k
    worst = max((model_cases.l2rel(g1[k], g0[k]), k) for k in g0)
    assert worst[0] < 2e-3, worst           # two summation orders of the same batch sums (BN-affine gradients are near-cancelling sums)
    del runs, o0, o1, g0, g1
    # --- MC fast path on the doubled batch
    m.load_state_dict(sd0)
    m.train()
    mc = [{k: keep((2 * B,) + shp, p) for k, (shp, p) in sites.items()}]
    m.set_dropout_masks(masks)
    m(x)
    fast = m.mc_dropout_logits(x, passes=1, reps=2, masks=mc)
    s_fast = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    m.load_state_dict(sd0)
    m.set_dropout_masks(masks)
    with torch.no_grad():
        m(x)
        m._recent = []
        plain = m.mc_dropout_logits(x, passes=1, reps=2, masks=mc)
    assert model_cases.rel(fast, plain) < 1e-4
    for k, v in m.state_dict().items():
        if "running" in k:
            assert model_cases.rel(s_fast[k], v) < 1e-4, k


# ---- BASELINE.json configs[4]: the ResNet-101 variant at its full per-GPU shape, 8 images of 512 x 512
def test_resnet101_per_gpu_batch_8_at_512_properties():
    """Size-independent properties at the full configs[4] shape (the CPU oracle runs the 128^2 / 256^2 fixtures, not this):
      * eval forward: batch independence - images 2..3 of the batch of 8 alone give the same outputs;
      * training forward + backward: permuting the images (and their injected dropout masks) permutes the outputs and leaves the
        parameter gradients and the running statistics unchanged (batch statistics and weight gradients are sums over the batch);
      * gradients are finite and every parameter receives one."""
    B, S = 8, 512
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(B, 3, S, S, generator=g, device=DEV)
    keep = lambda shp, p: (torch.rand(shp, generator=g, device=DEV) >= p).to(torch.uint8)
    sites = {"aspp.dropout": ((256, S // 16, S // 16), 0.5), "decoder.last_conv_boundary.3": ((256, S // 4, S // 4), 0.5),
             "decoder.last_conv_boundary.7": ((256, S // 4, S // 4), 0.1), "decoder.last_conv.2": ((305, S // 4, S // 4), 0.1)}
    masks = {k: keep((B,) + shp, p) for k, (shp, p) in sites.items()}
    m = model_cases.seeded_model(perturb=True, backbone="resnet").to(DEV)
    m.eval()
    with torch.no_grad():
        full = m(x)
        part = m(x[2:4].contiguous())
    for n, a, b in zip(model_cases.NAMES, full, part):
        assert model_cases.rel(a[2:4], b) < 1e-5, n
    del full, part
    tmap = (torch.rand(B, 2, S, S, generator=g, device=DEV) > 0.5).float()
    tbd = torch.rand(B, 1, S, S, generator=g, device=DEV)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    runs = []
    from uda_clr_amd import ops
    for p in (None, perm):
        m.load_state_dict(sd0)
        m.train()
        for q in m.parameters():
            q.grad = None
        sel = (lambda t: t) if p is None else (lambda t: t[p].contiguous())
        m.set_dropout_masks({k: sel(v) for k, v in masks.items()})
        out = m(sel(x))
        ops.seg_loss(out[0], out[1], sel(tmap), sel(tbd)).backward()
        runs.append((out[0].detach(), {k: q.grad.clone() for k, q in m.named_parameters()},
                     {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
        del out
    (o0, g0, r0), (o1, g1, r1) = runs
    assert all(bool(torch.isfinite(v).all()) for v in g0.values()) and len(g0) == len(list(m.parameters()))
    assert model_cases.rel(o1, o0[perm]) < 2e-4
    for k in r0:      # (two summation orders of the batch sums below 101 BN layers: measured 5e-5 at the ASPP, 1e-5 holds for MobileNetV2)
        assert model_cases.rel(r1[k], r0[k]) < 3e-4, k
    # two summation orders of the same batch sums through 101 layers of training-mode BN: every tensor moves by ~1e-3 (measured:
    # median 1.3e-3, worst 2.6e-3 - the level at which the fp32 oracle itself sits from its fp64 run on this network);
    # a wrong batch-sum term would be O(1) on the tensors it feeds
    errs = sorted((model_cases.l2rel(g1[k], g0[k]), k) for k in g0)
    print("resnet-101 B=8 512^2: permutation test, median %.2e worst %s" % (errs[len(errs) // 2][0], errs[-1]))
    assert errs[len(errs) // 2][0] < 5e-3 and errs[-1][0] < 2e-2, (errs[len(errs) // 2], errs[-1])
