"""-m gpu: BASELINE.json configs[3] as a WORKLOAD - the per-GPU batch of 32 source + 32 target images at 512 x 512 - through the
prototype kernels and one full Trainer_prototype_full.train_step (Trainer_prototype_full.py:358-398, utils/Utils.py:159-225).
No CPU oracle runs at this size; the checks are size-independent properties plus the torch statement of the kernels
(tests/kernel_spec.py) evaluated on the device in fp64:

  * mc_stats on [8 * 32, 2, 512, 512]: a chunked evaluation (8 images at a time) is bit-identical, values equal torch's;
  * proto_weights / proto_reduce / proto_finalize / proto_bwd at P = 32 * 128^2: equal to the fp64 statement; centroids are
    invariant under a permutation of the images, sums are additive over image chunks (what the data-parallel all-reduce relies on);
  * one full train_step at 32 + 32 (MC fast path on the doubled batch of 64, conv operands above the 2^29-element launch limit
    run as image groups): permuting the images of both domains (and their dropout masks) leaves the six logged losses, the
    centroids and the generator / discriminator updates unchanged.
"""
import pytest
import torch

import model_cases
from kernel_spec import SpecKernels
from uda_clr_amd import ops
from uda_clr_amd.networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator
from uda_clr_amd.train_process import Trainer_prototype_full

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
B, S, T = 32, 512, 8


def _synthetic_mc_predictions(g):
    """[T * B, 2, S, S] logits: a smooth per-image field + per-pass noise whose amplitude varies over the image, so that the
    std < 0.04 gate and the sigmoid > 0.75 pseudo label both take both values on a good share of the pixels."""
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S, device=DEV), torch.linspace(-1, 1, S, device=DEV), indexing="ij")
    cx = torch.rand(B, 2, 1, 1, generator=g, device=DEV) - 0.5
    base = 4.0 - 9.0 * ((yy - cx[:, :1]) ** 2 + (xx - cx[:, 1:]) ** 2)          # [B, 1, S, S]
    base = torch.cat([base - 1.5, base], 1)                                       # cup inside disc
    amp = 0.02 + 0.5 * (xx > 0).float()                                           # quiet half, noisy half
    noise = torch.randn(T, B, 2, S, S, generator=g, device=DEV) * amp
    return (base[None] + noise).reshape(T * B, 2, S, S).contiguous()


def test_prototype_kernels_at_per_gpu_batch_32():
    K, spec = ops.kernels(), SpecKernels()
    g = torch.Generator(device=DEV).manual_seed(11)
    preds = _synthetic_mc_predictions(g)
    std, mean = K.mc_stats(preds, T)
    assert std.shape == (B, 2, S, S)
    # chunked evaluation: the statistics are per pixel over the T passes, so any subset of images gives the same bits
    p5 = preds.reshape(T, B, 2, S, S)
    for b0 in (0, 8, 24):
        s_c, m_c = K.mc_stats(p5[:, b0:b0 + 8].contiguous().reshape(T * 8, 2, S, S), T)
        assert torch.equal(s_c, std[b0:b0 + 8]) and torch.equal(m_c, mean[b0:b0 + 8])
    s_ref, m_ref = spec.mc_stats(preds[: T * B].double(), T)
    assert (std - s_ref.float()).abs().max() < 2e-6 and (mean - m_ref.float()).abs().max() < 1e-6
    del s_ref, m_ref, p5
    # retrified weights at 128^2 from the 512^2 statistics; the gate and the pseudo label both fire on both sides
    h = w = S // 4
    P = B * h * w
    oT_before = (torch.randn(P, 2, generator=g, device=DEV) * 2.0 + 1.0).contiguous()
    wts, m0, m1 = K.proto_weights(2, B, h, w, logits=oT_before, std_map=std, mean_map=mean)
    w_ref, m0_ref, m1_ref = spec.proto_weights(2, B, h, w, logits=oT_before, std_map=std, mean_map=mean)
    gate = m0 / 2
    assert 0.1 < gate.mean().item() < 0.9 and 0.05 < (wts[:, 0] > 0).float().mean().item() < 0.95
    flips = ((m0 != m0_ref) | (m1 != m1_ref)).float().mean().item()       # the std < 0.04 gate on a bilinear blend: ulp-level ties only
    assert flips < 1e-4, flips
    agree = (m0 == m0_ref) & (m1 == m1_ref)
    assert (wts - w_ref)[agree].abs().max() < 1e-6
    # the masked sums over P = 524,288 rows of 305 channels against fp64 torch
    C = 305
    feat = torch.randn(P, 308, generator=g, device=DEV)[:, :C]             # NHWC rows with the engine's padded row stride
    sums = torch.zeros(4, C + 1, dtype=torch.float64, device=DEV)
    K.proto_reduce(feat, wts, sums)
    ref = torch.zeros_like(sums)
    spec.proto_reduce(feat, wts, ref)
    assert model_cases.l2rel(sums, ref) < 1e-6, model_cases.l2rel(sums, ref)      # fp32 partial sums over 96 rows, then fp64
    cent = K.proto_finalize(sums)
    assert model_cases.l2rel(cent, spec.proto_finalize(ref)) < 1e-5
    # additivity over image chunks (per-rank partial sums -> all-reduce) and invariance under a permutation of the images
    part = torch.zeros_like(sums)
    for b0 in range(0, B, 8):
        r0, r1 = b0 * h * w, (b0 + 8) * h * w
        K.proto_reduce(feat[r0:r1], wts[r0:r1], part)
    assert model_cases.l2rel(part, sums) < 1e-6
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).to(DEV)
    featp = torch.empty(P, 308, device=DEV)[:, :C]
    featp.copy_(feat.reshape(B, h * w, C)[perm].reshape(P, C))
    wtsp = wts.reshape(B, h * w, 4)[perm].reshape(P, 4).contiguous()
    sp = torch.zeros_like(sums)
    K.proto_reduce(featp, wtsp, sp)
    assert model_cases.l2rel(K.proto_finalize(sp), cent) < 1e-5
    # backward: gradient into the features and into the weights
    dC = torch.randn(4, C, generator=g, device=DEV)
    d_feat = torch.empty(P, 308, device=DEV)[:, :C]
    d_w = K.proto_bwd(feat, wts, sums, dC, d_feat, False, True)
    d_ref = torch.empty(P, C, device=DEV)
    dw_ref = spec.proto_bwd(feat, wts, ref, dC, d_ref, False, True)
    assert model_cases.rel(d_feat, d_ref) < 1e-5 and model_cases.rel(d_w, dw_ref) < 1e-5


class _SeededMasks(torch.nn.Module):
    """The product generator with injected dropout keep-masks drawn on the device from a seeded generator, one draw per call in
    call order; with ``perm`` the images' masks are permuted like the images (for the MC passes on x.repeat(2): per copy)."""
    SITES = {"aspp.dropout": ((256, S // 16, S // 16), 0.5), "decoder.last_conv_boundary.3": ((256, S // 4, S // 4), 0.5),
             "decoder.last_conv_boundary.7": ((256, S // 4, S // 4), 0.1), "decoder.last_conv.2": ((305, S // 4, S // 4), 0.1)}

    def __init__(self, model, perm=None):
        super().__init__()
        self.model, self.perm = model, perm
        self.g = torch.Generator(device=DEV).manual_seed(123)

    def _draw(self, n, copies=1):
        out = {}
        for k, (shp, p) in self.SITES.items():
            m = (torch.rand((n,) + shp, generator=self.g, device=DEV) >= p).to(torch.uint8)
            if self.perm is not None:
                per = n // copies
                m = torch.cat([m[c * per:(c + 1) * per][self.perm] for c in range(copies)]).contiguous()
            out[k] = m
        return out

    def forward(self, x):
        self.model.set_dropout_masks(self._draw(x.shape[0]))
        return self.model(x)

    def mc_dropout_logits(self, x, passes=4, reps=2):
        masks = [self._draw(reps * x.shape[0], copies=reps) for _ in range(passes)]
        return self.model.mc_dropout_logits(x, passes=passes, reps=reps, masks=masks)

    def shared_weight_layouts(self):
        return self.model.shared_weight_layouts()

    def note_params_changed(self):
        self.model.note_params_changed()

    def pop_nonfinite(self):
        return self.model.pop_nonfinite()


def _one_step(perm, tmp, batches):
    img, tmap, tbd, imgT = batches
    sel = (lambda t: t) if perm is None else (lambda t: t[perm].contiguous())
    m = _SeededMasks(model_cases.seeded_model().to(DEV), perm)
    torch.manual_seed(1338)
    d1, d2 = BoundaryDiscriminator().to(DEV), UncertaintyDiscriminator().to(DEV)
    og = torch.optim.SGD(m.parameters(), lr=0.05)                 # SGD: the update is linear in the gradient (Adam turns the sign of
    od = torch.optim.SGD(d1.parameters(), lr=0.01)                # every near-zero gradient into a full step)
    od2 = torch.optim.SGD(d2.parameters(), lr=0.01)
    tr = Trainer_prototype_full.Trainer(
        cuda=True, model_gen=m, model_dis=d1, model_uncertainty_dis=d2, optimizer_gen=og, optimizer_dis=od,
        optimizer_uncertainty_dis=od2, val_loader=[], domain_loaderS=[], domain_loaderT=[], out=str(tmp), max_epoch=1,
        use_global=True, use_pid=True, retrify_pesudo=True, global_pro_weight=0.9, pro_weight=0.1, batch_size=B, warmup_epoch=-1)
    tr.epoch = 0
    m.train(); d1.train(); d2.train()
    init = {k: v.detach().clone() for k, v in m.model.named_parameters()}
    rows = [tr.train_step({"image": sel(img), "map": sel(tmap), "boundary": sel(tbd)}, {"image": sel(imgT)})]
    torch.cuda.synchronize()
    rec = {"rows": rows, "gen": {k: (v.detach() - init[k]) for k, v in m.model.named_parameters()},
           "dis": {k: v.detach().clone() for k, v in list(d1.named_parameters()) + [("u." + k, v) for k, v in d2.named_parameters()]},
           "src": tr.src_centroids.matrix.clone(), "tgt": tr.tgt_centroids.matrix.clone(),
           "mask0": tr.mask_0.clone(), "std": tr.target_std_map.clone()}
    del tr, m, d1, d2, og, od, od2
    torch.cuda.empty_cache()
    return rec


def test_train_step_at_32_plus_32_is_a_function_of_the_batch_as_a_set(tmp_path):
    from bench import synth_batch
    img, tmap, tbd = synth_batch(B, S, 1337, DEV)
    imgT = synth_batch(B, S, 4242, DEV)[0]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(DEV)
    torch.cuda.reset_peak_memory_stats()
    a = _one_step(None, tmp_path / "a", (img, tmap, tbd, imgT))
    b = _one_step(perm, tmp_path / "b", (img, tmap, tbd, imgT))
    print("rows", a["rows"], b["rows"], "peak GiB %.1f" % (torch.cuda.max_memory_allocated() / 2 ** 30))
    ra, rb = a["rows"][0], b["rows"][0]
    assert len(ra) == 6 and all(v == v and abs(v) < 1e6 for v in ra)
    for i, (x, y) in enumerate(zip(ra, rb)):
        assert abs(x - y) <= 2e-4 * max(abs(x), 1e-6), (i, x, y)       # seg, adv, D_same, D_diff, intra, inter
    assert model_cases.rel(b["src"], a["src"]) < 1e-4 and model_cases.rel(b["tgt"], a["tgt"]) < 1e-3
    # per-image outputs follow the permutation: the MC-dropout std map and the reliability mask of the target images
    assert model_cases.rel(b["std"], a["std"][perm]) < 1e-3
    assert (b["mask0"] != a["mask0"][perm]).float().mean().item() < 1e-3
    # the generator update (lr x averaged gradient): two summation orders of the same batch sums
    worst = max((model_cases.l2rel(b["gen"][k], a["gen"][k]), k) for k in a["gen"])
    print("worst generator update difference", worst)
    assert worst[0] < 5e-3, worst
    for k in a["dis"]:
        assert model_cases.l2rel(b["dis"][k], a["dis"][k]) < 1e-5, k
