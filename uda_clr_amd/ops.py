"""Differentiable front-ends of the loss / prototype / metric kernels (torch.autograd.Function
wrappers around uda_clr_amd.kernels.HipKernels).  HIP only: CPU tensors raise.

Reference call sites these replace:
  seg_loss               Trainer_prototype_full.py:292-294, Trainer_baseline.py:206-208
  prototypes             utils/Utils.py:108-131 (gen_prototype), :159-225 (gen_prototype_retrify)
  seg_counts / dice ...  utils/metrics.py:118-168
"""
from __future__ import annotations

import torch

from .acts import nchw_view, round4
from .parallel import all_reduce_sum_

_K = None


def kernels():
    global _K
    if _K is None:
        from .kernels import HipKernels
        _K = HipKernels()
    return _K


def rows_view(t: torch.Tensor) -> torch.Tensor:
    """Logical [B, C, H, W] -> [P, C] NHWC rows with a 16-byte aligned, multiple-of-4 row stride.
    Zero-copy for the channels-last views the generator returns; one packing copy otherwise."""
    B, C, H, W = t.shape
    ld = t.stride(3) if W > 1 else (t.stride(2) // max(W, 1) if H > 1 else round4(C))
    ok = (t.stride(1) == 1 or C == 1) and ld % 4 == 0 and ld >= round4(C) and t.stride(2) == W * ld and \
        t.stride(0) == H * W * ld and (t.data_ptr() % 16 == 0) and t.dtype == torch.float32
    if ok:
        return t.as_strided((B * H * W, C), (ld, 1), t.storage_offset())
    buf = torch.empty(B * H * W, round4(C), dtype=torch.float32, device=t.device)
    v = buf[:, :C]
    v.copy_(t.permute(0, 2, 3, 1).reshape(B * H * W, C))
    return v


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o, b, tmap, tbd):
        o, b, tmap, tbd = (t.contiguous().float() for t in (o, b, tmap, tbd))
        loss3 = kernels().seg_loss_fwd(o, tmap, b, tbd)
        ctx.save_for_backward(o, b, tmap, tbd)
        return loss3[0]

    @staticmethod
    def backward(ctx, g):
        o, b, tmap, tbd = ctx.saved_tensors
        d_o, d_b = kernels().seg_loss_bwd(o, tmap, b, tbd, g.reshape(1).contiguous().float())
        return d_o, d_b, None, None


def seg_loss(oS, boundaryS, target_map, target_boundary):
    """BCELoss(sigmoid(oS), map) + MSELoss(sigmoid(boundaryS), boundary): one fused pass, and one
    fused pass for both gradients."""
    return _SegLossFn.apply(oS, boundaryS, target_map, target_boundary)


class _ProtoFn(torch.autograd.Function):
    """centroids[4, C] = (sum_p w_k[p] f[p,:]) / (sum_p w_k[p]); sums are all-reduced over the data
    parallel ranks before the division so every rank holds the global centroids."""

    @staticmethod
    def forward(ctx, feat, wts, pred):
        K = kernels()
        rows = rows_view(feat)
        P, C = rows.shape
        sums = torch.zeros(4, C + 1, dtype=torch.float64, device=feat.device)
        K.proto_reduce(rows, wts, sums)
        all_reduce_sum_(sums)
        ctx.save_for_backward(rows, wts, sums)
        ctx.geom = tuple(feat.shape)
        ctx.want_dw = pred is not None and pred.requires_grad
        return K.proto_finalize(sums)

    @staticmethod
    def backward(ctx, dC):
        rows, wts, sums = ctx.saved_tensors
        B, C, H, W = ctx.geom
        d_feat = None
        d_rows = None
        if ctx.needs_input_grad[0]:
            d_rows = torch.empty(rows.shape[0], round4(C), dtype=torch.float32, device=rows.device)[:, :C]
            d_feat = nchw_view(d_rows, B, H, W)
        d_w = kernels().proto_bwd(rows, wts, sums, dC.contiguous().float(), d_rows, False, ctx.want_dw)
        d_pred = None
        if ctx.want_dw:      # w = (p0, p1, 1-p0, 1-p1)
            d_pred = torch.stack([d_w[:, 0] - d_w[:, 2], d_w[:, 1] - d_w[:, 3]], 1).reshape(B, H, W, 2).permute(0, 3, 1, 2)
        return d_feat, None, d_pred


class Centroids(tuple):
    """(cup_obj, disc_obj, cup_bck, disc_bck) as [1,C,1,1] views, like the reference's four return values, plus ``.matrix``,
    the [4, C] tensor they are rows of (what the fused alignment kernel takes)."""
    matrix = None
    centroids = None       # gen_prototype_retrify's 7-tuple: its first four entries as a Centroids of their own


def _split(cent):
    C = cent.shape[1]
    out = Centroids(cent[k].reshape(1, C, 1, 1) for k in range(4))
    out.matrix = cent
    return out


def _matrix(cents):
    m = getattr(cents, "matrix", None)
    return m if m is not None else torch.cat([t.reshape(1, -1) for t in cents], 0)


class _AlignFn(torch.autograd.Function):
    """EMA of both domains' centroids + intra / inter losses in one launch (uda_proto_align_fwd); backward: one launch for both
    current-centroid gradients (the EMA keeps gradient only through the current term, quirk Q4)."""

    @staticmethod
    def forward(ctx, cur_src, cur_tgt, prev_src, prev_tgt, decay):
        K = kernels()
        new_src, new_tgt, losses = K.proto_align_fwd(cur_src.contiguous().float(), cur_tgt.contiguous().float(),
                                                     prev_src, prev_tgt, decay)
        ctx.save_for_backward(new_src, new_tgt)
        ctx.w = (1.0 if prev_src is None else decay, 1.0 if prev_tgt is None else decay)
        ctx.mark_non_differentiable(new_src, new_tgt)
        ctx.set_materialize_grads(False)        # an unused output arrives as None, not as a zero tensor: no device read needed below
        return losses[0], losses[1], new_src, new_tgt

    @staticmethod
    def backward(ctx, g_intra, g_inter, _a, _b):
        new_src, new_tgt = ctx.saved_tensors
        if g_inter is not None:
            raise NotImplementedError("inter_loss is logged only (Trainer_prototype_full.py:443-449, :465); no gradient is built for it")
        if g_intra is None:
            return None, None, None, None, None
        d_src, d_tgt = kernels().proto_align_bwd(new_src, new_tgt, g_intra.reshape(1).contiguous().float(), ctx.w[0], ctx.w[1])
        return d_src, d_tgt, None, None, None


def proto_align(cur_src, cur_tgt, prev_src, prev_tgt, decay):
    """Trainer_prototype_full.py:335-355, 378-398, 428-444 fused: returns (intra_loss, inter_loss, src, tgt) where src / tgt
    are the DETACHED EMA centroids to store for the next iteration (``Centroids``); ``prev_*`` is that stored state or None."""
    ps = None if prev_src is None else _matrix(prev_src).detach().contiguous()
    pt = None if prev_tgt is None else _matrix(prev_tgt).detach().contiguous()
    intra, inter, new_src, new_tgt = _AlignFn.apply(_matrix(cur_src), _matrix(cur_tgt), ps, pt, float(decay))
    return intra, inter, _split(new_src), _split(new_tgt)


class _AdvLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d1, d2, label, scale):
        d1, d2 = d1.contiguous().float(), d2.contiguous().float()
        ctx.save_for_backward(d1, d2)
        ctx.args = (label, scale)
        return kernels().adv_loss_fwd(d1, d2, label, scale)[0]

    @staticmethod
    def backward(ctx, g):
        d1, d2 = ctx.saved_tensors
        g1, g2 = kernels().adv_loss_bwd(d1, d2, ctx.args[0], ctx.args[1], g.reshape(1).contiguous().float())
        return g1, g2, None, None


def adv_loss(d_out1, d_out2, label, scale=1.0):
    """scale * (BCEWithLogits(d_out1, label) + BCEWithLogits(d_out2, label)) on the two patch-discriminator outputs
    (Trainer_prototype_full.py:456-458, 479-513): one launch forward, one for both gradients."""
    return _AdvLossFn.apply(d_out1, d_out2, float(label), float(scale))


def gen_prototype(pred, feature):
    """utils/Utils.py:108-131.  pred [B,2,h,w] (ch0 cup, ch1 disc; hard labels or probabilities),
    feature [B,C,h,w] -> (cup_obj, disc_obj, cup_bck, disc_bck) centroids [1,C,1,1]."""
    B, _, h, w = pred.shape
    wts, _, _ = kernels().proto_weights(0, B, h, w, map_=pred.detach().contiguous().float())
    return _split(_ProtoFn.apply(feature, wts, pred))


def gen_prototype_from_labels(target_map, feature):
    """Fused form of Trainer_prototype_full.py:330-334: nearest-resize of the full-resolution label
    map to the feature size happens inside the weights kernel."""
    B, _, h, w = feature.shape
    wts, _, _ = kernels().proto_weights(0, B, h, w, map_=target_map.detach().contiguous().float())
    return _split(_ProtoFn.apply(feature, wts, None))


def gen_prototype_retrify(oT_before, xt_feature, preds, features, T, stride):
    """utils/Utils.py:159-225.  ``features`` is accepted for signature parity and ignored (its mean
    is dead in the reference, quirk Q5).  Gradient reaches only ``xt_feature`` (quirk Q6)."""
    K = kernels()
    B, C, h, w = xt_feature.shape
    assert preds.shape[0] == T * stride and stride == B
    std_map, mean_map = K.mc_stats(preds.detach().contiguous().float(), T)
    wts, m0, m1 = K.proto_weights(2, B, h, w, logits=rows_view(oT_before.detach()), std_map=std_map, mean_map=mean_map)
    cents = _split(_ProtoFn.apply(xt_feature, wts, None))
    out = Centroids(tuple(cents) + (std_map, m0.reshape(B, 1, h, w), m1.reshape(B, 1, h, w)))     # the reference's 7 return values
    out.matrix, out.centroids = None, cents
    return out


class _DiscriminativeFn(torch.autograd.Function):
    """Prototype-guided discriminative loss on the source features (no shipped
    source - parity unpinned).  D(f,c) = mean_c (f-c)^2, so D(f,c_obj) - D(f,c_bck) is AFFINE in f:
        diff_k[p] = -(2/C) f[p,:].(c_obj - c_bck) + (|c_obj|^2 - |c_bck|^2)/C
    one pass over the 305-channel feature (uda_feat_dot4) gives both classes' differences, and the
    gradient is a per-pixel scalar times the constant vector (c_bck - c_obj) (uda_feat_rank4)."""

    @staticmethod
    def forward(ctx, feature, cents, labels, margin):
        K = kernels()
        rows = rows_view(feature)
        P, C = rows.shape
        B, _, H, W = feature.shape
        c = cents.detach().float()                      # [4, C]: cup_obj, disc_obj, cup_bck, disc_bck
        coef = torch.zeros(4, C + 1, dtype=torch.float32, device=rows.device)
        for k in (0, 1):
            coef[k, :C] = (-2.0 / C) * (c[k] - c[k + 2])
            coef[k, C] = (c[k].pow(2).sum() - c[k + 2].pow(2).sum()) / C
        diff = K.feat_dot4(rows, coef)[:, :2]                                  # [P, 2]
        m = labels.detach().permute(0, 2, 3, 1).reshape(P, 2).float()
        pos, neg = diff + margin, margin - diff
        loss = ((m * torch.relu(pos)).sum(0) + ((1 - m) * torch.relu(neg)).sum(0)).sum() / P
        ctx.save_for_backward(coef, m, diff)
        ctx.geom, ctx.margin = (B, C, H, W), margin
        return loss

    @staticmethod
    def backward(ctx, g):
        coef, m, diff = ctx.saved_tensors
        B, C, H, W = ctx.geom
        P = m.shape[0]
        s = (m * (diff + ctx.margin > 0).float() - (1 - m) * (ctx.margin - diff > 0).float()) * (g / P)
        wts = torch.zeros(P, 4, dtype=torch.float32, device=m.device)
        wts[:, :2] = s
        d_rows = torch.empty(P, round4(C), dtype=torch.float32, device=m.device)[:, :C]
        kernels().feat_rank4(wts, coef, d_rows, False)
        return nchw_view(d_rows, B, H, W), None, None, None


def discriminative_loss(feature, centroids, labels, margin=0.01):
    """centroids: the 4-tuple (cup_obj, disc_obj, cup_bck, disc_bck) of [1,C,1,1] prototypes (detached)."""
    cents = torch.cat([t.reshape(1, -1) for t in centroids], 0)
    return _DiscriminativeFn.apply(feature, cents, labels, margin)


def normalize_tf(image_u8, label_u8):
    """Device-side ``Normalize_tf`` + ``ToTensor`` (dataloaders/custom_transforms.py:432-466,504-507) of a uint8 batch:
    image_u8 [B,H,W,3], label_u8 [B,H,W] (grey-coded mask) -> image [B,3,H,W] in [-1,1], map [B,2,H,W], boundary [B,1,H,W].
    Bit-identical to the scipy.ndimage calls the reference makes per sample on the CPU (tests/test_input_gpu.py)."""
    return kernels().normalize_tf(image_u8.contiguous(), label_u8.contiguous())


def elastic_deform(image_u8, label_u8, apply=None, noise=None, generator=None):
    """Device-side ``elastic_transform`` (custom_transforms.py:95-147) of a uint8 batch: per sample with ``apply[b]`` set
    (default: drawn with p = 0.5) a displacement field alpha * gaussian_filter(U(-1,1), sigma), alpha = 2 * W, sigma = 0.08 * W,
    bilinear resampling of the image (0 outside) and of the label (nearest edge).  Everything from the uniform noise on is the
    reference's float64 arithmetic in the reference's order: with ``noise`` ([2,B,H,W] float64 in [-1,1), the two fields numpy's
    ``RandomState.rand`` gave the reference) the output bytes equal the reference's (tests/test_input_golden_gpu.py).  Without it
    the noise comes from torch's device generator - the reference seeds its RandomState from OS entropy, so there is no
    stream to reproduce; the distribution is the same."""
    K = kernels()
    B, H, W, _ = image_u8.shape
    dev = image_u8.device
    if apply is None:
        apply = (torch.rand(B, generator=generator) > 0.5).to(torch.uint8).to(dev)
    if noise is None:
        noise = torch.rand(2, B, H, W, device=dev, dtype=torch.float64) * 2.0 - 1.0
    field = K.field_smooth(noise.to(torch.float64).contiguous(), 0.08 * W, 2.0 * W)
    return K.elastic_warp(image_u8.contiguous(), label_u8.contiguous(), field[0], field[1], apply)


def photometric_u8(image_u8, sp_pos, sp_count, sp_value, lut, erase_box):
    """add_salt_pepper_noise -> adjust_light -> eraser (custom_transforms.py:150-250) on a uint8 batch, in place, with the
    outcomes the dataloader workers drew (see dataloaders.custom_transforms.DEVICE_TAIL)."""
    return kernels().photometric_u8(image_u8, sp_pos.contiguous(), sp_count.contiguous().view(-1), sp_value.contiguous().view(-1),
                                    lut.contiguous(), erase_box.contiguous())


def photometric_augment(images, generator=None):
    """Device-side photometric augmentation of a [-1,1] image batch in the spirit of utils/Utils.py:33-43
    (brightness/contrast + saturation jitter with p=0.8, grayscale with p=0.2, 5x5 Gaussian blur with
    p=0.5; geometry and labels unchanged).  The reference does this per image on the CPU with
    albumentations/cv2; the exact random streams are not reproducible, so this is 'same family, same
    probabilities' (parity unpinned)."""
    B = images.shape[0]
    dev = images.device
    r = lambda *s: torch.rand(*s, generator=generator, device="cpu").to(dev)
    x = (images + 1.0) * 0.5
    on = (r(B, 1, 1, 1) < 0.8).float()
    alpha = 1.0 + on * (r(B, 1, 1, 1) * 0.4 - 0.2)               # contrast  in [0.8, 1.2]
    beta = on * (r(B, 1, 1, 1) * 0.4 - 0.2)                        # brightness in [-0.2, 0.2]
    x = (x * alpha + beta * x.mean((1, 2, 3), keepdim=True)).clamp(0, 1)
    gray = x.mean(1, keepdim=True)
    sat = 1.0 + on * (r(B, 1, 1, 1) * 0.6 - 0.3)
    x = (gray + (x - gray) * sat).clamp(0, 1)
    tg = (r(B, 1, 1, 1) < 0.2).float()
    x = tg * gray.expand_as(x) + (1 - tg) * x
    k = torch.tensor([1., 4., 6., 4., 1.], device=dev)
    k = (k[:, None] * k[None, :] / 256.0).expand(3, 1, 5, 5)
    blur = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), k, groups=3)
    bl = (r(B, 1, 1, 1) < 0.5).float()
    x = bl * blur + (1 - bl) * x
    return x * 2.0 - 1.0


def consistency_threshold(epoch, rampup=200):
    import math
    phase = 1.0 - min(max(float(epoch), 0.0), rampup) / rampup
    return (0.85 + 0.25 * math.exp(-5.0 * phase * phase)) * math.log(2.0)


def consistency_loss(oT_aug, oT, mask_0, mask_1, epoch, aug_weight=1.0):
    """aug_weight * sum(mask * BCE(sigmoid(oT_aug), [sigmoid(oT) > tau(epoch)])) / sum(mask), mask = nearest-
    upsampled cat(mask_0, mask_1) (parity unpinned).  Elementwise torch ops on
    two 512x512x2 maps per image - not worth a kernel of their own yet."""
    F = torch.nn.functional
    y = (torch.sigmoid(oT.detach()) > consistency_threshold(epoch)).to(oT.dtype)
    mask = F.interpolate(torch.cat((mask_0, mask_1), 1), size=oT.shape[2:], mode="nearest")
    bce = F.binary_cross_entropy(torch.sigmoid(oT_aug), y, reduction="none")
    return aug_weight * (mask * bce).sum() / mask.sum().clamp_min(1e-12)


def seg_counts(pred, target, thr=0.75):
    """int64 [C,3] on the host: (intersection, predicted, ground truth) for sigmoid(pred) > thr."""
    return kernels().seg_counts(pred.detach().contiguous().float(), target.detach().contiguous().float(), thr).cpu()
