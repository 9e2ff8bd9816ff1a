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


def _split(cent):
    C = cent.shape[1]
    return tuple(cent[k].reshape(1, C, 1, 1) for k in range(4))


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
    return cents + (std_map, m0.reshape(B, 1, h, w), m1.reshape(B, 1, h, w))


def seg_counts(pred, target, thr=0.75):
    """int64 [C,3] on the host: (intersection, predicted, ground truth) for sigmoid(pred) > thr."""
    return kernels().seg_counts(pred.detach().contiguous().float(), target.detach().contiguous().float(), thr).cpu()
