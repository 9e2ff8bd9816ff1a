"""CPU restatement of the category-prototype functions (TEST INFRASTRUCTURE).

Reference anchors
  utils/Utils.py:108-131   gen_prototype
  utils/Utils.py:159-225   gen_prototype_retrify
  train_process/Trainer_prototype_full.py:335-355, 378-398   EMA of centroids (quirk Q4)
  train_process/Trainer_prototype_full.py:428-444            intra / inter losses

The restatement works on (weights, feature) pairs: every reference variant is a
weighted masked mean ``sum_p w_k[p] f[p, :] / sum_p w_k[p]`` for four weight maps
k = (cup obj, disc obj, cup bck, disc bck); only the weights differ.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _weighted_centroids(weights, feat):
    """weights: 4 tensors [B,1,h,w]; feat [B,C,h,w] -> 4 centroids [1,C,1,1]."""
    out = []
    for w in weights:
        s = torch.sum(feat * w, dim=[0, 2, 3], keepdim=True)
        n = torch.sum(w, dim=[0, 2, 3], keepdim=True)
        out.append(s / n)
    return tuple(out)


def gen_prototype(pred, feat):
    """Utils.py:108-131.  pred [B,2,h,w] (ch0 cup, ch1 disc), feat [B,C,h,w].
    Order of results: cup_obj, disc_obj, cup_bck, disc_bck."""
    cup, disc = pred[:, 0:1], pred[:, 1:]
    return _weighted_centroids((cup, disc, 1.0 - cup, 1.0 - disc), feat)


def mc_statistics(preds, T, stride):
    """Utils.py:161-168: preds [T*stride,2,H,W] logits of T stochastic passes.
    Returns (std_map of sigmoid(x/2) over T, unbiased; mean of sigmoid(x) over T)."""
    p = preds.reshape(T, stride, 2, preds.shape[2], preds.shape[3])
    return torch.std(torch.sigmoid(p / 2.0), dim=0), torch.mean(torch.sigmoid(p), dim=0)


def retrify_weights(oT_before, std_map, prediction, size):
    """Utils.py:170-206: pseudo label (sigmoid>0.75), reliability mask (std<0.04 after
    bilinear align_corners resize) and confidence (resized mean prediction)."""
    pred_s = F.interpolate(prediction, size=size, mode="bilinear", align_corners=True)
    std_s = F.interpolate(std_map, size=size, mode="bilinear", align_corners=True)
    pl = (torch.sigmoid(oT_before) > 0.75).to(oT_before.dtype)     # no gradient (quirk Q6)
    m = (std_s < 0.04).to(oT_before.dtype)
    m0, m1 = m[:, 0:1], m[:, 1:]
    w = (m0 * pl[:, 0:1] * pred_s[:, 0:1],
         m1 * pl[:, 1:] * pred_s[:, 1:],
         m0 * (1.0 - pl[:, 0:1]) * (1.0 - pred_s[:, 0:1]),
         m1 * (1.0 - pl[:, 1:]) * (1.0 - pred_s[:, 1:]))
    return w, 2.0 * m0, 2.0 * m1        # mask_0 / mask_1 take values {0, 2} (Utils.py:205-206)


def gen_prototype_retrify(oT_before, xt_feature, preds, T, stride):
    """Utils.py:159-225 without the dead ``features`` mean (quirk Q5).
    Returns (4 centroids..., std_map, mask_0, mask_1)."""
    std_map, prediction = mc_statistics(preds, T, stride)
    w, mask_0, mask_1 = retrify_weights(oT_before.detach(), std_map, prediction,
                                        xt_feature.shape[2:])
    return _weighted_centroids(w, xt_feature) + (std_map, mask_0, mask_1)


class PrototypeBank:
    """EMA state of the eight centroids (Trainer_prototype_full.py:335-355, 378-398).
    ``update`` returns the centroids entering the loss: first call -> current, later
    (1-decay)*stored.detach() + decay*current; the stored copy is always detached."""

    def __init__(self, decay):
        self.decay = decay
        self.state = {"src": None, "tgt": None}

    def update(self, which, current):
        prev = self.state[which]
        if prev is None:
            new = tuple(current)
        else:
            d = self.decay
            new = tuple((1 - d) * p + d * c for p, c in zip(prev, current))
        self.state[which] = tuple(t.detach() for t in new)
        return new


def alignment_losses(src, tgt):
    """Trainer_prototype_full.py:428-444.  src/tgt: (cup_obj, disc_obj, cup_bck, disc_bck)."""
    intra = sum(F.mse_loss(s, t) for s, t in zip(src, tgt))
    inter = F.mse_loss(src[1], src[3]) + F.mse_loss(src[0], src[2])
    return intra, inter
