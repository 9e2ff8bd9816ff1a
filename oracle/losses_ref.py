"""CPU restatement of the two losses that are NOT in the reference's shipped source
(SURVEY.md section 8 row a15 / Appendix B).  TEST INFRASTRUCTURE.

                     *****  PARITY UNPINNED  *****
The only specification is the instruction/constant order of a stale bytecode file
(train_process/__pycache__/Trainer_prototype_mt.cpython-38.pyc, read as data, never executed) and the
paper; there is no source line, fixture or test in the reference to check these against.  The HIP
path is compared with THIS file, which pins the product to our reading of Appendix B, not to the
reference.

Decisions taken where Appendix B is silent:
  * the prototypes enter the discriminative loss detached (fixed anchors);
  * D(f, c) is the channel MEAN of (f - c)^2, margin 0.01, plain mean over B*h*w positions per term.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def discriminative_loss(feature, centroids, labels, margin=0.01):
    """feature [B,C,h,w]; centroids = (cup_obj, disc_obj, cup_bck, disc_bck) [1,C,1,1]; labels [B,2,h,w]
    (nearest-resized source labels).  Sum over k in {cup, disc} of
        mean(m_k * relu(D(f,c_k_obj) - D(f,c_k_bck) + margin)) + mean((1-m_k) * relu(D(f,c_k_bck) - D(f,c_k_obj) + margin))"""
    c = [t.detach() for t in centroids]
    dist = lambda cc: ((feature - cc) ** 2).mean(1, keepdim=True)
    loss = feature.new_zeros(())
    for k, (obj, bck) in enumerate(((c[0], c[2]), (c[1], c[3]))):
        m = labels[:, k:k + 1]
        a, b = dist(obj), dist(bck)
        loss = loss + (m * F.relu(a - b + margin)).mean() + ((1 - m) * F.relu(b - a + margin)).mean()
    return loss


def consistency_threshold(epoch, rampup=200):
    phase = 1.0 - min(max(float(epoch), 0.0), rampup) / rampup
    return (0.85 + 0.25 * math.exp(-5.0 * phase * phase)) * math.log(2.0)


def consistency_loss(oT_aug, oT, mask_0, mask_1, epoch, aug_weight=1.0):
    """pseudo label = sigmoid(oT) > tau(epoch); reliability mask = nearest-upsampled cat(mask_0, mask_1)
    (values {0, 2}); loss = aug_weight * sum(mask * BCE(sigmoid(oT_aug), label)) / sum(mask)."""
    y = (torch.sigmoid(oT.detach()) > consistency_threshold(epoch)).to(oT.dtype)
    mask = F.interpolate(torch.cat((mask_0, mask_1), 1), size=oT.shape[2:], mode="nearest")
    bce = F.binary_cross_entropy(torch.sigmoid(oT_aug), y, reduction="none")
    return aug_weight * (mask * bce).sum() / mask.sum().clamp_min(1e-12)
