"""Validation metrics - drop-in for the reference's ``utils/metrics.py`` (same function names and
return conventions).  The thresholding and the three pixel counts per class run in one HIP kernel
(uda_seg_counts); Dice, pixel accuracy and IoU are closed forms of those counts:

    dice = (2I + 1) / (1 + S + G)                                 (metrics.py:71-101, +1 smoothing)
    2x2 confusion (rows = label, cols = prediction): TP = I, FP = S - I, FN = G - I, TN = n - S - G + I
    PA = (TP + TN) / n,  IoU_fg = TP / (TP + FP + FN),  IoU_bg = TN / (TN + FP + FN)   (:149-168)
"""
import numpy as np
import torch

from .. import ops


def dice_from_counts(inter, seg, gt):
    return (2.0 * float(inter) + 1.0) / (1.0 + float(seg) + float(gt))


def dice_coeff_2label(pred, target):
    """(cup dice, disc dice) at sigmoid(pred) > 0.75 over the whole batch (metrics.py:118-132)."""
    c = ops.seg_counts(pred, target, 0.75)
    return dice_from_counts(*c[0].tolist()), dice_from_counts(*c[1].tolist())


def dice_coeff(pred, target):
    """Single-label Dice at threshold 0.5 (metrics.py:104-116) over all channels together."""
    c = ops.seg_counts(pred, target, 0.5).sum(0)
    return dice_from_counts(*c.tolist())


def _pa_miou(inter, seg, gt, n):
    tp, fp, fn = float(inter), float(seg - inter), float(gt - inter)
    tn = float(n) - tp - fp - fn
    pa = (tp + tn) / float(n)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.array([tn / (tn + fp + fn) if (tn + fp + fn) else np.nan,
                        tp / (tp + fp + fn) if (tp + fp + fn) else np.nan])
    return pa, float(np.nanmean(iou))


def pixel_acc(pred, target):
    """(PA_cup, PA_disc, IoU_cup, IoU_disc) as metrics.py:149-168."""
    c = ops.seg_counts(pred, target, 0.75)
    n = pred.shape[0] * pred.shape[2] * pred.shape[3]
    pa_c, iou_c = _pa_miou(*c[0].tolist(), n)
    pa_d, iou_d = _pa_miou(*c[1].tolist(), n)
    return pa_c, pa_d, iou_c, iou_d


def DiceLoss(input, target):
    smooth = 1.0
    i, t = input.contiguous().view(-1), target.contiguous().view(-1)
    return 1 - ((2.0 * (i * t).sum() + smooth) / (i.sum() + t.sum() + smooth))
