"""CPU restatement of the validation metrics (TEST INFRASTRUCTURE).

Reference anchors
  utils/metrics.py:71-101    dice_coefficient_numpy  ((2I+1)/(1+S+G))
  utils/metrics.py:118-132   dice_coeff_2label       (sigmoid > 0.75, batch-level)
  utils/metrics.py:149-168   pixel_acc               (2x2 confusion: PA, mean IoU)
"""
from __future__ import annotations

import numpy as np
import torch


def dice_binary(seg, gt):
    seg = np.asarray(seg, dtype=bool)
    gt = np.asarray(gt, dtype=bool)
    inter = float(np.logical_and(seg, gt).sum())
    return (2.0 * inter + 1.0) / (1.0 + float(seg.sum()) + float(gt.sum()))


def _binarise(logits, thr=0.75):
    return (torch.sigmoid(logits.detach().cpu()) > thr).numpy()


def dice_coeff_2label(pred, target):
    p = _binarise(pred)
    t = target.detach().cpu().numpy()
    return dice_binary(p[:, 0], t[:, 0]), dice_binary(p[:, 1], t[:, 1])


def _confusion(pred, label):
    pred = pred.astype(np.int64).ravel()
    label = label.astype(np.int64).ravel()
    ok = (label >= 0) & (label < 2)
    return np.bincount(2 * label[ok] + pred[ok], minlength=4).reshape(2, 2).astype(np.float64)


def _pa_miou(cm):
    pa = np.diag(cm).sum() / cm.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
    return pa, np.nanmean(iou)


def pixel_acc(pred, target):
    """Returns (PA_cup, PA_disc, IoU_cup, IoU_disc) as metrics.py:149-168."""
    p = _binarise(pred)
    t = target.detach().cpu().numpy()
    pa_d, iou_d = _pa_miou(_confusion(p[:, 1], t[:, 1]))
    pa_c, iou_c = _pa_miou(_confusion(p[:, 0], t[:, 0]))
    return pa_c, pa_d, iou_c, iou_d
