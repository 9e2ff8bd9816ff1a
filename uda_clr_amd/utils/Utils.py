"""Prototype helpers and ramp functions - drop-in for the hot-path part of the reference's
``utils/Utils.py`` (same names and signatures, no cv2 / skimage / albumentations at import).

  gen_prototype(pred, feature)                                   Utils.py:108-131
  gen_prototype_retrify(oT_before, xt_feature, preds, features, T, stride)   Utils.py:159-225
  sigmoid_rampup / linear_rampup / cosine_rampdown               Utils.py:312-334

  postprocessing(prediction, threshold=0.75, dataset='G')         Utils.py:438-463 (+ get_largest_fillhole :427-436)

The visualisation helpers of the reference (Utils.py:349-426, 466-590) are outside the hot path and not built
(SURVEY.md section 2).
"""
import math

import numpy as np

from ..ops import gen_prototype, gen_prototype_from_labels, gen_prototype_retrify  # noqa: F401
from .metrics import *  # noqa: F401,F403  (the reference re-exports its metrics the same way)


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - t)^2) ramp of https://arxiv.org/abs/1610.02242"""
    if rampup_length == 0:
        return 1.0
    current = float(np.clip(current, 0.0, rampup_length))
    phase = 1.0 - current / rampup_length
    return float(math.exp(-5.0 * phase * phase))


def linear_rampup(current, rampup_length):
    assert current >= 0 and rampup_length >= 0
    return 1.0 if current >= rampup_length else current / rampup_length


def cosine_rampdown(current, rampdown_length):
    assert 0 <= current <= rampdown_length
    return float(0.5 * (math.cos(math.pi * current / rampdown_length) + 1))


def adaptation_factor(m):
    return 1.0 / (1.0 + math.exp(-0.8 * (m + 1))) - 0.3


def postprocessing(prediction, threshold=0.75, dataset='G'):
    """Evaluation post-processing of ONE image's [2, H, W] (cup, disc) probability map, as utils/Utils.py:438-463: threshold
    (dataset names starting with 'D': disc > 0.5, cup > 0.1; otherwise both > ``threshold``), five 7x7 median filters, erosion
    by the radius-7 diamond, largest connected component, holes filled.  Runs on the device (``uda_postprocess``); a CPU
    tensor is moved there.  Returns a numpy array like the reference's: [2, H, W], float for the 'D' branch (it writes the
    masks into a copy of the probabilities), uint8 otherwise.  ``postprocessing_batch`` does a whole [B, 2, H, W] batch."""
    import torch
    from ..ops import kernels
    pred = torch.as_tensor(prediction)
    if pred.dim() != 3 or pred.shape[0] != 2:
        raise ValueError("expected a [2, H, W] prediction")
    out = postprocessing_batch(pred[None], threshold, dataset)[0].cpu().numpy()
    return out.astype(np.float32) if dataset[0] == 'D' else out


def postprocessing_batch(predictions, threshold=0.75, dataset='G'):
    """[B, 2, H, W] probabilities -> uint8 [B, 2, H, W] post-processed masks, on the device."""
    import torch
    from ..ops import kernels
    pred = predictions
    if not pred.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("uda_clr_amd post-processing computes only on the MI355X HIP kernels (there is no CPU fallback)")
        pred = pred.cuda()
    thr_cup, thr_disc = (0.1, 0.5) if dataset[0] == 'D' else (threshold, threshold)
    return kernels().postprocess(pred.contiguous().float(), thr_cup, thr_disc)
