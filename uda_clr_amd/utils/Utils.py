"""Prototype helpers and ramp functions - drop-in for the hot-path part of the reference's
``utils/Utils.py`` (same names and signatures, no cv2 / skimage / albumentations at import).

  gen_prototype(pred, feature)                                   Utils.py:108-131
  gen_prototype_retrify(oT_before, xt_feature, preds, features, T, stride)   Utils.py:159-225
  sigmoid_rampup / linear_rampup / cosine_rampdown               Utils.py:312-334

Visualisation / post-processing helpers of the reference (Utils.py:349-590) are outside the hot
path and not built.
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
