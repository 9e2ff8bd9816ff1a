"""-m gpu: the HIP prototype / metric kernels, through the product front-ends and the C ABI, against the fixtures the
REFERENCE itself wrote (tests/golden/proto.npz: utils/Utils.py:108-131,159-225; metrics.json: utils/metrics.py:118-168)."""
import pytest
import torch

import proto_cases
from uda_clr_amd import ops
from uda_clr_amd.utils import metrics

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_hip_gen_prototype_matches_reference_fixture():
    """hard labels, soft predictions, and the soft branch's gradient into BOTH the prediction (quirk Q6,
    Trainer_prototype_full.py:375-377) and the feature."""
    proto_cases.check_gen_prototype(ops, DEV)


def test_hip_gen_prototype_retrify_matches_reference_fixture():
    """centroids 1e-5, std_map, exact reliability masks, |grad xt_feature|, zero gradient into oT_before (quirk Q6)."""
    proto_cases.check_gen_prototype_retrify(ops, DEV)


def test_hip_metrics_match_reference_fixture():
    proto_cases.check_metrics(metrics, DEV)
