"""-m gpu: every HIP kernel, called through the C ABI, against its fp32 torch statement."""
import pytest
import torch

from kernel_cases import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", CASES, ids=[c[0] for c in CASES])
def test_kernel_matches_spec(name, fn):
    err, tol = fn(torch.device("cuda:0"))
    torch.cuda.synchronize()
    assert err <= tol, "%s: rel err %.3e > %.1e" % (name, err, tol)
