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


# ---- the dense-conv cases again with the wide tiles on the OTHER matrix instructions than the session default (UDA_CLR_MFMA):
# "bf16x3" = fp32 emulated on the bf16 pipe (exact 3-way operand split, six bf16 MFMAs per fp32 product block, fp32 accumulation),
# "f32" = v_mfma_f32_32x32x2_f32.  Same statements, SAME tolerances for both.
_DENSE = [c for c in CASES if c[0].startswith(("conv", "dgrad", "wgrad"))]


@pytest.mark.parametrize("name,fn", _DENSE, ids=["other-mfma " + c[0] for c in _DENSE])
def test_kernel_matches_spec_on_the_other_matrix_instructions(name, fn):
    import kernel_cases
    K = kernel_cases.hip()
    keep = K.mfma
    K.mfma = K.MFMA_F32 if keep == K.MFMA_BF16X3 else K.MFMA_BF16X3
    try:
        err, tol = fn(torch.device("cuda:0"))
        torch.cuda.synchronize()
    finally:
        K.mfma = keep
    assert err <= tol, "%s (mfma mode %d): rel err %.3e > %.1e" % (name, 1 - keep, err, tol)
