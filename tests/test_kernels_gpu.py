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


def test_operands_beyond_the_32_bit_offsets_run_as_image_groups():
    """The conv / weight-gradient kernels take at most 2^29 elements per operand; HipKernels splits larger batches into groups
    of whole images.  With the limit lowered so that a 6-image batch needs 2-3 groups, results equal the single launch (outputs
    bit for bit: same per-row arithmetic; statistics and weight gradients up to summation order)."""
    import kernel_cases
    from kernel_cases import make_src, act_to, gen, padded, to_dev, rel
    dev = torch.device("cuda:0")
    K = kernel_cases.hip()
    g = gen(5)
    N, H, W, Cin, Cout = 6, 24, 20, 64, 160
    src = act_to(make_src(N, H, W, Cin, g, True, 1, True), dev)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev) / 24.0
    wl = K.relayout_ohwi(w)
    P = N * H * W
    dy = torch.randn(P, Cout, generator=g).to(dev)
    res = {}
    for mode in (K.MFMA_F32, K.MFMA_BF16X3):
        keep_mode, K.mfma = K.mfma, mode
        for tag, lim in (("one", 1 << 29), ("groups", (2 * H * W + 300) * max(src.x.stride(0), Cout))):
            keep = K.ELEM_LIMIT
            type(K).ELEM_LIMIT = lim
            try:
                if tag == "groups":
                    assert len(K._image_groups(N, H * W, max(src.x.stride(0), Cout), Cin if mode else 0)) >= 2
                out = torch.empty(P, Cout, device=dev)
                st = torch.zeros(16, 2, Cout, dtype=torch.float64, device=dev)
                K.conv(src, wl, 3, 1, out, stats=st)
                dw = torch.empty(Cout, Cin, 3, 3, device=dev)
                K.conv_wgrad(src, dy, 3, 1, dw)
                res[(mode, tag)] = (out, st.sum(0), dw)
            finally:
                type(K).ELEM_LIMIT = keep
        K.mfma = keep_mode
        a, b = res[(mode, "one")], res[(mode, "groups")]
        assert torch.equal(a[0], b[0])
        assert rel(b[1], a[1]) < 1e-6 and rel(b[2], a[2]) < 1e-5      # per-tile fp32 partials of the statistics shift with the tile origin


def test_stride_two_is_refused_where_it_is_not_built():
    """stride 2 lives in the wide-tile loaders only; the narrow kernels must say so instead of computing a stride-1 result"""
    import kernel_cases
    from uda_clr_amd.acts import Act
    dev = torch.device("cuda:0")
    K = kernel_cases.hip()
    x = torch.randn(2 * 16 * 16, 64, device=dev)
    w = K.relayout_ohwi(torch.randn(32, 64, 3, 3, device=dev))
    with pytest.raises(RuntimeError, match="stride 2"):
        K.conv(Act(x, 2, 16, 16), w, 3, 1, torch.empty(2 * 8 * 8, 32, device=dev), stride=2)
    with pytest.raises(RuntimeError, match="stride 2"):
        K.conv_wgrad(Act(x, 2, 16, 16), torch.randn(2 * 8 * 8, 32, device=dev), 3, 1, torch.empty(32, 64, 3, 3, device=dev), stride=2)
