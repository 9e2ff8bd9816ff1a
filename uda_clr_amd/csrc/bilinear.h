// Bilinear (align_corners=True) index / weight arithmetic shared by resample.hip and upconv.hip.
#pragma once
#include "common.h"

// PyTorch's area_pixel_compute_source_index for align_corners=True: src = scale * dst (fp32)
__device__ __forceinline__ void bil_src(int o, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
    // contraction off: the product is rounded to fp32 BEFORE the subtraction below, as in PyTorch.  Fused into an fma
    // (hipcc's default; __fmul_rn is a plain `*` in the HIP headers and fuses too) the lerp weight carries the unrounded
    // product and the interpolated values sit ~1e-6 (relative) away from the reference's instead of ~5e-8.
#pragma clang fp contract(off)
    const float real = scale * (float)o;
    i0 = (int)real;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = real - (float)i0;
    l0 = 1.f - l1;
}
static inline float bil_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

// weight with which output index o reads input index i (0 if it does not)
__device__ __forceinline__ float bil_weight(int o, float scale, int n_in, int i) {
    int i0, i1;
    float l0, l1;
    bil_src(o, scale, n_in, i0, i1, l0, l1);
    float wgt = 0.f;
    if (i0 == i) wgt += l0;
    if (i1 == i) wgt += l1;
    return wgt;
}
// conservative range of outputs that can touch input i
__device__ __forceinline__ void bil_range(int i, float scale, int n_out, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = n_out - 1; return; }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > n_out - 1) hi = n_out - 1;
}

