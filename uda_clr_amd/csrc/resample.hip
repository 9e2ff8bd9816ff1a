// Bilinear (align_corners=True) resampling, global average pooling / row broadcast and the
// dropout keep-mask generator.  All HBM-bound gather/stream kernels.
//   upsample_*        NHWC -> NHWC (decoder.py:50, 32x32 -> 128x128 x 256 ch into a channel window)
//   head_upsample_*   NHWC (C<=4) -> contiguous NCHW (deeplabv3.py:39-40, 128^2 -> 512^2 logits)
//   gap / broadcast   aspp.py:55-58,70-71 (image-level branch) and their adjoints
// The backward kernels GATHER (every input pixel re-derives, with the forward's own index
// arithmetic, which outputs touched it) - deterministic, no float atomics.
#include "common.h"
#include "bilinear.h"

// STATS: per-channel (sum, sum of squares) of the OUTPUT accumulated on the way (a thread's channel group is fixed over its
// grid-stride iterations because 256 % (C / 4) == 0; the host checks it): per-thread fp32 partial sums, the 256 / G threads of a
// channel group combined through LDS, then fp64 atomics into one of the UDA_STAT_SLOTS replicas of double[2][stat_C].
template <bool STATS, bool STORE = true>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, int64_t ldx, int N, int h, int w,
                                                           int C, float* __restrict__ out, int64_t ldo, int H, int W,
                                                           float sh, float sw, double* __restrict__ stats, int stat_C) {
    const int G = C >> 2;
    const int64_t total = (int64_t)N * H * W * G;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const int ow = (int)(p % W), oh = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        int h0, h1, w0, w1;
        float lh0, lh1, lw0, lw1;
        bil_src(oh, sh, h, h0, h1, lh0, lh1);
        bil_src(ow, sw, w, w0, w1, lw0, lw1);
        const float* b = x + (int64_t)n * h * w * ldx + cg * 4;
        const float4 a00 = uda_ld4(b + ((int64_t)h0 * w + w0) * ldx), a01 = uda_ld4(b + ((int64_t)h0 * w + w1) * ldx);
        const float4 a10 = uda_ld4(b + ((int64_t)h1 * w + w0) * ldx), a11 = uda_ld4(b + ((int64_t)h1 * w + w1) * ldx);
        float4 r;
        r.x = lh0 * (lw0 * a00.x + lw1 * a01.x) + lh1 * (lw0 * a10.x + lw1 * a11.x);
        r.y = lh0 * (lw0 * a00.y + lw1 * a01.y) + lh1 * (lw0 * a10.y + lw1 * a11.y);
        r.z = lh0 * (lw0 * a00.z + lw1 * a01.z) + lh1 * (lw0 * a10.z + lw1 * a11.z);
        r.w = lh0 * (lw0 * a00.w + lw1 * a01.w) + lh1 * (lw0 * a10.w + lw1 * a11.w);
        if (STORE) uda_st4(out + p * ldo + cg * 4, r);
        if (STATS) {
            s1[0] += r.x; s1[1] += r.y; s1[2] += r.z; s1[3] += r.w;
            s2[0] += r.x * r.x; s2[1] += r.y * r.y; s2[2] += r.z * r.z; s2[3] += r.w * r.w;
        }
    }
    if (STATS) {
        __shared__ float red[256 * 8];
        const int tid = threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[tid * 8 + j] = s1[j];
            red[tid * 8 + 4 + j] = s2[j];
        }
        __syncthreads();
        // channel c = 4 * cg + j belongs to the threads tid = cg, cg + G, ... (cg = tid % G for every iteration of a thread)
        for (int e = tid; e < 2 * C; e += 256) {
            const int q = e / C, c = e - q * C, cg = c >> 2, j = c & 3;
            float t = 0.f;
            for (int th = cg; th < 256; th += G) t += red[th * 8 + q * 4 + j];
            atomicAdd(&stats[((int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 2 + q) * stat_C + c], (double)t);
        }
    }
}

// Statistics of the upsampled tensor WITHOUT forming it (the stochastic passes: uda_mc_seg_head re-derives the values later).
// Every output pixel of cell (i, j) - the output rows whose upper source row is i, the output columns whose left source column
// is j - is a bilinear form of the cell's four corners, u = sum_{r,c} a_r(oh) b_c(ow) f_rc, so over the cell's pixels
//     sum u   = sum_{r,c} (sum_oh a_r) (sum_ow b_c) f_rc
//     sum u^2 = sum_{r,r',c,c'} (sum_oh a_r a_r') (sum_ow b_c b_c') f_rc f_r'c'
// with 2 + 3 row moments and 2 + 3 column moments that depend on the cell only: ~30 multiply-adds per cell and channel instead
// of ~10 per OUTPUT PIXEL and channel (16 pixels per cell at x4): the pixel-wise pass was bound by its arithmetic (195 us at 32
// images for 33 MB of input).  One thread: 4 channels of one cell per grid-stride iteration; fp32 partial sums, LDS, fp64 atomics.
__device__ __forceinline__ void bil_cell_moments(int i, float scale, int n_in, int n_out, float& m0, float& m1, float& m00, float& m01, float& m11) {
    m0 = m1 = m00 = m01 = m11 = 0.f;
    int lo, hi;
    if (scale <= 0.f) { lo = 0; hi = n_out - 1; }
    else {
        lo = max((int)floorf((float)i / scale) - 1, 0);
        hi = min((int)ceilf((float)(i + 1) / scale) + 1, n_out - 1);
    }
    for (int o = lo; o <= hi; ++o) {
        int i0, i1;
        float l0, l1;
        bil_src(o, scale, n_in, i0, i1, l0, l1);
        if (i0 != i) continue;
        m0 += l0; m1 += l1;
        m00 += l0 * l0; m01 += l0 * l1; m11 += l1 * l1;
    }
}

__global__ __launch_bounds__(256) void upsample_stats_cells_kernel(const float* __restrict__ x, int64_t ldx, int N, int h, int w, int C,
                                                                   int H, int W, float sh, float sw, double* __restrict__ stats, int stat_C) {
    const int G = C >> 2;
    const int64_t total = (int64_t)N * h * w * G;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t q = e / G;
        const int j = (int)(q % w), i = (int)((q / w) % h), n = (int)(q / ((int64_t)w * h));
        float a0, a1, a00, a01, a11, b0, b1, b00, b01, b11;
        bil_cell_moments(i, sh, h, H, a0, a1, a00, a01, a11);
        bil_cell_moments(j, sw, w, W, b0, b1, b00, b01, b11);
        if ((a0 == 0.f && a1 == 0.f) || (b0 == 0.f && b1 == 0.f)) continue;      // no output pixel has this cell (the last row / column of an exact ratio)
        const int i1 = min(i + 1, h - 1), j1 = min(j + 1, w - 1);
        const float* bp = x + (int64_t)n * h * w * ldx + cg * 4;
        const float4 f00 = uda_ld4(bp + ((int64_t)i * w + j) * ldx), f01 = uda_ld4(bp + ((int64_t)i * w + j1) * ldx);
        const float4 f10 = uda_ld4(bp + ((int64_t)i1 * w + j) * ldx), f11 = uda_ld4(bp + ((int64_t)i1 * w + j1) * ldx);
        const float u00[4] = {f00.x, f00.y, f00.z, f00.w}, u01[4] = {f01.x, f01.y, f01.z, f01.w};
        const float u10[4] = {f10.x, f10.y, f10.z, f10.w}, u11[4] = {f11.x, f11.y, f11.z, f11.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s1[k] += a0 * (b0 * u00[k] + b1 * u01[k]) + a1 * (b0 * u10[k] + b1 * u11[k]);
            // column forms Q(u_r, u_r') = b00 u_r0 u_r'0 + b01 (u_r0 u_r'1 + u_r1 u_r'0) + b11 u_r1 u_r'1
            const float q00 = b00 * u00[k] * u00[k] + 2.f * b01 * u00[k] * u01[k] + b11 * u01[k] * u01[k];
            const float q01 = b00 * u00[k] * u10[k] + b01 * (u00[k] * u11[k] + u01[k] * u10[k]) + b11 * u01[k] * u11[k];
            const float q11 = b00 * u10[k] * u10[k] + 2.f * b01 * u10[k] * u11[k] + b11 * u11[k] * u11[k];
            s2[k] += a00 * q00 + 2.f * a01 * q01 + a11 * q11;
        }
    }
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid * 8 + k] = s1[k];
        red[tid * 8 + 4 + k] = s2[k];
    }
    __syncthreads();
    for (int e = tid; e < 2 * C; e += 256) {
        const int qd = e / C, c = e - qd * C, cg = c >> 2, k = c & 3;
        float t = 0.f;
        for (int th = cg; th < 256; th += G) t += red[th * 8 + qd * 4 + k];
        atomicAdd(&stats[((int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 2 + qd) * stat_C + c], (double)t);
    }
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dout, int64_t ldo, int N, int H, int W,
                                                           int C, float* __restrict__ dx, int64_t ldx, int h, int w,
                                                           float sh, float sw) {
    const int G = C >> 2;
    const int64_t total = (int64_t)N * h * w * G;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const int iw = (int)(p % w), ih = (int)((p / w) % h), n = (int)(p / ((int64_t)w * h));
        int hlo, hhi, wlo, whi;
        bil_range(ih, sh, H, hlo, hhi);
        bil_range(iw, sw, W, wlo, whi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oh = hlo; oh <= hhi; ++oh) {
            const float wh = bil_weight(oh, sh, h, ih);
            if (wh == 0.f) continue;
            for (int ow = wlo; ow <= whi; ++ow) {
                const float ww = bil_weight(ow, sw, w, iw);
                if (ww == 0.f) continue;
                const float4 g = uda_ld4(dout + (((int64_t)n * H + oh) * W + ow) * ldo + cg * 4);
                const float k = wh * ww;
                acc.x += k * g.x; acc.y += k * g.y; acc.z += k * g.z; acc.w += k * g.w;
            }
        }
        uda_st4(dx + p * ldx + cg * 4, acc);
    }
}

extern "C" int uda_upsample_fwd(const float* x, int64_t ldx, int N, int h, int w, int C, float* out, int64_t ldo, int H,
                                int W, void* stream) {
    UDA_REQUIRE(x && out && uda_aligned16(x) && uda_aligned16(out) && ldx % 4 == 0 && ldo % 4 == 0 && C % 4 == 0 && C > 0 &&
                    ldx >= C && ldo >= C, "uda_upsample_fwd: C and lds must be multiples of 4, pointers 16-byte aligned");
    UDA_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "uda_upsample_fwd: bad geometry");
    const int64_t total = (int64_t)N * H * W * (C / 4);
    int grid = uda_cdiv(total, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(upsample_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, N, h, w, C, out, ldo, H, W,
                       bil_scale(h, H), bil_scale(w, W), (double*)nullptr, 0);
    UDA_LAUNCH_CHECK("upsample_fwd");
    return 0;
}

/* uda_upsample_fwd that also accumulates the per-channel (sum, sum of squares) of its OUTPUT into channels [0, C) of
 * stats = double[UDA_STAT_SLOTS][2][stat_C] (ADDED into; the BatchNorm(305) of decoder.py:23 over cat(up(x), low, boundary):
 * the 256 upsampled channels' statistics come from the pass that writes them).  Needs 256 % (C / 4) == 0. */
extern "C" int uda_upsample_fwd_stats(const float* x, int64_t ldx, int N, int h, int w, int C, float* out, int64_t ldo, int H,
                                      int W, double* stats, int stat_C, void* stream) {
    UDA_REQUIRE(x && uda_aligned16(x) && ldx % 4 == 0 && C % 4 == 0 && C > 0 && ldx >= C &&
                    (!out || (uda_aligned16(out) && ldo % 4 == 0 && ldo >= C)),
                "uda_upsample_fwd_stats: C and lds must be multiples of 4, pointers 16-byte aligned");
    UDA_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "uda_upsample_fwd_stats: bad geometry");
    UDA_REQUIRE(stats && stat_C >= C && 256 % (C / 4) == 0, "uda_upsample_fwd_stats: needs an accumulator and 256 %% (C / 4) == 0");
    const int64_t total = (int64_t)N * H * W * (C / 4);
    int grid = uda_cdiv(total, 256);
    if (grid > 4096) grid = 4096;                 // (fewer, longer threads: one LDS reduction + 2C fp64 atomics per workgroup)
    if (out)
        hipLaunchKernelGGL((upsample_fwd_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, N, h, w, C, out, ldo,
                           H, W, bil_scale(h, H), bil_scale(w, W), stats, stat_C);
    else {      // out = NULL: the statistics of the upsampled tensor without forming it (uda_mc_seg_head re-derives its values): per source cell
        static const int cells = getenv("UDA_UPSAMPLE_STATS_CELLS") ? atoi(getenv("UDA_UPSAMPLE_STATS_CELLS")) : 1;      // A/B switch
        int gc = uda_cdiv((int64_t)N * h * w * (C / 4), 256);
        if (gc > 1024) gc = 1024;
        if (cells)
            hipLaunchKernelGGL(upsample_stats_cells_kernel, dim3(gc), dim3(256), 0, (hipStream_t)stream, x, ldx, N, h, w, C, H, W,
                               bil_scale(h, H), bil_scale(w, W), stats, stat_C);
        else
            hipLaunchKernelGGL((upsample_fwd_kernel<true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, N, h, w, C, out, ldo,
                               H, W, bil_scale(h, H), bil_scale(w, W), stats, stat_C);
    }
    UDA_LAUNCH_CHECK("upsample_fwd_stats");
    return 0;
}

// ------------------------------------------------------------------------------------------
// The segmentation head of a NO-GRAD stochastic pass (Trainer_prototype_full.py:358-368 -> decoder.py:23-32,51-53 + :53's cat):
//     x1b[p, o] = bias[o] + sum_c w[o, c] * mask[p, c] * ms * act(scale[c] * xf[p, c] + shift[c]),
//     xf[p, :] = cat(up(feature)[p, 0:Cf], low[p mod P_low, 0:Cl], boundary[p])
// with the 305-channel x_feature matrix xf never written: the upsampled channels are interpolated on the fly from the small
// feature map (4 cached 16-byte reads per 4 channels instead of a 1 KB row per pixel), the low-level channels come from the
// un-repeated [P_low, Cl] rows (the repeated batch shares them), the boundary logit from its own column.  16 lanes per pixel,
// coefficients and weights in LDS - the structure of conv_heads_kernel (igemm_conv.hip), same summation order.
struct McHeadArgs {
    const float* feat; int64_t ld_feat; int N, h, w, Cf;
    const float* low; int64_t ld_low; int Cl; int64_t P_low;
    const float* bnd; int64_t ld_bnd;
    int H, W;
    const float* scale; const float* shift;
    int act;
    const uint8_t* mask; int64_t ldm; float mask_scale;
    const float* wgt; int64_t ldw;
    const float* bias;
    float* out; int64_t ldo;
    float sh, sw;
};

__global__ __launch_bounds__(256) void mc_seg_head_kernel(McHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];       // [4][Kc]: scale, shift, w[0], w[1]
    const int C = a.Cf + a.Cl + 1, Kc = ((C + 3) >> 2) << 2;
    for (int e = threadIdx.x; e < Kc; e += 256) {
        const bool in = e < C;
        hsm[e] = (in && a.scale) ? a.scale[e] : 1.f;
        hsm[Kc + e] = (in && a.shift) ? a.shift[e] : 0.f;
        hsm[2 * Kc + e] = in ? a.wgt[e] : 0.f;
        hsm[3 * Kc + e] = in ? a.wgt[a.ldw + e] : 0.f;
    }
    __syncthreads();
    const int l16 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int64_t P = (int64_t)a.N * a.H * a.W;
    const float alo = a.act == ACT_NONE ? -INFINITY : 0.f, ahi = a.act == ACT_RELU6 ? 6.f : INFINITY;
    const float ms = a.mask_scale;
    const int G = Kc >> 2;
    for (int it = 0; it < 8; ++it) {
        const int64_t p = (int64_t)blockIdx.x * 128 + it * 16 + pl;
        float acc0 = 0.f, acc1 = 0.f;
        if (p < P) {
            const int ow = (int)(p % a.W), oh = (int)((p / a.W) % a.H);
            const int64_t n = p / ((int64_t)a.W * a.H);
            int h0, h1, w0, w1;
            float lh0, lh1, lw0, lw1;
            bil_src(oh, a.sh, a.h, h0, h1, lh0, lh1);
            bil_src(ow, a.sw, a.w, w0, w1, lw0, lw1);
            const float* fb = a.feat + n * a.h * a.w * a.ld_feat;
            const float* f00 = fb + ((int64_t)h0 * a.w + w0) * a.ld_feat, *f01 = fb + ((int64_t)h0 * a.w + w1) * a.ld_feat;
            const float* f10 = fb + ((int64_t)h1 * a.w + w0) * a.ld_feat, *f11 = fb + ((int64_t)h1 * a.w + w1) * a.ld_feat;
            const float* lr = a.low + (p % a.P_low) * a.ld_low;
            for (int g = l16; g < G; g += 16) {
                const int c = g * 4;
                float4 xv;
                if (c < a.Cf) {
                    const float4 a00 = uda_ld4(f00 + c), a01 = uda_ld4(f01 + c), a10 = uda_ld4(f10 + c), a11 = uda_ld4(f11 + c);
                    xv.x = lh0 * (lw0 * a00.x + lw1 * a01.x) + lh1 * (lw0 * a10.x + lw1 * a11.x);
                    xv.y = lh0 * (lw0 * a00.y + lw1 * a01.y) + lh1 * (lw0 * a10.y + lw1 * a11.y);
                    xv.z = lh0 * (lw0 * a00.z + lw1 * a01.z) + lh1 * (lw0 * a10.z + lw1 * a11.z);
                    xv.w = lh0 * (lw0 * a00.w + lw1 * a01.w) + lh1 * (lw0 * a10.w + lw1 * a11.w);
                } else if (c < a.Cf + a.Cl) {
                    xv = uda_ld4(lr + (c - a.Cf));
                } else {
                    xv = make_float4(a.bnd[p * a.ld_bnd], 0.f, 0.f, 0.f);
                }
                const float4 sc = uda_ld4(&hsm[c]), shv = uda_ld4(&hsm[Kc + c]);
                float u[4] = {__builtin_amdgcn_fmed3f(xv.x * sc.x + shv.x, alo, ahi), __builtin_amdgcn_fmed3f(xv.y * sc.y + shv.y, alo, ahi),
                              __builtin_amdgcn_fmed3f(xv.z * sc.z + shv.z, alo, ahi), __builtin_amdgcn_fmed3f(xv.w * sc.w + shv.w, alo, ahi)};
                if (a.mask) {
                    uint32_t mk = *reinterpret_cast<const uint32_t*>(a.mask + p * a.ldm + c);
                    if (c + 4 > C) mk &= 0xffffffffu >> (8 * (c + 4 - C));
                    u[0] *= (float)(mk & 0xffu) * ms; u[1] *= (float)((mk >> 8) & 0xffu) * ms;
                    u[2] *= (float)((mk >> 16) & 0xffu) * ms; u[3] *= (float)(mk >> 24) * ms;
                }
                const float4 w0v = uda_ld4(&hsm[2 * Kc + c]), w1v = uda_ld4(&hsm[3 * Kc + c]);      // zero beyond C
                acc0 += u[0] * w0v.x + u[1] * w0v.y + u[2] * w0v.z + u[3] * w0v.w;
                acc1 += u[0] * w1v.x + u[1] * w1v.y + u[2] * w1v.z + u[3] * w1v.w;
            }
        }
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) {
            acc0 += __shfl_xor(acc0, d);
            acc1 += __shfl_xor(acc1, d);
        }
        if (l16 == 0 && p < P) {
            a.out[p * a.ldo] = acc0 + (a.bias ? a.bias[0] : 0.f);
            a.out[p * a.ldo + 1] = acc1 + (a.bias ? a.bias[1] : 0.f);
        }
    }
}

extern "C" int uda_mc_seg_head(const float* feature, int64_t ld_feat, int N, int h, int w, int Cf, const float* low, int64_t ld_low,
                               int Cl, int64_t P_low, const float* boundary, int64_t ld_bnd, int H, int W, const float* scale,
                               const float* shift, int act, const uint8_t* mask, int64_t ldm, float mask_scale, const float* weight,
                               int64_t ldw, const float* bias, float* out, int64_t ldo, void* stream) {
    UDA_REQUIRE(feature && low && boundary && weight && out && uda_aligned16(feature) && uda_aligned16(low) && ld_feat % 4 == 0 &&
                    ld_low % 4 == 0 && Cf > 0 && Cf % 4 == 0 && Cl > 0 && Cl % 4 == 0 && ld_feat >= Cf && ld_low >= Cl && ld_bnd >= 1 &&
                    ldo >= 2 && ldw >= Cf + Cl + 1, "uda_mc_seg_head: bad operands (channel counts multiples of 4, 16-byte aligned rows)");
    UDA_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && P_low > 0 && ((int64_t)N * H * W) % P_low == 0,
                "uda_mc_seg_head: bad geometry (the low-level rows must tile the pixels)");
    UDA_REQUIRE(!mask || (((reinterpret_cast<uintptr_t>(mask)) & 3u) == 0 && ldm % 4 == 0 && ldm >= ((Cf + Cl + 1 + 3) / 4) * 4),
                "uda_mc_seg_head: mask rows must be 4-byte aligned and hold round4(C) bytes");
    McHeadArgs a;
    a.feat = feature; a.ld_feat = ld_feat; a.N = N; a.h = h; a.w = w; a.Cf = Cf;
    a.low = low; a.ld_low = ld_low; a.Cl = Cl; a.P_low = P_low;
    a.bnd = boundary; a.ld_bnd = ld_bnd; a.H = H; a.W = W;
    a.scale = scale; a.shift = shift; a.act = act; a.mask = mask; a.ldm = ldm; a.mask_scale = mask_scale;
    a.wgt = weight; a.ldw = ldw; a.bias = bias; a.out = out; a.ldo = ldo;
    a.sh = bil_scale(h, H); a.sw = bil_scale(w, W);
    const int Kc = ((Cf + Cl + 1 + 3) / 4) * 4;
    const int64_t P = (int64_t)N * H * W;
    hipLaunchKernelGGL(mc_seg_head_kernel, dim3((unsigned)uda_cdiv(P, 128)), dim3(256), (size_t)4 * Kc * sizeof(float), (hipStream_t)stream, a);
    UDA_LAUNCH_CHECK("mc_seg_head");
    return 0;
}

extern "C" int uda_upsample_bwd(const float* dout, int64_t ldo, int N, int H, int W, int C, float* dx, int64_t ldx, int h,
                                int w, void* stream) {
    UDA_REQUIRE(dout && dx && uda_aligned16(dout) && uda_aligned16(dx) && ldx % 4 == 0 && ldo % 4 == 0 && C % 4 == 0 && C > 0 &&
                    ldx >= C && ldo >= C, "uda_upsample_bwd: C and lds must be multiples of 4, pointers 16-byte aligned");
    UDA_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "uda_upsample_bwd: bad geometry");
    const int64_t total = (int64_t)N * h * w * (C / 4);
    int grid = uda_cdiv(total, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, ldo, N, H, W, C, dx, ldx, h, w,
                       bil_scale(h, H), bil_scale(w, W));
    UDA_LAUNCH_CHECK("upsample_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_upsample_fwd_kernel(const float* __restrict__ x, int64_t ldx, int N, int h, int w,
                                                                int C, float* __restrict__ out, int H, int W, float sh,
                                                                float sw) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int ow = (int)(p % W), oh = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        int h0, h1, w0, w1;
        float lh0, lh1, lw0, lw1;
        bil_src(oh, sh, h, h0, h1, lh0, lh1);
        bil_src(ow, sw, w, w0, w1, lw0, lw1);
        const float* b = x + (int64_t)n * h * w * ldx;
        for (int c = 0; c < C; ++c) {
            const float a00 = b[((int64_t)h0 * w + w0) * ldx + c], a01 = b[((int64_t)h0 * w + w1) * ldx + c];
            const float a10 = b[((int64_t)h1 * w + w0) * ldx + c], a11 = b[((int64_t)h1 * w + w1) * ldx + c];
            out[(((int64_t)n * C + c) * H + oh) * W + ow] = lh0 * (lw0 * a00 + lw1 * a01) + lh1 * (lw0 * a10 + lw1 * a11);
        }
    }
}

__global__ __launch_bounds__(256) void head_upsample_bwd_kernel(const float* __restrict__ dout, int N, int C, int H, int W,
                                                                float* dx, int64_t ldx, int h, int w, int accumulate,
                                                                float sh, float sw) {
    const int64_t total = (int64_t)N * h * w;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int iw = (int)(p % w), ih = (int)((p / w) % h), n = (int)(p / ((int64_t)w * h));
        int hlo, hhi, wlo, whi;
        bil_range(ih, sh, H, hlo, hhi);
        bil_range(iw, sw, W, wlo, whi);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int oh = hlo; oh <= hhi; ++oh) {
            const float wh = bil_weight(oh, sh, h, ih);
            if (wh == 0.f) continue;
            for (int ow = wlo; ow <= whi; ++ow) {
                const float ww = bil_weight(ow, sw, w, iw);
                if (ww == 0.f) continue;
                const float k = wh * ww;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < C) acc[c] += k * dout[(((int64_t)n * C + c) * H + oh) * W + ow];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) {
                float* d = dx + p * ldx + c;
                *d = accumulate ? (*d + acc[c]) : acc[c];
            }
    }
}

extern "C" int uda_head_upsample_fwd(const float* x, int64_t ldx, int N, int h, int w, int C, float* out, int H, int W,
                                     void* stream) {
    UDA_REQUIRE(x && out && C >= 1 && C <= 4 && ldx >= C && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "uda_head_upsample_fwd: bad args");
    int grid = uda_cdiv((int64_t)N * H * W, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(head_upsample_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, N, h, w, C, out, H, W,
                       bil_scale(h, H), bil_scale(w, W));
    UDA_LAUNCH_CHECK("head_upsample_fwd");
    return 0;
}

extern "C" int uda_head_upsample_bwd(const float* dout, int N, int C, int H, int W, float* dx, int64_t ldx, int h, int w,
                                     int accumulate, void* stream) {
    UDA_REQUIRE(dout && dx && C >= 1 && C <= 4 && ldx >= C && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "uda_head_upsample_bwd: bad args");
    int grid = uda_cdiv((int64_t)N * h * w, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(head_upsample_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, N, C, H, W, dx, ldx, h, w,
                       accumulate, bil_scale(h, H), bil_scale(w, W));
    UDA_LAUNCH_CHECK("head_upsample_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------
// out[n, c] = scale * sum_{pixels of image n} x[p, c];  one workgroup per (image, 256-channel block)
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, int64_t ldx, int HW, int C, float scale,
                                                  float* __restrict__ out, int64_t ldo) {
    __shared__ float red[4][64 * 4];
    const int n = blockIdx.x, cb = blockIdx.y * 256;
    const int Cb = min(256, C - cb), G = (Cb + 3) >> 2;     // <= 64 groups
    const int tid = threadIdx.x, cg = tid % 64, pl = tid / 64;   // 4 pixel lanes
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (cg < G) {
        const int c0 = cb + cg * 4;
        // eight independent partial sums: eight loads in flight (one workgroup walks a whole image: with a single chain it ran at one
        // load latency per pixel, 72 us for 21 MB)
        float part[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[u][j] = 0.f;
        const float* base = x + (int64_t)n * HW * ldx + c0;
        int p = pl;
        for (; p + 28 < HW; p += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 v = uda_ld4(base + (int64_t)(p + 4 * u) * ldx);
                part[u][0] += v.x; part[u][1] += v.y; part[u][2] += v.z; part[u][3] += v.w;
            }
        }
        for (; p < HW; p += 4) {
            const float4 v = uda_ld4(base + (int64_t)p * ldx);
            part[0][0] += v.x; part[0][1] += v.y; part[0][2] += v.z; part[0][3] += v.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] = ((part[0][j] + part[1][j]) + (part[2][j] + part[3][j])) + ((part[4][j] + part[5][j]) + (part[6][j] + part[7][j]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[pl][cg * 4 + j] = acc[j];
    __syncthreads();
    if (tid < Cb) out[(int64_t)n * ldo + cb + tid] = scale * (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

extern "C" int uda_gap_fwd(const float* x, int64_t ldx, int N, int HW, int C, float scale, float* out, int64_t ldo, void* stream) {
    UDA_REQUIRE(x && out && uda_aligned16(x) && ldx % 4 == 0 && ldx >= ((C + 3) / 4) * 4 && N > 0 && HW > 0 && C > 0 && ldo >= C,
                "uda_gap_fwd: x must be 16-byte aligned with ldx %% 4 == 0");
    hipLaunchKernelGGL(gap_kernel, dim3(N, uda_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, HW, C, scale, out, ldo);
    UDA_LAUNCH_CHECK("gap");
    return 0;
}

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ g, int64_t ldg, int N, int HW, int C,
                                                             float scale, const float* addend, int64_t ld_add, float* out,
                                                             int64_t ldo) {
    const int G = C >> 2;
    const int64_t total = (int64_t)N * HW * G;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const int n = (int)(p / HW);
        const float4 v = uda_ld4(g + (int64_t)n * ldg + cg * 4);
        float4 r = make_float4(scale * v.x, scale * v.y, scale * v.z, scale * v.w);
        if (addend) {
            const float4 a = uda_ld4(addend + p * ld_add + cg * 4);
            r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
        }
        uda_st4(out + p * ldo + cg * 4, r);
    }
}

extern "C" int uda_broadcast_rows(const float* g, int64_t ldg, int N, int HW, int C, float scale, const float* addend,
                                  int64_t ld_add, float* out, int64_t ldo, void* stream) {
    UDA_REQUIRE(g && out && uda_aligned16(g) && uda_aligned16(out) && ldg % 4 == 0 && ldo % 4 == 0 && C % 4 == 0 && C > 0,
                "uda_broadcast_rows: C and lds must be multiples of 4, pointers 16-byte aligned");
    if (addend) UDA_REQUIRE(uda_aligned16(addend) && ld_add % 4 == 0, "uda_broadcast_rows: bad addend layout");
    const int64_t total = (int64_t)N * HW * (C / 4);
    int grid = uda_cdiv(total, 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, ldg, N, HW, C, scale, addend,
                       ld_add, out, ldo);
    UDA_LAUNCH_CHECK("broadcast_rows");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10: one counter -> 8 keep bytes (4 x 32 random bits, one 16-bit compare each: the keep probability is exact to
// 2^-17, i.e. 6e-6 at p = 0.1 and exact at p = 0.5).  The generator, not the store, sets this kernel's rate (10 rounds of two
// 32 x 32 -> 64 bit multiplies), so every 32-bit word is spent on two elements.  The bit stream is implementation-defined.
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, int64_t ldm, int64_t P, int C,
                                                           uint32_t thresh16, uint64_t seed, uint64_t offset) {
    const int G = (C + 7) >> 3, C4 = ((C + 3) >> 2) << 2;
    const int64_t total = P * G;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        uint32_t c0 = (uint32_t)e, c1 = (uint32_t)(e >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        // keep with probability 1 - p: element 2j from the low half of word j, element 2j + 1 from its high half
        const uint32_t lo = ((c0 & 0xffffu) >= thresh16 ? 1u : 0u) | ((c0 >> 16) >= thresh16 ? 0x100u : 0u) |
                            ((c1 & 0xffffu) >= thresh16 ? 0x10000u : 0u) | ((c1 >> 16) >= thresh16 ? 0x1000000u : 0u);
        const uint32_t hi = ((c2 & 0xffffu) >= thresh16 ? 1u : 0u) | ((c2 >> 16) >= thresh16 ? 0x100u : 0u) |
                            ((c3 & 0xffffu) >= thresh16 ? 0x10000u : 0u) | ((c3 >> 16) >= thresh16 ? 0x1000000u : 0u);
        uint8_t* row = mask + p * ldm + cg * 8;
        *reinterpret_cast<uint32_t*>(row) = lo;
        if (cg * 8 + 4 < C4) *reinterpret_cast<uint32_t*>(row + 4) = hi;
    }
}

extern "C" int uda_dropout_mask(uint8_t* mask, int64_t ldm, int64_t P, int C, float p, uint64_t seed, uint64_t offset, void* stream) {
    UDA_REQUIRE(mask && (reinterpret_cast<uintptr_t>(mask) & 3u) == 0 && ldm % 4 == 0 && ldm >= ((C + 3) / 4) * 4 && P > 0 && C > 0,
                "uda_dropout_mask: mask must be 4-byte aligned with ldm %% 4 == 0 and >= round4(C)");
    UDA_REQUIRE(p >= 0.f && p < 1.f, "uda_dropout_mask: p must be in [0, 1)");
    const uint32_t thresh16 = (uint32_t)((double)p * 65536.0 + 0.5);
    int grid = uda_cdiv(P * ((C + 7) / 8), 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, ldm, P, C, thresh16, seed, offset);
    UDA_LAUNCH_CHECK("dropout_mask");
    return 0;
}
