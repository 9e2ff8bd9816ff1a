// Depthwise 3x3 convolution (MobileNetV2 inverted residual, mobilenet.py:39,53) and the 3->32
// stem convolution (mobilenet.py:10) for gfx950.  All of these are HBM-bound (1.7 / 9.8 FLOP per
// byte): the design goal is one coalesced 16-byte access per lane per tap, the BN
// affine + ReLU6 of the PRODUCER applied on load, and the BN statistics of THIS conv accumulated
// in the epilogue so the activation is written once and never re-read for normalisation.
//
// Thread mapping (shared by the depthwise kernels): channel group cg = tid % G (G = C/4 float4
// groups), pixel lane pl = tid / G; a workgroup walks ITER strips of PP = 256/G consecutive
// output pixels, so the 64 lanes of a wave read G*16 contiguous bytes per pixel.
#include <stdlib.h>
#include "common.h"

int uda_reduce_partials(const float* part, int nrows, int ncols, double* out, hipStream_t st);

#define DW_ITER_FWD 8
#define DW_ITER_RED 32

struct DwArgs {
    uda_src_t src;
    const float* w9c;       // [9][C]
    int stride, dil, border_mode;
    int Ho, Wo;
    float* y;               // fwd: output; wgrad: unused
    int64_t ldy;
    const float* dy;        // wgrad: gradient of the output
    int64_t lddy;
    float* part;            // wgrad: [nWG][9][C]
    double* stats;          // fwd: [UDA_STAT_SLOTS][2][C] or null (fp64 atomics)
};

__device__ __forceinline__ float4 dw_transform(float4 v, const Xf4& xf, bool has_xf, int act) {
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float u = r[j];
        if (has_xf) u = u * xf.sc[j] + xf.sh[j];
        r[j] = uda_act(u, act);
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

__global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwArgs a) {
    __shared__ float red[2 * 1024];
    const int C = a.src.C, G = C >> 2, PP = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    const bool active = pl < PP;
    const int H = a.src.H, W = a.src.W;
    const int64_t Pout = (int64_t)a.src.N * a.Ho * a.Wo;
    const int c0 = cg * 4;
    const bool has_xf = a.src.scale != nullptr;
    const int act = a.src.act;
    Xf4 xf;
    uda_load_xf4(xf, a.src.scale, a.src.shift, c0, C);
    float4 bval = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.border_mode == 1)
        bval = make_float4(uda_act(xf.sh[0], act), uda_act(xf.sh[1], act), uda_act(xf.sh[2], act), uda_act(xf.sh[3], act));
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = active ? uda_ld4(a.w9c + t * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const int64_t base = (int64_t)blockIdx.x * (PP * DW_ITER_FWD);
    for (int it = 0; it < DW_ITER_FWD; ++it) {
        const int64_t po = base + (int64_t)it * PP + pl;
        if (!active || po >= Pout) continue;
        const int ow = (int)(po % a.Wo), oh = (int)((po / a.Wo) % a.Ho), n = (int)(po / ((int64_t)a.Wo * a.Ho));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * a.stride + (kh - 1) * a.dil;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * a.stride + (kw - 1) * a.dil;
                float4 u = bval;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W)
                    u = dw_transform(uda_ld4(a.src.x + (((int64_t)n * H + ih) * W + iw) * a.src.ldx + c0), xf, has_xf, act);
                const float4 ww = w[kh * 3 + kw];
                acc.x += ww.x * u.x; acc.y += ww.y * u.y; acc.z += ww.z * u.z; acc.w += ww.w * u.w;
            }
        }
        uda_st4(a.y + po * a.ldy + c0, acc);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x += acc.x * acc.x; s2.y += acc.y * acc.y; s2.z += acc.z * acc.z; s2.w += acc.w * acc.w;
    }
    if (a.stats == nullptr) return;
    if (active) {
        uda_st4(&red[(pl * 2 + 0) * C + c0], s1);
        uda_st4(&red[(pl * 2 + 1) * C + c0], s2);
    }
    __syncthreads();
    double* dst = a.stats + (int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 2 * C;
    for (int e = tid; e < 2 * C; e += 256) {
        float t = 0.f;
        for (int p = 0; p < PP; ++p) t += red[p * 2 * C + e];
        atomicAdd(&dst[e], (double)t);
    }
}

// ------------------------------------------------------------------------------------------
// LDS-tiled forward: a workgroup owns TH x TW output pixels x 32 channels.  The input halo tile is
// loaded ONCE (16 B per lane, one 128-B line per pixel), the producer's BN affine + ReLU6 (or the
// quirk-Q1 border value) is applied ONCE per element on the way into LDS, and the 9 taps are read
// back with ds_read_b128.  (The first version re-read and re-transformed every input 9 times from
// L1/L2: 1.3-2.1 TB/s of algorithmic bytes; this one is bounded by the 1.3-1.6x halo over-read.)
#define DWT_CB 32
// Halo tile into LDS: every thread's loads are issued back to back with clamped coordinates (the value of an out-of-image position is
// replaced afterwards) - with the bounds test AROUND the load a thread had one load in flight at a time and a workgroup spent six to
// twelve load latencies on staging (the tiled kernels streamed at 2.4-3.7 TB/s).
template <int S, int TH, int TW>
__device__ __forceinline__ void dw_stage_tile(float* tile, const float* xb, int64_t ldx, int H, int W, int gy0, int gx0, int IH, int IW, bool cok,
                                              const Xf4& xf, bool has_xf, int act, float4 bval, int pl, int cg) {
    constexpr int MAXL = (((TH - 1) * S + 5) * ((TW - 1) * S + 5) + 31) / 32;      // dilation <= 2
    float4 v[MAXL];
    const int npix = IH * IW;
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        const int idx = min(pl + 32 * l, npix - 1);
        const int iy = idx / IW, ix = idx - iy * IW;
        const int gy = min(max(gy0 + iy, 0), H - 1), gx = min(max(gx0 + ix, 0), W - 1);
        v[l] = uda_ld4(xb + ((int64_t)gy * W + gx) * ldx);
    }
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        const int idx = pl + 32 * l;
        if (idx < npix) {
            const int iy = idx / IW, ix = idx - iy * IW;
            const int gy = gy0 + iy, gx = gx0 + ix;
            const bool in = cok && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const float4 u = in ? dw_transform(v[l], xf, has_xf, act) : bval;
            uda_st4(&tile[idx * DWT_CB + cg * 4], u);
        }
    }
}

template <int S, int TH, int TW>
__global__ __launch_bounds__(256) void dwconv_fwd_tiled_kernel(DwArgs a, int tilesX, int tilesY) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [IH*IW][32] then [32 lanes][2][32] for the stats
    const int d = a.dil;
    const int IH = (TH - 1) * S + 2 * d + 1, IW = (TW - 1) * S + 2 * d + 1;
    const int C = a.src.C, H = a.src.H, W = a.src.W;
    const int tid = threadIdx.x, cg = tid & 7, pl = tid >> 3;
    int bt = blockIdx.x;
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int c0 = blockIdx.y * DWT_CB + cg * 4;
    const bool cok = c0 < C;
    const bool has_xf = a.src.scale != nullptr;
    const int act = a.src.act;
    Xf4 xf;
    uda_load_xf4(xf, a.src.scale, a.src.shift, c0, C);
    float4 bval = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.border_mode == 1)
        bval = make_float4(uda_act(xf.sh[0], act), uda_act(xf.sh[1], act), uda_act(xf.sh[2], act), uda_act(xf.sh[3], act));
    // ---- stage the halo tile
    const int gy0 = oy0 * S - d, gx0 = ox0 * S - d;
    const float* xb = a.src.x + (int64_t)n * H * W * a.src.ldx + (cok ? c0 : 0);
    dw_stage_tile<S, TH, TW>(tile, xb, a.src.ldx, H, W, gy0, gx0, IH, IW, cok, xf, has_xf, act, bval, pl, cg);
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = cok ? uda_ld4(a.w9c + t * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- 3x3 from LDS
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int k = 0; k < (TH * TW) / 32; ++k) {
        const int op = pl + 32 * k;
        const int oy = op / TW, ox = op % TW;
        if (!cok || oy0 + oy >= a.Ho || ox0 + ox >= a.Wo) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float4 u = uda_ld4(&tile[((oy * S + kh * d) * IW + ox * S + kw * d) * DWT_CB + cg * 4]);
                const float4 ww = w[kh * 3 + kw];
                acc.x += ww.x * u.x; acc.y += ww.y * u.y; acc.z += ww.z * u.z; acc.w += ww.w * u.w;
            }
        uda_st4(a.y + (((int64_t)n * a.Ho + oy0 + oy) * a.Wo + ox0 + ox) * a.ldy + c0, acc);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x += acc.x * acc.x; s2.y += acc.y * acc.y; s2.z += acc.z * acc.z; s2.w += acc.w * acc.w;
    }
    if (a.stats == nullptr) return;
    __syncthreads();                       // the tile is dead: reuse it for the reduction
    uda_st4(&tile[(pl * 2 + 0) * DWT_CB + cg * 4], s1);
    uda_st4(&tile[(pl * 2 + 1) * DWT_CB + cg * 4], s2);
    __syncthreads();
    if (tid < 2 * DWT_CB) {
        const int q = tid / DWT_CB, cl = tid % DWT_CB;
        const int c = blockIdx.y * DWT_CB + cl;
        if (c < C) {
            float t = 0.f;
            for (int p = 0; p < 32; ++p) t += tile[(p * 2 + q) * DWT_CB + cl];
            atomicAdd(&a.stats[((int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 2 + q) * C + c], (double)t);
        }
    }
}

// Weight gradient on the same halo tile: dw[c][t] = sum_p u(p + off_t) * dy[p]; per-thread 9 x float4
// accumulators over its output pixels, a 32-lane LDS reduction, then fp64 atomics into slot replicas.
template <int S, int TH, int TW>
__global__ __launch_bounds__(256) void dwconv_wgrad_tiled_kernel(DwArgs a, int tilesX, int tilesY, double* sums) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // max([IH*IW][32], [32 lanes][9][32])
    const int d = a.dil;
    const int IH = (TH - 1) * S + 2 * d + 1, IW = (TW - 1) * S + 2 * d + 1;
    const int C = a.src.C, H = a.src.H, W = a.src.W;
    const int tid = threadIdx.x, cg = tid & 7, pl = tid >> 3;
    int bt = blockIdx.x;
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int c0 = blockIdx.y * DWT_CB + cg * 4;
    const bool cok = c0 < C;
    const bool has_xf = a.src.scale != nullptr;
    const int act = a.src.act;
    Xf4 xf;
    uda_load_xf4(xf, a.src.scale, a.src.shift, c0, C);
    float4 bval = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.border_mode == 1)
        bval = make_float4(uda_act(xf.sh[0], act), uda_act(xf.sh[1], act), uda_act(xf.sh[2], act), uda_act(xf.sh[3], act));
    const int gy0 = oy0 * S - d, gx0 = ox0 * S - d;
    const float* xb = a.src.x + (int64_t)n * H * W * a.src.ldx + (cok ? c0 : 0);
    dw_stage_tile<S, TH, TW>(tile, xb, a.src.ldx, H, W, gy0, gx0, IH, IW, cok, xf, has_xf, act, bval, pl, cg);
    __syncthreads();
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gv[(TH * TW) / 32];           // this thread's dy values, loaded together (clamped positions, zero weight outside)
#pragma unroll
    for (int k = 0; k < (TH * TW) / 32; ++k) {
        const int op = pl + 32 * k;
        const int oy = min(oy0 + op / TW, a.Ho - 1), ox = min(ox0 + op % TW, a.Wo - 1);
        gv[k] = uda_ld4(a.dy + (((int64_t)n * a.Ho + oy) * a.Wo + ox) * a.lddy + (cok ? c0 : 0));
    }
#pragma unroll
    for (int k = 0; k < (TH * TW) / 32; ++k) {
        const int op = pl + 32 * k;
        const int oy = op / TW, ox = op % TW;
        if (!cok || oy0 + oy >= a.Ho || ox0 + ox >= a.Wo) continue;
        const float4 g = gv[k];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float4 u = uda_ld4(&tile[((oy * S + kh * d) * IW + ox * S + kw * d) * DWT_CB + cg * 4]);
                float4& s = acc[kh * 3 + kw];
                s.x += g.x * u.x; s.y += g.y * u.y; s.z += g.z * u.z; s.w += g.w * u.w;
            }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) uda_st4(&tile[(pl * 9 + t) * DWT_CB + cg * 4], acc[t]);
    __syncthreads();
    for (int e = tid; e < 9 * DWT_CB; e += 256) {
        const int t = e / DWT_CB, cl = e % DWT_CB;
        const int c = blockIdx.y * DWT_CB + cl;
        if (c < C) {
            float v = 0.f;
            for (int p = 0; p < 32; ++p) v += tile[(p * 9 + t) * DWT_CB + cl];
            atomicAdd(&sums[((int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 9 + t) * C + c], (double)v);
        }
    }
}

// dw[c][t] (float) = sum over slots of sums[slot][t][c] (double)
__global__ void dw_wgrad_store_slots_kernel(const double* __restrict__ sums, int C, float* __restrict__ dw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 9 * C) {
        double v = 0.0;
        for (int k = 0; k < UDA_STAT_SLOTS; ++k) v += sums[(int64_t)k * 9 * C + e];
        dw[(e % C) * 9 + e / C] = (float)v;
    }
}

template <int S, int TH, int TW>
static int launch_dw_wgrad_tiled(DwArgs& a, double* sums, float* dw, hipStream_t st) {
    const int d = a.dil;
    const int IH = (TH - 1) * S + 2 * d + 1, IW = (TW - 1) * S + 2 * d + 1;
    size_t lds = (size_t)IH * IW * DWT_CB * sizeof(float);
    if (lds < 32 * 9 * DWT_CB * sizeof(float)) lds = 32 * 9 * DWT_CB * sizeof(float);
    auto fn = dwconv_wgrad_tiled_kernel<S, TH, TW>;
    static size_t reserved = 0;
    if (lds > reserved && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return uda_set_error("dwconv_wgrad: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        reserved = lds;
    }
    const int C = a.src.C;
    (void)hipMemsetAsync(sums, 0, (size_t)UDA_STAT_SLOTS * 9 * C * sizeof(double), st);
    const int tilesX = uda_cdiv(a.Wo, TW), tilesY = uda_cdiv(a.Ho, TH);
    hipLaunchKernelGGL(fn, dim3(tilesX * tilesY * a.src.N, uda_cdiv(C, DWT_CB)), dim3(256), lds, st, a, tilesX, tilesY, sums);
    UDA_LAUNCH_CHECK("dwconv_wgrad_tiled");
    hipLaunchKernelGGL(dw_wgrad_store_slots_kernel, dim3(uda_cdiv(9 * C, 256)), dim3(256), 0, st, sums, C, dw);
    UDA_LAUNCH_CHECK("dw_wgrad_store");
    return 0;
}

template <int S, int TH, int TW>
static int launch_dw_tiled(DwArgs& a, hipStream_t st) {
    const int d = a.dil;
    const int IH = (TH - 1) * S + 2 * d + 1, IW = (TW - 1) * S + 2 * d + 1;
    size_t lds = (size_t)IH * IW * DWT_CB * sizeof(float);
    if (lds < 32 * 2 * DWT_CB * sizeof(float)) lds = 32 * 2 * DWT_CB * sizeof(float);
    auto fn = dwconv_fwd_tiled_kernel<S, TH, TW>;
    static size_t reserved = 0;
    if (lds > reserved && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return uda_set_error("dwconv_fwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        reserved = lds;
    }
    const int tilesX = uda_cdiv(a.Wo, TW), tilesY = uda_cdiv(a.Ho, TH);
    hipLaunchKernelGGL(fn, dim3(tilesX * tilesY * a.src.N, uda_cdiv(a.src.C, DWT_CB)), dim3(256), lds, st, a, tilesX, tilesY);
    UDA_LAUNCH_CHECK("dwconv_fwd_tiled");
    return 0;
}

// gradient w.r.t. the interior H x W positions of the (padded) depthwise input
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ w9c, int C, int stride,
                                                            int dil, int N, int H, int W, int Ho, int Wo,
                                                            float* __restrict__ dx, int64_t lddx) {
    const int G = C >> 2;
    const int64_t total = (int64_t)N * H * W * G;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const int iw = (int)(p % W), ih = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        const int c0 = cg * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (stride == 2 && dil == 1) {
            // stride 2 (the four down-sampling blocks): an even coordinate is read by the centre tap only, an odd one by the two outer
            // taps - at most 2 x 2 taps exist.  The four candidates are loaded back to back from clamped positions and weighted 0
            // where they do not exist (behind the general loop's tests a thread had one load in flight at a time).
            const int khA = (ih & 1) ? 0 : 1, ohA = (ih + 1) >> 1, ohB = (ih - 1) >> 1;        // tap khA reads oh = (ih + 1 - khA) / 2; B: kh = 2
            const int kwA = (iw & 1) ? 0 : 1, owA = (iw + 1) >> 1, owB = (iw - 1) >> 1;
            const int oha = (ih & 1) ? ohA : (ih >> 1), owa = (iw & 1) ? owA : (iw >> 1);
            const bool hA = oha < Ho, hB = (ih & 1) && ohB >= 0 && ohB < Ho;
            const bool wA = owa < Wo, wB = (iw & 1) && owB >= 0 && owB < Wo;
            const int r0 = min(oha, Ho - 1), r1 = min(max(ohB, 0), Ho - 1), q0 = min(owa, Wo - 1), q1 = min(max(owB, 0), Wo - 1);
            const float* base = dy + (int64_t)n * Ho * Wo * lddy + c0;
            const float4 g00 = uda_ld4(base + ((int64_t)r0 * Wo + q0) * lddy), g01 = uda_ld4(base + ((int64_t)r0 * Wo + q1) * lddy);
            const float4 g10 = uda_ld4(base + ((int64_t)r1 * Wo + q0) * lddy), g11 = uda_ld4(base + ((int64_t)r1 * Wo + q1) * lddy);
            const float4 w00 = uda_ld4(w9c + (khA * 3 + kwA) * C + c0), w01 = uda_ld4(w9c + (khA * 3 + 2) * C + c0);
            const float4 w10 = uda_ld4(w9c + (2 * 3 + kwA) * C + c0), w11 = uda_ld4(w9c + (2 * 3 + 2) * C + c0);
            const float k00 = (hA && wA) ? 1.f : 0.f, k01 = (hA && wB) ? 1.f : 0.f, k10 = (hB && wA) ? 1.f : 0.f, k11 = (hB && wB) ? 1.f : 0.f;
            // same order as the general loop: kh ascending, kw ascending
            acc.x = k00 * (w00.x * g00.x); acc.y = k00 * (w00.y * g00.y); acc.z = k00 * (w00.z * g00.z); acc.w = k00 * (w00.w * g00.w);
            acc.x += k01 * (w01.x * g01.x); acc.y += k01 * (w01.y * g01.y); acc.z += k01 * (w01.z * g01.z); acc.w += k01 * (w01.w * g01.w);
            acc.x += k10 * (w10.x * g10.x); acc.y += k10 * (w10.y * g10.y); acc.z += k10 * (w10.z * g10.z); acc.w += k10 * (w10.w * g10.w);
            acc.x += k11 * (w11.x * g11.x); acc.y += k11 * (w11.y * g11.y); acc.z += k11 * (w11.z * g11.z); acc.w += k11 * (w11.w * g11.w);
            uda_st4(dx + p * lddx + c0, acc);
            continue;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int nh = ih - (kh - 1) * dil;
            if (nh < 0 || nh % stride) continue;
            const int oh = nh / stride;
            if (oh >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int nw = iw - (kw - 1) * dil;
                if (nw < 0 || nw % stride) continue;
                const int ow = nw / stride;
                if (ow >= Wo) continue;
                const float4 g = uda_ld4(dy + (((int64_t)n * Ho + oh) * Wo + ow) * lddy + c0);
                const float4 ww = uda_ld4(w9c + (kh * 3 + kw) * C + c0);
                acc.x += ww.x * g.x; acc.y += ww.y * g.y; acc.z += ww.z * g.z; acc.w += ww.w * g.w;
            }
        }
        uda_st4(dx + p * lddx + c0, acc);
    }
}

__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwArgs a) {
    __shared__ float red[9 * 1024];
    const int C = a.src.C, G = C >> 2, PP = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    const bool active = pl < PP;
    const int H = a.src.H, W = a.src.W;
    const int64_t Pout = (int64_t)a.src.N * a.Ho * a.Wo;
    const int c0 = cg * 4;
    const bool has_xf = a.src.scale != nullptr;
    const int act = a.src.act;
    Xf4 xf;
    uda_load_xf4(xf, a.src.scale, a.src.shift, c0, C);
    float4 bval = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.border_mode == 1)
        bval = make_float4(uda_act(xf.sh[0], act), uda_act(xf.sh[1], act), uda_act(xf.sh[2], act), uda_act(xf.sh[3], act));
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t base = (int64_t)blockIdx.x * (PP * DW_ITER_RED);
    for (int it = 0; it < DW_ITER_RED; ++it) {
        const int64_t po = base + (int64_t)it * PP + pl;
        if (!active || po >= Pout) continue;
        const int ow = (int)(po % a.Wo), oh = (int)((po / a.Wo) % a.Ho), n = (int)(po / ((int64_t)a.Wo * a.Ho));
        const float4 g = uda_ld4(a.dy + po * a.lddy + c0);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * a.stride + (kh - 1) * a.dil;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * a.stride + (kw - 1) * a.dil;
                float4 u = bval;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W)
                    u = dw_transform(uda_ld4(a.src.x + (((int64_t)n * H + ih) * W + iw) * a.src.ldx + c0), xf, has_xf, act);
                float4& s = acc[kh * 3 + kw];
                s.x += g.x * u.x; s.y += g.y * u.y; s.z += g.z * u.z; s.w += g.w * u.w;
            }
        }
    }
    if (active) {
#pragma unroll
        for (int t = 0; t < 9; ++t) uda_st4(&red[(pl * 9 + t) * C + c0], acc[t]);
    }
    __syncthreads();
    for (int e = tid; e < 9 * C; e += 256) {
        float t = 0.f;
        for (int p = 0; p < PP; ++p) t += red[p * 9 * C + e];
        a.part[(int64_t)blockIdx.x * 9 * C + e] = t;
    }
}

// dw[c][t] (float) = sums[t][c] (double)
__global__ void dw_wgrad_store_kernel(const double* __restrict__ sums, int C, float* __restrict__ dw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 9 * C) dw[(e % C) * 9 + e / C] = (float)sums[e];
}

static int dw_check(const uda_src_t* s, const char* who) {
    UDA_REQUIRE(s && s->x && uda_aligned16(s->x) && s->ldx % 4 == 0 && s->ldx >= s->C, "%s: bad src", who);
    UDA_REQUIRE(s->C % 4 == 0 && s->C >= 4 && s->C <= 1024, "%s: C=%d must be a multiple of 4 in [4,1024]", who, s->C);
    UDA_REQUIRE((s->scale == nullptr) == (s->shift == nullptr), "%s: scale/shift must come together", who);
    UDA_REQUIRE(s->mask == nullptr, "%s: dropout masks are not supported on depthwise inputs", who);
    return 0;
}

static inline int dw_pixels_per_wg(int C, int iter) { return (256 / (C / 4)) * iter; }

extern "C" uint64_t uda_dwconv_workspace_bytes(int64_t Pout, int C) {
    if (C < 4) return 0;
    const uint64_t flat = (uint64_t)uda_cdiv(Pout, dw_pixels_per_wg(C, DW_ITER_RED)) * 9 * C * sizeof(float) + 9 * C * sizeof(double);
    const uint64_t tiled = (uint64_t)UDA_STAT_SLOTS * 9 * C * sizeof(double);
    return flat > tiled ? flat : tiled;
}

extern "C" int uda_dwconv_fwd(const uda_src_t* src, const float* w9c, int stride, int dil, int border_mode,
                              float* y, int64_t ldy, double* stats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (int e = dw_check(src, "uda_dwconv_fwd")) return e;
    UDA_REQUIRE(w9c && uda_aligned16(w9c) && y && uda_aligned16(y) && ldy % 4 == 0 && ldy >= src->C, "uda_dwconv_fwd: bad pointers");
    UDA_REQUIRE((stride == 1 || stride == 2) && dil >= 1, "uda_dwconv_fwd: stride must be 1 or 2");
    UDA_REQUIRE(border_mode == 0 || (border_mode == 1 && src->shift), "uda_dwconv_fwd: border_mode 1 needs shift");
    DwArgs a;
    a.src = *src;
    a.w9c = w9c;
    a.stride = stride; a.dil = dil; a.border_mode = border_mode;
    a.Ho = (src->H - 1) / stride + 1;
    a.Wo = (src->W - 1) / stride + 1;
    a.y = y; a.ldy = ldy; a.dy = nullptr; a.lddy = 0;
    const int64_t Pout = (int64_t)src->N * a.Ho * a.Wo;
    const int nwg = uda_cdiv(Pout, dw_pixels_per_wg(src->C, DW_ITER_FWD));
    a.part = nullptr;
    a.stats = stats;
    static const bool flat = getenv("UDA_DW_FLAT") != nullptr;      // diagnostics: the untiled kernel
    if (!flat && dil <= 2) return stride == 1 ? launch_dw_tiled<1, 8, 16>(a, st) : launch_dw_tiled<2, 8, 8>(a, st);
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(nwg), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("dwconv_fwd");
    return 0;
}

extern "C" int uda_dwconv_dgrad(const float* dy, int64_t lddy, const float* w9c, int C, int stride, int dil,
                                int N, int H, int W, float* dx, int64_t lddx, void* stream) {
    UDA_REQUIRE(dy && w9c && dx && uda_aligned16(dy) && uda_aligned16(dx) && uda_aligned16(w9c), "uda_dwconv_dgrad: pointers must be 16-byte aligned");
    UDA_REQUIRE(C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && lddy >= C && lddx >= C, "uda_dwconv_dgrad: C and lds must be multiples of 4");
    UDA_REQUIRE((stride == 1 || stride == 2) && dil >= 1 && N > 0 && H > 0 && W > 0, "uda_dwconv_dgrad: bad geometry");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t total = (int64_t)N * H * W * (C / 4);
    int grid = uda_cdiv(total, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(dwconv_dgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, lddy, w9c, C, stride, dil,
                       N, H, W, Ho, Wo, dx, lddx);
    UDA_LAUNCH_CHECK("dwconv_dgrad");
    return 0;
}

extern "C" int uda_dwconv_wgrad(const uda_src_t* src, const float* dy, int64_t lddy, int stride, int dil,
                                int border_mode, float* dw, float* workspace, uint64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (int e = dw_check(src, "uda_dwconv_wgrad")) return e;
    UDA_REQUIRE(dy && uda_aligned16(dy) && lddy % 4 == 0 && lddy >= src->C && dw, "uda_dwconv_wgrad: bad pointers");
    UDA_REQUIRE((stride == 1 || stride == 2) && dil >= 1, "uda_dwconv_wgrad: stride must be 1 or 2");
    UDA_REQUIRE(border_mode == 0 || (border_mode == 1 && src->shift), "uda_dwconv_wgrad: border_mode 1 needs shift");
    DwArgs a;
    a.src = *src;
    a.w9c = nullptr;
    a.stride = stride; a.dil = dil; a.border_mode = border_mode;
    a.Ho = (src->H - 1) / stride + 1;
    a.Wo = (src->W - 1) / stride + 1;
    a.y = nullptr; a.ldy = 0; a.dy = dy; a.lddy = lddy; a.stats = nullptr;
    const int C = src->C;
    const int64_t Pout = (int64_t)src->N * a.Ho * a.Wo;
    const int nwg = uda_cdiv(Pout, dw_pixels_per_wg(C, DW_ITER_RED));
    UDA_REQUIRE(workspace && workspace_bytes >= uda_dwconv_workspace_bytes(Pout, C), "uda_dwconv_wgrad: workspace too small");
    static const bool flat = getenv("UDA_DW_FLAT") != nullptr;
    if (!flat && dil <= 2) {
        double* slot_sums = reinterpret_cast<double*>(workspace);
        return stride == 1 ? launch_dw_wgrad_tiled<1, 8, 16>(a, slot_sums, dw, st) : launch_dw_wgrad_tiled<2, 8, 8>(a, slot_sums, dw, st);
    }
    double* sums = reinterpret_cast<double*>(workspace);          // [9][C] first (8-byte aligned)
    a.part = workspace + 2 * 9 * C;
    (void)hipMemsetAsync(sums, 0, 9 * C * sizeof(double), st);
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(nwg), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("dwconv_wgrad");
    if (int e = uda_reduce_partials(a.part, nwg, 9 * C, sums, st)) return e;
    hipLaunchKernelGGL(dw_wgrad_store_kernel, dim3(uda_cdiv(9 * C, 256)), dim3(256), 0, st, sums, C, dw);
    UDA_LAUNCH_CHECK("dw_wgrad_store");
    return 0;
}

// ==========================================================================================
// Stem: 3x3 stride 2 pad 1, 3 -> 32, NCHW image -> NHWC.  8 lanes (4 channels each) per output
// pixel: a wave writes 8 pixels x 128 B contiguous; the 27-tap window is re-read from L1.
struct StemArgs {
    const float* x;   // [N][3][H][W]
    int N, H, W, Ho, Wo;
    const float* w;   // [32][3][3][3]
    float* y;
    int64_t ldy;
    float* part;      // wgrad: [nWG][864]
    double* stats;    // fwd: [UDA_STAT_SLOTS][2][32] or null
    const float* dy;
    int64_t lddy;
};

#define STEM_PIX_PER_WG 256   // 32 pixels per pass x 8 passes

__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
    __shared__ float wsm[27 * 32];     // [tap][co]
    __shared__ float red[32 * 2 * 32];
    const int tid = threadIdx.x;
    for (int e = tid; e < 27 * 32; e += 256) wsm[e] = a.w[(e % 32) * 27 + e / 32];
    __syncthreads();
    const int cg = tid & 7, pl = tid >> 3, c0 = cg * 4;
    const int64_t Pout = (int64_t)a.N * a.Ho * a.Wo;
    const int64_t plane = (int64_t)a.H * a.W;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (int it = 0; it < 8; ++it) {
        const int64_t po = (int64_t)blockIdx.x * STEM_PIX_PER_WG + it * 32 + pl;
        if (po >= Pout) continue;
        const int ow = (int)(po % a.Wo), oh = (int)((po / a.Wo) % a.Ho), n = (int)(po / ((int64_t)a.Wo * a.Ho));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = oh * 2 - 1 + kh;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = ow * 2 - 1 + kw;
                    float v = 0.f;
                    if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) v = a.x[((int64_t)n * 3 + ci) * plane + (int64_t)ih * a.W + iw];
                    const float4 ww = uda_ld4(&wsm[(ci * 9 + kh * 3 + kw) * 32 + c0]);
                    acc.x += ww.x * v; acc.y += ww.y * v; acc.z += ww.z * v; acc.w += ww.w * v;
                }
            }
        uda_st4(a.y + po * a.ldy + c0, acc);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x += acc.x * acc.x; s2.y += acc.y * acc.y; s2.z += acc.z * acc.z; s2.w += acc.w * acc.w;
    }
    if (a.stats == nullptr) return;
    uda_st4(&red[(pl * 2 + 0) * 32 + c0], s1);
    uda_st4(&red[(pl * 2 + 1) * 32 + c0], s2);
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
        for (int p = 0; p < 32; ++p) t += red[p * 64 + tid];
        atomicAdd(&a.stats[(int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 64 + tid], (double)t);
    }
}

// dw[co][ci][kh][kw] partials: thread = (co = tid&31, tap group tg = tid>>5 -> taps tg, tg+8, tg+16, tg+24)
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemArgs a) {
    __shared__ float xs[32][28];     // per pass: 32 pixels x 27 taps
    __shared__ float gs[32][33];     // 32 pixels x 32 channels
    const int tid = threadIdx.x, co = tid & 31, tg = tid >> 5;
    const int64_t Pout = (int64_t)a.N * a.Ho * a.Wo;
    const int64_t plane = (int64_t)a.H * a.W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < 8; ++it) {
        const int64_t pbase = (int64_t)blockIdx.x * STEM_PIX_PER_WG + it * 32;
        __syncthreads();
        for (int e = tid; e < 32 * 27; e += 256) {
            const int pp = e / 27, tap = e % 27;
            const int64_t po = pbase + pp;
            float v = 0.f;
            if (po < Pout) {
                const int ow = (int)(po % a.Wo), oh = (int)((po / a.Wo) % a.Ho), n = (int)(po / ((int64_t)a.Wo * a.Ho));
                const int ci = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) v = a.x[((int64_t)n * 3 + ci) * plane + (int64_t)ih * a.W + iw];
            }
            xs[pp][tap] = v;
        }
        for (int e = tid; e < 32 * 32; e += 256) {
            const int pp = e >> 5, c = e & 31;
            const int64_t po = pbase + pp;
            gs[pp][c] = po < Pout ? a.dy[po * a.lddy + c] : 0.f;
        }
        __syncthreads();
        for (int pp = 0; pp < 32; ++pp) {
            const float g = gs[pp][co];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tap = tg + 8 * q;
                if (tap < 27) acc[q] += g * xs[pp][tap];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tap = tg + 8 * q;
        if (tap < 27) a.part[(int64_t)blockIdx.x * 864 + co * 27 + tap] = acc[q];
    }
}

// Row-staged variants for output rows that are a multiple of 256 pixels wide (the 512 x 512 training images: Wo = 256): a workgroup
// owns 256 consecutive pixels of ONE output row, stages the three input rows x three channels it reads (513 columns each) in LDS
// with coalesced loads, and computes from there.  (The generic kernels above fetch every tap with its own strided global load:
// 27 loads per 16 bytes of output, 1.0-1.1 TB/s; these run at the rate of their 184 MB of traffic.)
#define STEM_ROW_LD 516
__device__ __forceinline__ void stem_stage_rows(const StemArgs& a, int n, int oh, int ow0, float* xin) {
    const int64_t plane = (int64_t)a.H * a.W;
    for (int e = threadIdx.x; e < 9 * 513; e += 256) {
        const int r = e / 513, j = e - r * 513;                 // r = ci * 3 + kh
        const int ci = r / 3, kh = r - ci * 3;
        const int ih = oh * 2 - 1 + kh, iw = ow0 * 2 - 1 + j;
        float v = 0.f;
        if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) v = a.x[((int64_t)n * 3 + ci) * plane + (int64_t)ih * a.W + iw];
        xin[r * STEM_ROW_LD + j] = v;
    }
}

__global__ __launch_bounds__(256) void stem_fwd_rows_kernel(StemArgs a) {
    __shared__ float wsm[27 * 32];     // [tap][co]
    __shared__ float xin[9 * STEM_ROW_LD];
    __shared__ float red[32 * 2 * 32];
    const int tid = threadIdx.x;
    const int segs = a.Wo >> 8, seg = blockIdx.x % segs, row = blockIdx.x / segs, oh = row % a.Ho, n = row / a.Ho, ow0 = seg << 8;
    for (int e = tid; e < 27 * 32; e += 256) wsm[e] = a.w[(e % 32) * 27 + e / 32];
    stem_stage_rows(a, n, oh, ow0, xin);
    __syncthreads();
    const int cg = tid & 7, pl = tid >> 3, c0 = cg * 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const int64_t pbase = ((int64_t)n * a.Ho + oh) * a.Wo + ow0;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int px = it * 32 + pl;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float v = xin[r * STEM_ROW_LD + 2 * px + kw];
                const float4 ww = uda_ld4(&wsm[(r * 3 + kw) * 32 + c0]);
                acc.x += ww.x * v; acc.y += ww.y * v; acc.z += ww.z * v; acc.w += ww.w * v;
            }
        uda_st4(a.y + (pbase + px) * a.ldy + c0, acc);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x += acc.x * acc.x; s2.y += acc.y * acc.y; s2.z += acc.z * acc.z; s2.w += acc.w * acc.w;
    }
    if (a.stats == nullptr) return;
    uda_st4(&red[(pl * 2 + 0) * 32 + c0], s1);
    uda_st4(&red[(pl * 2 + 1) * 32 + c0], s2);
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
        for (int p = 0; p < 32; ++p) t += red[p * 64 + tid];
        atomicAdd(&a.stats[(int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 64 + tid], (double)t);
    }
}

__global__ __launch_bounds__(256) void stem_wgrad_rows_kernel(StemArgs a) {
    __shared__ float xin[9 * STEM_ROW_LD];
    __shared__ float gs[256 * 33];     // 256 pixels x 32 channels
    const int tid = threadIdx.x, co = tid & 31, tg = tid >> 5;
    const int segs = a.Wo >> 8, seg = blockIdx.x % segs, row = blockIdx.x / segs, oh = row % a.Ho, n = row / a.Ho, ow0 = seg << 8;
    const int64_t pbase = ((int64_t)n * a.Ho + oh) * a.Wo + ow0;
    stem_stage_rows(a, n, oh, ow0, xin);
    for (int e = tid; e < 256 * 8; e += 256) {
        const int pp = e >> 3, c4 = (e & 7) * 4;
        const float4 g = uda_ld4(a.dy + (pbase + pp) * a.lddy + c4);
        gs[pp * 33 + c4] = g.x; gs[pp * 33 + c4 + 1] = g.y; gs[pp * 33 + c4 + 2] = g.z; gs[pp * 33 + c4 + 3] = g.w;
    }
    __syncthreads();
    int xo[4];                          // LDS offset of tap tg + 8q at pixel 0 (a tap beyond 26 reads tap 0 and is not stored)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tap = tg + 8 * q < 27 ? tg + 8 * q : 0;
        xo[q] = (tap / 3) * STEM_ROW_LD + tap % 3;               // tap = (ci * 3 + kh) * 3 + kw
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int pp = 0; pp < 256; ++pp) {
        const float g = gs[pp * 33 + co];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += g * xin[xo[q] + 2 * pp];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tap = tg + 8 * q;
        if (tap < 27) a.part[(int64_t)blockIdx.x * 864 + co * 27 + tap] = acc[q];
    }
}

__global__ void cast_d2f_kernel(const double* __restrict__ in, int n, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = (float)in[e];
}

extern "C" uint64_t uda_stem_workspace_bytes(int64_t Pout) {
    return (uint64_t)uda_cdiv(Pout, STEM_PIX_PER_WG) * 864 * sizeof(float) + 864 * sizeof(double);
}

extern "C" int uda_stem_fwd(const float* x, int N, int H, int W, const float* w, float* y, int64_t ldy,
                            double* stats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(x && w && y && uda_aligned16(y) && ldy % 4 == 0 && ldy >= 32 && N > 0 && H > 1 && W > 1, "uda_stem_fwd: bad args");
    StemArgs a;
    a.x = x; a.N = N; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.w = w; a.y = y; a.ldy = ldy; a.dy = nullptr; a.lddy = 0;
    const int64_t Pout = (int64_t)N * a.Ho * a.Wo;
    const int nwg = uda_cdiv(Pout, STEM_PIX_PER_WG);
    a.part = nullptr;
    a.stats = stats;
    if (a.Wo % 256 == 0) hipLaunchKernelGGL(stem_fwd_rows_kernel, dim3(nwg), dim3(256), 0, st, a);       // nwg = N * Ho * (Wo / 256)
    else hipLaunchKernelGGL(stem_fwd_kernel, dim3(nwg), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("stem_fwd");
    return 0;
}

extern "C" int uda_stem_wgrad(const float* x, int N, int H, int W, const float* dy, int64_t lddy, float* dw,
                              float* workspace, uint64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(x && dy && dw && lddy >= 32 && N > 0 && H > 1 && W > 1, "uda_stem_wgrad: bad args");
    StemArgs a;
    a.x = x; a.N = N; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.w = nullptr; a.y = nullptr; a.ldy = 0; a.dy = dy; a.lddy = lddy; a.stats = nullptr;
    const int64_t Pout = (int64_t)N * a.Ho * a.Wo;
    const int nwg = uda_cdiv(Pout, STEM_PIX_PER_WG);
    UDA_REQUIRE(workspace && workspace_bytes >= uda_stem_workspace_bytes(Pout), "uda_stem_wgrad: workspace too small");
    double* sums = reinterpret_cast<double*>(workspace);
    a.part = workspace + 2 * 864;
    (void)hipMemsetAsync(sums, 0, 864 * sizeof(double), st);
    if (a.Wo % 256 == 0 && uda_aligned16(dy) && lddy % 4 == 0) hipLaunchKernelGGL(stem_wgrad_rows_kernel, dim3(nwg), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nwg), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("stem_wgrad");
    if (int e = uda_reduce_partials(a.part, nwg, 864, sums, st)) return e;
    hipLaunchKernelGGL(cast_d2f_kernel, dim3(uda_cdiv(864, 256)), dim3(256), 0, st, sums, 864, dw);
    UDA_LAUNCH_CHECK("stem_wgrad_store");
    return 0;
}
