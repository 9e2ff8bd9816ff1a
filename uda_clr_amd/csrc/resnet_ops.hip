// ResNet-101 stem and geometry helpers for gfx950 (networks/backbone/resnet.py of the reference):
//   stem7_fwd / stem7_wgrad   conv1 7x7 stride 2 pad 3, 3 -> 64, NCHW image in, NHWC out (resnet.py:59,114)
//   maxpool_fwd / maxpool_bwd MaxPool2d(3, 2, 1) on the pending relu(bn1(.)) (resnet.py:63,117)
//   rows_stride               pick every s-th pixel / its transpose (zero insertion): how the two
//                             stride-2 bottlenecks (layer2.0, layer3.0) use the stride-1 MFMA kernels
// All HBM / VALU bound and < 3 % of the ResNet step; the bottleneck convs run on igemm_*.hip.
#include "common.h"

int uda_reduce_partials(const float* part, int nrows, int ncols, double* out, hipStream_t st);

// ------------------------------------------------------------------------------------------
struct Stem7Args {
    const float* x;   // [N][3][H][W]
    int N, H, W, Ho, Wo;
    const float* w;   // [64][3][7][7]
    float* y;
    int64_t ldy;
    double* stats;    // [UDA_STAT_SLOTS][2][64] or null
    const float* dy;
    int64_t lddy;
    float* part;      // wgrad: [nWG][9408]
};

#define S7_TAPS 147
#define S7_PIX_PER_WG 256     // 4 passes x 16 pixel quads x 4 pixels
#define S7W_PIX_PER_WG 1024   // 32 passes x 32 pixels

// thread = (cg = tid & 15: 4 output channels, pq = tid >> 4: a quad of 4 outputs adjacent along W).
// Per input row the 13 columns under the quad are loaded once and feed 7 taps x 4 pixels x 4 channels.
__global__ __launch_bounds__(256) void stem7_fwd_kernel(Stem7Args a) {
    __shared__ __attribute__((aligned(16))) float wsm[S7_TAPS * 64];     // [tap][co]
    __shared__ float red[16 * 2 * 64];
    const int tid = threadIdx.x;
    for (int e = tid; e < S7_TAPS * 64; e += 256) wsm[e] = a.w[(e & 63) * S7_TAPS + (e >> 6)];
    __syncthreads();
    const int cg = tid & 15, pq = tid >> 4, c0 = cg * 4;
    const int64_t Pout = (int64_t)a.N * a.Ho * a.Wo;
    const int64_t plane = (int64_t)a.H * a.W;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < 4; ++it) {
        const int64_t po = (int64_t)blockIdx.x * S7_PIX_PER_WG + (it * 16 + pq) * 4;
        if (po >= Pout) continue;
        const int ow0 = (int)(po % a.Wo), oh = (int)((po / a.Wo) % a.Ho), n = (int)(po / ((int64_t)a.Wo * a.Ho));
        float acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = 0.f;
        for (int ci = 0; ci < 3; ++ci)
            for (int kh = 0; kh < 7; ++kh) {
                const int ih = oh * 2 - 3 + kh;
                if (ih < 0 || ih >= a.H) continue;
                const float* row = a.x + ((int64_t)n * 3 + ci) * plane + (int64_t)ih * a.W;
                float xv[13];
#pragma unroll
                for (int t = 0; t < 13; ++t) {
                    const int iw = ow0 * 2 - 3 + t;
                    xv[t] = (iw >= 0 && iw < a.W) ? row[iw] : 0.f;
                }
                const float* wr = &wsm[(ci * 49 + kh * 7) * 64 + c0];
#pragma unroll
                for (int kw = 0; kw < 7; ++kw) {
                    const float4 ww = uda_ld4(wr + kw * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = xv[2 * j + kw];
                        acc[j][0] += ww.x * v; acc[j][1] += ww.y * v; acc[j][2] += ww.z * v; acc[j][3] += ww.w * v;
                    }
                }
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uda_st4(a.y + (po + j) * a.ldy + c0, make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += acc[j][c];
                s2[c] += acc[j][c] * acc[j][c];
            }
        }
    }
    if (a.stats == nullptr) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        red[(pq * 2 + 0) * 64 + c0 + c] = s1[c];
        red[(pq * 2 + 1) * 64 + c0 + c] = s2[c];
    }
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
        for (int p = 0; p < 16; ++p) t += red[p * 128 + tid];
        atomicAdd(&a.stats[(int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 128 + tid], (double)t);
    }
}

// dw[co][tap] partials: thread = (co = tid & 63, tg = tid >> 6 -> taps tg + 4q, q < 37); the tap value is
// wave-uniform so xs[][] reads are LDS broadcasts.
__global__ __launch_bounds__(256) void stem7_wgrad_kernel(Stem7Args a) {
    __shared__ float xs[32][S7_TAPS + 1];
    __shared__ float gs[32][65];
    const int tid = threadIdx.x, co = tid & 63, tg = tid >> 6;
    const int64_t Pout = (int64_t)a.N * a.Ho * a.Wo;
    const int64_t plane = (int64_t)a.H * a.W;
    float acc[37];
#pragma unroll
    for (int q = 0; q < 37; ++q) acc[q] = 0.f;
    for (int it = 0; it < S7W_PIX_PER_WG / 32; ++it) {
        const int64_t pbase = (int64_t)blockIdx.x * S7W_PIX_PER_WG + it * 32;
        if (pbase >= Pout) break;
        __syncthreads();
        for (int e = tid; e < 32 * S7_TAPS; e += 256) {
            const int pp = e / S7_TAPS, tap = e % S7_TAPS;
            const int64_t po = pbase + pp;
            float v = 0.f;
            if (po < Pout) {
                const int ow = (int)(po % a.Wo), oh = (int)((po / a.Wo) % a.Ho), n = (int)(po / ((int64_t)a.Wo * a.Ho));
                const int ci = tap / 49, kh = (tap % 49) / 7, kw = tap % 7;
                const int ih = oh * 2 - 3 + kh, iw = ow * 2 - 3 + kw;
                if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) v = a.x[((int64_t)n * 3 + ci) * plane + (int64_t)ih * a.W + iw];
            }
            xs[pp][tap] = v;
        }
        for (int e = tid; e < 32 * 64; e += 256) {
            const int pp = e >> 6, c = e & 63;
            const int64_t po = pbase + pp;
            gs[pp][c] = po < Pout ? a.dy[po * a.lddy + c] : 0.f;
        }
        __syncthreads();
        for (int pp = 0; pp < 32; ++pp) {
            const float g = gs[pp][co];
#pragma unroll
            for (int q = 0; q < 37; ++q) {
                const int tap = tg + 4 * q;
                if (tap < S7_TAPS) acc[q] += g * xs[pp][tap];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 37; ++q) {
        const int tap = tg + 4 * q;
        if (tap < S7_TAPS) a.part[(int64_t)blockIdx.x * (64 * S7_TAPS) + co * S7_TAPS + tap] = acc[q];
    }
}

__global__ void s7_cast_d2f_kernel(const double* __restrict__ in, int n, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = (float)in[e];
}

extern "C" uint64_t uda_stem7_workspace_bytes(int64_t Pout) {
    return (uint64_t)uda_cdiv(Pout, S7W_PIX_PER_WG) * 64 * S7_TAPS * sizeof(float) + 64 * S7_TAPS * sizeof(double);
}

extern "C" int uda_stem7_fwd(const float* x, int N, int H, int W, const float* w, float* y, int64_t ldy, double* stats,
                             void* stream) {
    UDA_REQUIRE(x && w && y && uda_aligned16(y) && ldy % 4 == 0 && ldy >= 64 && N > 0 && H >= 8 && W >= 8 && W % 8 == 0,
                "uda_stem7_fwd: bad args (W %% 8 == 0, ldy >= 64)");
    Stem7Args a;
    a.x = x; a.N = N; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.w = w; a.y = y; a.ldy = ldy; a.stats = stats; a.dy = nullptr; a.lddy = 0; a.part = nullptr;
    const int64_t Pout = (int64_t)N * a.Ho * a.Wo;
    hipLaunchKernelGGL(stem7_fwd_kernel, dim3(uda_cdiv(Pout, S7_PIX_PER_WG)), dim3(256), 0, (hipStream_t)stream, a);
    UDA_LAUNCH_CHECK("stem7_fwd");
    return 0;
}

extern "C" int uda_stem7_wgrad(const float* x, int N, int H, int W, const float* dy, int64_t lddy, float* dw,
                               float* workspace, uint64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(x && dy && dw && lddy >= 64 && N > 0 && H >= 8 && W >= 8, "uda_stem7_wgrad: bad args");
    Stem7Args a;
    a.x = x; a.N = N; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.w = nullptr; a.y = nullptr; a.ldy = 0; a.stats = nullptr; a.dy = dy; a.lddy = lddy;
    const int64_t Pout = (int64_t)N * a.Ho * a.Wo;
    const int nwg = uda_cdiv(Pout, S7W_PIX_PER_WG), nel = 64 * S7_TAPS;
    UDA_REQUIRE(workspace && workspace_bytes >= uda_stem7_workspace_bytes(Pout), "uda_stem7_wgrad: workspace too small");
    double* sums = reinterpret_cast<double*>(workspace);
    a.part = workspace + 2 * nel;
    (void)hipMemsetAsync(sums, 0, nel * sizeof(double), st);
    hipLaunchKernelGGL(stem7_wgrad_kernel, dim3(nwg), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("stem7_wgrad");
    if (int e = uda_reduce_partials(a.part, nwg, nel, sums, st)) return e;
    hipLaunchKernelGGL(s7_cast_d2f_kernel, dim3(uda_cdiv(nel, 256)), dim3(256), 0, st, sums, nel, dw);
    UDA_LAUNCH_CHECK("stem7_wgrad_store");
    return 0;
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, pad 1) of u = act(x*scale+shift).  One thread = 4 channels of one output
// pixel; the winning tap (first maximum in scan order, as ATen) is kept as one byte per element.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(uda_src_t s, int Ho, int Wo, float* __restrict__ out, int64_t ldo,
                                                          uint8_t* __restrict__ idx, int64_t ldi) {
    const int C4 = s.C >> 2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t Pout = (int64_t)s.N * Ho * Wo;
    if (e >= Pout * C4) return;
    const int64_t po = e / C4;
    const int c0 = (int)(e % C4) * 4;
    const int ow = (int)(po % Wo), oh = (int)((po / Wo) % Ho), n = (int)(po / ((int64_t)Wo * Ho));
    Xf4 xf;
    uda_load_xf4(xf, s.scale, s.shift, c0, s.C);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        if (ih < 0 || ih >= s.H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            if (iw < 0 || iw >= s.W) continue;
            const float4 xv = uda_ld4(s.x + (((int64_t)n * s.H + ih) * s.W + iw) * s.ldx + c0);
            const float v[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = uda_act(v[j] * xf.sc[j] + xf.sh[j], s.act);
                if (u > best[j] || u != u) {
                    best[j] = u;
                    bi[j] = kh * 3 + kw;
                }
            }
        }
    }
    uda_st4(out + po * ldo + c0, make_float4(best[0], best[1], best[2], best[3]));
    *reinterpret_cast<uint32_t*>(idx + po * ldi + c0) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

// gather form of the backward: an input pixel lies in at most 2 x 2 windows
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dz, int64_t lddz, const uint8_t* __restrict__ idx,
                                                          int64_t ldi, int N, int H, int W, int Ho, int Wo, int C4,
                                                          float* __restrict__ du, int64_t ldu) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t P = (int64_t)N * H * W;
    if (e >= P * C4) return;
    const int64_t p = e / C4;
    const int c0 = (int)(e % C4) * 4;
    const int iw = (int)(p % W), ih = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = ih >> 1; oh <= (ih + 1) >> 1; ++oh) {
        if (oh >= Ho) continue;
        const int kh = ih - (oh * 2 - 1);
        for (int ow = iw >> 1; ow <= (iw + 1) >> 1; ++ow) {
            if (ow >= Wo) continue;
            const uint32_t t = (uint32_t)(kh * 3 + (iw - (ow * 2 - 1)));
            const int64_t po = ((int64_t)n * Ho + oh) * Wo + ow;
            const uint32_t bi = *reinterpret_cast<const uint32_t*>(idx + po * ldi + c0);
            const float4 g = uda_ld4(dz + po * lddz + c0);
            if ((bi & 0xffu) == t) acc[0] += g.x;
            if (((bi >> 8) & 0xffu) == t) acc[1] += g.y;
            if (((bi >> 16) & 0xffu) == t) acc[2] += g.z;
            if ((bi >> 24) == t) acc[3] += g.w;
        }
    }
    uda_st4(du + p * ldu + c0, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

extern "C" int uda_maxpool_fwd(const uda_src_t* src, float* out, int64_t ldo, uint8_t* idx, int64_t ldi, void* stream) {
    UDA_REQUIRE(src && src->x && uda_aligned16(src->x) && src->ldx % 4 == 0 && src->C > 0 && src->C % 4 == 0 && !src->mask,
                "uda_maxpool_fwd: src must be aligned, C %% 4 == 0, no mask");
    UDA_REQUIRE(out && idx && uda_aligned16(out) && ldo % 4 == 0 && ldi % 4 == 0 && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0 &&
                    ldo >= src->C && ldi >= src->C, "uda_maxpool_fwd: bad out/idx layout");
    const int Ho = (src->H - 1) / 2 + 1, Wo = (src->W - 1) / 2 + 1;
    const int64_t n = (int64_t)src->N * Ho * Wo * (src->C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(uda_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, *src, Ho, Wo, out, ldo, idx, ldi);
    UDA_LAUNCH_CHECK("maxpool_fwd");
    return 0;
}

extern "C" int uda_maxpool_bwd(const float* dz, int64_t lddz, const uint8_t* idx, int64_t ldi, int N, int H, int W, int C,
                               float* du, int64_t ldu, void* stream) {
    UDA_REQUIRE(dz && idx && du && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && lddz % 4 == 0 && ldi % 4 == 0 && ldu % 4 == 0 &&
                    uda_aligned16(dz) && uda_aligned16(du), "uda_maxpool_bwd: bad args");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t n = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(uda_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dz, lddz, idx, ldi, N, H, W,
                       Ho, Wo, C / 4, du, ldu);
    UDA_LAUNCH_CHECK("maxpool_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------
// scatter == 0:  dst[(n,oh,ow), c] = src[(n, oh*s, ow*s), c]          (dst has Ho x Wo pixels)
// scatter == 1:  dst[(n,ih,iw), c] = src[(n, ih/s, iw/s), c] if s | ih and s | iw else 0   (transpose)
__global__ __launch_bounds__(256) void rows_stride_kernel(const float* __restrict__ src, int64_t lds_, int N, int H, int W, int C4,
                                                          int s, int scatter, float* __restrict__ dst, int64_t ldd) {
    const int Ho = (H - 1) / s + 1, Wo = (W - 1) / s + 1;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t Pd = scatter ? (int64_t)N * H * W : (int64_t)N * Ho * Wo;
    if (e >= Pd * C4) return;
    const int64_t p = e / C4;
    const int c0 = (int)(e % C4) * 4;
    if (!scatter) {
        const int ow = (int)(p % Wo), oh = (int)((p / Wo) % Ho), n = (int)(p / ((int64_t)Wo * Ho));
        uda_st4(dst + p * ldd + c0, uda_ld4(src + (((int64_t)n * H + oh * s) * W + ow * s) * lds_ + c0));
    } else {
        const int iw = (int)(p % W), ih = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ih % s == 0 && iw % s == 0) v = uda_ld4(src + (((int64_t)n * Ho + ih / s) * Wo + iw / s) * lds_ + c0);
        uda_st4(dst + p * ldd + c0, v);
    }
}

extern "C" int uda_rows_stride(const float* src, int64_t ld_src, int N, int H, int W, int C, int stride, int scatter, float* dst,
                               int64_t ld_dst, void* stream) {
    UDA_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && stride >= 1 && ld_src % 4 == 0 && ld_dst % 4 == 0 &&
                    ld_src >= C && ld_dst >= C && uda_aligned16(src) && uda_aligned16(dst), "uda_rows_stride: bad args (C %% 4 == 0)");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t n = (scatter ? (int64_t)N * H * W : (int64_t)N * Ho * Wo) * (C / 4);
    hipLaunchKernelGGL(rows_stride_kernel, dim3(uda_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, N, H, W, C / 4,
                       stride, scatter, dst, ld_dst);
    UDA_LAUNCH_CHECK("rows_stride");
    return 0;
}
