// Loss, metric and category-prototype kernels of the training step (all HBM-bound streaming /
// reduction kernels; LDS + wave reductions).  Sums: per-thread and per-workgroup partials in fp64; the prototype sums are
// combined by a second kernel in a fixed order (bit-reproducible), the seg-loss sums by fp64 atomicAdd across workgroups
// (order-dependent in the last bits of the fp64 sum, i.e. ~1e-16 relative - far below the fp32 value the loss is rounded
// to - but not bit-reproducible by construction); the pixel counts are integer atomics (exact).
//   seg loss       BCELoss(sigmoid(o), map) + MSELoss(sigmoid(b), boundary)  Trainer_prototype_full.py:292-294
//   seg counts     dice_coeff_2label / pixel_acc ingredients                  utils/metrics.py:118-168
//   mc stats       std over T of sigmoid(x/2), mean over T of sigmoid(x)      utils/Utils.py:164-168
//   proto weights  hard (nearest-downsampled labels), soft (sigmoid logits), retrified (pseudo label
//                  x reliability mask x confidence, with the two 512->128 bilinear resizes folded in)
//                                                                             utils/Utils.py:108-131,170-206
//   proto reduce   4 weighted sums over (N,H,W) of the 305-channel feature + 4 counts in ONE pass
//                  (the reference reads the feature 8 times)                  utils/Utils.py:114-130,207-223
//   proto backward d feature / d weights of centroid = sum / count
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ double block_sum(double v, double* red) {
    // 256 threads: wave64 shuffle reduction then 4 partials through LDS
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------ seg loss
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                           int64_t n_o, const float* __restrict__ b,
                                                           const float* __restrict__ tb, int64_t n_b, double* __restrict__ acc) {
    __shared__ double red[4];
    double s_bce = 0.0, s_mse = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_o; e += stride) {
        const float x = sigmoidf_(o[e]), tt = t[e];
        const float l1 = fmaxf(logf(x), -100.f), l0 = fmaxf(logf(1.f - x), -100.f);     // BCELoss clamps log at -100
        s_bce += (double)(-(tt * l1 + (1.f - tt) * l0));
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_b; e += stride) {
        const float d = sigmoidf_(b[e]) - tb[e];
        s_mse += (double)(d * d);
    }
    const double a0 = block_sum(s_bce, red);
    const double a1 = block_sum(s_mse, red);
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], a0);
        atomicAdd(&acc[1], a1);
    }
}
__global__ void seg_loss_final_kernel(const double* __restrict__ acc, int64_t n_o, int64_t n_b, float* __restrict__ loss) {
    loss[0] = (float)(acc[0] / (double)n_o + acc[1] / (double)n_b);
    loss[1] = (float)(acc[0] / (double)n_o);
    loss[2] = (float)(acc[1] / (double)n_b);
}
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                           int64_t n_o, const float* __restrict__ b,
                                                           const float* __restrict__ tb, int64_t n_b,
                                                           const float* __restrict__ gscale, float* __restrict__ d_o,
                                                           float* __restrict__ d_b) {
    const float g = gscale[0];
    const float go = g / (float)n_o, gb = 2.f * g / (float)n_b;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_o; e += stride) {
        const float x = sigmoidf_(o[e]);
        const float v = x * (1.f - x);
        // binary_cross_entropy_backward: (x - t) / max(x(1-x), 1e-12), then sigmoid' = x(1-x)
        d_o[e] = go * (x - t[e]) / fmaxf(v, 1e-12f) * v;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_b; e += stride) {
        const float x = sigmoidf_(b[e]);
        d_b[e] = gb * (x - tb[e]) * x * (1.f - x);
    }
}

static inline int stream_grid(int64_t n) {
    int g = uda_cdiv(n, 256 * 4);
    if (g > 2048) g = 2048;
    return g < 1 ? 1 : g;
}

extern "C" int uda_seg_loss_fwd(const float* o, const float* map, int64_t n_o, const float* b, const float* boundary,
                                int64_t n_b, float* loss3, double* workspace2, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(o && map && b && boundary && loss3 && workspace2 && n_o > 0 && n_b > 0, "uda_seg_loss_fwd: bad args");
    (void)hipMemsetAsync(workspace2, 0, 2 * sizeof(double), st);
    hipLaunchKernelGGL(seg_loss_fwd_kernel, dim3(stream_grid(n_o)), dim3(256), 0, st, o, map, n_o, b, boundary, n_b, workspace2);
    UDA_LAUNCH_CHECK("seg_loss_fwd");
    hipLaunchKernelGGL(seg_loss_final_kernel, dim3(1), dim3(1), 0, st, workspace2, n_o, n_b, loss3);
    UDA_LAUNCH_CHECK("seg_loss_final");
    return 0;
}
extern "C" int uda_seg_loss_bwd(const float* o, const float* map, int64_t n_o, const float* b, const float* boundary,
                                int64_t n_b, const float* gscale, float* d_o, float* d_b, void* stream) {
    UDA_REQUIRE(o && map && b && boundary && gscale && d_o && d_b && n_o > 0 && n_b > 0, "uda_seg_loss_bwd: bad args");
    hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3(stream_grid(n_o)), dim3(256), 0, (hipStream_t)stream, o, map, n_o, b, boundary,
                       n_b, gscale, d_o, d_b);
    UDA_LAUNCH_CHECK("seg_loss_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------ metrics
// counts[c][0..2] = (intersection, predicted-positive, ground-truth-positive) for sigmoid(logit) > thr
__global__ __launch_bounds__(256) void seg_counts_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                         int B, int C, int64_t HW, float thr, unsigned long long* counts) {
    __shared__ double red[4];
    const int c = blockIdx.y;
    double ci = 0, cs = 0, cg = 0;
    const int64_t n = (int64_t)B * HW, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t bi = e / HW, p = e % HW;
        const int64_t idx = (bi * C + c) * HW + p;
        const bool pr = sigmoidf_(logits[idx]) > thr, gt = target[idx] != 0.f;
        ci += (pr && gt); cs += pr; cg += gt;
    }
    const double a = block_sum(ci, red), s = block_sum(cs, red), g = block_sum(cg, red);
    if (threadIdx.x == 0) {
        atomicAdd(&counts[c * 3 + 0], (unsigned long long)(a + 0.5));
        atomicAdd(&counts[c * 3 + 1], (unsigned long long)(s + 0.5));
        atomicAdd(&counts[c * 3 + 2], (unsigned long long)(g + 0.5));
    }
}
extern "C" int uda_seg_counts(const float* logits, const float* target, int B, int C, int64_t HW, float thr,
                              uint64_t* counts, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(logits && target && counts && B > 0 && C > 0 && C <= 64 && HW > 0, "uda_seg_counts: bad args");
    (void)hipMemsetAsync(counts, 0, (size_t)C * 3 * sizeof(uint64_t), st);
    hipLaunchKernelGGL(seg_counts_kernel, dim3(stream_grid((int64_t)B * HW), C), dim3(256), 0, st, logits, target, B, C, HW, thr,
                       reinterpret_cast<unsigned long long*>(counts));
    UDA_LAUNCH_CHECK("seg_counts");
    return 0;
}

// ------------------------------------------------------------------------------------------ MC statistics
// preds [T][n] logits -> std_map[n] = unbiased std over T of sigmoid(x/2), mean_map[n] = mean of sigmoid(x)
template <int T>
__global__ __launch_bounds__(256) void mc_stats_kernel(const float* __restrict__ preds, int64_t n, float* __restrict__ std_map,
                                                       float* __restrict__ mean_map) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        float h[T];
        float m1 = 0.f, mh = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float x = preds[(int64_t)t * n + e];
            m1 += sigmoidf_(x);
            h[t] = sigmoidf_(x * 0.5f);
            mh += h[t];
        }
        mh /= (float)T;
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) ss += (h[t] - mh) * (h[t] - mh);
        std_map[e] = sqrtf(ss / (float)(T - 1));
        mean_map[e] = m1 / (float)T;
    }
}
extern "C" int uda_mc_stats(const float* preds, int T, int64_t n, float* std_map, float* mean_map, void* stream) {
    UDA_REQUIRE(preds && std_map && mean_map && n > 0, "uda_mc_stats: bad args");
    UDA_REQUIRE(T == 8, "uda_mc_stats: built for T = 8 stochastic passes (Trainer_prototype_full.py:359)");
    hipLaunchKernelGGL((mc_stats_kernel<8>), dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, preds, n, std_map, mean_map);
    UDA_LAUNCH_CHECK("mc_stats");
    return 0;
}

// ------------------------------------------------------------------------------------------ prototype weights
// mode 0: hard  - labels map [B,2,H,W] nearest-downsampled to h x w : w = (m0, m1, 1-m0, 1-m1)
// mode 1: soft  - p = sigmoid(logits[p, 0..1])                      : w = (p0, p1, 1-p0, 1-p1)
// mode 2: retrify - pl = sigmoid(logit) > 0.75, m = bilinear(std) < 0.04, q = bilinear(mean):
//                   w = (m0*pl0*q0, m1*pl1*q1, m0*(1-pl0)*(1-q0), m1*(1-pl1)*(1-q1)); masks = 2*m
struct PwArgs {
    int mode, B, h, w, H, W;
    const float* map;       // mode 0: [B,2,H,W]
    const float* logits;    // modes 1,2: [P, ldl] (2 channels)
    int64_t ldl;
    const float* std_map;   // mode 2: [B,2,H,W]
    const float* mean_map;  // mode 2: [B,2,H,W]
    float* wts;             // [P,4]
    float* mask0;           // mode 2: [P]
    float* mask1;
};

__device__ __forceinline__ float bil_sample(const float* plane, int H, int W, int oh, int ow, float sh, float sw) {
#pragma clang fp contract(off)      // source coordinates rounded to fp32 before the subtraction, as PyTorch (see bilinear.h)
    const float rh = sh * (float)oh, rw = sw * (float)ow;
    int h0 = (int)rh, w0 = (int)rw;
    if (h0 > H - 1) h0 = H - 1;
    if (w0 > W - 1) w0 = W - 1;
    const int h1 = h0 + (h0 < H - 1 ? 1 : 0), w1 = w0 + (w0 < W - 1 ? 1 : 0);
    const float lh1 = rh - (float)h0, lw1 = rw - (float)w0, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    return lh0 * (lw0 * plane[(int64_t)h0 * W + w0] + lw1 * plane[(int64_t)h0 * W + w1]) +
           lh1 * (lw0 * plane[(int64_t)h1 * W + w0] + lw1 * plane[(int64_t)h1 * W + w1]);
}

__global__ __launch_bounds__(256) void proto_weights_kernel(PwArgs a) {
    const int64_t P = (int64_t)a.B * a.h * a.w;
    const int64_t plane = (int64_t)a.H * a.W;
    const float sh = a.h > 1 ? (float)(a.H - 1) / (float)(a.h - 1) : 0.f;
    const float sw = a.w > 1 ? (float)(a.W - 1) / (float)(a.w - 1) : 0.f;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const int ow = (int)(p % a.w), oh = (int)((p / a.w) % a.h), n = (int)(p / ((int64_t)a.w * a.h));
        float w0, w1, w2, w3;
        if (a.mode == 0) {
            // F.interpolate(mode='nearest'): src = floor(dst * (H / h))
            const int ih = min((int)floorf((float)oh * ((float)a.H / (float)a.h)), a.H - 1);
            const int iw = min((int)floorf((float)ow * ((float)a.W / (float)a.w)), a.W - 1);
            const float m0 = a.map[((int64_t)n * 2 + 0) * plane + (int64_t)ih * a.W + iw];
            const float m1 = a.map[((int64_t)n * 2 + 1) * plane + (int64_t)ih * a.W + iw];
            w0 = m0; w1 = m1; w2 = 1.f - m0; w3 = 1.f - m1;
        } else if (a.mode == 1) {
            const float p0 = sigmoidf_(a.logits[p * a.ldl + 0]), p1 = sigmoidf_(a.logits[p * a.ldl + 1]);
            w0 = p0; w1 = p1; w2 = 1.f - p0; w3 = 1.f - p1;
        } else {
            const float pl0 = sigmoidf_(a.logits[p * a.ldl + 0]) > 0.75f ? 1.f : 0.f;
            const float pl1 = sigmoidf_(a.logits[p * a.ldl + 1]) > 0.75f ? 1.f : 0.f;
            const float* s0 = a.std_map + ((int64_t)n * 2 + 0) * plane;
            const float* s1 = a.std_map + ((int64_t)n * 2 + 1) * plane;
            const float* q0p = a.mean_map + ((int64_t)n * 2 + 0) * plane;
            const float* q1p = a.mean_map + ((int64_t)n * 2 + 1) * plane;
            const float m0 = bil_sample(s0, a.H, a.W, oh, ow, sh, sw) < 0.04f ? 1.f : 0.f;
            const float m1 = bil_sample(s1, a.H, a.W, oh, ow, sh, sw) < 0.04f ? 1.f : 0.f;
            const float q0 = bil_sample(q0p, a.H, a.W, oh, ow, sh, sw), q1 = bil_sample(q1p, a.H, a.W, oh, ow, sh, sw);
            w0 = m0 * pl0 * q0; w1 = m1 * pl1 * q1;
            w2 = m0 * (1.f - pl0) * (1.f - q0); w3 = m1 * (1.f - pl1) * (1.f - q1);
            a.mask0[p] = 2.f * m0;
            a.mask1[p] = 2.f * m1;
        }
        uda_st4(a.wts + p * 4, make_float4(w0, w1, w2, w3));
    }
}

extern "C" int uda_proto_weights(int mode, int B, int h, int w, int H, int W, const float* map, const float* logits,
                                 int64_t ldl, const float* std_map, const float* mean_map, float* wts, float* mask0,
                                 float* mask1, void* stream) {
    UDA_REQUIRE(mode >= 0 && mode <= 2 && B > 0 && h > 0 && w > 0 && wts && uda_aligned16(wts), "uda_proto_weights: bad args");
    UDA_REQUIRE(mode != 0 || (map && H > 0 && W > 0), "uda_proto_weights: mode 0 needs the label map");
    UDA_REQUIRE(mode == 0 || (logits && ldl >= 2), "uda_proto_weights: modes 1/2 need the [P,2] logits");
    UDA_REQUIRE(mode != 2 || (std_map && mean_map && mask0 && mask1 && H > 0 && W > 0), "uda_proto_weights: mode 2 needs std/mean maps and mask outputs");
    PwArgs a{mode, B, h, w, H, W, map, logits, ldl, std_map, mean_map, wts, mask0, mask1};
    hipLaunchKernelGGL(proto_weights_kernel, dim3(stream_grid((int64_t)B * h * w * 4)), dim3(256), 0, (hipStream_t)stream, a);
    UDA_LAUNCH_CHECK("proto_weights");
    return 0;
}

// ------------------------------------------------------------------------------------------ prototype reduce
// part[wg][k][c] = sum_p w_k[p] f[p,c]  (c < C),  part[wg][k][C] = sum_p w_k[p]
#define PR_ITER 32
__global__ __launch_bounds__(256) void proto_reduce_kernel(const float* __restrict__ f, int64_t ldf, int64_t P, int C,
                                                           const float* __restrict__ wts, float* __restrict__ part) {
    __shared__ float red[4 * 1024 + 64];
    const int G = (C + 3) >> 2, PP = 256 / G;                 // C = 305: G = 77, PP = 3
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    const bool active = pl < PP;
    const int c0 = cg * 4;
    float acc[4][4];
    float cnt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    const int64_t base = (int64_t)blockIdx.x * (PP * PR_ITER);
    for (int it = 0; it < PR_ITER; ++it) {
        const int64_t p = base + (int64_t)it * PP + pl;
        if (!active || p >= P) continue;
        const float4 w = uda_ld4(wts + p * 4);
        const float4 v4 = uda_ld4(f + p * ldf + c0);
        const float v[4] = {c0 + 0 < C ? v4.x : 0.f, c0 + 1 < C ? v4.y : 0.f, c0 + 2 < C ? v4.z : 0.f, c0 + 3 < C ? v4.w : 0.f};
        const float wk[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] += wk[k] * v[j];
            if (cg == 0) cnt[k] += wk[k];
        }
    }
    const int Cp = G * 4;
    if (active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(k * PP + pl) * Cp + c0 + j] = acc[k][j];
            if (cg == 0) red[4 * 1024 + k * PP + pl] = cnt[k];
        }
    }
    __syncthreads();
    const int C1 = C + 1;
    for (int e = tid; e < 4 * C1; e += 256) {
        const int k = e / C1, c = e % C1;
        float t = 0.f;
        if (c < C) for (int p = 0; p < PP; ++p) t += red[(k * PP + p) * Cp + c];
        else for (int p = 0; p < PP; ++p) t += red[4 * 1024 + k * PP + p];
        part[((int64_t)blockIdx.x * 4 + k) * C1 + c] = t;
    }
}

int uda_reduce_partials(const float* part, int nrows, int ncols, double* out, hipStream_t st);

static inline int proto_nwg(int64_t P, int C) { return uda_cdiv(P, (int64_t)(256 / ((C + 3) / 4)) * PR_ITER); }

extern "C" uint64_t uda_proto_workspace_bytes(int64_t P, int C) { return (uint64_t)proto_nwg(P, C) * 4 * (C + 1) * sizeof(float); }

/* sums (double [4][C+1], ADDED into): per class k the C weighted channel sums followed by the count */
extern "C" int uda_proto_reduce(const float* feat, int64_t ldf, int64_t P, int C, const float* wts, double* sums,
                                float* workspace, uint64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(feat && uda_aligned16(feat) && ldf % 4 == 0 && ldf >= ((C + 3) / 4) * 4 && C > 0 && C <= 1024 && P > 0 && wts &&
                    uda_aligned16(wts) && sums, "uda_proto_reduce: bad args");
    UDA_REQUIRE(workspace && workspace_bytes >= uda_proto_workspace_bytes(P, C), "uda_proto_reduce: workspace too small");
    const int nwg = proto_nwg(P, C);
    hipLaunchKernelGGL(proto_reduce_kernel, dim3(nwg), dim3(256), 0, st, feat, ldf, P, C, wts, workspace);
    UDA_LAUNCH_CHECK("proto_reduce");
    return uda_reduce_partials(workspace, nwg, 4 * (C + 1), sums, st);
}

// centroid[k][c] = sums[k][c] / sums[k][C]
__global__ void proto_finalize_kernel(const double* __restrict__ sums, int C, float* __restrict__ cent) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 4 * C) {
        const int k = e / C, c = e % C;
        cent[e] = (float)(sums[k * (C + 1) + c] / sums[k * (C + 1) + C]);
    }
}
extern "C" int uda_proto_finalize(const double* sums, int C, float* centroids, void* stream) {
    UDA_REQUIRE(sums && centroids && C > 0, "uda_proto_finalize: bad args");
    hipLaunchKernelGGL(proto_finalize_kernel, dim3(uda_cdiv(4 * C, 256)), dim3(256), 0, (hipStream_t)stream, sums, C, centroids);
    UDA_LAUNCH_CHECK("proto_finalize");
    return 0;
}

// coef[k][c] = dC[k][c] / cnt_k  (c < C);  coef[k][C] = -(sum_c dC[k][c] * sums[k][c]) / cnt_k^2
__global__ __launch_bounds__(256) void proto_bwd_coef_kernel(const double* __restrict__ sums, const float* __restrict__ dC,
                                                             int C, float* __restrict__ coef) {
    __shared__ double red[4];
    const int k = blockIdx.x;
    const double cnt = sums[k * (C + 1) + C];
    double dot = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        coef[k * (C + 1) + c] = (float)((double)dC[k * C + c] / cnt);
        dot += (double)dC[k * C + c] * sums[k * (C + 1) + c];
    }
    const double tot = block_sum(dot, red);
    if (threadIdx.x == 0) coef[k * (C + 1) + C] = (float)(-tot / (cnt * cnt));
}

// d_feat[p,c] (+)= sum_k w_k[p] coef[k][c];   d_w[p,k] = sum_c f[p,c] coef[k][c] + coef[k][C]
__global__ __launch_bounds__(256) void proto_bwd_kernel(const float* __restrict__ f, int64_t ldf, int64_t P, int C,
                                                        const float* __restrict__ wts, const float* __restrict__ coef,
                                                        float* d_feat, int64_t ldd, int accumulate, float* __restrict__ d_w) {
    // one wave per pixel: lanes stride the channels, the per-pixel dot products are wave-reduced
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int C1 = C + 1;
    for (int64_t p = wave; p < P; p += nwaves) {
        const float4 w = d_feat ? uda_ld4(wts + p * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float k0 = coef[c], k1 = coef[C1 + c], k2 = coef[2 * C1 + c], k3 = coef[3 * C1 + c];
            if (d_feat) {
                const float g = w.x * k0 + w.y * k1 + w.z * k2 + w.w * k3;
                float* dst = d_feat + p * ldd + c;
                *dst = accumulate ? (*dst + g) : g;
            }
            if (d_w) {
                const float v = f[p * ldf + c];
                d0 += v * k0; d1 += v * k1; d2 += v * k2; d3 += v * k3;
            }
        }
        if (d_w) {
            for (int o = 32; o > 0; o >>= 1) {
                d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o);
            }
            if (lane == 0)
                uda_st4(d_w + p * 4, make_float4(d0 + coef[C], d1 + coef[C1 + C], d2 + coef[2 * C1 + C], d3 + coef[3 * C1 + C]));
        }
    }
}

/* dC: gradient w.r.t. the 4 centroids [4][C].  coef_ws: float [4][C+1] scratch. */
extern "C" int uda_proto_bwd(const float* feat, int64_t ldf, int64_t P, int C, const float* wts, const double* sums,
                             const float* dC, float* coef_ws, float* d_feat, int64_t ldd, int accumulate, float* d_w,
                             void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(feat && wts && sums && dC && coef_ws && uda_aligned16(wts) && P > 0 && C > 0, "uda_proto_bwd: bad args");
    UDA_REQUIRE(d_feat || d_w, "uda_proto_bwd: nothing to compute");
    if (d_w) UDA_REQUIRE(uda_aligned16(d_w), "uda_proto_bwd: d_w must be 16-byte aligned");
    hipLaunchKernelGGL(proto_bwd_coef_kernel, dim3(4), dim3(256), 0, st, sums, dC, C, coef_ws);
    UDA_LAUNCH_CHECK("proto_bwd_coef");
    int grid = uda_cdiv(P, 4);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(proto_bwd_kernel, dim3(grid), dim3(256), 0, st, feat, ldf, P, C, wts, coef_ws, d_feat, ldd, accumulate, d_w);
    UDA_LAUNCH_CHECK("proto_bwd");
    return 0;
}

/* out[p][k] = sum_c feat[p,c] * coef[k][c] + coef[k][C]   (4 affine functionals of every pixel's feature vector;
 * the prototype-guided discriminative loss needs D(f,c_obj) - D(f,c_bck), which is affine in f) */
extern "C" int uda_feat_dot4(const float* feat, int64_t ldf, int64_t P, int C, const float* coef, float* out, void* stream) {
    UDA_REQUIRE(feat && coef && out && uda_aligned16(out) && P > 0 && C > 0 && ldf >= C, "uda_feat_dot4: bad args");
    int grid = uda_cdiv(P, 4);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(proto_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, ldf, P, C, (const float*)nullptr, coef,
                       (float*)nullptr, (int64_t)0, 0, out);
    UDA_LAUNCH_CHECK("feat_dot4");
    return 0;
}
/* d_feat[p,c] (+)= sum_k wts[p][k] * coef[k][c]   (adjoint of uda_feat_dot4) */
extern "C" int uda_feat_rank4(const float* wts, const float* coef, int64_t P, int C, float* d_feat, int64_t ldd, int accumulate,
                              void* stream) {
    UDA_REQUIRE(wts && uda_aligned16(wts) && coef && d_feat && P > 0 && C > 0 && ldd >= C, "uda_feat_rank4: bad args");
    int grid = uda_cdiv(P, 4);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(proto_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, (int64_t)0, P, C, wts,
                       coef, d_feat, ldd, accumulate, (float*)nullptr);
    UDA_LAUNCH_CHECK("feat_rank4");
    return 0;
}

// ------------------------------------------------------------------------------------------ fused Adam
// torch.optim.Adam (no weight decay, no amsgrad) over one flat fp32 buffer, with the scalars rounded where torch rounds them
// (python doubles 1 - beta, lr / bias_correction1, sqrt(bias_correction2) cast to fp32 when they meet the tensors):
//   m = m + (1-b1) * (g - m)   [lerp];   v = b2 * v + (1-b2) * g * g;   p -= (lr/bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float step_size, float w1, float b2, float w2,
                                                   float eps, float bc2_sqrt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const float gg = g[e];
        const float m0 = m[e];
        const float mm = m0 + w1 * (gg - m0);
        const float vv = b2 * v[e] + w2 * gg * gg;
        m[e] = mm;
        v[e] = vv;
        p[e] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
}
extern "C" int uda_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                             double beta1, double beta2, double eps, int64_t step, void* stream) {
    UDA_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "uda_adam_step: bad args");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                       (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2));
    UDA_LAUNCH_CHECK("adam");
    return 0;
}

// ------------------------------------------------------------------------------------------ prototype EMA + alignment losses
// Trainer_prototype_full.py:335-355, 378-398 (EMA of the eight centroids, gradient only through the current term - quirk Q4)
// and :428-444 (intra = sum_k MSE(src_k, tgt_k); inter = MSE(src_1, src_3) + MSE(src_0, src_2), logged only), as ONE launch on
// the two [4][C] centroid matrices (k = cup_obj, disc_obj, cup_bck, disc_bck):
//     new = has_prev ? keep * prev + decay * cur : cur                (per domain; keep = 1 - decay evaluated by the caller in
//                                                                      double like the reference's Python expression)
//     losses[0] = sum_k mean_c (new_src[k][c] - new_tgt[k][c])^2,   losses[1] = mean_c (s1 - s3)^2 + mean_c (s0 - s2)^2
// Backward of losses[0] w.r.t. the CURRENT centroids: d cur_src = g * w_src * 2 (new_src - new_tgt) / C, d cur_tgt = -(...) with
// w = decay (or 1 on first use).  One workgroup; fp64 block sums.
__global__ __launch_bounds__(256) void proto_align_fwd_kernel(const float* __restrict__ cur_src, const float* __restrict__ cur_tgt,
                                                              const float* __restrict__ prev_src, const float* __restrict__ prev_tgt,
                                                              float keep, float decay, int C, float* __restrict__ new_src,
                                                              float* __restrict__ new_tgt, float* __restrict__ losses) {
#pragma clang fp contract(off)
    __shared__ double red[4];
    double intra = 0.0, inter = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s[4], t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float cs = cur_src[k * C + c], ct = cur_tgt[k * C + c];
            // (1-decay) * stored + decay * current (Trainer_prototype_full.py:348-351), two products then one add, no fma
            s[k] = prev_src ? keep * prev_src[k * C + c] + decay * cs : cs;
            t[k] = prev_tgt ? keep * prev_tgt[k * C + c] + decay * ct : ct;
            new_src[k * C + c] = s[k];
            new_tgt[k * C + c] = t[k];
            const float d = s[k] - t[k];
            intra += (double)(d * d);
        }
        const float d13 = s[1] - s[3], d02 = s[0] - s[2];
        inter += (double)(d13 * d13) + (double)(d02 * d02);
    }
    const double a = block_sum(intra, red);
    const double b = block_sum(inter, red);
    if (threadIdx.x == 0) {
        losses[0] = (float)(a / (double)C);
        losses[1] = (float)(b / (double)C);
    }
}

__global__ __launch_bounds__(256) void proto_align_bwd_kernel(const float* __restrict__ new_src, const float* __restrict__ new_tgt,
                                                              const float* __restrict__ g, float w_src, float w_tgt, int C,
                                                              float* __restrict__ d_cur_src, float* __restrict__ d_cur_tgt) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 4 * C) return;
    const float v = g[0] * 2.f * (new_src[e] - new_tgt[e]) / (float)C;
    d_cur_src[e] = w_src * v;
    d_cur_tgt[e] = -w_tgt * v;
}

extern "C" int uda_proto_align_fwd(const float* cur_src, const float* cur_tgt, const float* prev_src, const float* prev_tgt,
                                   float keep, float decay, int C, float* new_src, float* new_tgt, float* losses2, void* stream) {
    UDA_REQUIRE(cur_src && cur_tgt && new_src && new_tgt && losses2 && C > 0, "uda_proto_align_fwd: bad args");
    hipLaunchKernelGGL(proto_align_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cur_src, cur_tgt, prev_src, prev_tgt, keep,
                       decay, C, new_src, new_tgt, losses2);
    UDA_LAUNCH_CHECK("proto_align_fwd");
    return 0;
}

extern "C" int uda_proto_align_bwd(const float* new_src, const float* new_tgt, const float* g_intra, float w_src, float w_tgt, int C,
                                   float* d_cur_src, float* d_cur_tgt, void* stream) {
    UDA_REQUIRE(new_src && new_tgt && g_intra && d_cur_src && d_cur_tgt && C > 0, "uda_proto_align_bwd: bad args");
    hipLaunchKernelGGL(proto_align_bwd_kernel, dim3(uda_cdiv(4 * C, 256)), dim3(256), 0, (hipStream_t)stream, new_src, new_tgt, g_intra,
                       w_src, w_tgt, C, d_cur_src, d_cur_tgt);
    UDA_LAUNCH_CHECK("proto_align_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------ adversarial loss on the patch logits
// scale * (BCEWithLogits(d1, label) + BCEWithLogits(d2, label)), both 'mean' (Trainer_prototype_full.py:456-458 with scale 0.01 and
// label 1; :479-513 with scale 1 and labels 1 / 0), on the two small [N,1,Ho,Wo] discriminator outputs: one launch forward
// (numerically as torch: max(x,0) - x*y + log1p(exp(-|x|))), one launch for both gradients (sigmoid(x) - y) * scale * g / n.
__global__ __launch_bounds__(256) void adv_loss_fwd_kernel(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2,
                                                           float label, float scale, float* __restrict__ loss) {
    __shared__ double red[4];
    double s1 = 0.0, s2 = 0.0;
    for (int e = threadIdx.x; e < n1; e += 256) {
        const float x = d1[e];
        s1 += (double)(fmaxf(x, 0.f) - x * label + log1pf(expf(-fabsf(x))));
    }
    for (int e = threadIdx.x; e < n2; e += 256) {
        const float x = d2[e];
        s2 += (double)(fmaxf(x, 0.f) - x * label + log1pf(expf(-fabsf(x))));
    }
    const double a = block_sum(s1, red);
    const double b = block_sum(s2, red);
    if (threadIdx.x == 0) loss[0] = scale * ((float)(a / (double)n1) + (float)(b / (double)n2));
}

__global__ __launch_bounds__(256) void adv_loss_bwd_kernel(const float* __restrict__ d1, int n1, const float* __restrict__ d2, int n2,
                                                           float label, float scale, const float* __restrict__ g,
                                                           float* __restrict__ g1, float* __restrict__ g2) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float gs = g[0] * scale;
    if (e < n1) g1[e] = (sigmoidf_(d1[e]) - label) * gs / (float)n1;
    if (e < n2) g2[e] = (sigmoidf_(d2[e]) - label) * gs / (float)n2;
}

extern "C" int uda_adv_loss_fwd(const float* d1, int n1, const float* d2, int n2, float label, float scale, float* loss, void* stream) {
    UDA_REQUIRE(d1 && d2 && loss && n1 > 0 && n2 > 0, "uda_adv_loss_fwd: bad args");
    hipLaunchKernelGGL(adv_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d1, n1, d2, n2, label, scale, loss);
    UDA_LAUNCH_CHECK("adv_loss_fwd");
    return 0;
}

extern "C" int uda_adv_loss_bwd(const float* d1, int n1, const float* d2, int n2, float label, float scale, const float* g, float* g1,
                                float* g2, void* stream) {
    UDA_REQUIRE(d1 && d2 && g && g1 && g2 && n1 > 0 && n2 > 0, "uda_adv_loss_bwd: bad args");
    const int n = n1 > n2 ? n1 : n2;
    hipLaunchKernelGGL(adv_loss_bwd_kernel, dim3(uda_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d1, n1, d2, n2, label, scale, g, g1, g2);
    UDA_LAUNCH_CHECK("adv_loss_bwd");
    return 0;
}
