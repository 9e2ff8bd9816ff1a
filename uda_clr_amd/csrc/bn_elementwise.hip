// Training-mode batch-norm pieces around the convolutions (F.batch_norm at every BatchNorm(...) of
// mobilenet.py / aspp.py / decoder.py), all HBM-bound streaming kernels on [P, C] NHWC matrices:
//   bn_finalize       (sum, sumsq) -> scale/shift for the consumer prologue, mean/invstd for
//                     backward, running-stat update (momentum, unbiased variance)
//   bn_apply          materialise act(bn(x))*mask (+ residual)   [block outputs, concat windows]
//   colstats          per-channel sum / sumsq of an already materialised tensor (BN(305))
//   bnbwd_reduce      per-channel sum g, sum g*xhat, sum dU   (g = dU * mask * act'(bn(x)))
//   bnbwd_finalize    adds the quirk-Q1 border term, -> c1, c2, dgamma, dbeta
//   bnbwd_apply       dx = addend + scale*(g - c1 - xhat*c2)
// Reductions are two-stage and deterministic: per-workgroup fp32 partials, then an fp64 sum.
// Thread mapping: cg = tid % G float4 channel groups, pl = tid / G pixel lanes (coalesced rows).
#include "common.h"
#include <stdlib.h>

int uda_reduce_partials(const float* part, int nrows, int ncols, double* out, hipStream_t st);

#define RED_ITER 32
#define RED_CBLK 1024
// channel block of the column REDUCTIONS: 128 channels per workgroup (8 pixel lanes x RED_ITER pixels): every
// workgroup ends with 3 x Cb fp64 atomics, so wide layers (C = 576, 960, 1024) must not put one pixel lane on 1024
// channels (2880 atomics per 32 pixels); channels beyond 128 go to blockIdx.y
#define RED_CBLK_SUM 128

__device__ __forceinline__ void st4_guard(float* p, const float v[4], int nvalid) {
    if (nvalid >= 4) {
        uda_st4(p, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) p[j] = v[j];
    }
}

// ------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const double* __restrict__ stats, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < UDA_STAT_SLOTS; ++k) {
        s1 += stats[(k * 2 + 0) * C + c];
        s2 += stats[(k * 2 + 1) * C + c];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double istd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * istd;
    mean[c] = (float)m;
    invstd[c] = (float)istd;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - m * sc);
    const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, int C,
                                      float eps, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double sc = (double)gamma[c] / sqrt((double)rvar[c] + (double)eps);
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - (double)rmean[c] * sc);
}

extern "C" int uda_bn_finalize(const double* stats, int C, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* scale,
                               float* shift, float* mean, float* invstd, void* stream) {
    UDA_REQUIRE(stats && gamma && beta && running_mean && running_var && scale && shift && mean && invstd && C > 0 && count > 0,
                "uda_bn_finalize: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(uda_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, C, count,
                       gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd);
    UDA_LAUNCH_CHECK("bn_finalize");
    return 0;
}

// k further momentum updates of the running statistics with the SAME batch statistics
// (closed form of k repetitions of  r <- (1-m) r + m s):  r <- (1-m)^k r + (1 - (1-m)^k) s
__global__ void bn_running_replay_kernel(const float* __restrict__ mean, const float* __restrict__ invstd, int C,
                                         double count, int k, float momentum, float eps, float* __restrict__ rmean,
                                         float* __restrict__ rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double is = (double)invstd[c];
    double var = 1.0 / (is * is) - (double)eps;
    if (var < 0.0) var = 0.0;
    const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
    const double a = pow(1.0 - (double)momentum, (double)k);
    rmean[c] = (float)(a * (double)rmean[c] + (1.0 - a) * (double)mean[c]);
    rvar[c] = (float)(a * (double)rvar[c] + (1.0 - a) * unbiased);
}

extern "C" int uda_bn_running_replay(const float* mean, const float* invstd, int C, double count, int k, float momentum,
                                     float eps, float* running_mean, float* running_var, void* stream) {
    UDA_REQUIRE(mean && invstd && running_mean && running_var && C > 0 && count > 0 && k >= 1, "uda_bn_running_replay: bad args");
    hipLaunchKernelGGL(bn_running_replay_kernel, dim3(uda_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, invstd, C,
                       count, k, momentum, eps, running_mean, running_var);
    UDA_LAUNCH_CHECK("bn_running_replay");
    return 0;
}

extern "C" int uda_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, int C, float eps, float* scale, float* shift, void* stream) {
    UDA_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "uda_bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(uda_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, C, eps, scale, shift);
    UDA_LAUNCH_CHECK("bn_eval_coeffs");
    return 0;
}

// ------------------------------------------------------------------------------------------
// TransNorm (--use_TN; networks/sync_batchnorm/batchnorm.py:436-520).  One workgroup walks all channels:
//   ratio_h[c] = mean_h / sqrt(var_h + eps)   (training: UNBIASED batch variance of domain half h; eval: running statistics)
//   prob[c] = 1 / (1 + |ratio_0 - ratio_1|),  gain[c] = 1 + C * prob[c] / sum_c prob
// and scales the per-half BN coefficients (already written by uda_bn_finalize / uda_bn_eval_coeffs) by gain[c].
__device__ __forceinline__ double tn_block_sum(double v, double* red) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void tn_gain_kernel(const double* __restrict__ st0, const double* __restrict__ st1, int C,
                                                      double n0, double n1, float eps, float* __restrict__ scale0,
                                                      float* __restrict__ shift0, float* __restrict__ scale1,
                                                      float* __restrict__ shift1, float* __restrict__ gain) {
    __shared__ double red[4];
    double part = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        double r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double* st = h ? st1 : st0;
            const double n = h ? n1 : n0;
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < UDA_STAT_SLOTS; ++k) {
                s1 += st[(k * 2 + 0) * C + c];
                s2 += st[(k * 2 + 1) * C + c];
            }
            const double m = s1 / n;
            double var = s2 / n - m * m;
            if (var < 0.0) var = 0.0;
            r[h] = m / sqrt(var * (n / (n - 1.0)) + (double)eps);
        }
        const double prob = 1.0 / (1.0 + fabs(r[0] - r[1]));
        gain[c] = (float)prob;            // parked; rescaled below
        part += prob;
    }
    const double total = tn_block_sum(part, red);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = (float)(1.0 + (double)C * (double)gain[c] / total);
        gain[c] = g;
        scale0[c] *= g; shift0[c] *= g;
        scale1[c] *= g; shift1[c] *= g;
    }
}

extern "C" int uda_tn_gain(const double* stats0, const double* stats1, int C, double count0, double count1, float eps,
                           float* scale0, float* shift0, float* scale1, float* shift1, float* gain, void* stream) {
    UDA_REQUIRE(stats0 && stats1 && scale0 && shift0 && scale1 && shift1 && gain && C > 0, "uda_tn_gain: bad args");
    UDA_REQUIRE(count0 > 1 && count1 > 1, "uda_tn_gain: each domain half needs more than 1 value per channel (got %g, %g)", count0, count1);
    hipLaunchKernelGGL(tn_gain_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats0, stats1, C, count0, count1, eps,
                       scale0, shift0, scale1, shift1, gain);
    UDA_LAUNCH_CHECK("tn_gain");
    return 0;
}

__global__ __launch_bounds__(256) void tn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ rms, const float* __restrict__ rvs,
                                                             const float* __restrict__ rmt, const float* __restrict__ rvt, int C,
                                                             float eps, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[4];
    double part = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const double rs = (double)rms[c] / sqrt((double)rvs[c] + (double)eps);
        const double rt = (double)rmt[c] / sqrt((double)rvt[c] + (double)eps);
        const double prob = 1.0 / (1.0 + fabs(rs - rt));
        scale[c] = (float)prob;           // parked; replaced below
        part += prob;
    }
    const double total = tn_block_sum(part, red);
    for (int c = threadIdx.x; c < C; c += 256) {
        const double g = 1.0 + (double)C * (double)scale[c] / total;
        const double sc = (double)gamma[c] / sqrt((double)rvt[c] + (double)eps);      // the TARGET statistics normalise
        scale[c] = (float)(sc * g);
        shift[c] = (float)(((double)beta[c] - (double)rmt[c] * sc) * g);
    }
}

extern "C" int uda_tn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean_source,
                                  const float* running_var_source, const float* running_mean_target,
                                  const float* running_var_target, int C, float eps, float* scale, float* shift, void* stream) {
    UDA_REQUIRE(gamma && beta && running_mean_source && running_var_source && running_mean_target && running_var_target &&
                scale && shift && C > 0, "uda_tn_eval_coeffs: bad args");
    hipLaunchKernelGGL(tn_eval_coeffs_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean_source,
                       running_var_source, running_mean_target, running_var_target, C, eps, scale, shift);
    UDA_LAUNCH_CHECK("tn_eval_coeffs");
    return 0;
}

// ------------------------------------------------------------------------------------------
#define EW_ITER 16
// Elementwise kernels: channel group cg = tid % G is FIXED per thread (its per-channel coefficients
// live in registers), pixel lane pl = tid / G walks EW_ITER strips of PP = 256/G pixels; channels
// beyond 1024 go to blockIdx.y.
__global__ __launch_bounds__(256) void bn_apply_kernel(uda_src_t s, const float* __restrict__ res, int64_t ldr,
                                                       float* __restrict__ out, int64_t ldo, int64_t P) {
    const int C = s.C, cblk0 = blockIdx.y * RED_CBLK;
    const int Cb = min(RED_CBLK, C - cblk0), G = (Cb + 3) >> 2, PP = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    if (pl >= PP) return;
    const int c0 = cblk0 + cg * 4;
    Xf4 xf;
    uda_load_xf4(xf, s.scale, s.shift, c0, C);
    const bool has_xf = s.scale != nullptr;
    const int64_t base = (int64_t)blockIdx.x * (PP * EW_ITER);
#pragma unroll 4
    for (int it = 0; it < EW_ITER; ++it) {
        const int64_t p = base + (int64_t)it * PP + pl;
        if (p >= P) break;
        const float4 xv = uda_ld4(s.x + p * s.ldx + c0);
        float v[4] = {xv.x, xv.y, xv.z, xv.w};
        uint32_t mk = 0x01010101u;
        if (s.mask) mk = *reinterpret_cast<const uint32_t*>(s.mask + p * s.ldm + c0);
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (res) {
            const float4 rv = uda_ld4(res + p * ldr + c0);
            r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u = v[j];
            if (has_xf) u = u * xf.sc[j] + xf.sh[j];
            u = uda_act(u, s.act);
            if (s.mask) u *= (float)((mk >> (8 * j)) & 0xffu) * s.mask_scale;
            v[j] = u + r[j];
        }
        st4_guard(out + p * ldo + c0, v, C - c0);
    }
}

static int src_check(const uda_src_t* s, const char* who) {
    UDA_REQUIRE(s && s->x && uda_aligned16(s->x) && s->ldx % 4 == 0 && s->ldx >= ((s->C + 3) / 4) * 4 && s->C > 0,
                "%s: src must be 16-byte aligned with ldx %% 4 == 0 and >= round4(C)", who);
    UDA_REQUIRE((s->scale == nullptr) == (s->shift == nullptr), "%s: scale/shift must come together", who);
    if (s->mask) UDA_REQUIRE(s->ldm % 4 == 0 && s->ldm >= ((s->C + 3) / 4) * 4 && (reinterpret_cast<uintptr_t>(s->mask) & 3u) == 0,
                             "%s: bad mask layout", who);
    return 0;
}

static inline dim3 ew_grid2(int64_t P, int C) {
    const int Cb = C < RED_CBLK ? C : RED_CBLK;
    const int PP = 256 / ((Cb + 3) / 4);
    return dim3(uda_cdiv(P, (int64_t)PP * EW_ITER), uda_cdiv(C, RED_CBLK));
}

extern "C" int uda_bn_apply(const uda_src_t* src, const float* residual, int64_t ldr, float* out, int64_t ldo, void* stream) {
    if (int e = src_check(src, "uda_bn_apply")) return e;
    UDA_REQUIRE(out && uda_aligned16(out) && ldo % 4 == 0 && ldo >= src->C, "uda_bn_apply: out must be 16-byte aligned, ldo %% 4 == 0");
    if (residual) UDA_REQUIRE(uda_aligned16(residual) && ldr % 4 == 0, "uda_bn_apply: residual must be 16-byte aligned");
    const int64_t P = (int64_t)src->N * src->H * src->W;
    hipLaunchKernelGGL(bn_apply_kernel, ew_grid2(P, src->C), dim3(256), 0, (hipStream_t)stream, *src, residual, ldr, out, ldo, P);
    UDA_LAUNCH_CHECK("bn_apply");
    return 0;
}

// Bottleneck output (resnet.py:37-41):  out = relu(ta(a) + tb(b)),  ta/tb the pending per-channel
// transforms of the two branches (bn3 of the main path; identity or the shortcut's BN).
__global__ __launch_bounds__(256) void bn_add_relu_kernel(uda_src_t a, uda_src_t b, float* __restrict__ out,
                                                          int64_t ldo, int64_t P) {
    const int C = a.C, cblk0 = blockIdx.y * RED_CBLK;
    const int Cb = min(RED_CBLK, C - cblk0), G = (Cb + 3) >> 2, PP = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    if (pl >= PP) return;
    const int c0 = cblk0 + cg * 4;
    Xf4 xa, xb;
    uda_load_xf4(xa, a.scale, a.shift, c0, C);
    uda_load_xf4(xb, b.scale, b.shift, c0, C);
    const int64_t base = (int64_t)blockIdx.x * (PP * EW_ITER);
#pragma unroll 4
    for (int it = 0; it < EW_ITER; ++it) {
        const int64_t p = base + (int64_t)it * PP + pl;
        if (p >= P) break;
        const float4 av = uda_ld4(a.x + p * a.ldx + c0), bv = uda_ld4(b.x + p * b.ldx + c0);
        const float va[4] = {av.x, av.y, av.z, av.w}, vb[4] = {bv.x, bv.y, bv.z, bv.w};
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = fmaxf(uda_act(va[j] * xa.sc[j] + xa.sh[j], a.act) + uda_act(vb[j] * xb.sc[j] + xb.sh[j], b.act), 0.f);
        st4_guard(out + p * ldo + c0, v, C - c0);
    }
}

extern "C" int uda_bn_add_relu(const uda_src_t* a, const uda_src_t* b, float* out, int64_t ldo, void* stream) {
    if (int e = src_check(a, "uda_bn_add_relu")) return e;
    if (int e = src_check(b, "uda_bn_add_relu")) return e;
    UDA_REQUIRE(a->C == b->C && a->N == b->N && a->H == b->H && a->W == b->W && !a->mask && !b->mask,
                "uda_bn_add_relu: operands must have one shape and no masks");
    UDA_REQUIRE(out && uda_aligned16(out) && ldo % 4 == 0 && ldo >= a->C, "uda_bn_add_relu: bad out");
    const int64_t P = (int64_t)a->N * a->H * a->W;
    hipLaunchKernelGGL(bn_add_relu_kernel, ew_grid2(P, a->C), dim3(256), 0, (hipStream_t)stream, *a, *b, out, ldo, P);
    UDA_LAUNCH_CHECK("bn_add_relu");
    return 0;
}

// backward of that ReLU: g = dz where the stored block output z is positive
__global__ __launch_bounds__(256) void relu_gate_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ z,
                                                        int64_t ldz, int64_t P, int C4, float* __restrict__ out, int64_t ldo) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= P * C4) return;
    const int64_t p = e / C4;
    const int c0 = (int)(e % C4) * 4;
    const float4 g = uda_ld4(dz + p * lddz + c0), v = uda_ld4(z + p * ldz + c0);
    uda_st4(out + p * ldo + c0, make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f,
                                            v.w > 0.f ? g.w : 0.f));
}

extern "C" int uda_relu_gate(const float* dz, int64_t lddz, const float* z, int64_t ldz, int64_t P, int C, float* out,
                             int64_t ldo, void* stream) {
    UDA_REQUIRE(dz && z && out && P > 0 && C > 0 && C % 4 == 0 && lddz % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 &&
                    uda_aligned16(dz) && uda_aligned16(z) && uda_aligned16(out), "uda_relu_gate: bad args (C %% 4 == 0, aligned rows)");
    hipLaunchKernelGGL(relu_gate_kernel, dim3(uda_cdiv(P * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, dz, lddz, z, ldz,
                       P, C / 4, out, ldo);
    UDA_LAUNCH_CHECK("relu_gate");
    return 0;
}

// ------------------------------------------------------------------------------------------
// generic [P, C] column reductions.  MODE 0: (sum x [, sum x^2]);  MODE 1: BN backward sums.
struct RedArgs {
    const float* x;      // MODE 0: tensor; MODE 1: dU
    int64_t ldx;
    int64_t P;
    int C, nq;
    uda_src_t y;         // MODE 1
    const float* mean;
    const float* invstd;
    double* out;         // [UDA_STAT_SLOTS][nq][outC], fp64 atomics
    int outC;            // row length of the accumulator (C, or more when x's channels are a window of a wider statistic)
    // MODE 1, low-rank upstream gradient: dU[p, c] = sum_o lr_d[p, o] * lr_w[o, c], o < lr_k <= 2 (the input gradient of a 1x1 conv
    // to one or two outputs - the decoder's heads, decoder.py:32,41 - is an outer product; it is formed here instead of being
    // written by a conv and read back).  lr_d = null: dU is read from x.
    const float* lr_d;
    int64_t lr_ld;
    int lr_k;
    const float* lr_w;
};

// ITER: pixel strips per workgroup.  32 for the large layers (fewest atomics); 8 when 32 would leave fewer than ~2 workgroups per
// CU (the 32x32-map layers at B = 16: a workgroup then walks 256 rows with 8 pixel lanes - latency-bound on an under-filled chip)
template <int MODE, int ITER>
__global__ __launch_bounds__(256) void colreduce_kernel(RedArgs a) {
    __shared__ float red[3 * 1024];
    const int cblk0 = blockIdx.y * RED_CBLK_SUM;
    const int Cb = min(RED_CBLK_SUM, a.C - cblk0);
    const int G = (Cb + 3) >> 2, PP = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    const bool active = pl < PP;
    const int c0 = cblk0 + cg * 4;
    float acc[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] = 0.f;
    float sc[4], sh[4], mu[4], is[4], lw[2][4];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (c0 + j) < a.C;
            sc[j] = ok ? a.y.scale[c0 + j] : 1.f;
            sh[j] = ok ? a.y.shift[c0 + j] : 0.f;
            mu[j] = ok ? a.mean[c0 + j] : 0.f;
            is[j] = ok ? a.invstd[c0 + j] : 0.f;
#pragma unroll
            for (int o = 0; o < 2; ++o) lw[o][j] = (a.lr_d && ok && o < a.lr_k) ? a.lr_w[(int64_t)o * a.C + c0 + j] : 0.f;
        }
    }
    const int64_t base = (int64_t)blockIdx.x * (PP * ITER);
    // Loads are UNCONDITIONAL (a row beyond the matrix / an idle lane reads the last row with weight 0): with a branch around them the
    // compiler keeps one iteration's two loads in flight and the pass ran at 3.4-4.4 TB/s; four iterations' loads issued back to back
    // stream at the rate of a plain elementwise kernel.
    const int64_t plast = a.P - 1;
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        const int64_t pr = base + (int64_t)it * PP + pl;
        const bool ok = active && pr < a.P;
        const int64_t p = ok ? pr : plast;
        const float wgt = ok ? 1.f : 0.f;
        float v[4];
        if (MODE == 1 && a.lr_d) {
            const float d0 = a.lr_d[p * a.lr_ld] * wgt, d1 = a.lr_k > 1 ? a.lr_d[p * a.lr_ld + 1] * wgt : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = d0 * lw[0][j] + d1 * lw[1][j];
        } else {
            const float4 xv = uda_ld4(a.x + p * a.ldx + c0);
            v[0] = xv.x * wgt; v[1] = xv.y * wgt; v[2] = xv.z * wgt; v[3] = xv.w * wgt;
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = (c0 + j) < a.C ? v[j] : 0.f;
                acc[0][j] += t;
                acc[1][j] += t * t;
            }
        } else {
            const float4 yv4 = uda_ld4(a.y.x + p * a.y.ldx + c0);
            const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
            uint32_t mk = 0x01010101u;
            if (a.y.mask) mk = *reinterpret_cast<const uint32_t*>(a.y.mask + p * a.y.ldm + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((c0 + j) < a.C) {
                    const float du = v[j];
                    float g = du * uda_act_gate(yv[j] * sc[j] + sh[j], a.y.act);
                    if (a.y.mask) g *= ((mk >> (8 * j)) & 0xffu) ? a.y.mask_scale : 0.f;
                    acc[0][j] += g;
                    acc[1][j] += g * ((yv[j] - mu[j]) * is[j]);
                    acc[2][j] += du;
                }
            }
        }
    }
    const int Cp = G * 4;
    if (active) {
        for (int q = 0; q < a.nq; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(q * PP + pl) * Cp + cg * 4 + j] = acc[q][j];
    }
    __syncthreads();
    for (int e = tid; e < a.nq * Cb; e += 256) {
        const int q = e / Cb, c = e % Cb;
        float t = 0.f;
        for (int p = 0; p < PP; ++p) t += red[(q * PP + p) * Cp + c];
        atomicAdd(&a.out[((int64_t)(blockIdx.x % UDA_STAT_SLOTS) * a.nq + q) * a.outC + cblk0 + c], (double)t);
    }
}

static inline int red_nwg(int64_t P, int C, int iter) {
    const int Cb = C < RED_CBLK_SUM ? C : RED_CBLK_SUM;
    const int PP = 256 / ((Cb + 3) / 4);
    return uda_cdiv(P, (int64_t)PP * iter);
}
static inline bool red_short(int64_t P, int C) {
    static const int thr = getenv("UDA_RED_SHORT_WGS") ? atoi(getenv("UDA_RED_SHORT_WGS")) : 2048;
    return (int64_t)red_nwg(P, C, RED_ITER) * uda_cdiv(C, RED_CBLK_SUM) < thr;
}

extern "C" int uda_colstats_window(const float* x, int64_t ldx, int64_t P, int C, int nq, double* out, int out_C, void* stream);
extern "C" int uda_colstats(const float* x, int64_t ldx, int64_t P, int C, int nq, double* out, void* stream) {
    return uda_colstats_window(x, ldx, P, C, nq, out, C, stream);
}

/* uda_colstats into a channel WINDOW of a wider accumulator: out points at the window's first channel of slot 0 / quantity 0,
 * rows of the accumulator are out_C doubles long (double[UDA_STAT_SLOTS][nq][out_C]). */
extern "C" int uda_colstats_window(const float* x, int64_t ldx, int64_t P, int C, int nq, double* out, int out_C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(x && uda_aligned16(x) && ldx % 4 == 0 && ldx >= ((C + 3) / 4) * 4 && P > 0 && C > 0 && (nq == 1 || nq == 2) && out &&
                    out_C >= C, "uda_colstats: bad args");
    RedArgs a = {};
    a.x = x; a.ldx = ldx; a.P = P; a.C = C; a.nq = nq; a.mean = nullptr; a.invstd = nullptr; a.out = out; a.outC = out_C;
    if (red_short(P, C))
        hipLaunchKernelGGL((colreduce_kernel<0, 8>), dim3(red_nwg(P, C, 8), uda_cdiv(C, RED_CBLK_SUM)), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((colreduce_kernel<0, RED_ITER>), dim3(red_nwg(P, C, RED_ITER), uda_cdiv(C, RED_CBLK_SUM)), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("colstats");
    return 0;
}

static int bnbwd_reduce_launch(const float* dU, int64_t ldu, const float* lr_d, int64_t lr_ld, int lr_k, const float* lr_w,
                               const uda_src_t* y, const float* mean, const float* invstd, double* sums, void* stream);

extern "C" int uda_bnbwd_reduce(const float* dU, int64_t ldu, const uda_src_t* y, const float* mean, const float* invstd,
                                double* sums, void* stream) {
    if (int e = src_check(y, "uda_bnbwd_reduce")) return e;
    UDA_REQUIRE(dU && uda_aligned16(dU) && ldu % 4 == 0 && ldu >= ((y->C + 3) / 4) * 4, "uda_bnbwd_reduce: bad dU layout");
    return bnbwd_reduce_launch(dU, ldu, nullptr, 0, 0, nullptr, y, mean, invstd, sums, stream);
}

/* uda_bnbwd_reduce with the upstream gradient given in low-rank form dU[p, c] = sum_{o < k} d[p, o] * w[o, c], k = 1 or 2: the
 * input gradient of a 1x1 conv to k outputs (decoder.py:32 305 -> 2, :41 256 -> 1) never has to exist as a [P, C] matrix. */
extern "C" int uda_bnbwd_reduce_lowrank(const float* d, int64_t ldd, int k, const float* w, const uda_src_t* y, const float* mean,
                                        const float* invstd, double* sums, void* stream) {
    if (int e = src_check(y, "uda_bnbwd_reduce_lowrank")) return e;
    UDA_REQUIRE(d && w && (k == 1 || k == 2) && ldd >= k, "uda_bnbwd_reduce_lowrank: d [P, k], w [k, C], k = 1 or 2");
    return bnbwd_reduce_launch(nullptr, 0, d, ldd, k, w, y, mean, invstd, sums, stream);
}

static int bnbwd_reduce_launch(const float* dU, int64_t ldu, const float* lr_d, int64_t lr_ld, int lr_k, const float* lr_w,
                               const uda_src_t* y, const float* mean, const float* invstd, double* sums, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(y->scale && mean && invstd && sums, "uda_bnbwd_reduce: needs scale/shift/mean/invstd");
    const int64_t P = (int64_t)y->N * y->H * y->W;
    RedArgs a = {};
    a.x = dU; a.ldx = ldu; a.P = P; a.C = y->C; a.nq = 3; a.y = *y; a.mean = mean; a.invstd = invstd; a.out = sums; a.outC = y->C;
    a.lr_d = lr_d; a.lr_ld = lr_ld; a.lr_k = lr_k; a.lr_w = lr_w;
    if (red_short(P, y->C))
        hipLaunchKernelGGL((colreduce_kernel<1, 8>), dim3(red_nwg(P, y->C, 8), uda_cdiv(y->C, RED_CBLK_SUM)), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((colreduce_kernel<1, RED_ITER>), dim3(red_nwg(P, y->C, RED_ITER), uda_cdiv(y->C, RED_CBLK_SUM)), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("bnbwd_reduce");
    return 0;
}

// ------------------------------------------------------------------------------------------
__global__ void bnbwd_finalize_kernel(const double* __restrict__ sums, int C, double count, int q1, int act,
                                      const float* __restrict__ shift, const float* __restrict__ mean,
                                      const float* __restrict__ invstd, const float* __restrict__ q1_total,
                                      float* __restrict__ c1, float* __restrict__ c2,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sg = 0.0, sgx = 0.0, sdu = 0.0;
#pragma unroll
    for (int k = 0; k < UDA_STAT_SLOTS; ++k) {
        sg += sums[(k * 3 + 0) * C + c];
        sgx += sums[(k * 3 + 1) * C + c];
        sdu += sums[(k * 3 + 2) * C + c];
    }
    if (q1) {
        // quirk Q1: the zero border of the padded block input went through this BN.  Its upstream gradients sum to
        // (sum over ALL padded positions) - (sum of interior dU); the total is 0 when the depthwise BatchNorm behind it is in
        // training mode (its backward is mean-free) and is handed in as q1_total when that BatchNorm is frozen
        // (= colsum(dy_dw) * sum of the 9 depthwise taps, every tap of every output lands inside the padded input).
        const double gb = ((q1_total ? (double)q1_total[c] : 0.0) - sdu) * (double)uda_act_gate(shift[c], act);
        sg += gb;
        sgx += gb * (-(double)mean[c] * (double)invstd[c]);
    }
    c1[c] = (float)(sg / count);
    c2[c] = (float)(sgx / count);
    dgamma[c] = (float)sgx;
    dbeta[c] = (float)sg;
}

extern "C" int uda_bnbwd_finalize(const double* sums, int C, double count, int q1_border, int act, const float* shift,
                                  const float* mean, const float* invstd, const float* q1_total, float* c1, float* c2,
                                  float* dgamma, float* dbeta, void* stream) {
    UDA_REQUIRE(sums && shift && mean && invstd && c1 && c2 && dgamma && dbeta && C > 0 && count > 0, "uda_bnbwd_finalize: bad args");
    hipLaunchKernelGGL(bnbwd_finalize_kernel, dim3(uda_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, C, count,
                       q1_border, act, shift, mean, invstd, q1_total, c1, c2, dgamma, dbeta);
    UDA_LAUNCH_CHECK("bnbwd_finalize");
    return 0;
}

struct LowRank {            // dU[p, c] = sum_{o < k} d[p, o] * w[o, c] (see RedArgs); d = null: dU is a matrix
    const float* d;
    int64_t ld;
    int k;
    const float* w;
};

__global__ __launch_bounds__(256) void bnbwd_apply_kernel(const float* __restrict__ dU, int64_t ldu, uda_src_t y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ c1, const float* __restrict__ c2,
                                                          const float* addend, int64_t ld_add, float* out, int64_t ldo,
                                                          int64_t P, LowRank lr) {
    const int C = y.C, cblk0 = blockIdx.y * RED_CBLK;
    const int Cb = min(RED_CBLK, C - cblk0), G = (Cb + 3) >> 2, PP = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    if (pl >= PP) return;
    const int c0 = cblk0 + cg * 4;
    // dx = ad + sc*(g - c1 - xhat*c2),  xhat = (y - mu)*is   ==>   dx = ad + sc*g - (k0 + k1*y)
    float sc[4], sh[4], k0[4], k1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = (c0 + j) < C;
        sc[j] = ok ? y.scale[c0 + j] : 0.f;
        sh[j] = ok ? y.shift[c0 + j] : 0.f;
        const float mu = ok ? mean[c0 + j] : 0.f, is = ok ? invstd[c0 + j] : 0.f;
        const float a1 = ok ? c1[c0 + j] : 0.f, a2 = ok ? c2[c0 + j] : 0.f;
        k1[j] = sc[j] * a2 * is;
        k0[j] = sc[j] * a1 - k1[j] * mu;
    }
    float lw[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) lw[o][j] = (lr.d && o < lr.k && (c0 + j) < C) ? lr.w[(int64_t)o * C + c0 + j] : 0.f;
    const int64_t base = (int64_t)blockIdx.x * (PP * EW_ITER);
    // (loads unconditional - a strip beyond the matrix re-reads the last row and stores nothing - so that four strips' loads are in flight
    // together, as in colreduce_kernel)
#pragma unroll 4
    for (int it = 0; it < EW_ITER; ++it) {
        const int64_t pr = base + (int64_t)it * PP + pl;
        const bool ok = pr < P;
        const int64_t p = ok ? pr : P - 1;
        float du[4];
        if (lr.d) {
            const float d0 = lr.d[p * lr.ld], d1 = lr.k > 1 ? lr.d[p * lr.ld + 1] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) du[j] = d0 * lw[0][j] + d1 * lw[1][j];
        } else {
            const float4 dv = uda_ld4(dU + p * ldu + c0);
            du[0] = dv.x; du[1] = dv.y; du[2] = dv.z; du[3] = dv.w;
        }
        const float4 yv4 = uda_ld4(y.x + p * y.ldx + c0);
        const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
        uint32_t mk = 0x01010101u;
        if (y.mask) mk = *reinterpret_cast<const uint32_t*>(y.mask + p * y.ldm + c0);
        float ad[4] = {0.f, 0.f, 0.f, 0.f};
        if (addend) {
            const float4 av = uda_ld4(addend + p * ld_add + c0);
            ad[0] = av.x; ad[1] = av.y; ad[2] = av.z; ad[3] = av.w;
        }
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float g = du[j] * uda_act_gate(yv[j] * sc[j] + sh[j], y.act);
            if (y.mask) g *= (float)((mk >> (8 * j)) & 0xffu) * y.mask_scale;
            r[j] = ad[j] + (sc[j] * g - (k0[j] + k1[j] * yv[j]));
        }
        if (ok) st4_guard(out + p * ldo + c0, r, C - c0);
    }
}

extern "C" int uda_bnbwd_apply(const float* dU, int64_t ldu, const uda_src_t* y, const float* mean, const float* invstd,
                               const float* c1, const float* c2, const float* addend, int64_t ld_add, float* out,
                               int64_t ldo, void* stream) {
    if (int e = src_check(y, "uda_bnbwd_apply")) return e;
    UDA_REQUIRE(y->scale && mean && invstd && c1 && c2, "uda_bnbwd_apply: needs scale/shift/mean/invstd/c1/c2");
    UDA_REQUIRE(dU && uda_aligned16(dU) && ldu % 4 == 0 && out && uda_aligned16(out) && ldo % 4 == 0, "uda_bnbwd_apply: bad dU/out layout");
    if (addend) UDA_REQUIRE(uda_aligned16(addend) && ld_add % 4 == 0, "uda_bnbwd_apply: bad addend layout");
    const int64_t P = (int64_t)y->N * y->H * y->W;
    LowRank lr = {nullptr, 0, 0, nullptr};
    hipLaunchKernelGGL(bnbwd_apply_kernel, ew_grid2(P, y->C), dim3(256), 0, (hipStream_t)stream, dU, ldu, *y, mean, invstd,
                       c1, c2, addend, ld_add, out, ldo, P, lr);
    UDA_LAUNCH_CHECK("bnbwd_apply");
    return 0;
}

/* uda_bnbwd_apply with the low-rank upstream gradient of uda_bnbwd_reduce_lowrank */
extern "C" int uda_bnbwd_apply_lowrank(const float* d, int64_t ldd, int k, const float* w, const uda_src_t* y, const float* mean,
                                       const float* invstd, const float* c1, const float* c2, const float* addend, int64_t ld_add,
                                       float* out, int64_t ldo, void* stream) {
    if (int e = src_check(y, "uda_bnbwd_apply_lowrank")) return e;
    UDA_REQUIRE(y->scale && mean && invstd && c1 && c2, "uda_bnbwd_apply_lowrank: needs scale/shift/mean/invstd/c1/c2");
    UDA_REQUIRE(d && w && (k == 1 || k == 2) && ldd >= k && out && uda_aligned16(out) && ldo % 4 == 0, "uda_bnbwd_apply_lowrank: bad args");
    if (addend) UDA_REQUIRE(uda_aligned16(addend) && ld_add % 4 == 0, "uda_bnbwd_apply_lowrank: bad addend layout");
    const int64_t P = (int64_t)y->N * y->H * y->W;
    LowRank lr = {d, ldd, k, w};
    hipLaunchKernelGGL(bnbwd_apply_kernel, ew_grid2(P, y->C), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, (int64_t)0, *y,
                       mean, invstd, c1, c2, addend, ld_add, out, ldo, P, lr);
    UDA_LAUNCH_CHECK("bnbwd_apply_lowrank");
    return 0;
}
