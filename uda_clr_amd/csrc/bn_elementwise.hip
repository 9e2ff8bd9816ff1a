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

int uda_reduce_partials(const float* part, int nrows, int ncols, double* out, hipStream_t st);

#define RED_ITER 32
#define RED_CBLK 1024

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
    const double m = stats[c] / count;
    double var = stats[C + c] / count - m * m;
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

extern "C" int uda_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, int C, float eps, float* scale, float* shift, void* stream) {
    UDA_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "uda_bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(uda_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, C, eps, scale, shift);
    UDA_LAUNCH_CHECK("bn_eval_coeffs");
    return 0;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_apply_kernel(uda_src_t s, const float* __restrict__ res, int64_t ldr,
                                                       float* __restrict__ out, int64_t ldo, int64_t P) {
    const int C = s.C, G = (C + 3) >> 2;
    const int64_t total = P * G;
    const bool has_xf = s.scale != nullptr;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const int c0 = cg * 4;
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
            const int c = c0 + j;
            float u = v[j];
            if (c < C) {
                if (has_xf) u = u * s.scale[c] + s.shift[c];
                u = uda_act(u, s.act);
                if (s.mask) u *= ((mk >> (8 * j)) & 0xffu) ? s.mask_scale : 0.f;
                u += r[j];
            }
            v[j] = u;
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

static inline int ew_grid(int64_t total) {
    int g = uda_cdiv(total, 256);
    if (g > 16384) g = 16384;
    return g < 1 ? 1 : g;
}

extern "C" int uda_bn_apply(const uda_src_t* src, const float* residual, int64_t ldr, float* out, int64_t ldo, void* stream) {
    if (int e = src_check(src, "uda_bn_apply")) return e;
    UDA_REQUIRE(out && uda_aligned16(out) && ldo % 4 == 0 && ldo >= src->C, "uda_bn_apply: out must be 16-byte aligned, ldo %% 4 == 0");
    if (residual) UDA_REQUIRE(uda_aligned16(residual) && ldr % 4 == 0, "uda_bn_apply: residual must be 16-byte aligned");
    const int64_t P = (int64_t)src->N * src->H * src->W;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(P * ((src->C + 3) / 4))), dim3(256), 0, (hipStream_t)stream, *src,
                       residual, ldr, out, ldo, P);
    UDA_LAUNCH_CHECK("bn_apply");
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
    float* part;         // [nWG][nq][C]
};

template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(RedArgs a) {
    __shared__ float red[3 * 1024];
    const int cblk0 = blockIdx.y * RED_CBLK;
    const int Cb = min(RED_CBLK, a.C - cblk0);
    const int G = (Cb + 3) >> 2, PP = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, pl = tid / G;
    const bool active = pl < PP;
    const int c0 = cblk0 + cg * 4;
    float acc[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] = 0.f;
    float sc[4], sh[4], mu[4], is[4];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (c0 + j) < a.C;
            sc[j] = ok ? a.y.scale[c0 + j] : 1.f;
            sh[j] = ok ? a.y.shift[c0 + j] : 0.f;
            mu[j] = ok ? a.mean[c0 + j] : 0.f;
            is[j] = ok ? a.invstd[c0 + j] : 0.f;
        }
    }
    const int64_t base = (int64_t)blockIdx.x * (PP * RED_ITER);
    for (int it = 0; it < RED_ITER; ++it) {
        const int64_t p = base + (int64_t)it * PP + pl;
        if (!active || p >= a.P) continue;
        const float4 xv = uda_ld4(a.x + p * a.ldx + c0);
        const float v[4] = {xv.x, xv.y, xv.z, xv.w};
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
        a.part[((int64_t)blockIdx.x * a.nq + q) * a.C + cblk0 + c] = t;
    }
}

static inline int red_nwg(int64_t P, int C) {
    const int Cb = C < RED_CBLK ? C : RED_CBLK;
    const int PP = 256 / ((Cb + 3) / 4);
    return uda_cdiv(P, (int64_t)PP * RED_ITER);
}

extern "C" uint64_t uda_reduce_workspace_bytes(int64_t P, int C, int nq) {
    return (uint64_t)red_nwg(P, C) * nq * C * sizeof(float);
}

extern "C" int uda_colstats(const float* x, int64_t ldx, int64_t P, int C, int nq, double* out, float* workspace,
                            uint64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(x && uda_aligned16(x) && ldx % 4 == 0 && ldx >= ((C + 3) / 4) * 4 && P > 0 && C > 0 && (nq == 1 || nq == 2) && out,
                "uda_colstats: bad args");
    UDA_REQUIRE(workspace && workspace_bytes >= uda_reduce_workspace_bytes(P, C, nq), "uda_colstats: workspace too small");
    RedArgs a;
    a.x = x; a.ldx = ldx; a.P = P; a.C = C; a.nq = nq; a.mean = nullptr; a.invstd = nullptr; a.part = workspace;
    const int nwg = red_nwg(P, C);
    hipLaunchKernelGGL((colreduce_kernel<0>), dim3(nwg, uda_cdiv(C, RED_CBLK)), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("colstats");
    return uda_reduce_partials(workspace, nwg, nq * C, out, st);
}

extern "C" int uda_bnbwd_reduce(const float* dU, int64_t ldu, const uda_src_t* y, const float* mean, const float* invstd,
                                double* sums, float* workspace, uint64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (int e = src_check(y, "uda_bnbwd_reduce")) return e;
    UDA_REQUIRE(y->scale && mean && invstd && sums, "uda_bnbwd_reduce: needs scale/shift/mean/invstd");
    UDA_REQUIRE(dU && uda_aligned16(dU) && ldu % 4 == 0 && ldu >= ((y->C + 3) / 4) * 4, "uda_bnbwd_reduce: bad dU layout");
    const int64_t P = (int64_t)y->N * y->H * y->W;
    UDA_REQUIRE(workspace && workspace_bytes >= uda_reduce_workspace_bytes(P, y->C, 3), "uda_bnbwd_reduce: workspace too small");
    RedArgs a;
    a.x = dU; a.ldx = ldu; a.P = P; a.C = y->C; a.nq = 3; a.y = *y; a.mean = mean; a.invstd = invstd; a.part = workspace;
    const int nwg = red_nwg(P, y->C);
    hipLaunchKernelGGL((colreduce_kernel<1>), dim3(nwg, uda_cdiv(y->C, RED_CBLK)), dim3(256), 0, st, a);
    UDA_LAUNCH_CHECK("bnbwd_reduce");
    return uda_reduce_partials(workspace, nwg, 3 * y->C, sums, st);
}

// ------------------------------------------------------------------------------------------
__global__ void bnbwd_finalize_kernel(const double* __restrict__ sums, int C, double count, int q1, int act,
                                      const float* __restrict__ shift, const float* __restrict__ mean,
                                      const float* __restrict__ invstd, float* __restrict__ c1, float* __restrict__ c2,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sg = sums[c], sgx = sums[C + c];
    if (q1) {
        // quirk Q1: the zero border of the padded block input went through this BN; its upstream
        // gradients sum to -(sum of interior dU) because the depthwise BN backward is mean-free.
        const double gb = -sums[2 * C + c] * (double)uda_act_gate(shift[c], act);
        sg += gb;
        sgx += gb * (-(double)mean[c] * (double)invstd[c]);
    }
    c1[c] = (float)(sg / count);
    c2[c] = (float)(sgx / count);
    dgamma[c] = (float)sgx;
    dbeta[c] = (float)sg;
}

extern "C" int uda_bnbwd_finalize(const double* sums, int C, double count, int q1_border, int act, const float* shift,
                                  const float* mean, const float* invstd, float* c1, float* c2, float* dgamma,
                                  float* dbeta, void* stream) {
    UDA_REQUIRE(sums && shift && mean && invstd && c1 && c2 && dgamma && dbeta && C > 0 && count > 0, "uda_bnbwd_finalize: bad args");
    hipLaunchKernelGGL(bnbwd_finalize_kernel, dim3(uda_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, C, count,
                       q1_border, act, shift, mean, invstd, c1, c2, dgamma, dbeta);
    UDA_LAUNCH_CHECK("bnbwd_finalize");
    return 0;
}

__global__ __launch_bounds__(256) void bnbwd_apply_kernel(const float* __restrict__ dU, int64_t ldu, uda_src_t y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ c1, const float* __restrict__ c2,
                                                          const float* addend, int64_t ld_add, float* out, int64_t ldo,
                                                          int64_t P) {
    const int C = y.C, G = (C + 3) >> 2;
    const int64_t total = P * G;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const int c0 = cg * 4;
        const float4 dv = uda_ld4(dU + p * ldu + c0);
        const float4 yv4 = uda_ld4(y.x + p * y.ldx + c0);
        const float du[4] = {dv.x, dv.y, dv.z, dv.w};
        const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
        uint32_t mk = 0x01010101u;
        if (y.mask) mk = *reinterpret_cast<const uint32_t*>(y.mask + p * y.ldm + c0);
        float ad[4] = {0.f, 0.f, 0.f, 0.f};
        if (addend) {
            const float4 av = uda_ld4(addend + p * ld_add + c0);
            ad[0] = av.x; ad[1] = av.y; ad[2] = av.z; ad[3] = av.w;
        }
        float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            if (c < C) {
                const float sc = y.scale[c];
                float g = du[j] * uda_act_gate(yv[j] * sc + y.shift[c], y.act);
                if (y.mask) g *= ((mk >> (8 * j)) & 0xffu) ? y.mask_scale : 0.f;
                const float xhat = (yv[j] - mean[c]) * invstd[c];
                r[j] = ad[j] + sc * (g - c1[c] - xhat * c2[c]);
            }
        }
        st4_guard(out + p * ldo + c0, r, C - c0);
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
    hipLaunchKernelGGL(bnbwd_apply_kernel, dim3(ew_grid(P * ((y->C + 3) / 4))), dim3(256), 0, (hipStream_t)stream, dU, ldu,
                       *y, mean, invstd, c1, c2, addend, ld_add, out, ldo, P);
    UDA_LAUNCH_CHECK("bnbwd_apply");
    return 0;
}
