// Patch-discriminator geometry (networks/GAN.py:86-148 of the reference): every layer is
// Conv2d(k=4, stride 2, pad 2, no bias) [+ LeakyReLU(0.2)].  A 4x4 stride-2 convolution of x equals a 2x2
// stride-1 convolution of the space-to-depth image of the zero-padded x,
//     z[n, i, j, (a, b, c)] = xpad[n, 2i + a, 2j + b, c] = x[n, 2i + a - 2, 2j + b - 2, c],
// with weights wz[co, (u, v), (a, b, c)] = w[co, c, 2u + a, 2v + b] - which is what the implicit-GEMM kernels of
// igemm_*.hip run with ksize = 2.  The two kernels here build z (fusing the previous layer's LeakyReLU and the
// crop of its output grid to the valid Ho x Wo region) and route gradients back (fusing the LeakyReLU gate,
// read from the sign of z).  HBM-bound streaming passes, ~3 % of a discriminator forward+backward.
#include "common.h"

struct S2dArgs {
    const float* src;     // forward: [N, Hs, Ws, C] rows (ld) or NCHW [N, C, Hs, Ws];  backward: dz [N, Hz, Wz, 4C]
    int64_t ld_src;
    const float* zsign;   // backward: z (same layout as dz), its sign is the LeakyReLU gate; null = no gate
    float* dst;           // forward: z [N, Hz, Wz, 4C];  backward: [N, Hs, Ws, C] rows or NCHW
    int64_t ld_dst;
    int N, Hs, Ws, C, vh, vw, Hz, Wz;
    int nchw;             // the un-s2d side (src forward / dst backward) is NCHW
    float slope;
    // first layer fed by generator LOGITS (Trainer_prototype_full.py:452-454, 479-487): 0 = plain values,
    // 1 = sigmoid(x), 2 = uncertainty map -sigmoid(x) * log(sigmoid(x) + 1e-7).  Backward multiplies the routed gradient by the
    // derivative at the logits `pre_x` (NCHW, the forward's source).
    int pre_op;
    const float* pre_x;
    // backward, alternative gate: rows [N, Hs, Ws, C] of the forward's SOURCE (z = lrelu(source), so the signs agree) - used when
    // the fp32 z image was never written (bf16x3 mode packs it straight from the source, uda_x3_pack_s2d_fwd)
    const float* gate;
    int64_t ld_gate;
};

#define ADV_SMOOTH 1e-7f
__device__ __forceinline__ float adv_pre(float x, int op) {
    if (op == 0) return x;
    const float s = 1.f / (1.f + expf(-x));
    return op == 1 ? s : -1.f * s * logf(s + ADV_SMOOTH);
}
__device__ __forceinline__ float adv_pre_grad(float x, int op) {          // d adv_pre / dx
    if (op == 0) return 1.f;
    const float s = 1.f / (1.f + expf(-x));
    const float ds = s * (1.f - s);
    return op == 1 ? ds : -(logf(s + ADV_SMOOTH) + s / (s + ADV_SMOOTH)) * ds;
}

// forward: one thread per (n, i, j, a, b, c)
__global__ __launch_bounds__(256) void s2d_fwd_kernel(S2dArgs p) {
    const int C4 = 4 * p.C;
    const int64_t total = (int64_t)p.N * p.Hz * p.Wz * C4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int q = (int)(e % C4);
        const int64_t pz = e / C4;
        const int c = q % p.C, ab = q / p.C;
        const int j = (int)(pz % p.Wz), i = (int)((pz / p.Wz) % p.Hz), n = (int)(pz / ((int64_t)p.Wz * p.Hz));
        const int h = 2 * i + (ab >> 1) - 2, w = 2 * j + (ab & 1) - 2;
        float v = 0.f;
        if (h >= 0 && h < p.vh && w >= 0 && w < p.vw) {
            v = p.nchw ? p.src[(((int64_t)n * p.C + c) * p.Hs + h) * p.Ws + w]
                       : p.src[(((int64_t)n * p.Hs + h) * p.Ws + w) * p.ld_src + c];
            if (p.pre_op) v = adv_pre(v, p.pre_op);
            v = v > 0.f ? v : v * p.slope;
        }
        p.dst[pz * p.ld_dst + q] = v;
    }
}

// same, 4 channels per thread (C % 4 == 0, rows-in)
__global__ __launch_bounds__(256) void s2d_fwd4_kernel(S2dArgs p) {
    const int Cq = p.C >> 2, C4q = 4 * Cq;
    const int64_t total = (int64_t)p.N * p.Hz * p.Wz * C4q;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int q = (int)(e % C4q);
        const int64_t pz = e / C4q;
        const int c = (q % Cq) * 4, ab = q / Cq;
        const int j = (int)(pz % p.Wz), i = (int)((pz / p.Wz) % p.Hz), n = (int)(pz / ((int64_t)p.Wz * p.Hz));
        const int h = 2 * i + (ab >> 1) - 2, w = 2 * j + (ab & 1) - 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= 0 && h < p.vh && w >= 0 && w < p.vw) {
            v = uda_ld4(p.src + (((int64_t)n * p.Hs + h) * p.Ws + w) * p.ld_src + c);
            v.x = v.x > 0.f ? v.x : v.x * p.slope;
            v.y = v.y > 0.f ? v.y : v.y * p.slope;
            v.z = v.z > 0.f ? v.z : v.z * p.slope;
            v.w = v.w > 0.f ? v.w : v.w * p.slope;
        }
        uda_st4(p.dst + pz * p.ld_dst + ab * p.C + c, v);
    }
}

// backward: one thread per element of the un-s2d side (n, h, w, c); every such element owns exactly one z slot
__global__ __launch_bounds__(256) void s2d_bwd_kernel(S2dArgs p) {
    const int64_t total = (int64_t)p.N * p.Hs * p.Ws * p.C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int n, h, w, c;
        if (p.nchw) {
            w = (int)(e % p.Ws); h = (int)((e / p.Ws) % p.Hs); c = (int)((e / ((int64_t)p.Ws * p.Hs)) % p.C);
            n = (int)(e / ((int64_t)p.Ws * p.Hs * p.C));
        } else {
            c = (int)(e % p.C); w = (int)((e / p.C) % p.Ws); h = (int)((e / ((int64_t)p.C * p.Ws)) % p.Hs);
            n = (int)(e / ((int64_t)p.C * p.Ws * p.Hs));
        }
        float g = 0.f;
        if (h < p.vh && w < p.vw) {
            const int64_t pz = ((int64_t)n * p.Hz + ((h + 2) >> 1)) * p.Wz + ((w + 2) >> 1);
            const int q = ((((h + 2) & 1) << 1) | ((w + 2) & 1)) * p.C + c;
            g = p.src[pz * p.ld_src + q];
            if (p.zsign && !(p.zsign[pz * p.ld_src + q] > 0.f)) g *= p.slope;
            if (p.gate && !(p.gate[(((int64_t)n * p.Hs + h) * p.Ws + w) * p.ld_gate + c] > 0.f)) g *= p.slope;
            if (p.pre_op) g *= adv_pre_grad(p.pre_x[e], p.pre_op);        // nchw only (checked by the entry point): e indexes the logits
        }
        if (p.nchw) p.dst[e] = g;
        else p.dst[(((int64_t)n * p.Hs + h) * p.Ws + w) * p.ld_dst + c] = g;
    }
}

__global__ __launch_bounds__(256) void s2d_bwd4_kernel(S2dArgs p) {
    const int Cq = p.C >> 2;
    const int64_t total = (int64_t)p.N * p.Hs * p.Ws * Cq;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % Cq) * 4, w = (int)((e / Cq) % p.Ws), h = (int)((e / ((int64_t)Cq * p.Ws)) % p.Hs);
        const int n = (int)(e / ((int64_t)Cq * p.Ws * p.Hs));
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h < p.vh && w < p.vw) {
            const int64_t pz = ((int64_t)n * p.Hz + ((h + 2) >> 1)) * p.Wz + ((w + 2) >> 1);
            const int q = ((((h + 2) & 1) << 1) | ((w + 2) & 1)) * p.C + c;
            g = uda_ld4(p.src + pz * p.ld_src + q);
            if (p.zsign) {
                const float4 z = uda_ld4(p.zsign + pz * p.ld_src + q);
                if (!(z.x > 0.f)) g.x *= p.slope;
                if (!(z.y > 0.f)) g.y *= p.slope;
                if (!(z.z > 0.f)) g.z *= p.slope;
                if (!(z.w > 0.f)) g.w *= p.slope;
            }
            if (p.gate) {
                const float4 z = uda_ld4(p.gate + (((int64_t)n * p.Hs + h) * p.Ws + w) * p.ld_gate + c);
                if (!(z.x > 0.f)) g.x *= p.slope;
                if (!(z.y > 0.f)) g.y *= p.slope;
                if (!(z.z > 0.f)) g.z *= p.slope;
                if (!(z.w > 0.f)) g.w *= p.slope;
            }
        }
        uda_st4(p.dst + (((int64_t)n * p.Hs + h) * p.Ws + w) * p.ld_dst + c, g);
    }
}

static int s2d_check(const char* who, const void* a, const void* b, int N, int Hs, int Ws, int C, int vh, int vw, int Hz, int Wz) {
    UDA_REQUIRE(a && b && N > 0 && Hs > 0 && Ws > 0 && C > 0 && vh > 0 && vw > 0 && vh <= Hs && vw <= Ws, "%s: bad dims", who);
    UDA_REQUIRE(Hz == (vh + 5) / 2 && Wz == (vw + 5) / 2, "%s: the z grid of a %dx%d region is %dx%d, got %dx%d", who, vh, vw,
                (vh + 5) / 2, (vw + 5) / 2, Hz, Wz);
    return 0;
}

extern "C" int uda_s2d_fwd(const float* src, int64_t ld_src, int nchw_in, int N, int Hs, int Ws, int C, int valid_h, int valid_w,
                           float slope, float* z, int64_t ld_z, int Hz, int Wz, void* stream) {
    if (int e = s2d_check("uda_s2d_fwd", src, z, N, Hs, Ws, C, valid_h, valid_w, Hz, Wz)) return e;
    UDA_REQUIRE(ld_z >= 4 * C && (nchw_in || ld_src >= C), "uda_s2d_fwd: leading dimensions too small");
    S2dArgs p = {};
    p.src = src; p.ld_src = ld_src; p.zsign = nullptr; p.dst = z; p.ld_dst = ld_z;
    p.N = N; p.Hs = Hs; p.Ws = Ws; p.C = C; p.vh = valid_h; p.vw = valid_w; p.Hz = Hz; p.Wz = Wz; p.nchw = nchw_in; p.slope = slope;
    p.pre_op = 0; p.pre_x = nullptr;
    const bool v4 = !nchw_in && C % 4 == 0 && ld_src % 4 == 0 && ld_z % 4 == 0 && uda_aligned16(src) && uda_aligned16(z);
    const int64_t total = (int64_t)N * Hz * Wz * (v4 ? C : 4 * C);
    const int grid = (int)(uda_cdiv(total, 256) > 16384 ? 16384 : uda_cdiv(total, 256));
    if (v4) hipLaunchKernelGGL(s2d_fwd4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(s2d_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    UDA_LAUNCH_CHECK("s2d_fwd");
    return 0;
}

extern "C" int uda_s2d_bwd(const float* dz, const float* z_sign, int64_t ld_z, int Hz, int Wz, float slope, int N, int Hs, int Ws,
                           int C, int valid_h, int valid_w, float* dst, int64_t ld_dst, int nchw_out, void* stream) {
    if (int e = s2d_check("uda_s2d_bwd", dz, dst, N, Hs, Ws, C, valid_h, valid_w, Hz, Wz)) return e;
    UDA_REQUIRE(ld_z >= 4 * C && (nchw_out || ld_dst >= C), "uda_s2d_bwd: leading dimensions too small");
    S2dArgs p = {};
    p.src = dz; p.ld_src = ld_z; p.zsign = z_sign; p.dst = dst; p.ld_dst = ld_dst;
    p.N = N; p.Hs = Hs; p.Ws = Ws; p.C = C; p.vh = valid_h; p.vw = valid_w; p.Hz = Hz; p.Wz = Wz; p.nchw = nchw_out; p.slope = slope;
    p.pre_op = 0; p.pre_x = nullptr;
    const bool v4 = !nchw_out && C % 4 == 0 && ld_z % 4 == 0 && ld_dst % 4 == 0 && uda_aligned16(dz) && uda_aligned16(dst) &&
                    (!z_sign || uda_aligned16(z_sign));
    const int64_t total = (int64_t)N * Hs * Ws * (v4 ? C / 4 : C);
    const int grid = (int)(uda_cdiv(total, 256) > 16384 ? 16384 : uda_cdiv(total, 256));
    if (v4) hipLaunchKernelGGL(s2d_bwd4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(s2d_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    UDA_LAUNCH_CHECK("s2d_bwd");
    return 0;
}

/* uda_s2d_bwd with the LeakyReLU gate read from the sign of the forward's SOURCE rows [N, Hs, Ws, C] instead of the z image
 * (same signs: z = lrelu(source)); rows out. */
extern "C" int uda_s2d_bwd_gate(const float* dz, int64_t ld_z, int Hz, int Wz, const float* gate_rows, int64_t ld_gate, float slope,
                                int N, int Hs, int Ws, int C, int valid_h, int valid_w, float* dst, int64_t ld_dst, void* stream) {
    if (int e = s2d_check("uda_s2d_bwd_gate", dz, dst, N, Hs, Ws, C, valid_h, valid_w, Hz, Wz)) return e;
    UDA_REQUIRE(ld_z >= 4 * C && ld_dst >= C && gate_rows && ld_gate >= C, "uda_s2d_bwd_gate: leading dimensions too small / no gate");
    S2dArgs p = {};
    p.src = dz; p.ld_src = ld_z; p.zsign = nullptr; p.dst = dst; p.ld_dst = ld_dst;
    p.N = N; p.Hs = Hs; p.Ws = Ws; p.C = C; p.vh = valid_h; p.vw = valid_w; p.Hz = Hz; p.Wz = Wz; p.nchw = 0; p.slope = slope;
    p.pre_op = 0; p.pre_x = nullptr; p.gate = gate_rows; p.ld_gate = ld_gate;
    const bool v4 = C % 4 == 0 && ld_z % 4 == 0 && ld_dst % 4 == 0 && ld_gate % 4 == 0 && uda_aligned16(dz) && uda_aligned16(dst) &&
                    uda_aligned16(gate_rows);
    const int64_t total = (int64_t)N * Hs * Ws * (v4 ? C / 4 : C);
    const int grid = (int)(uda_cdiv(total, 256) > 16384 ? 16384 : uda_cdiv(total, 256));
    if (v4) hipLaunchKernelGGL(s2d_bwd4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(s2d_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    UDA_LAUNCH_CHECK("s2d_bwd_gate");
    return 0;
}

// First discriminator layer fed by generator logits: z = s2d(pre(logits)) and its adjoint d logits = route(dz) * pre'(logits),
// pre = sigmoid (boundary branch) or the uncertainty map (Trainer_prototype_full.py:452-454); replaces the elementwise
// sigmoid / log / mul chain and its autograd backward on the full-resolution maps.
extern "C" int uda_adv_s2d_fwd(const float* logits_nchw, int N, int C, int H, int W, int pre_op, float* z, int64_t ld_z, int Hz, int Wz,
                               void* stream) {
    if (int e = s2d_check("uda_adv_s2d_fwd", logits_nchw, z, N, H, W, C, H, W, Hz, Wz)) return e;
    UDA_REQUIRE(ld_z >= 4 * C && (pre_op == 1 || pre_op == 2), "uda_adv_s2d_fwd: bad leading dimension or pre_op");
    S2dArgs p = {};
    p.src = logits_nchw; p.ld_src = 0; p.zsign = nullptr; p.dst = z; p.ld_dst = ld_z;
    p.N = N; p.Hs = H; p.Ws = W; p.C = C; p.vh = H; p.vw = W; p.Hz = Hz; p.Wz = Wz; p.nchw = 1; p.slope = 1.f;
    p.pre_op = pre_op; p.pre_x = nullptr;
    const int64_t total = (int64_t)N * Hz * Wz * 4 * C;
    const int grid = (int)(uda_cdiv(total, 256) > 16384 ? 16384 : uda_cdiv(total, 256));
    hipLaunchKernelGGL(s2d_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    UDA_LAUNCH_CHECK("adv_s2d_fwd");
    return 0;
}

extern "C" int uda_adv_s2d_bwd(const float* dz, int64_t ld_z, int Hz, int Wz, const float* logits_nchw, int N, int C, int H, int W,
                               int pre_op, float* d_logits_nchw, void* stream) {
    if (int e = s2d_check("uda_adv_s2d_bwd", dz, d_logits_nchw, N, H, W, C, H, W, Hz, Wz)) return e;
    UDA_REQUIRE(ld_z >= 4 * C && logits_nchw && (pre_op == 1 || pre_op == 2), "uda_adv_s2d_bwd: bad args");
    S2dArgs p = {};
    p.src = dz; p.ld_src = ld_z; p.zsign = nullptr; p.dst = d_logits_nchw; p.ld_dst = 0;
    p.N = N; p.Hs = H; p.Ws = W; p.C = C; p.vh = H; p.vw = W; p.Hz = Hz; p.Wz = Wz; p.nchw = 1; p.slope = 1.f;
    p.pre_op = pre_op; p.pre_x = logits_nchw;
    const int64_t total = (int64_t)N * H * W * C;
    const int grid = (int)(uda_cdiv(total, 256) > 16384 ? 16384 : uda_cdiv(total, 256));
    hipLaunchKernelGGL(s2d_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    UDA_LAUNCH_CHECK("adv_s2d_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------
// z-space weight operands of a [O, C, 4, 4] discriminator weight, in the tap-chunked K order of the implicit GEMM
// (k = (cc*4 + t)*32 + q % 32, cc = q / 32 over the operand's channel index q, zero padded to a multiple of 32):
//   forward : rows o,       taps t = (u, v),          channels q = (a, b, c):  w[o, c, 2u+a, 2v+b]
//   dgrad   : rows (a,b,c), taps t = 3 - (u, v) flip, channels q = o        :  the same element
__global__ void relayout_s2d_kernel(const float* __restrict__ w, int O, int C, int dgrad, int Kr, int Kc, int64_t total,
                                    float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(e % Kr), row = (int)(e / Kr);
        int t, q;
        if (Kc < 32) { t = k / Kc; q = k - t * Kc; }          // fewer than 32 channels: tap-major, unpadded
        else { const int chunk = k >> 5; t = chunk & 3; q = (chunk >> 2) * 32 + (k & 31); }
        int o, abc, uv;
        if (!dgrad) { o = row; abc = q; uv = t; }
        else { o = q; abc = row; uv = 3 - t; }
        float v = 0.f;
        if (o < O && abc < 4 * C) {
            const int c = abc % C, ab = abc / C;
            const int kh = 2 * (uv >> 1) + (ab >> 1), kw = 2 * (uv & 1) + (ab & 1);
            v = w[(((int64_t)o * C + c) * 4 + kh) * 4 + kw];
        }
        out[e] = v;
    }
}

extern "C" int uda_relayout_s2d(const float* w, int O, int C, int dgrad, float* out, void* stream) {
    UDA_REQUIRE(w && out && O > 0 && C > 0, "uda_relayout_s2d: bad args");
    const int chan = dgrad ? O : 4 * C, rows = dgrad ? 4 * C : O;
    const int Kc = ((chan + 3) / 4) * 4;
    const int Kr = Kc < 32 ? 4 * Kc : ((Kc + 31) / 32) * 4 * 32;
    const int64_t total = (int64_t)rows * Kr;
    const int grid = (int)(uda_cdiv(total, 256) > 8192 ? 8192 : uda_cdiv(total, 256));
    hipLaunchKernelGGL(relayout_s2d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, O, C, dgrad, Kr, Kc, total, out);
    UDA_LAUNCH_CHECK("relayout_s2d");
    return 0;
}
