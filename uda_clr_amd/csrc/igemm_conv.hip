// FP32-MFMA implicit-GEMM convolution for gfx950 (stride 1, 1x1 / 3x3, any dilation), NHWC.
//
//   y[p, co] = bias[co] + addend[p, co] + sum_k A(p, k) * Wt[co, k]
//   A(p, k)  = u(p + off_t, ci)  (prologue-transformed input, 0 outside the image / ci >= C)
//   K order ("tap-chunked"): chunk = cc*T + t covers channels 32cc .. 32cc+31 of tap t, so all T taps of one
//   channel slice are consecutive chunks and re-read the workgroup's pixel strip while it is L2-resident
//   (PMC: 1.6x the algorithmic bytes on the memory side; the tap-major order k = t*Kc + ci measured 4.6x).
//
// Roofline: MFMA-bound for the decoder / ASPP 3x3 convs (560 / 284 FLOP per byte) on
// v_mfma_f32_32x32x2_f32 (exact fp32, 157.3 TF peak, 64 cycles per instruction per SIMD); the
// backbone 1x1 convs are HBM-bound and use the narrow-N configurations.
//
// Work decomposition (wave64, 4 waves = 256 threads per workgroup, one wave per SIMD):
//   workgroup tile BM x BN = (32*TM*WM) x (32*TN*WN) outputs, K walked in chunks of BK = 32.
//   Both operand tiles are staged global -> registers -> LDS as [row][BK+4] fp32 (row stride 36
//   dwords: conflict-free for the 16-lane groups of ds_read_b128).  The next chunk's global loads
//   are issued before the MFMAs of the current chunk (async-STAGE split, guide T14) and the BN
//   affine / ReLU / dropout prologue runs in registers just before the LDS write.
//   k-permutation: inside each group of 8 k, lane half h supplies k = 8g + 4h + s to MFMA step s
//   for BOTH operands, so one ds_read_b128 per operand row feeds 4 MFMAs.
//   Tiles are ordered n-fastest and XCD-chunked (uda_xcd_remap) so the two N tiles of a pixel
//   tile and spatially adjacent pixel tiles share one XCD's L2.
#include <algorithm>
#include "common.h"
#include <stdlib.h>
#include "igemm_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));


// PIPE: two LDS tile images and two register sets - chunk c is computed from one image while chunk c + 1 is staged into the other and
// the global loads of chunk c + 2 are in flight, one barrier per chunk (round 3: the long-K project convs of the 32x32-map layers run
// one workgroup per CU and sat at the global-load latency of every chunk: 12 chunks x 1.7 us against 0.47 us of MFMA work each).
// PIPE = false: one image, loads one chunk ahead, two barriers per chunk (short K: fewer registers and half the LDS per workgroup).
// Registers: without a bound the compiler spends 125 VGPRs + 48 AGPRs on <1,3,4,1> (two workgroups per CU, and the output-bound expand
// convs wrote at 2.4 TB/s: load, MFMA and store phases of two workgroups do not cover each other); the lean form is asked for 4 / 3
// waves per SIMD by accumulator count, the pipelined form keeps 2.
#define IGC_MIN_WAVES(TM, TN, PIPE) ((PIPE) ? 2 : ((TM) * (TN) <= 3 ? 4 : 3))
// XF: -1 = general (any kernel size, transform and keep-mask decided at run time); 0 / 1 = 1x1 conv without a keep-mask, 0 also without
// transform or activation (a gradient matrix): the tap arithmetic, the four mask loads and - for 0 - the eight coefficient loads per
// chunk and thread are not compiled in (every load of the general form is unconditional so that the compiler can count them; the
// price was ~20 load instructions per chunk where 7 carry data).
template <int TM, int TN, int WM, int WN, bool PIPE = false, int XF = -1>
__global__ __launch_bounds__(256) void igemm_conv_kernel(ConvKArgs a) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int TILE = (BM + BN) * IG_LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // (PIPE ? 2 : 1) * TILE floats (launch_conv)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lid = uda_xcd_remap(blockIdx.x, a.nMt * a.nNt);
    const int mt = lid / a.nNt, nt = lid % a.nNt;
    const int H = a.src.H, W = a.src.W, C = a.src.C;
    const int64_t P = (int64_t)a.src.N * H * W;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;

    // ---- loader mapping: row = (tid>>3) + 32*i, 4 consecutive k at kv
    const int lrow = tid >> 3, kv = (tid & 7) * 4;
    int ph[A_IT], pw[A_IT];
    bool pok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int64_t p = m0 + lrow + 32 * i;
        pok[i] = p < P;
        const int64_t q = pok[i] ? p : 0;
        if (XF < 0) {               // (1x1: no tap leaves the pixel - its coordinates, two 64-bit divisions per row, are not needed)
            pw[i] = (int)(q % W);
            ph[i] = (int)((q / W) % H);
        } else {
            pw[i] = ph[i] = 0;
        }
    }

    struct Stg {                // one chunk of operand loads in registers + what its staging step needs
        float4 areg[A_IT], breg[B_IT];
        uint32_t amask[A_IT];
        unsigned aok;           // bit i: areg[i] holds a real load (else zero)
        Xf4 xf;
        int c_ci;
        bool c_kval;
    };
    Stg R0, R1;

    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src.x), 0, (int)min((int64_t)0x7fffffff, (P * a.src.ldx) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.src.mask ? a.src.mask : reinterpret_cast<const uint8_t*>(a.src.x)), 0,
        a.src.mask ? (int)min((int64_t)0x7fffffff, P * a.src.ldm) : 0, 0x00020000);
    // weights and per-channel coefficients through descriptors too: every load of a chunk is then an unconditional buffer load (a
    // missing operand = an out-of-range offset = zeros), the loop has no branch around a load and the compiler can count vmcnt
    // across the two register sets (with branches it waited vmcnt(0) before every staging step, i.e. for the NEWEST loads too:
    // 1.2 us per chunk whatever the prefetch distance - measured, tests/tools/sweep_narrow.py)
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, (int)min((int64_t)0x7fffffff, (int64_t)a.Cout * a.Ktot * 4), 0x00020000);
    const bool has_coef = a.src.scale != nullptr;
    const __amdgpu_buffer_rsrc_t scres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_coef ? a.src.scale : a.src.x), 0, has_coef ? C * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t shres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_coef ? a.src.shift : a.src.x), 0, has_coef ? C * 4 : 0, 0x00020000);
    constexpr int OOB = 0x7ffffff0;

    auto issue = [&](int chunk, Stg& r) {
        float4 (&areg)[A_IT] = r.areg;
        float4 (&breg)[B_IT] = r.breg;
        uint32_t (&amask)[A_IT] = r.amask;
        unsigned& aok = r.aok;
        Xf4& xf = r.xf;
        int& c_ci = r.c_ci;
        bool& c_kval = r.c_kval;
        const int k0 = chunk * IG_BK + kv;          // position in the weight row
        int t = 0, ci = k0;
        if (XF < 0 && a.ksize >= 2) {
            if (a.Kc >= IG_BK) {                     // tap-chunked K: chunk = cc * T + t
                const int T = a.ksize * a.ksize;
                t = chunk % T;
                ci = (chunk / T) * IG_BK + kv;
            } else {                                 // fewer than 32 channels: k = t * Kc + ci
                t = k0 / a.Kc;
                ci = k0 - t * a.Kc;
            }
        }
        c_kval = k0 < a.Ktot && ci < a.Kc;
        c_ci = ci;
        int dh = 0, dw = 0;
        if (XF < 0 && a.ksize >= 2) {
            const int th = a.ksize == 3 ? t / 3 : t >> 1;
            dh = (th - a.cen) * a.dil;
            dw = (t - th * a.ksize - a.cen) * a.dil;
        }
        // coefficients of the 4 channels ci .. ci + 3 (identity where there is no transform / beyond C: those lanes are zeroed in
        // the staging step anyway)
        // (the descriptors' own range check returns zeros beyond C and without a transform; the staging step applies the
        // coefficients only when there is a transform and zeroes the lanes beyond C itself - no condition, no select here)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (XF != 0) {
                xf.sc[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scres, (ci + j) * 4, 0, 0));
                xf.sh[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shres, (ci + j) * 4, 0, 0));
            } else {
                xf.sc[j] = 1.f;
                xf.sh[j] = 0.f;
            }
        }
        aok = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int hh = ph[i] + dh, ww = pw[i] + dw;
            const bool ok = XF >= 0 ? (c_kval && pok[i]) : (c_kval && pok[i] && hh >= 0 && hh < H && ww >= 0 && ww < W);
            // buffer loads: an out-of-range offset returns zeros (no branch around the load)
            const int q = (int)(m0 + lrow + 32 * i) + dh * W + dw;
            areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? (q * (int)a.src.ldx + ci) * 4 : OOB, 0, 0));
            if (XF < 0) amask[i] = __builtin_amdgcn_raw_buffer_load_b32(mres, ok ? q * (int)a.src.ldm + ci : OOB, 0, 0);      // (no mask: a zero-size descriptor)
            else amask[i] = 0x01010101u;
            aok |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + lrow + 32 * i;
            // (rows beyond Cout fall outside the descriptor; only the padded tail of the last chunk needs the select)
            breg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wres, k0 < a.Ktot ? (n * a.Ktot + k0) * 4 : OOB, 0, 0));
        }
    };

    auto stage = [&](Stg& r, float* As, float* Bs) {
        float4 (&areg)[A_IT] = r.areg;
        float4 (&breg)[B_IT] = r.breg;
        uint32_t (&amask)[A_IT] = r.amask;
        const unsigned aok = r.aok;
        const Xf4& xf = r.xf;
        const int c_ci = r.c_ci;
        const bool has_xf = XF != 0 && a.src.scale != nullptr;
        const int act = XF == 0 ? ACT_NONE : a.src.act;
        const float ms = a.src.mask_scale;
        // packed pairs (v_pk_fma / v_pk_mul) and a one-instruction clamp for none / ReLU / ReLU6
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc01 = {xf.sc[0], xf.sc[1]}, sc23 = {xf.sc[2], xf.sc[3]};       // identity when there is no transform
        const f32x2 sh01 = {xf.sh[0], xf.sh[1]}, sh23 = {xf.sh[2], xf.sh[3]};
        const float alo = act == ACT_NONE ? -INFINITY : 0.f, ahi = act == ACT_RELU6 ? 6.f : INFINITY;
        const bool masked = XF < 0 && a.src.mask != nullptr;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = (aok >> i) & 1u;
            f32x2 t01 = {areg[i].x, areg[i].y}, t23 = {areg[i].z, areg[i].w};
            if (has_xf) {
                t01 = t01 * sc01 + sh01;
                t23 = t23 * sc23 + sh23;
            }
            t01 = f32x2{__builtin_amdgcn_fmed3f(t01.x, alo, ahi), __builtin_amdgcn_fmed3f(t01.y, alo, ahi)};
            t23 = f32x2{__builtin_amdgcn_fmed3f(t23.x, alo, ahi), __builtin_amdgcn_fmed3f(t23.y, alo, ahi)};
            if (masked) {
                const uint32_t mk = amask[i];
                t01 *= f32x2{(float)(mk & 0xffu), (float)((mk >> 8) & 0xffu)} * ms;
                t23 *= f32x2{(float)((mk >> 16) & 0xffu), (float)(mk >> 24)} * ms;
            }
            float v[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(ok && (c_ci + j) < C)) v[j] = 0.f;
            uda_st4(&As[(lrow + 32 * i) * IG_LD + kv], make_float4(v[0], v[1], v[2], v[3]));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) uda_st4(&Bs[(lrow + 32 * i) * IG_LD + kv], breg[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (a.Ktot + IG_BK - 1) / IG_BK;
    const int arow = wm * TM * 32 + (lane & 31), brow = wn * TN * 32 + (lane & 31);
    const int koff = 4 * (lane >> 5);

    auto math = [&](const float* As, const float* Bs) {
#pragma unroll
        for (int g = 0; g < IG_BK / 8; ++g) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = uda_ld4(&As[(arow + 32 * i) * IG_LD + g * 8 + koff]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = uda_ld4(&Bs[(brow + 32 * j) * IG_LD + g * 8 + koff]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };
    float* As = smem;
    float* Bs = smem + BM * IG_LD;
    if constexpr (PIPE) {
        float* As1 = smem + TILE;
        float* Bs1 = As1 + BM * IG_LD;
        // chunk c lives in register set c & 1 and in image c & 1; chunks c + 1 (staged) and c + 2 (in flight) are ahead of the math
        // Every issue / stage below is UNCONDITIONAL: a chunk beyond the last one loads through out-of-range offsets (zeros, no
        // memory access) and stages zeros into an image nobody reads.  With `if (c + 3 < nchunks) issue(...)` the number of loads in
        // flight differs between the paths that meet at the next staging step, and the compiler's vmcnt has to cover the smaller
        // count - i.e. it waits for the NEWEST register set too and the second set buys nothing (seen in the ISA: vmcnt(5) ... (0)).
        issue(0, R0);
        issue(1, R1);
        stage(R0, As, Bs);
        issue(2, R0);
        for (int c = 0; c < nchunks; c += 2) {
            __syncthreads();        // image 0 holds chunk c; image 1 is no longer read
            stage(R1, As1, Bs1);
            issue(c + 3, R1);
            math(As, Bs);
            __syncthreads();        // image 1 holds chunk c + 1; image 0 is no longer read
            stage(R0, As, Bs);
            issue(c + 4, R0);
            if (c + 1 < nchunks) math(As1, Bs1);
        }
    } else {
        issue(0, R0);
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();            // every wave finished reading the previous chunk
            stage(R0, As, Bs);
            __syncthreads();
            if (c + 1 < nchunks) issue(c + 1, R0);   // in flight under the MFMAs below
            math(As, Bs);
        }
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Each 32x32 block goes through a per-wave LDS patch and leaves as 16-byte stores (8 rows x 128 B per
    // instruction instead of 2 rows x 128 B with 4-byte stores: the short-K convs are bound by their output).
    const int colb = n0 + wn * TN * 32 + (lane & 31);
    const bool vec_ok = uda_aligned16_dev(a.y) && (a.ldy & 3) == 0 && (!a.addend || (uda_aligned16_dev(a.addend) && (a.ld_add & 3) == 0));
    __syncthreads();                       // every wave is done with the operand tiles: smem becomes staging space
    float* patch = smem + wave * (32 * IG_LD);
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        s1[j] = 0.f;
        s2[j] = 0.f;
        const int col = colb + 32 * j;
        const bool cok = col < a.Cout;
        const float bv = (cok && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int64_t rbase = m0 + wm * TM * 32 + 32 * i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[i][j][r] + bv;
                if (cok && rbase + rl < P) {
                    s1[j] += v;
                    s2[j] += v * v;
                }
                if (vec_ok) {
                    patch[rl * IG_LD + (lane & 31)] = v;
                } else if (cok && rbase + rl < P) {
                    a.y[(rbase + rl) * a.ldy + col] = a.addend ? v + a.addend[(rbase + rl) * a.ld_add + col] : v;
                }
            }
            if (vec_ok) {
                const int cb = n0 + wn * TN * 32 + 32 * j + (lane & 7) * 4;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int rl = rq * 8 + (lane >> 3);
                    const int64_t row = rbase + rl;
                    float4 o = uda_ld4(&patch[rl * IG_LD + (lane & 7) * 4]);
                    if (row < P && cb < a.Cout) {
                        if (cb + 3 < a.Cout) {
                            if (a.addend) {
                                const float4 ad = uda_ld4(a.addend + row * a.ld_add + cb);
                                o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                            }
                            uda_st4(a.y + row * a.ldy + cb, o);
                        } else {
                            const float ov[4] = {o.x, o.y, o.z, o.w};
                            for (int q = 0; q < 4; ++q)
                                if (cb + q < a.Cout) a.y[row * a.ldy + cb + q] = a.addend ? ov[q] + a.addend[row * a.ld_add + cb + q] : ov[q];
                        }
                    }
                }
            }
        }
    }
    if (a.stats) {   // per-tile BN statistics: combine lane halves, the WM waves through LDS, then fp64 atomics
        __syncthreads();
        float* red = smem;   // [WM][2][BN]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32);
            const float t2 = s2[j] + __shfl_xor(s2[j], 32);
            if (lane < 32) {
                const int cl = wn * TN * 32 + 32 * j + lane;
                red[(wm * 2 + 0) * BN + cl] = t1;
                red[(wm * 2 + 1) * BN + cl] = t2;
            }
        }
        __syncthreads();
        double* dst = a.stats + (int64_t)(mt % UDA_STAT_SLOTS) * 2 * a.Cout;
        for (int e = tid; e < 2 * BN; e += 256) {
            const int qd = e / BN, cl = e % BN;
            if (n0 + cl < a.Cout) {
                float t = 0.f;
#pragma unroll
                for (int m = 0; m < WM; ++m) t += red[(m * 2 + qd) * BN + cl];
                atomicAdd(&dst[qd * a.Cout + n0 + cl], (double)t);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// out[col] += sum_rows part[row][col]   (double accumulation; rows split over blockIdx.y)
__global__ void reduce_partials_kernel(const float* __restrict__ part, int nrows, int ncols,
                                       int rows_per_seg, double* __restrict__ out) {
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(nrows, r0 + rows_per_seg);
    double s = 0.0;
    if (col < ncols)
        for (int r = r0 + rg; r < r1; r += 4) s += (double)part[(int64_t)r * ncols + col];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < ncols) atomicAdd(&out[col], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

int uda_reduce_partials(const float* part, int nrows, int ncols, double* out, hipStream_t st) {
    int segs = nrows / 64;
    if (segs < 1) segs = 1;
    if (segs > 64) segs = 64;
    const int rps = (nrows + segs - 1) / segs;
    dim3 grid(uda_cdiv(ncols, 64), uda_cdiv(nrows, rps));
    hipLaunchKernelGGL(reduce_partials_kernel, grid, dim3(256), 0, st, part, nrows, ncols, rps, out);
    UDA_LAUNCH_CHECK("reduce_partials");
    return 0;
}

// floats per weight row of a conv with C input channels and ksize x ksize taps
static inline int uda_k_row(int C, int ksize) {
    const int Kc = ((C + 3) / 4) * 4;
    if (ksize == 1 || Kc < IG_BK) return ksize * ksize * Kc;      // fewer than 32 channels: tap-major, unpadded
    return ((Kc + IG_BK - 1) / IG_BK) * ksize * ksize * IG_BK;
}

static int check_src(const uda_src_t& s, const char* who) {
    UDA_REQUIRE(s.x && uda_aligned16(s.x), "%s: src.x must be 16-byte aligned", who);
    UDA_REQUIRE(s.ldx % 4 == 0 && s.ldx >= ((s.C + 3) / 4) * 4, "%s: src.ldx=%lld must be a multiple of 4 and >= round4(C=%d)",
                who, (long long)s.ldx, s.C);
    UDA_REQUIRE(s.N > 0 && s.H > 0 && s.W > 0 && s.C > 0, "%s: bad src dims", who);
    UDA_REQUIRE((s.scale == nullptr) == (s.shift == nullptr), "%s: scale/shift must come together", who);
    if (s.mask) UDA_REQUIRE(s.ldm % 4 == 0 && (reinterpret_cast<uintptr_t>(s.mask) & 3u) == 0 && s.ldm >= ((s.C + 3) / 4) * 4,
                            "%s: mask must be 4-byte aligned with ldm %% 4 == 0", who);
    return 0;
}

// Cout = 1 on a raw operand (the discriminators' last layer, GAN.py:100: 2048 -> 1 over 4 taps): a dot product
// per pixel, one wave per pixel, 16-byte loads along the contiguous channels; HBM-bound (reads the operand once),
// where a 128 x 32 MFMA tile would run 46 workgroups with one useful column.
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvKArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = a.src.H, W = a.src.W;
    const int64_t P = (int64_t)a.src.N * H * W;
    if (p >= P) return;
    const int pw = (int)(p % W), ph = (int)((p / W) % H);
    const int T = a.ksize * a.ksize;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        const int th = t / a.ksize;
        const int hh = ph + (th - a.cen) * a.dil, ww = pw + (t - th * a.ksize - a.cen) * a.dil;
        if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
        const float* xr = a.src.x + (p + (int64_t)(hh - ph) * W + (ww - pw)) * a.src.ldx;
        for (int c = lane * 4; c < a.Kc; c += 256) {
            // tap-chunked weight row (T > 1): k = ((c / 32) * T + t) * 32 + c % 32; zero beyond C
            const int kq = (T == 1 || a.Kc < IG_BK) ? t * a.Kc + c : ((c / IG_BK) * T + t) * IG_BK + (c % IG_BK);
            const float4 xv = uda_ld4(xr + c), wv = uda_ld4(a.w + kq);
            float s = xv.x * wv.x;
            if (c + 1 < a.src.C) s += xv.y * wv.y;
            if (c + 2 < a.src.C) s += xv.z * wv.z;
            if (c + 3 < a.src.C) s += xv.w * wv.w;
            acc += s;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        float v = acc + (a.bias ? a.bias[0] : 0.f);
        if (a.addend) v += a.addend[p * a.ld_add];
        a.y[p * a.ldy] = v;
    }
}

// 1x1 convs with one or two outputs on a lazily transformed operand (the segmentation / boundary heads, decoder.py:32,41:
// BN + ReLU + dropout on 305 / 256 channels -> 2 / 1 logits): a per-pixel dot product, 16 lanes per pixel, 16-byte loads
// along the channels, per-channel coefficients and weights in LDS.  HBM-bound (the operand and its mask are read once),
// where the 128 x 32 MFMA tile spends 32 columns on 1-2 useful ones.
template <int NO>
__global__ __launch_bounds__(256) void conv_heads_kernel(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];       // [2 + NO][Kc]: scale, shift, w[0..NO)
    const int Kc = a.Kc, C = a.src.C;
    for (int e = threadIdx.x; e < Kc; e += 256) {
        const bool in = e < C;
        hsm[e] = (in && a.src.scale) ? a.src.scale[e] : 1.f;
        hsm[Kc + e] = (in && a.src.shift) ? a.src.shift[e] : 0.f;
#pragma unroll
        for (int o = 0; o < NO; ++o) hsm[(2 + o) * Kc + e] = in ? a.w[(int64_t)o * a.Ktot + e] : 0.f;
    }
    __syncthreads();
    const int l16 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int64_t P = (int64_t)a.src.N * a.src.H * a.src.W;
    const float alo = a.src.act == ACT_NONE ? -INFINITY : 0.f, ahi = a.src.act == ACT_RELU6 ? 6.f : INFINITY;
    const float ms = a.src.mask_scale;
    const int G = Kc >> 2;
    for (int it = 0; it < 8; ++it) {
        const int64_t p = (int64_t)blockIdx.x * 128 + it * 16 + pl;
        float acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = 0.f;
        if (p < P) {
            const float* xr = a.src.x + p * a.src.ldx;
            for (int g = l16; g < G; g += 16) {
                const int c = g * 4;
                float4 xv = uda_ld4(xr + c);
                const float4 sc = uda_ld4(&hsm[c]), sh = uda_ld4(&hsm[Kc + c]);
                if (c + 4 > C) {        // padding lanes of the last granule hold whatever the buffer held (Inf * 0 would be NaN)
                    if (c + 1 >= C) xv.y = 0.f;
                    if (c + 2 >= C) xv.z = 0.f;
                    xv.w = 0.f;
                }
                float u[4] = {__builtin_amdgcn_fmed3f(xv.x * sc.x + sh.x, alo, ahi), __builtin_amdgcn_fmed3f(xv.y * sc.y + sh.y, alo, ahi),
                              __builtin_amdgcn_fmed3f(xv.z * sc.z + sh.z, alo, ahi), __builtin_amdgcn_fmed3f(xv.w * sc.w + sh.w, alo, ahi)};
                if (a.src.mask) {
                    uint32_t mk = *reinterpret_cast<const uint32_t*>(a.src.mask + p * a.src.ldm + c);
                    if (c + 4 > C) mk &= 0xffffffffu >> (8 * (c + 4 - C));
                    u[0] *= (float)(mk & 0xffu) * ms; u[1] *= (float)((mk >> 8) & 0xffu) * ms;
                    u[2] *= (float)((mk >> 16) & 0xffu) * ms; u[3] *= (float)(mk >> 24) * ms;
                }
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const float4 wv = uda_ld4(&hsm[(2 + o) * Kc + c]);      // zero beyond C
                    acc[o] += u[0] * wv.x + u[1] * wv.y + u[2] * wv.z + u[3] * wv.w;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
#pragma unroll
            for (int d = 8; d > 0; d >>= 1) acc[o] += __shfl_xor(acc[o], d);
        }
        if (l16 == 0 && p < P) {
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                float v = acc[o] + (a.bias ? a.bias[o] : 0.f);
                if (a.addend) v += a.addend[p * a.ld_add + o];
                a.y[p * a.ldy + o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Short-K 1x1 convs over many pixels (the backbone's expand convs 16 -> 96, 24 -> 144, 32 -> 192 and the input gradients of their
// project twins, mobilenet.py:43-57): output-bound - 0.5 GB written for 0.07 GB read - and the tiled kernel above wrote at 2.4 TB/s
// whatever its options, because one workgroup spends ~4000 instructions of generic staging / epilogue code on 48 MFMAs.  Here:
//   * one WAVE owns 32 pixels x all Cout columns per trip of a persistent loop; no LDS, no barriers;
//   * its operands go straight into the MFMA register layout: lane (row r = l & 31, half h = l >> 5) loads channels [h K/2, (h+1) K/2)
//     of its pixel with K/8 16-byte loads - the contraction order is permuted (k-slot 0 of step kk is channel kk, k-slot 1 channel
//     K/2 + kk; the weights, held in registers for the whole launch, are read the same way), which changes nothing but the rounding order;
//   * the accumulators leave as 4-byte stores of two 128-byte row segments per instruction through a buffer descriptor (a row or
//     column outside the matrix gets an out-of-range offset), the addend arrives the same way;
//   * the BatchNorm statistics stay in registers over the loop and are added once per wave (fp64 atomics).
// ~350 instructions per 32 pixels instead of ~1000.
// A wave keeps ONE group of NB column blocks for the whole launch (its weights live in registers): wave w serves column group w % ngroups
// and the pixel tiles w / ngroups, + nwaves / ngroups, ... (Cout = 192 = two groups of 96: 48 weight + 48 accumulator registers per wave
// instead of 96 + 96, two waves per SIMD).
template <int KH, int NB>      // KH = K / 2 channels per lane (8, 12, 16), NB = 32-column blocks per group
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(ConvKArgs a, int64_t P, int ntiles, int ngroups) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const int K = 2 * KH;
    const int grp = wave % ngroups, col0 = grp * (NB * 32);
    // weights: b[nb][kk] = w[col][h KH + kk], col = col0 + 32 nb + r (zero beyond Cout)
    float b[NB][KH];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col0 + nb * 32 + r;
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < a.Cout) v = uda_ld4(a.w + (int64_t)col * K + h * KH + 4 * q);
            b[nb][4 * q] = v.x; b[nb][4 * q + 1] = v.y; b[nb][4 * q + 2] = v.z; b[nb][4 * q + 3] = v.w;
        }
    }
    const bool has_xf = a.src.scale != nullptr;      // (scale and shift are re-read per tile - 2 KH / 4 cached loads - instead of held in 2 KH registers)
    const bool xf16 = has_xf && ((reinterpret_cast<uintptr_t>(a.src.scale) | reinterpret_cast<uintptr_t>(a.src.shift)) & 15) == 0;
    const float alo = a.src.act == ACT_NONE ? -INFINITY : 0.f, ahi = a.src.act == ACT_RELU6 ? 6.f : INFINITY;
    // statistics: fp32 over one tile's 16 rows per lane (as long a chain as the tiled kernel's), fp64 across the tiles of the loop - an
    // fp32 chain over all of a wave's tiles (hundreds of values) showed up as 5x the noise in the near-cancelling BN-affine gradients
    double s1[NB], s2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) s1[nb] = s2[nb] = 0.0;
    constexpr int OOB = 0x7ffffff0;
    const int ldy4 = (int)a.ldy * 4, lda4 = (int)a.ld_add * 4;
    for (int t = wave / ngroups; t < ntiles; t += nwaves / ngroups) {
        const int64_t p0 = (int64_t)t * 32;
        const int64_t p = min(p0 + r, P - 1);                          // (rows beyond the matrix: computed, never stored or counted)
        float av[KH];
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float4 v = uda_ld4(a.src.x + p * a.src.ldx + h * KH + 4 * q);
            av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
        }
        if (has_xf) {
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) {
                float4 sc, sh;
                if (xf16) {
                    sc = uda_ld4(a.src.scale + h * KH + 4 * q);
                    sh = uda_ld4(a.src.shift + h * KH + 4 * q);
                } else {
                    const float* ps = a.src.scale + h * KH + 4 * q;
                    const float* pt = a.src.shift + h * KH + 4 * q;
                    sc = make_float4(ps[0], ps[1], ps[2], ps[3]);
                    sh = make_float4(pt[0], pt[1], pt[2], pt[3]);
                }
                av[4 * q] = av[4 * q] * sc.x + sh.x; av[4 * q + 1] = av[4 * q + 1] * sc.y + sh.y;
                av[4 * q + 2] = av[4 * q + 2] * sc.z + sh.z; av[4 * q + 3] = av[4 * q + 3] * sc.w + sh.w;
            }
        }
#pragma unroll
        for (int j = 0; j < KH; ++j) av[j] = __builtin_amdgcn_fmed3f(av[j], alo, ahi);
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b[nb][kk], acc[nb], 0, 0, 0);
        // C/D layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
        const int rows_left = (int)min((int64_t)32, P - p0);
        const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(a.y + p0 * a.ldy, 0, ((rows_left - 1) * (int)a.ldy + a.Cout) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t adres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.addend ? a.addend + p0 * a.ld_add : a.y), 0, a.addend ? ((rows_left - 1) * (int)a.ld_add + a.Cout) * 4 : 0, 0x00020000);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = col0 + nb * 32 + r;
            const bool cok = col < a.Cout;
            float ad[16];                                              // the block's addend values, loaded together
#pragma unroll
            for (int e = 0; e < 16; ++e) ad[e] = 0.f;
            if (a.addend) {                                            // (uniform)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = (e & 3) + 8 * (e >> 2);
                    const bool ok = cok && (rl + 4 * h) < rows_left;
                    ad[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(adres, ok ? (4 * h) * lda4 + col * 4 : OOB, rl * lda4, 0));
                }
            }
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2);                 // + 4 h (per lane)
                const float v = acc[nb][e];
                const bool ok = cok && (rl + 4 * h) < rows_left;
                t1 += ok ? v : 0.f;
                t2 += ok ? v * v : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + ad[e]), yres, ok ? (4 * h) * ldy4 + col * 4 : OOB, rl * ldy4, 0);
            }
            s1[nb] += (double)t1;
            s2[nb] += (double)t2;
        }
    }
    if (a.stats) {
        double* dst = a.stats + (int64_t)(wave % UDA_STAT_SLOTS) * 2 * a.Cout;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const double t1 = s1[nb] + __shfl_xor(s1[nb], 32), t2 = s2[nb] + __shfl_xor(s2[nb], 32);
            const int col = col0 + nb * 32 + r;
            if (h == 0 && col < a.Cout) {
                atomicAdd(&dst[col], t1);
                atomicAdd(&dst[a.Cout + col], t2);
            }
        }
    }
}

template <int KH, int NB>
static int launch_stream(ConvKArgs& k, int64_t P, hipStream_t st) {
    const int ntiles = (int)uda_cdiv(P, 32);
    static int resident_dev[UDA_MAX_DEVICES] = {};          // workgroups of this kernel the device holds at a time
    int& resident = resident_dev[uda_device_slot()];
    auto fn = conv1x1_stream_kernel<KH, NB>;
    if (!resident) {
        int dev = 0, ncu = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), 256, 0) != hipSuccess || ncu < 1 || per_cu < 1)
            return uda_set_error("conv1x1_stream: cannot query the device's residency");
        resident = ncu * per_cu;
    }
    const int ngroups = uda_cdiv(k.Cout, NB * 32);                       // 1 or 2 (conv_stream_shape)
    int grid = (int)std::min<int64_t>(resident, uda_cdiv((int64_t)ntiles * ngroups, 4));
    if (ngroups == 2 && grid > 1) grid &= ~1;                            // (waves per launch divisible by the groups: 4 per workgroup)
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, st, k, P, ntiles, ngroups);
    UDA_LAUNCH_CHECK("conv1x1_stream");
    return 0;
}

// the short-K streaming form applies: 1x1, stride 1, K = Cin in {16, 24, 32}, no keep-mask, no bias, >= 32768 pixels, aligned rows
static int conv_stream_shape(const uda_conv_args_t* a, int64_t P) {
    static const int on = getenv("UDA_CONV_STREAM") ? atoi(getenv("UDA_CONV_STREAM")) : 1;      // A/B switch
    if (!on || a->ksize != 1 || (a->stride > 1) || a->src.mask || a->bias || P < 32768) return 0;
    if (!uda_aligned16(a->src.x) || a->src.ldx % 4 || (a->src.C != 16 && a->src.C != 24 && a->src.C != 32)) return 0;
    if (P * a->ldy >= ((int64_t)1 << 29) || (a->addend && P * a->ld_add >= ((int64_t)1 << 29))) return 0;
    const int C = a->src.C, Cout = a->Cout;
    if (C == 16 && Cout <= 32) return 5;
    if (C == 16 && Cout <= 96) return 1;
    if (C == 24 && Cout <= 64) return 2;
    if (C == 24 && Cout <= 96) return 6;
    if (C == 24 && Cout <= 160) return 3;
    if (C == 32 && Cout > 96 && Cout <= 192 && P >= 262144) return 4;      // (at 65536 pixels, and towards few columns, the tiled kernel is as fast or faster)
    return 0;
}

template <int TM, int TN, int WM, int WN, bool PIPE, int XF>
static int launch_conv_one(ConvKArgs& k, hipStream_t st, size_t lds) {
    auto fn = igemm_conv_kernel<TM, TN, WM, WN, PIPE, XF>;
    if (lds > 64 * 1024) {                                  // beyond the default dynamic-LDS limit: raise it once per device
        static bool configured_dev[UDA_MAX_DEVICES] = {};
        bool& configured = configured_dev[uda_device_slot()];
        if (!configured) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return uda_set_error("igemm_conv: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            configured = true;
        }
    }
    hipLaunchKernelGGL(fn, dim3(k.nMt * k.nNt), dim3(256), lds, st, k);
    UDA_LAUNCH_CHECK("igemm_conv");
    return 0;
}

template <int TM, int TN, int WM, int WN>
static int launch_conv_xf(ConvKArgs& k, int xf, hipStream_t st, bool pipe, size_t tile_bytes) {
    if (pipe) return xf == 0 ? launch_conv_one<TM, TN, WM, WN, true, 0>(k, st, 2 * tile_bytes) : launch_conv_one<TM, TN, WM, WN, true, 1>(k, st, 2 * tile_bytes);
    return xf == 0 ? launch_conv_one<TM, TN, WM, WN, false, 0>(k, st, tile_bytes) : launch_conv_one<TM, TN, WM, WN, false, 1>(k, st, tile_bytes);
}

template <int TM, int TN, int WM, int WN>
static int launch_conv(ConvKArgs& k, int64_t P, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static_assert(BM == 128 || BM == 64, "tiles of 128 or 64 pixels");
    constexpr size_t tile_bytes = (size_t)(BM + BN) * IG_LD * sizeof(float);
    static_assert(2 * tile_bytes <= 160 * 1024, "two tile images must fit the LDS");
    k.nMt = uda_cdiv(P, BM);
    k.nNt = uda_cdiv(k.Cout, BN);
    // long K: the pipelined form (two tile images, loads two chunks ahead); short K keeps the lean one (more workgroups per CU)
    static const int pipe_min = getenv("UDA_CONV_PIPE_MIN_K") ? atoi(getenv("UDA_CONV_PIPE_MIN_K")) : 192;
    // 1x1 without a keep-mask: the lean loader (XF 0: also no transform and no activation - gradient matrices; XF 1: the rest)
    static const int lean = getenv("UDA_CONV_LEAN") ? atoi(getenv("UDA_CONV_LEAN")) : 1;       // A/B switch
    const int xf = (lean && k.ksize == 1 && !k.src.mask) ? ((!k.src.scale && k.src.act == ACT_NONE) ? 0 : 1) : -1;
    if (xf >= 0) return launch_conv_xf<TM, TN, WM, WN>(k, xf, st, k.Ktot >= pipe_min, tile_bytes);
    if (k.Ktot >= pipe_min) {
        auto fn = igemm_conv_kernel<TM, TN, WM, WN, true>;
        if (2 * tile_bytes > 64 * 1024) {                   // beyond the default dynamic-LDS limit: raise it once per device
            static bool configured_dev[UDA_MAX_DEVICES] = {};
            bool& configured = configured_dev[uda_device_slot()];
            if (!configured) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * tile_bytes));
                if (e != hipSuccess) return uda_set_error("igemm_conv: cannot reserve %zu B of LDS: %s", 2 * tile_bytes, hipGetErrorString(e));
                configured = true;
            }
        }
        hipLaunchKernelGGL(fn, dim3(k.nMt * k.nNt), dim3(256), 2 * tile_bytes, st, k);
    }
    else hipLaunchKernelGGL((igemm_conv_kernel<TM, TN, WM, WN, false>), dim3(k.nMt * k.nNt), dim3(256), tile_bytes, st, k);
    UDA_LAUNCH_CHECK("igemm_conv");
    return 0;
}

// true when uda_conv_fwd routes these arguments to the wide-tile (MFMA-bound) kernels
static bool conv_is_wide(const uda_conv_args_t* a, int Kc, int Ktot) {
    if (a->Cout == 1 && !a->src.scale && !a->src.mask && a->src.act == ACT_NONE && !a->stats && Ktot >= 1024) return false;
    if (a->Cout <= 2 && a->ksize == 1 && !a->stats && Kc >= 64 && Kc <= 2048) return false;
    if (a->Cout <= 96)      // narrow outputs: only the bf16x3 mode has a 64-column wide tile (long-K multi-tap convs)
        return a->mfma == UDA_MFMA_BF16X3 && a->ksize >= 2 && a->Cout >= 40 && a->Cout <= 64 && Kc >= 128 && Ktot >= 1024;
    if (Ktot <= 192 || (a->ksize >= 2 && Kc < IG_BK)) return false;
    return true;
}

/* 1 when uda_conv_fwd will run these arguments on the bf16x3 wide-tile kernel, i.e. needs a->x3_src / a->x3_w (uda_x3_pack) */
extern "C" int uda_conv_uses_x3(const uda_conv_args_t* a) {
    if (!a || a->mfma != UDA_MFMA_BF16X3 || a->ksize < 1 || a->ksize > 3) return 0;
    const int Kc = ((a->src.C + 3) / 4) * 4, Ktot = uda_k_row(a->src.C, a->ksize);
    ConvKArgs k;
    k.Kc = Kc; k.ksize = a->ksize; k.Cout = a->Cout;
    return conv_is_wide(a, Kc, Ktot) && conv_x3_eligible(k) ? 1 : 0;
}

extern "C" uint64_t uda_conv_fwd_workspace_bytes(const uda_conv_args_t* a) {
    if (!uda_conv_uses_x3(a)) return 0;
    ConvKArgs k;
    k.src = a->src; k.Cout = a->Cout; k.ksize = a->ksize; k.stats = a->stats;
    k.Kc = ((a->src.C + 3) / 4) * 4;
    k.Ktot = uda_k_row(a->src.C, a->ksize);
    const int sd = a->stride <= 1 ? 1 : a->stride;
    return conv_x3_workspace_bytes(k, (int64_t)a->src.N * ((a->src.H - 1) / sd + 1) * ((a->src.W - 1) / sd + 1));
}

extern "C" int uda_conv_fwd(const uda_conv_args_t* a, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(a != nullptr, "uda_conv_fwd: null args");
    if (int e = check_src(a->src, "uda_conv_fwd")) return e;
    UDA_REQUIRE(a->ksize >= 1 && a->ksize <= 3, "uda_conv_fwd: ksize must be 1, 2 or 3");
    UDA_REQUIRE(a->ksize != 2 || a->origin == 0 || a->origin == 1, "uda_conv_fwd: origin must be 0 or 1 for ksize 2");
    UDA_REQUIRE(a->Cout > 0 && a->dil >= 1 && a->y && a->w, "uda_conv_fwd: bad args");
    UDA_REQUIRE(uda_aligned16(a->w), "uda_conv_fwd: weights must be 16-byte aligned");
    UDA_REQUIRE(a->ldy >= a->Cout, "uda_conv_fwd: ldy < Cout");
    // stride 2 (resnet.py:66 conv2 of the first bottleneck of layer2 / layer3, :93 their 1x1 shortcut): output pixel (n, oh, ow) is
    // centred on input pixel (n, 2 oh, 2 ow); P counts OUTPUT rows from here on
    const int sd = a->stride <= 1 ? 1 : a->stride;
    UDA_REQUIRE(sd <= 2, "uda_conv_fwd: stride must be 1 or 2");
    const int Ho = (a->src.H - 1) / sd + 1, Wo = (a->src.W - 1) / sd + 1;
    const int64_t Pin = (int64_t)a->src.N * a->src.H * a->src.W, P = (int64_t)a->src.N * Ho * Wo;
    ConvKArgs k;
    k.src = a->src;
    k.stride = sd; k.Ho = Ho; k.Wo = Wo;
    k.w = a->w;
    k.Cout = a->Cout;
    k.ksize = a->ksize;
    k.dil = a->dil;
    k.cen = a->ksize == 3 ? 1 : (a->ksize == 2 ? a->origin : 0);
    k.Kc = ((a->src.C + 3) / 4) * 4;
    k.Ktot = uda_k_row(a->src.C, a->ksize);
    k.bias = a->bias;
    k.addend = a->addend;
    k.ld_add = a->ld_add;
    k.y = a->y;
    k.ldy = a->ldy;
    k.stats = a->stats;
    k.debug = 0;
    k.x3 = a->mfma == UDA_MFMA_BF16X3;
    UDA_REQUIRE((Pin + 128) * a->src.ldx < ((int64_t)1 << 29) && (Pin + 128) * (a->src.mask ? a->src.ldm : 1) < ((int64_t)1 << 31),
                "uda_conv_fwd: operand too large for 32-bit byte offsets (P * ld must stay below 2^29 elements)");
    int e;
    switch (conv_stream_shape(a, P)) {
        case 1: return launch_stream<8, 3>(k, P, st);
        case 2: return launch_stream<12, 2>(k, P, st);
        case 3: return launch_stream<12, 5>(k, P, st);      // 144 columns in one group of five blocks
        case 4: return launch_stream<16, 3>(k, P, st);      // two column groups of 96
        case 5: return launch_stream<8, 1>(k, P, st);
        case 6: return launch_stream<12, 3>(k, P, st);
        default: break;
    }
    if (sd != 1) {       // only the wide-tile kernels walk a strided output grid
        const bool wide = conv_is_wide(a, k.Kc, k.Ktot) && a->Cout > 96 && !(k.Ktot <= 192 || (a->ksize >= 2 && k.Kc < IG_BK));
        UDA_REQUIRE(wide, "uda_conv_fwd: stride 2 is built on the wide-tile kernels only (Cout > 96, K > 192)");
        return (k.x3 && conv_x3_eligible(k)) ? launch_conv_x3(k, P, a->x3_src, a->x3_w, st, a->workspace, a->workspace_bytes) : launch_conv_ws(k, P, st);
    }
    if (a->Cout == 1 && !a->src.scale && !a->src.mask && a->src.act == ACT_NONE && !a->stats && k.Ktot >= 1024) {
        hipLaunchKernelGGL(conv_cout1_kernel, dim3(uda_cdiv(P, 4)), dim3(256), 0, st, k);
        UDA_LAUNCH_CHECK("conv_cout1");
        return 0;
    }
    if (a->Cout <= 2 && a->ksize == 1 && !a->stats && k.Kc >= 64 && k.Kc <= 2048) {
        const size_t lds = (size_t)(2 + a->Cout) * k.Kc * sizeof(float);
        if (a->Cout == 1) hipLaunchKernelGGL(conv_heads_kernel<1>, dim3(uda_cdiv(P, 128)), dim3(256), lds, st, k);
        else hipLaunchKernelGGL(conv_heads_kernel<2>, dim3(uda_cdiv(P, 128)), dim3(256), lds, st, k);
        UDA_LAUNCH_CHECK("conv_heads");
        return 0;
    }
    // few pixels (the 32x32-map layers at B = 16: 128 tiles of 128 pixels for 256 CUs): 64-pixel tiles, twice the workgroups
    static const int low_env = getenv("UDA_CONV_LOW") ? atoi(getenv("UDA_CONV_LOW")) : 1;
    const bool low = low_env && P > 64 && uda_cdiv(P, 128) * uda_cdiv(a->Cout, a->Cout <= 64 ? 64 : 128) <= 192;
    // few pixels, wide output, long K (the project convs of the 32x32-map layers: 960 -> 160, 576 -> 160, 960 -> 320 at P = 16384): the
    // 128 x 128 wide tiles pad 160 columns to 256 and give one workgroup per CU; 64-pixel tiles of 192 or 320 columns give the same 256
    // workgroups with 17 % / no padding
    static const int few_env = getenv("UDA_CONV_FEW") ? atoi(getenv("UDA_CONV_FEW")) : 1;
    int few_w = 0;
    if (few_env && P > 64 && uda_cdiv(P, 128) <= 192 && a->Cout <= 320 && k.Ktot <= 1024) {      // (MobileNetV2's shapes; ResNet-101's wider / longer
        const int c192 = uda_cdiv(a->Cout, 192) * 192, c320 = uda_cdiv(a->Cout, 320) * 320;       // 1x1 convs stay on the wide-tile kernels: measured)
        const int64_t t192 = uda_cdiv(P, 64) * (c192 / 192), t320 = uda_cdiv(P, 64) * (c320 / 320);
        if (c192 <= c320 && t192 >= 192 && t192 <= 512) few_w = 192;
        else if (t320 >= 192 && t320 <= 512) few_w = 320;
    }
    const bool few = few_w != 0;
    if (uda_conv_uses_x3(a)) e = launch_conv_x3(k, P, a->x3_src, a->x3_w, st, a->workspace, a->workspace_bytes);
    else if (few && k.Ktot > 192 && a->ksize == 1 && a->Cout > 128 && few_w == 192) e = launch_conv<1, 3, 2, 2>(k, P, st);
    else if (few && k.Ktot > 192 && a->ksize == 1 && a->Cout > 128 && few_w == 320) e = launch_conv<1, 5, 2, 2>(k, P, st);
    else if (low && a->Cout <= 64) e = launch_conv<1, 1, 2, 2>(k, P, st);
    else if (low && a->Cout <= 128 && (k.Ktot <= 192 || a->Cout <= 96 || (a->ksize >= 2 && k.Kc < IG_BK))) e = launch_conv<1, 2, 2, 2>(k, P, st);
    else if (a->Cout <= 32) e = launch_conv<1, 1, 4, 1>(k, P, st);
    else if (a->Cout <= 64) e = launch_conv<1, 2, 4, 1>(k, P, st);
    else if (a->Cout <= 96) e = launch_conv<1, 3, 4, 1>(k, P, st);
    else if (k.Ktot <= 192 || (a->ksize >= 2 && k.Kc < IG_BK)) {     // (the wide-tile kernel only walks the tap-chunked K order)
        // short K (the backbone's expand convs): output-bound; pick the tile width that wastes the fewest columns
        // (Cout = 144 -> one 160-wide tile instead of two 128-wide ones, 576 -> six 96-wide tiles, ...)
        const int w96 = uda_cdiv(a->Cout, 96) * 96, w128 = uda_cdiv(a->Cout, 128) * 128, w160 = uda_cdiv(a->Cout, 160) * 160;
        if (w160 <= w128 && w160 <= w96) e = launch_conv<1, 5, 4, 1>(k, P, st);
        else if (w128 <= w96) e = launch_conv<2, 2, 2, 2>(k, P, st);
        else e = launch_conv<1, 3, 4, 1>(k, P, st);
    }
    else e = (k.x3 && conv_x3_eligible(k)) ? launch_conv_x3(k, P, a->x3_src, a->x3_w, st, a->workspace, a->workspace_bytes) : launch_conv_ws(k, P, st);
    return e;
}

// ==========================================================================================
// Weight gradient: dw[co][j] = sum_p dy[p, co] * A(p, j),  j = t*Kc + ci; reduction over pixels.
// Operands are staged [pixel][row] (row fastest) so the MFMA operand reads are plain ds_read_b32
// of 32 consecutive dwords.  The pixel range is split over blockIdx.y; every split writes its
// own fp32 slab and a second kernel sums the slabs (bitwise reproducible, no float atomics).

// pixels per K-chunk of the narrow (HBM-bound) weight-gradient kernel: one barrier pair per 64 pixels
#define WGN_BKP 64

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(WgradKArgs a) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int AV = BM / 4, BV = BN / 4;                 // float4 per staged row
    constexpr int ARPP = 256 / AV, BRPP = 256 / BV;          // pixel rows per pass
    constexpr int APASS = WGN_BKP / ARPP, BPASS = WGN_BKP / BRPP;
    __shared__ __attribute__((aligned(16))) float smem[WGN_BKP * (BM + BN)];
    float* As = smem;                    // [32][BM]
    float* Bs = smem + WGN_BKP * BM;      // [32][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int cot = blockIdx.x / a.nJt, jt = blockIdx.x % a.nJt;
    const int split = blockIdx.y;
    const int H = a.src.H, W = a.src.W, C = a.src.C;
    const int64_t P = (int64_t)a.src.N * H * W;

    const int acv = (tid % AV) * 4, apr = tid / AV;
    const int co = cot * BM + acv;
    const int bjv = (tid % BV) * 4, bpr = tid / BV;
    const int j0 = jt * BN + bjv;
    const bool jok = j0 < a.Jtot;
    int t = 0, ci = j0;
    if (a.ksize >= 2 && jok) {
        t = j0 / a.Kc;
        ci = j0 - t * a.Kc;
    }
    int dh = 0, dw = 0;
    if (a.ksize >= 2) {
        const int th = a.ksize == 3 ? t / 3 : t >> 1;
        dh = (th - a.cen) * a.dil;
        dw = (t - th * a.ksize - a.cen) * a.dil;
    }
    Xf4 xf;
    uda_load_xf4(xf, a.src.scale, a.src.shift, ci, jok ? C : 0);
    const bool has_xf = a.src.scale != nullptr;
    const int act = a.src.act;
    const float ms = a.src.mask_scale;

    float4 areg[APASS], breg[BPASS];
    uint32_t bmask[BPASS];
    unsigned bok = 0, aok = 0;

    auto issue = [&](int chunk) {
        const int64_t p0 = (int64_t)chunk * WGN_BKP;
        // every load is unconditional, from a clamped position; what does not exist is zeroed in the staging step (bit i of aok / bok).
        // Behind bounds tests the loads of a chunk sat in separate basic blocks and the kernel streamed at a third of what its bytes allow.
        const int64_t plast = P - 1;
        const int coc = min(co, a.Cout > 4 ? a.Cout - 4 : 0);
        aok = 0;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int64_t p = p0 + apr + i * ARPP;
            areg[i] = uda_ld4(a.dy + min(p, plast) * a.lddy + (co < a.Cout ? co : coc));
            aok |= ((p < P && co < a.Cout) ? 1u : 0u) << i;
        }
        bok = 0;
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int64_t p = p0 + bpr + i * BRPP;
            bool in = jok && p < P;
            int64_t q = min(p, plast);
            if (a.ksize >= 2) {          // (1x1: the tap is the pixel itself - no coordinates, i.e. no two 64-bit divisions per row and chunk)
                const int w0 = (int)(q % W), h0 = (int)((q / W) % H);
                const int hh = h0 + dh, ww = w0 + dw;
                const bool inside = hh >= 0 && hh < H && ww >= 0 && ww < W;
                in = in && inside;
                q = inside ? q + (int64_t)dh * W + dw : q;
            }
            const int cic = jok ? ci : 0;
            breg[i] = uda_ld4(a.src.x + q * a.src.ldx + cic);
            bmask[i] = a.src.mask ? *reinterpret_cast<const uint32_t*>(a.src.mask + q * a.src.ldm + cic) : 0x01010101u;
            bok |= (in ? 1u : 0u) << i;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            float v[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
            const bool oka = (aok >> i) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!oka || co + j >= a.Cout) v[j] = 0.f;
            uda_st4(&As[(apr + i * ARPP) * BM + acv], make_float4(v[0], v[1], v[2], v[3]));
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));       // packed prologue, as in the forward kernels
        const f32x2 sc01 = {xf.sc[0], xf.sc[1]}, sc23 = {xf.sc[2], xf.sc[3]};
        const f32x2 sh01 = {xf.sh[0], xf.sh[1]}, sh23 = {xf.sh[2], xf.sh[3]};
        const bool masked = a.src.mask != nullptr;
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const bool ok = (bok >> i) & 1u;
            f32x2 t01 = {breg[i].x, breg[i].y}, t23 = {breg[i].z, breg[i].w};
            if (has_xf) {
                t01 = t01 * sc01 + sh01;
                t23 = t23 * sc23 + sh23;
            }
            t01 = f32x2{uda_act(t01.x, act), uda_act(t01.y, act)};
            t23 = f32x2{uda_act(t23.x, act), uda_act(t23.y, act)};
            if (masked) {
                const uint32_t mk = bmask[i];
                t01 *= f32x2{(float)(mk & 0xffu), (float)((mk >> 8) & 0xffu)} * ms;
                t23 *= f32x2{(float)((mk >> 16) & 0xffu), (float)(mk >> 24)} * ms;
            }
            float v[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(ok && (ci + j) < C)) v[j] = 0.f;
            uda_st4(&Bs[(bpr + i * BRPP) * BN + bjv], make_float4(v[0], v[1], v[2], v[3]));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int c0 = split * a.chunks_per_split;
    const int c1 = min(a.nchunks, c0 + a.chunks_per_split);
    const int acol = wm * TM * 32 + (lane & 31), bcol = wn * TN * 32 + (lane & 31);
    const int kh = lane >> 5;
    if (c0 < c1) issue(c0);
    for (int c = c0; c < c1; ++c) {
        __syncthreads();
        stage();
        __syncthreads();
        if (c + 1 < c1) issue(c + 1);
#pragma unroll
        for (int kk = 0; kk < WGN_BKP / 2; ++kk) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[(2 * kk + kh) * BM + acol + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[(2 * kk + kh) * BN + bcol + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    float* slab = a.slab + (int64_t)split * a.Cout * a.Jtot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = jt * BN + wn * TN * 32 + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = cot * BM + wm * TM * 32 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < a.Cout && col < a.Jtot) slab[(int64_t)row * a.Jtot + col] = acc[i][j][r];
            }
        }
}

// dw[co][ci][t] = sum_s slab[s][co][t*Kc + ci].  256 / Q outputs x Q split-lanes per workgroup: lane q sums the
// slabs s = q, q+Q, ... (4 loads in flight), the Q partial sums meet in LDS in a fixed order (bitwise
// reproducible).  Q = 8 (32 outputs per workgroup); Q = 64 for the many-slab plans of tiny weights (S up to 1024).
// The serial one-thread-per-output form took 38 us on the small layers (S up to 256).
template <int Q>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int S, int Cout, int Cin, int T,
                                                           int Kc, float* __restrict__ dw) {
    constexpr int NO = 256 / Q;
    __shared__ float red[Q][NO + 1];
    const int64_t total = (int64_t)Cout * T * Cin;
    const int64_t stride = (int64_t)Cout * T * Kc;
    const int ol = threadIdx.x % NO, q = threadIdx.x / NO;
    for (int64_t e0 = (int64_t)blockIdx.x * NO; e0 < total; e0 += (int64_t)gridDim.x * NO) {
        const int64_t e = e0 + ol;
        float s = 0.f;
        int64_t dst = 0;
        if (e < total) {
            const int ci = (int)(e % Cin);
            const int t = (int)((e / Cin) % T);
            const int co = (int)(e / ((int64_t)Cin * T));
            dst = ((int64_t)co * Cin + ci) * T + t;
            const float* src = slab + ((int64_t)co * T + t) * Kc + ci;
            int k = q;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (; k + 3 * Q < S; k += 4 * Q) {
                s0 += src[(int64_t)k * stride];
                s1 += src[(int64_t)(k + Q) * stride];
                s2 += src[(int64_t)(k + 2 * Q) * stride];
                s3 += src[(int64_t)(k + 3 * Q) * stride];
            }
            for (; k < S; k += Q) s0 += src[(int64_t)k * stride];
            s = (s0 + s1) + (s2 + s3);
        }
        red[q][ol] = s;
        __syncthreads();
        if (q == 0 && e < total) {
            float acc[8];           // eight interleaved partial sums, combined as a tree: the same association for every Q
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int j = 0; j < Q; ++j) acc[j & 7] += red[j][ol];
            dw[dst] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
        __syncthreads();
    }
}

struct WgradPlan {
    int bm, bn, nCot, nJt, S, cps, nchunks;
};

static WgradPlan wgrad_plan(int64_t P, int Cout, int Cin, int ksize) {
    WgradPlan p;
    const int Kc = ((Cin + 3) / 4) * 4, J = ksize * ksize * Kc;
    const int cm = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    const int cn = J <= 32 ? 32 : (J <= 64 ? 64 : 128);
    const bool big = Cout >= 192 && J >= 256 && uda_cdiv(P, WG_BKP) >= 4096;      // (measured: a loss below ~100k pixels)
    if (big) { p.bm = 256; p.bn = 256; }
    else if (cm == 128 && cn == 32) { p.bm = 128; p.bn = 32; }
    else if (cm == 32 && cn == 128) { p.bm = 32; p.bn = 128; }
    else if (cm <= 64 && cn <= 64) { p.bm = 64; p.bn = 64; }
    else { p.bm = 128; p.bn = 128; }
    p.nCot = uda_cdiv(Cout, p.bm);
    p.nJt = uda_cdiv(J, p.bn);
    p.nchunks = uda_cdiv(P, ((p.bm == 128 && p.bn == 128) || p.bm == 256) ? WG_BKP : WGN_BKP);
    int S = (p.bm == 256 ? 512 : 1024) / (p.nCot * p.nJt);      // 256 x 256 tiles: one workgroup per CU, two rounds
    if (S > p.nchunks / 4) S = p.nchunks / 4;
    if (S < 1) S = 1;
    // up to 256 slabs; up to 1024 for a tiny weight (one or two tiles over a million pixels: the backbone's first blocks, the decoder's
    // 24 -> 48 conv): with 256 workgroups each walks thousands of pixels at one chunk's load latency per chunk, and its slabs are small
    const int scap = (int64_t)Cout * J <= 8192 ? 1024 : 256;
    if (S > scap) S = scap;
    p.cps = uda_cdiv(p.nchunks, S);
    p.S = uda_cdiv(p.nchunks, p.cps);
    return p;
}

extern "C" uint64_t uda_conv_wgrad_workspace_bytes(int64_t P, int Cout, int Cin, int ksize) {
    const WgradPlan p = wgrad_plan(P, Cout, Cin, ksize);
    const int Kc = ((Cin + 3) / 4) * 4;
    return (uint64_t)p.S * Cout * ksize * ksize * Kc * sizeof(float);
}

extern "C" int uda_conv_wgrad_uses_x3(const uda_wgrad_args_t* a) {
    if (!a || a->mfma != UDA_MFMA_BF16X3 || a->ksize < 1 || a->ksize > 3) return 0;
    const int sd = a->stride <= 1 ? 1 : a->stride;
    const int64_t P = (int64_t)a->src.N * ((a->src.H - 1) / sd + 1) * ((a->src.W - 1) / sd + 1);      // pixels of dy
    const WgradPlan p = wgrad_plan(P, a->Cout, a->src.C, a->ksize);
    const bool wide = p.bm == 256 || (p.bm == 128 && p.bn == 128);
    return wide && wgrad_x3_eligible(a->src.C, a->Cout, a->ksize, P) ? 1 : 0;
}

extern "C" int uda_conv_wgrad(const uda_wgrad_args_t* a, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    UDA_REQUIRE(a != nullptr, "uda_conv_wgrad: null args");
    if (int e = check_src(a->src, "uda_conv_wgrad")) return e;
    UDA_REQUIRE(a->ksize >= 1 && a->ksize <= 3, "uda_conv_wgrad: ksize must be 1, 2 or 3");
    UDA_REQUIRE(a->ksize != 2 || a->origin == 0 || a->origin == 1, "uda_conv_wgrad: origin must be 0 or 1 for ksize 2");
    UDA_REQUIRE(a->dy && uda_aligned16(a->dy) && a->lddy % 4 == 0 && a->lddy >= ((a->Cout + 3) / 4) * 4,
                "uda_conv_wgrad: dy must be 16-byte aligned with lddy %% 4 == 0 and >= round4(Cout)");
    // stride 2: dy lives on the output grid, its pixel (n, oh, ow) pairs with source pixel (n, 2 oh, 2 ow); P = pixels of dy
    const int sd = a->stride <= 1 ? 1 : a->stride;
    UDA_REQUIRE(sd <= 2, "uda_conv_wgrad: stride must be 1 or 2");
    const int Ho = (a->src.H - 1) / sd + 1, Wo = (a->src.W - 1) / sd + 1;
    const int64_t P = (int64_t)a->src.N * Ho * Wo;
    const WgradPlan p = wgrad_plan(P, a->Cout, a->src.C, a->ksize);
    UDA_REQUIRE(a->workspace && a->workspace_bytes >= uda_conv_wgrad_workspace_bytes(P, a->Cout, a->src.C, a->ksize),
                "uda_conv_wgrad: workspace too small");
    UDA_REQUIRE(sd == 1 || p.bm == 256 || (p.bm == 128 && p.bn == 128),
                "uda_conv_wgrad: stride 2 is built on the wide-tile kernels only (Cout > 64, K > 64)");
    WgradKArgs k;
    k.src = a->src;
    k.stride = sd; k.Ho = Ho; k.Wo = Wo;
    k.dy = a->dy;
    k.lddy = a->lddy;
    k.Cout = a->Cout;
    k.ksize = a->ksize;
    k.dil = a->dil;
    k.cen = a->ksize == 3 ? 1 : (a->ksize == 2 ? a->origin : 0);
    k.Kc = ((a->src.C + 3) / 4) * 4;
    k.Jtot = a->ksize * a->ksize * k.Kc;
    k.slab = a->workspace;
    k.nCot = p.nCot;
    k.nJt = p.nJt;
    k.chunks_per_split = p.cps;
    k.nchunks = p.nchunks;
    dim3 grid(p.nCot * p.nJt, p.S);
    int S_used = p.S;
    if (uda_conv_wgrad_uses_x3(a)) { if (int e = launch_wgrad_x3(k, P, p.S, a->x3_src, a->x3_dy, st, S_used)) return e; }
    else if (p.bm == 256) { if (int e = launch_wgrad_ws(k, p.S, true, st)) return e; }
    else if (p.bm == 128 && p.bn == 128) { if (int e = launch_wgrad_ws(k, p.S, false, st)) return e; }
    else if (p.bm == 64) hipLaunchKernelGGL((igemm_wgrad_kernel<1, 1, 2, 2>), grid, dim3(256), 0, st, k);
    else if (p.bm == 128) hipLaunchKernelGGL((igemm_wgrad_kernel<1, 1, 4, 1>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((igemm_wgrad_kernel<1, 1, 1, 4>), grid, dim3(256), 0, st, k);
    UDA_LAUNCH_CHECK("igemm_wgrad");
    const int T = a->ksize * a->ksize;
    const int64_t total = (int64_t)a->Cout * T * a->src.C;
    if (S_used > 256)
        hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3(uda_cdiv(total, 4) > 4096 ? 4096 : uda_cdiv(total, 4)), dim3(256), 0, st,
                           k.slab, S_used, a->Cout, a->src.C, T, k.Kc, a->dw);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3(uda_cdiv(total, 32) > 4096 ? 4096 : uda_cdiv(total, 32)), dim3(256), 0, st,
                           k.slab, S_used, a->Cout, a->src.C, T, k.Kc, a->dw);
    UDA_LAUNCH_CHECK("wgrad_reduce");
    return 0;
}

// ==========================================================================================
// weight re-layouts
// K order of a weight row: T == 1: [Kc];  T > 1 ("tap-chunked"): [nCC][T][32], k = (cc*T + t)*32 + c % 32 with
// cc = c / 32, channels zero-padded to a multiple of 32: all T taps of one 32-channel slice are consecutive
// K-chunks, so a workgroup re-reads its pixel strip while it is still L2-resident (the tap-major order re-read it from
// the memory side: 5x the algorithmic bytes on FETCH_SIZE).
__device__ __forceinline__ void uda_k_decode(int64_t e, int T, int Kr, int Kc, int& c, int& t, int& row) {
    // e indexes [row][Kr]; Kr = Kc (T == 1), T*Kc (fewer than 32 channels: tap-major, no padding) or nCC*T*32
    const int k = (int)(e % Kr);
    row = (int)(e / Kr);
    if (T == 1) {
        c = k;
        t = 0;
    } else if (Kc < IG_BK) {
        t = k / Kc;
        c = k - t * Kc;
    } else {
        const int chunk = k / IG_BK;
        t = chunk % T;
        c = (chunk / T) * IG_BK + (k % IG_BK);
    }
}
__global__ void relayout_ohwi_kernel(const float* __restrict__ w, int O, int I, int T, int Kr, float* __restrict__ out) {
    const int64_t total = (int64_t)O * Kr;
    const int Kc = ((I + 3) / 4) * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int ci, t, o;
        uda_k_decode(e, T, Kr, Kc, ci, t, o);
        out[e] = ci < I ? w[((int64_t)o * I + ci) * T + t] : 0.f;
    }
}
__global__ void relayout_dgrad_kernel(const float* __restrict__ w, int O, int I, int T, int Kr, float* __restrict__ out) {
    const int64_t total = (int64_t)I * Kr;
    const int Kc = ((O + 3) / 4) * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int o, t, ci;
        uda_k_decode(e, T, Kr, Kc, o, t, ci);
        out[e] = o < O ? w[((int64_t)o * I + ci) * T + (T - 1 - t)] : 0.f;
    }
}
__global__ void relayout_dw_kernel(const float* __restrict__ w, int C, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 9 * C) out[e] = w[(e % C) * 9 + e / C];
}

static inline int grid_for(int64_t total) {
    int g = uda_cdiv(total, 256);
    return g > 4096 ? 4096 : (g < 1 ? 1 : g);
}

extern "C" int uda_relayout_ohwi(const float* w, int O, int I, int k, float* out, void* stream) {
    UDA_REQUIRE(w && out && O > 0 && I > 0 && (k >= 1 && k <= 3), "uda_relayout_ohwi: bad args");
    const int Kr = uda_k_row(I, k);
    hipLaunchKernelGGL(relayout_ohwi_kernel, dim3(grid_for((int64_t)O * Kr)), dim3(256), 0, (hipStream_t)stream,
                       w, O, I, k * k, Kr, out);
    UDA_LAUNCH_CHECK("relayout_ohwi");
    return 0;
}
extern "C" int uda_relayout_dgrad(const float* w, int O, int I, int k, float* out, void* stream) {
    UDA_REQUIRE(w && out && O > 0 && I > 0 && (k >= 1 && k <= 3), "uda_relayout_dgrad: bad args");
    const int Kr = uda_k_row(O, k);
    hipLaunchKernelGGL(relayout_dgrad_kernel, dim3(grid_for((int64_t)I * Kr)), dim3(256), 0, (hipStream_t)stream,
                       w, O, I, k * k, Kr, out);
    UDA_LAUNCH_CHECK("relayout_dgrad");
    return 0;
}
extern "C" int uda_relayout_dw(const float* w, int C, float* out, void* stream) {
    UDA_REQUIRE(w && out && C > 0, "uda_relayout_dw: bad args");
    hipLaunchKernelGGL(relayout_dw_kernel, dim3(uda_cdiv(9 * C, 256)), dim3(256), 0, (hipStream_t)stream, w, C, out);
    UDA_LAUNCH_CHECK("relayout_dw");
    return 0;
}
