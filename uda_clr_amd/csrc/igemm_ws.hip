// Warp-specialised FP32-MFMA implicit-GEMM kernels for the wide (MFMA-bound) convolutions on gfx950.
//
// Measured on the single-role kernel (igemm_conv.hip, rocprofv3 PMC, decoder 3x3 at B=16): 6.5 VALU
// instructions per MFMA (address arithmetic + BN/ReLU/dropout prologue + LDS staging) and a
// matrix pipe busy 65 % of the time: with one role per wave the staging phases of the two co-resident
// waves of a SIMD leave the pipe idle.  Here a 512-thread workgroup puts TWO waves on every SIMD
// with fixed roles:
//     waves 0-3  "math"   : only ds_read_b128 operand fragments + v_mfma_f32_32x32x2_f32
//     waves 4-7  "loader" : global -> registers -> (prologue) -> LDS for the NEXT K-chunk
// on a double-buffered LDS tile pair with ONE barrier per chunk, so every SIMD always has a wave
// whose next instruction is an MFMA while its partner does the vector/memory work (the
// producer/consumer split of the guide's loader-ring engines, with barriers instead of flags since
// both roles live in one workgroup).  Loader waves are the later-dispatched half, i.e. they lose
// issue arbitration to the math waves (MI355X_MICROARCH.md "Two waves per SIMD").
//
// Tile: BM = 128 pixels x BN = 64*TN couts (TN = 2 -> 128, TN = 4 -> 256: whole Cout = 256 of the
// decoder / ASPP convs in one tile, so the activation tile is staged once), BK = 32.
#include "common.h"
#include <stdlib.h>
#include "igemm_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef UDA_WS_MATH_WAVES_EVEN
#define UDA_WS_MATH_WAVES_EVEN 8      // math waves of the even-width tiles (TN = 2, 4)
#endif

// XF: 0 = raw operand, 1 = BN affine + activation, 2 = same + dropout keep-mask
// MW = number of math waves (4: one per SIMD, wave tile 64 x 32*TN; 8: two per SIMD, wave tile 64 x 16*TN, so
// one math wave's fragment reads / waits are covered by the other's MFMAs); the 4 loader waves follow them.
// BM = 256 (8 math waves as 4 x 2, wave tile 64 x 32*TN): a third less staging per MFMA than BM = 128.
template <int KS, int XF, int TN, int MW, int BM>
__global__ __launch_bounds__((MW + 4) * 64) void igemm_conv_ws_kernel(ConvKArgs a) {
    constexpr int BN = 64 * TN, TM = 2;
    constexpr int WMM = BM / 64;                   // math waves along M
    constexpr int WNN = MW / WMM;                  // math waves along N
    constexpr int TNW = 2 * TN / WNN;              // 32-column blocks per math wave
    static_assert(WMM * WNN == MW && (2 * TN) % WNN == 0, "math-wave grid must tile the workgroup tile");
    constexpr int NTHR = (MW + 4) * 64;
    constexpr int A_IT = BM / 32;
    constexpr int B_IT = BN / 32;
    constexpr int TILE = (BM + BN) * IG_LD;        // floats per buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = __builtin_amdgcn_readfirstlane(wave) >= MW;    // provably wave-uniform
    const int lid = uda_xcd_remap(blockIdx.x, a.nMt * a.nNt);
    const int mt = lid / a.nNt, nt = lid % a.nNt;
    const int H = a.src.H, W = a.src.W, C = a.src.C;
    const int64_t P = (int64_t)a.src.N * a.Ho * a.Wo;          // output rows (= input pixels at stride 1)
    const int64_t Pin = (int64_t)a.src.N * H * W;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int nchunks = (a.Ktot + IG_BK - 1) / IG_BK;

    // ------------------------------------------------------------------ loader state
    // All per-row quantities are precomputed once (32-bit element offsets, a 9-bit tap-validity
    // mask per pixel row); a K-chunk then costs ~6 VALU per operand row instead of a 64-bit
    // multiply + four compares.  (PMC on the first version: loader waves, not the MFMA pipe, set the
    // chunk time: 11.4k-13.6k cycles against 8.2k cycles of MFMA work.)
    const int lt = (tid - MW * 64) & 255, lrow = lt >> 3, kv = (lt & 7) * 4;
    int rowoff[A_IT], rowoffm[A_IT], boff[B_IT];
    unsigned vmask[A_IT];
    float4 areg[A_IT], breg[B_IT];
    uint32_t amask[A_IT];
    unsigned aok = 0;
    Xf4 xf;
    int st_ci = 0;               // channel of the staged registers' first element
    int t_cur = 0, ci_cur = kv;  // (tap, channel) of the NEXT chunk to issue: tap-chunked K, chunk = cc * T + t
    int kb_cur = kv;             // its position in the weight row
    if (loader) {
        const int ldx = (int)a.src.ldx, ldm = (int)a.src.ldm;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int64_t p = m0 + lrow + 32 * i;
            const bool ok = p < P;
            const int qo = ok ? (int)p : 0;
            // output pixel (n, oh, ow) is centred on input pixel (n, oh * stride, ow * stride)
            const int pw = (qo % a.Wo) * a.stride, ph = ((qo / a.Wo) % a.Ho) * a.stride;
            const int q = ((qo / (a.Wo * a.Ho)) * H + ph) * W + pw;
            rowoff[i] = q * ldx;
            rowoffm[i] = q * ldm;
            unsigned vm = 0;
            if (ok) {
                if (KS == 3) {          // multi-tap: 3x3 (9 taps) or 2x2 (4 taps), runtime a.ksize
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int th = a.ksize == 3 ? t / 3 : t / 2, tw = a.ksize == 3 ? t % 3 : t % 2;
                        const int hh = ph + (th - a.cen) * a.dil, ww = pw + (tw - a.cen) * a.dil;
                        vm |= (t < a.ksize * a.ksize && hh >= 0 && hh < H && ww >= 0 && ww < W ? 1u : 0u) << t;
                    }
                } else {
                    vm = 1u;
                }
            }
            vmask[i] = vm;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = min(n0 + lrow + 32 * i, a.Cout - 1);
            boff[i] = n * a.Ktot;
        }
    }

    // Operand loads go through buffer descriptors: a lane whose tap falls outside the image (or beyond the channels)
    // presents an out-of-range offset and the hardware returns zeros - no branch, no select, no memory access.
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src.x), 0, (int)min((int64_t)0x7fffffff, (Pin * a.src.ldx) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.src.mask ? a.src.mask : reinterpret_cast<const uint8_t*>(a.src.x)), 0,
        a.src.mask ? (int)min((int64_t)0x7fffffff, Pin * a.src.ldm) : 0, 0x00020000);
    constexpr int OOB = 0x7ffffff0;

    auto issue = [&]() {       // loads of the chunk at (t_cur, ci_cur); then advance by BK
        const int t = t_cur, ci = ci_cur;
        const bool kval = ci < a.Kc;
        st_ci = ci;
        int tapoff = 0;
        if (KS == 3) {
            const int th = a.ksize == 3 ? t / 3 : t >> 1;
            tapoff = ((th - a.cen) * W + (t - th * a.ksize - a.cen)) * a.dil;
        }
        if (a.debug & 16) tapoff = 0;      // diagnostics: every tap re-reads the centre pixel (cache-resident operand)
        if (XF >= 1) uda_load_xf4(xf, a.src.scale, a.src.shift, ci, kval ? C : 0);
        const int xoff = tapoff * (int)a.src.ldx + ci;
        const int moff = tapoff * (int)a.src.ldm + ci;
        aok = 0;
        // weight rows of the tile padding read the last real row (their output columns are never stored), the K tail
        // of a 1x1 conv reads the row's last granule against zeroed activations
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = kval && ((vmask[i] >> (KS == 3 ? t : 0)) & 1u);
            areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? (rowoff[i] + xoff) * 4 : OOB, 0, 0));
            if (XF == 2) amask[i] = __builtin_amdgcn_raw_buffer_load_b32(mres, ok ? rowoffm[i] + moff : OOB, 0, 0);
            aok |= (ok ? 1u : 0u) << i;
        }
        const int k0 = min(kb_cur, a.Ktot - 4);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) breg[i] = uda_ld4(a.w + (boff[i] + k0));
        kb_cur += IG_BK;
        if (KS == 3) {
            if (++t_cur == a.ksize * a.ksize) {
                t_cur = 0;
                ci_cur += IG_BK;
            }
        } else {
            ci_cur += IG_BK;
        }
    };

    auto stage = [&](float* buf) {
        float* As = buf;
        float* Bs = buf + BM * IG_LD;
        const int act = a.src.act;
        const float ms = a.src.mask_scale;
        const bool ctail = (C & 3) != 0;
        const float alo = act == ACT_NONE ? -INFINITY : 0.f, ahi = act == ACT_RELU6 ? 6.f : INFINITY;
        // prologue on packed pairs (v_pk_fma_f32 / v_pk_mul_f32: half the VALU issue slots, which the loader shares
        // with the SIMD's MFMA stream - PMC: 3.0 VALU per MFMA with the scalar form of the masked prologue)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc01 = {xf.sc[0], xf.sc[1]}, sc23 = {xf.sc[2], xf.sc[3]};
        const f32x2 sh01 = {xf.sh[0], xf.sh[1]}, sh23 = {xf.sh[2], xf.sh[3]};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            float v[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
            const bool ok = (aok >> i) & 1u;
            if (XF >= 1) {
                f32x2 t01 = f32x2{v[0], v[1]} * sc01 + sh01, t23 = f32x2{v[2], v[3]} * sc23 + sh23;
                // none / ReLU / ReLU6 as ONE clamp (v_med3_f32) with per-launch bounds
                t01 = f32x2{__builtin_amdgcn_fmed3f(t01.x, alo, ahi), __builtin_amdgcn_fmed3f(t01.y, alo, ahi)};
                t23 = f32x2{__builtin_amdgcn_fmed3f(t23.x, alo, ahi), __builtin_amdgcn_fmed3f(t23.y, alo, ahi)};
                // multiplier: 0 outside the image, keep-mask * 1/(1-p) inside (v_cvt_f32_ubyteN; amask = 0 when !ok)
                f32x2 m01, m23;
                if (XF == 2) {
                    const uint32_t mk = amask[i];
                    m01 = f32x2{(float)(mk & 0xffu), (float)((mk >> 8) & 0xffu)} * ms;
                    m23 = f32x2{(float)((mk >> 16) & 0xffu), (float)(mk >> 24)} * ms;
                } else {
                    const float o = ok ? 1.f : 0.f;
                    m01 = f32x2{o, o};
                    m23 = m01;
                }
                t01 *= m01;
                t23 *= m23;
                v[0] = t01.x; v[1] = t01.y; v[2] = t23.x; v[3] = t23.y;
            }
            if (ctail) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((st_ci + j) >= C) v[j] = 0.f;
            }
            uda_st4(&As[(lrow + 32 * i) * IG_LD + kv], make_float4(v[0], v[1], v[2], v[3]));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) uda_st4(&Bs[(lrow + 32 * i) * IG_LD + kv], breg[i]);
    };

    // ------------------------------------------------------------------ math state
    const int wm = (wave / WNN) % WMM, wn = wave % WNN;
    const int arow = wm * 64 + (lane & 31), brow = wn * (32 * TNW) + (lane & 31);
    const int koff = 4 * (lane >> 5);

    // ------------------------------------------------------------------ pipeline
    // The role branch is OUTERMOST (wave-uniform, scalar branch) so the register allocation is the
    // maximum of the two roles' live sets, not their sum; both roles execute the same number of
    // workgroup barriers.
    float s1[TNW], s2[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) s1[j] = s2[j] = 0.f;
    if (loader) {
        issue();
        stage(smem);
        if (nchunks > 1) issue();
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks && !(a.debug & 2)) {
                if (!(a.debug & 8)) stage(smem + ((c + 1) & 1) * TILE);     // chunk c+1 (its loads flew during chunk c-1)
                if (c + 2 < nchunks && !(a.debug & 4)) issue();           // chunk c+2 flies during chunk c+1
            }
            __syncthreads();
        }
    } else {
        f32x16 acc[TM][TNW];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);      // math waves win issue arbitration over their loader partners
        for (int c = 0; c < nchunks; ++c) {
            if (a.debug & 1) {
                __syncthreads();
                continue;
            }
            const float* As = smem + (c & 1) * TILE;
            const float* Bs = As + BM * IG_LD;
            if constexpr (BM == 128) {
            // operand fragments are double-buffered in registers: group g+1 is read from LDS while
            // the 8*TN MFMAs of group g run, so only the first read after the barrier is exposed
            float4 af[2][TM], bf[2][TNW];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = uda_ld4(&As[(arow + 32 * i) * IG_LD + koff]);
#pragma unroll
            for (int j = 0; j < TNW; ++j) bf[0][j] = uda_ld4(&Bs[(brow + 32 * j) * IG_LD + koff]);
#pragma unroll
            for (int g = 0; g < IG_BK / 8; ++g) {
                const int cur = g & 1, nxt = cur ^ 1;
                if (g + 1 < IG_BK / 8) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[nxt][i] = uda_ld4(&As[(arow + 32 * i) * IG_LD + (g + 1) * 8 + koff]);
#pragma unroll
                    for (int j = 0; j < TNW; ++j) bf[nxt][j] = uda_ld4(&Bs[(brow + 32 * j) * IG_LD + (g + 1) * 8 + koff]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float4 a4 = af[cur][i];
                        const float av = s == 0 ? a4.x : s == 1 ? a4.y : s == 2 ? a4.z : a4.w;
#pragma unroll
                        for (int j = 0; j < TNW; ++j) {
                            const float4 b4 = bf[cur][j];
                            const float bv = s == 0 ? b4.x : s == 1 ? b4.y : s == 2 ? b4.z : b4.w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
            } else {
            // 168-VGPR budget (3 waves per SIMD, 128 accumulator registers): fragments single-buffered, the
            // read latency of one math wave is covered by the MFMAs of the other math wave of the SIMD
#pragma unroll
            for (int g = 0; g < IG_BK / 8; ++g) {
                float4 af[TM], bf[TNW];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = uda_ld4(&As[(arow + 32 * i) * IG_LD + g * 8 + koff]);
#pragma unroll
                for (int j = 0; j < TNW; ++j) bf[j] = uda_ld4(&Bs[(brow + 32 * j) * IG_LD + g * 8 + koff]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float av = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
                        for (int j = 0; j < TNW; ++j) {
                            const float bv = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
            }
            __syncthreads();
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int colb = n0 + wn * (32 * TNW) + (lane & 31);
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int col = colb + 32 * j;
            const bool cok = col < a.Cout;
            const float bv = (cok && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (cok && row < P) {
                        float v = acc[i][j][r] + bv;
                        s1[j] += v;
                        s2[j] += v * v;
                        if (a.addend) v += a.addend[row * a.ld_add + col];
                        a.y[row * a.ldy + col] = v;
                    }
                }
            }
        }
    }
    if (a.stats) {           // uniform over the workgroup
        float* red = smem;   // [WMM][2][BN]
        if (!loader) {
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const float t1 = s1[j] + __shfl_xor(s1[j], 32);
                const float t2 = s2[j] + __shfl_xor(s2[j], 32);
                if (lane < 32) {
                    const int cl = wn * (32 * TNW) + 32 * j + lane;
                    red[(wm * 2 + 0) * BN + cl] = t1;
                    red[(wm * 2 + 1) * BN + cl] = t2;
                }
            }
        }
        __syncthreads();
        double* dst = a.stats + (int64_t)(mt % UDA_STAT_SLOTS) * 2 * a.Cout;
        for (int e = tid; e < 2 * BN; e += NTHR) {
            const int qd = e / BN, cl = e % BN;
            if (n0 + cl < a.Cout) {
                float t = 0.f;
#pragma unroll
                for (int m = 0; m < WMM; ++m) t += red[(m * 2 + qd) * BN + cl];
                atomicAdd(&dst[qd * a.Cout + n0 + cl], (double)t);
            }
        }
    }
}

template <int KS, int XF, int TN, int BM>
static int launch_ws(ConvKArgs& k, int64_t P, hipStream_t st) {
    constexpr int BN = 64 * TN;
    constexpr int MW = BM == 64 ? 4 : ((BM == 256 || TN % 2 == 0) ? UDA_WS_MATH_WAVES_EVEN : 4);
    constexpr size_t lds = 2 * (BM + BN) * IG_LD * sizeof(float);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool configured_dev[UDA_MAX_DEVICES] = {};       // hipFuncSetAttribute is per device
    bool& configured = configured_dev[uda_device_slot()];
    auto fn = igemm_conv_ws_kernel<KS, XF, TN, MW, BM>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return uda_set_error("igemm_conv_ws: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        configured = true;
    }
    k.nMt = uda_cdiv(P, BM);
    k.nNt = uda_cdiv(k.Cout, BN);
    #ifdef UDA_DIAG          // diagnostic builds only (make DIAG=1): the ablation modes change the results
    static const int dbg = getenv("UDA_WS_DEBUG") ? atoi(getenv("UDA_WS_DEBUG")) : 0;
#else
    const int dbg = 0;
#endif
    k.debug = dbg;
    hipLaunchKernelGGL(fn, dim3(k.nMt * k.nNt), dim3((MW + 4) * 64), lds, st, k);
    UDA_LAUNCH_CHECK("igemm_conv_ws");
    return 0;
}

template <int KS, int TN, int BM>
static int launch_ws_xf(ConvKArgs& k, int64_t P, hipStream_t st) {
    if (k.src.mask) return launch_ws<KS, 2, TN, BM>(k, P, st);
    if (k.src.scale) return launch_ws<KS, 1, TN, BM>(k, P, st);
    if (k.src.act != ACT_NONE) return launch_ws<KS, 1, TN, BM>(k, P, st);
    return launch_ws<KS, 0, TN, BM>(k, P, st);
}

template <int KS>
static int launch_ws_tn(ConvKArgs& k, int64_t P, int tn, bool tall, bool low, hipStream_t st) {
    if (low) return launch_ws_xf<KS, 2, 64>(k, P, st);          // 64 x 128 tiles: few pixels (32x32 maps), fill the CUs first
    switch (tn) {
        case 2: return launch_ws_xf<KS, 2, 128>(k, P, st);      // two 128x128 workgroups per CU beat one 256x128 (measured)
        case 3: return launch_ws_xf<KS, 3, 128>(k, P, st);
        case 4: return tall ? launch_ws_xf<KS, 4, 256>(k, P, st) : launch_ws_xf<KS, 4, 128>(k, P, st);
        default: return launch_ws_xf<KS, 5, 128>(k, P, st);
    }
}

int launch_conv_ws(ConvKArgs& k, int64_t P, hipStream_t st) {
    const int64_t lim = (int64_t)1 << 31, Pin = (int64_t)k.src.N * k.src.H * k.src.W;      // (P: output rows)
    UDA_REQUIRE((Pin + 128) * k.src.ldx < lim / 4 && (Pin + 128) * (k.src.mask ? k.src.ldm : 1) < lim && (int64_t)(k.Cout + 320) * k.Ktot < lim,
                "uda_conv_fwd: operand too large for the 32-bit element offsets of the wide-tile kernel");
    // Tile width BN = 64*TN chosen by a wave-quantisation model: workgroups run one per CU, a K-chunk
    // costs ~TN MFMA-units, so time ~ ceil(#tiles / 256 CUs) * TN.  E.g. Cout = 304 at P = 262144 ->
    // TN = 5 (one 320-wide tile, 5 % padding); Cout = 320 at P = 16384 -> TN = 3 (256 workgroups).
    const int64_t nMt = uda_cdiv(P, 128);
    int best = 2;
    int64_t best_cost = -1;
    for (int tn = 2; tn <= 5; ++tn) {
        const int64_t tiles = nMt * uda_cdiv(k.Cout, 64 * tn);
        const int64_t cost = ((tiles + 255) / 256) * tn * 16 + (tn == 2 ? 3 : 0);   // BN=128 stages A twice as often
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = tn;
        }
    }
    // 256-pixel tiles (a third less operand staging per MFMA) once they still fill the chip twice over
    static const int tall_env = getenv("UDA_WS_TALL") ? atoi(getenv("UDA_WS_TALL")) : 1;
    const bool tall = tall_env && best == 4 && uda_cdiv(P, 256) * uda_cdiv(k.Cout, 256) >= 512;
    // few pixels (ResNet's 32x32-map layers at B = 8: 64 tiles of 128 rows for 256 CUs): 64-row tiles, twice the workgroups
    static const int low_env = getenv("UDA_WS_LOW") ? atoi(getenv("UDA_WS_LOW")) : 1;
    const bool low = low_env && nMt * uda_cdiv(k.Cout, 64 * best) <= 192 && nMt * uda_cdiv(k.Cout, 128) <= 256;
    return k.ksize >= 2 ? launch_ws_tn<3>(k, P, best, tall, low, st) : launch_ws_tn<1>(k, P, best, tall, low, st);
}

// ==========================================================================================
// Weight gradient, same two-role structure.  Operand tiles are [pixel][row].  BIG = false: 128 (co) x 128 (j) tiles,
// 4 math waves of 64 x 64; BIG = true: 256 x 256 tiles, 8 math waves of 64 x 128 (half the staging per MFMA, one
// ds_read_b64 + one ds_read_b128 per 8 MFMAs), taken for the large layers (Cout >= 192, J >= 256, many pixel chunks).
template <int KS, int XF, bool BIG>
__global__ __launch_bounds__(BIG ? 768 : 512) void igemm_wgrad_ws_kernel(WgradKArgs a) {
    constexpr int BM = BIG ? 256 : 128, BN = BM, TM = 2, TN = BIG ? 4 : 2, MW = BIG ? 8 : 4;
    constexpr int RV = BM / 4, RP = 256 / RV, NP = WG_BKP / RP;     // float4 per staged row, pixel rows per pass, passes
    constexpr int TILE = WG_BKP * (BM + BN);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = __builtin_amdgcn_readfirstlane(wave) >= MW;
    const int cot = blockIdx.x / a.nJt, jt = blockIdx.x % a.nJt;
    const int split = blockIdx.y;
    const int H = a.src.H, W = a.src.W, C = a.src.C;
    const int Ho = a.Ho, Wo = a.Wo, sd = a.stride;              // grid of dy; its pixel (n, oh, ow) pairs with source pixel (n, oh * sd, ow * sd)
    const int64_t P = (int64_t)a.src.N * Ho * Wo, Pin = (int64_t)a.src.N * H * W;
    const int c0 = split * a.chunks_per_split;
    const int c1 = min(a.nchunks, c0 + a.chunks_per_split);

    // loader mapping: RV float4 per staged row, RP pixel rows per pass, NP passes
    const int lt = (tid - MW * 64) & 255;
    const int cv = (lt % RV) * 4, pr = lt / RV;
    const int co = cot * BM + cv;
    const int j0 = jt * BN + cv;
    const bool jok = j0 < a.Jtot;
    int t = 0, ci = j0;
    if (KS == 3 && jok) {
        t = j0 / a.Kc;
        ci = j0 - t * a.Kc;
    }
    int dh = 0, dw = 0;
    if (KS == 3) {
        const int th = a.ksize == 3 ? t / 3 : t >> 1;
        dh = (th - a.cen) * a.dil;
        dw = (t - th * a.ksize - a.cen) * a.dil;
    }
    Xf4 xf;
    if (XF >= 1) uda_load_xf4(xf, a.src.scale, a.src.shift, ci, jok ? C : 0);
    const int act = a.src.act;
    const float ms = a.src.mask_scale;
    const bool ctail = (C & 3) != 0, cotail = (a.Cout & 3) != 0;
    float4 areg[NP], breg[NP];
    uint32_t bmask[NP];
    unsigned bok = 0;
    // per staged pixel row: pixel index and (h, w), advanced by 32 pixels per chunk without divisions
    int pp[NP], hh0[NP], ww0[NP], sp[NP];                       // dy pixel, its (oh, ow), source pixel
    if (loader) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t p = (int64_t)c0 * WG_BKP + pr + i * RP;
            const int q = (int)(p < P ? p : 0);
            pp[i] = (int)p;
            ww0[i] = q % Wo;
            hh0[i] = (q / Wo) % Ho;
            sp[i] = ((q / (Wo * Ho)) * H + hh0[i] * sd) * W + ww0[i] * sd + (int)(p - q);      // (beyond P: never read)
        }
    }
    const int rowstep = sd * (W - Wo), imgstep = (H - sd * Ho) * W;      // source-pixel corrections at a row / image wrap (0 at stride 1)
    const int ldx = (int)a.src.ldx, ldm = (int)a.src.ldm, lddy = (int)a.lddy;
    const int tapoff = dh * W + dw;

    // buffer descriptors: lanes outside the pixel range / the image / the channel range present an out-of-range offset
    // and receive zeros from the hardware (no branches, no selects)
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy), 0, (int)min((int64_t)0x7fffffff, (P * a.lddy) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src.x), 0, (int)min((int64_t)0x7fffffff, (Pin * a.src.ldx) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.src.mask ? a.src.mask : reinterpret_cast<const uint8_t*>(a.src.x)), 0,
        a.src.mask ? (int)min((int64_t)0x7fffffff, Pin * a.src.ldm) : 0, 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    auto issue = [&]() {
        bok = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool pin = pp[i] < (int)P;
            areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                yres, (pin && co < a.Cout) ? (pp[i] * lddy + co) * 4 : OOB, 0, 0));
            const int hh = hh0[i] * sd + dh, ww = ww0[i] * sd + dw;
            const bool ok = jok && pin && hh >= 0 && hh < H && ww >= 0 && ww < W;
            const int q = sp[i] + tapoff;
            breg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? (q * ldx + ci) * 4 : OOB, 0, 0));
            if (XF == 2) bmask[i] = __builtin_amdgcn_raw_buffer_load_b32(mres, ok ? q * ldm + ci : OOB, 0, 0);
            bok |= (ok ? 1u : 0u) << i;
            // advance this row by one chunk (32 pixels)
            pp[i] += WG_BKP;
            sp[i] += WG_BKP * sd;
            ww0[i] += WG_BKP;
            while (ww0[i] >= Wo) {
                ww0[i] -= Wo;
                sp[i] += rowstep;
                if (++hh0[i] >= Ho) {
                    hh0[i] = 0;
                    sp[i] += imgstep;
                }
            }
        }
    };
    auto stage = [&](float* buf) {
        float* As = buf;
        float* Bs = buf + WG_BKP * BM;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float v[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
            if (cotail) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (co + j >= a.Cout) v[j] = 0.f;
            }
            uda_st4(&As[(pr + i * RP) * BM + cv], make_float4(v[0], v[1], v[2], v[3]));
        }
        // prologue on packed pairs with a one-instruction clamp, as in the forward kernel's loader
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc01 = {xf.sc[0], xf.sc[1]}, sc23 = {xf.sc[2], xf.sc[3]};
        const f32x2 sh01 = {xf.sh[0], xf.sh[1]}, sh23 = {xf.sh[2], xf.sh[3]};
        const float alo = act == ACT_NONE ? -INFINITY : 0.f, ahi = act == ACT_RELU6 ? 6.f : INFINITY;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float v[4] = {breg[i].x, breg[i].y, breg[i].z, breg[i].w};
            const bool ok = (bok >> i) & 1u;
            if (XF >= 1) {
                f32x2 t01 = f32x2{v[0], v[1]} * sc01 + sh01, t23 = f32x2{v[2], v[3]} * sc23 + sh23;
                t01 = f32x2{__builtin_amdgcn_fmed3f(t01.x, alo, ahi), __builtin_amdgcn_fmed3f(t01.y, alo, ahi)};
                t23 = f32x2{__builtin_amdgcn_fmed3f(t23.x, alo, ahi), __builtin_amdgcn_fmed3f(t23.y, alo, ahi)};
                f32x2 m01, m23;
                if (XF == 2) {
                    const uint32_t mk = bmask[i];          // 0 when !ok
                    m01 = f32x2{(float)(mk & 0xffu), (float)((mk >> 8) & 0xffu)} * ms;
                    m23 = f32x2{(float)((mk >> 16) & 0xffu), (float)(mk >> 24)} * ms;
                } else {
                    const float o = ok ? 1.f : 0.f;
                    m01 = f32x2{o, o};
                    m23 = m01;
                }
                t01 *= m01;
                t23 *= m23;
                v[0] = t01.x; v[1] = t01.y; v[2] = t23.x; v[3] = t23.y;
            }
            if (ctail) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((ci + j) >= C) v[j] = 0.f;
            }
            uda_st4(&Bs[(pr + i * RP) * BN + cv], make_float4(v[0], v[1], v[2], v[3]));
        }
    };

    const int wm = BIG ? (wave >> 1) & 3 : (wave & 3) >> 1, wn = wave & 1;
    // MFMA row slot r of a wave owns tile rows 2r and 2r+1 (blocks i = 0, 1), column slot r owns columns TN*r .. TN*r+TN-1:
    // the values a lane needs per k-step are adjacent in LDS and come with ONE ds_read_b64 / ds_read_b128 per operand
    const int acol = wm * 64 + 2 * (lane & 31), bcol = wn * (32 * TN) + TN * (lane & 31), kh = lane >> 5;
    const int n = c1 - c0;
    if (loader) {
        if (n > 0) {
            issue();
            stage(smem);
            if (n > 1) issue();
        }
        __syncthreads();
        for (int c = 0; c < n; ++c) {
            if (c + 1 < n) {
                stage(smem + ((c + 1) & 1) * TILE);
                if (c + 2 < n) issue();
            }
            __syncthreads();
        }
    } else {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();
        for (int c = 0; c < n; ++c) {
            const float* As = smem + (c & 1) * TILE;
            const float* Bs = As + WG_BKP * BM;
#pragma unroll
            for (int kk = 0; kk < WG_BKP / 2; ++kk) {
                const float2 a2 = *reinterpret_cast<const float2*>(&As[(2 * kk + kh) * BM + acol]);
                const float af[TM] = {a2.x, a2.y};
                float bf[TN];
                if constexpr (BIG) {
                    const float4 b4 = uda_ld4(&Bs[(2 * kk + kh) * BN + bcol]);
                    bf[0] = b4.x; bf[1] = b4.y; bf[2] = b4.z; bf[3] = b4.w;
                } else {
                    const float2 b2 = *reinterpret_cast<const float2*>(&Bs[(2 * kk + kh) * BN + bcol]);
                    bf[0] = b2.x; bf[1] = b2.y;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
        float* slab = a.slab + (int64_t)split * a.Cout * a.Jtot;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = jt * BN + wn * (32 * TN) + TN * (lane & 31) + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = cot * BM + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + i;
                    if (row < a.Cout && col < a.Jtot) slab[(int64_t)row * a.Jtot + col] = acc[i][j][r];
                }
            }
    }
}

template <int KS, int XF, bool BIG>
static int launch_wg(WgradKArgs& k, int S, hipStream_t st) {
    constexpr size_t lds = 2 * WG_BKP * (BIG ? 512 : 256) * sizeof(float);
    static bool configured_dev[UDA_MAX_DEVICES] = {};       // hipFuncSetAttribute is per device
    bool& configured = configured_dev[uda_device_slot()];
    auto fn = igemm_wgrad_ws_kernel<KS, XF, BIG>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return uda_set_error("igemm_wgrad_ws: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        configured = true;
    }
    hipLaunchKernelGGL(fn, dim3(k.nCot * k.nJt, S), dim3(BIG ? 768 : 512), lds, st, k);
    UDA_LAUNCH_CHECK("igemm_wgrad_ws");
    return 0;
}

template <int KS, bool BIG>
static int launch_wg_xf(WgradKArgs& k, int S, int xf, hipStream_t st) {
    return xf == 2 ? launch_wg<KS, 2, BIG>(k, S, st) : xf == 1 ? launch_wg<KS, 1, BIG>(k, S, st) : launch_wg<KS, 0, BIG>(k, S, st);
}

int launch_wgrad_ws(WgradKArgs& k, int S, bool big, hipStream_t st) {
    const int64_t lim = (int64_t)1 << 31, P = (int64_t)k.src.N * k.src.H * k.src.W;
    UDA_REQUIRE((P + 64) * k.src.ldx < lim / 4 && (P + 64) * k.lddy < lim / 4 && (P + 64) * (k.src.mask ? k.src.ldm : 1) < lim,
                "uda_conv_wgrad: operand too large for the 32-bit byte offsets of the wide-tile kernel (P * ld must stay below 2^29 elements)");
    const int xf = k.src.mask ? 2 : ((k.src.scale || k.src.act != ACT_NONE) ? 1 : 0);
    if (k.ksize >= 2) return big ? launch_wg_xf<3, true>(k, S, xf, st) : launch_wg_xf<3, false>(k, S, xf, st);
    return big ? launch_wg_xf<1, true>(k, S, xf, st) : launch_wg_xf<1, false>(k, S, xf, st);
}
