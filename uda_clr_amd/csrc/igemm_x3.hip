// fp32 implicit-GEMM convolution on the BF16 matrix pipe by exact three-way operand splitting ("bf16x3").
//
// gfx950 has no TF32 and its fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  Every fp32 value splits
// EXACTLY into three bf16 pieces, a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = a - a1 - a2: 3 x 8 significand bits,
// the residuals are exact in fp32), and every product of two pieces is exact in the fp32 accumulator of the bf16 MFMA.  Of the
// nine piece products of a * b the six with i + j <= 4 are kept:
//     a*b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1),      dropped terms <= 2^-24 |a b| (+ 2^-32): one fp32 rounding's worth,
// so a dot product carries the same ~sqrt(K) * 2^-24 error as the fp32 fma chain it replaces (tests/bench_x3.py and
// tests/noise_report.py measure both against float64).  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) do the work of eight
// v_mfma_f32_32x32x2_f32 (64 cycles each): 2.67x the matrix-pipe rate for the same fp32-level result.  Inputs, outputs,
// accumulators, BN statistics and the epilogue stay fp32 / fp64.
//
// Structure = igemm_ws.hip's (8 math waves + 4 loader waves, double-buffered LDS tile, one barrier per K-chunk, buffer-descriptor
// operand loads with hardware zero-fill for out-of-image taps, XCD-chunked tile order), with these differences:
//   * operands are split ONCE, by a packing pass (x3_pack_kernel: applies the pending BN / activation / dropout transform and
//     writes the three pieces), not per tap and per workgroup: the first version split in the loader and was loader-bound at
//     175-195 TF-equivalent; a 3x3 conv re-splits every activation nine times that way and every workgroup re-splits the weights;
//   * the loader waves therefore only copy: one buffer_load_b128 + one ds_write_b128 per 16 bytes, no VALU but addresses;
//   * a K-chunk is 16 deep (one bf16 MFMA k-step): tile = (BM + BN) rows x 112 B = 56 KiB for 256 x 256, two buffers;
//   * the loader keeps TWO chunks of global loads in flight (two register sets): with one, its loads were issued only a barrier
//     before they were needed and the kernel ran at the load latency (2.3 us per chunk against 1.5 us of MFMA work).
#include "common.h"
#include <stdlib.h>
#include "igemm_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define X3_BK 16
#define X3_ROW 56         // bf16 per staged row: 3 pieces x 16 + 8 pad = 112 B: a row's 96 B arrive as they lie in the packed operand
                          // (linear, conflict-free ds_write_b128), and the fragment reads of 16 lanes (rows r, byte r*112 + q*32 + h*16)
                          // fall on 64 distinct banks

// Packed operand ("x3 layout"): per row (pixel, or weight row) and per 16-wide K block the three bf16 pieces, 96 contiguous bytes:
//     element (row, k, piece q) at ((row * nb + k / 16) * 3 + q) * 16 + k % 16,   nb = K blocks per row.
// Activations: k = channel (nb = ceil(C / 16)).  Weights: k = position in the tap-chunked K order of the fp32 layout.

// (v0, v1) -> the three bf16 pieces of each, packed (v0 in the low half)
__device__ __forceinline__ void x3_split2(float v0, float v1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    bf16x2 a = {(__bf16)v0, (__bf16)v1};                       // v_cvt_pk_bf16_f32 (round to nearest even)
    p1 = __builtin_bit_cast(uint32_t, a);
    const float r0 = v0 - __builtin_bit_cast(float, p1 << 16), r1 = v1 - __builtin_bit_cast(float, p1 & 0xffff0000u);   // exact
    bf16x2 b = {(__bf16)r0, (__bf16)r1};
    p2 = __builtin_bit_cast(uint32_t, b);
    const float s0 = r0 - __builtin_bit_cast(float, p2 << 16), s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);   // exact
    bf16x2 c = {(__bf16)s0, (__bf16)s1};                       // s has <= 8 significant bits left: exact
    p3 = __builtin_bit_cast(uint32_t, c);
}

// one thread: 8 consecutive k of one row -> 3 x 16 B; a workgroup's 256 octets are consecutive in the output ([row][block] order,
// 48 B each), so the 12 KB it produces are staged in LDS in output order and stored as three fully coalesced 4 KB rows.
// XF as in the conv kernels (0 raw, 1 BN + act, 2 + keep-mask).
struct X3PackArgs {
    uda_src_t src;        // activations [P, ldx] with the pending transform; for weight rows: x = rows, C = Ktot, no transform
    int64_t P;
    int nb;               // 16-wide blocks per row
    uint32_t* out;        // packed, [P][nb][3][16] bf16
    // XF 3 / 4: the operand is a space-to-depth image of the patch discriminators (gan_ops.hip) that is never written in fp32:
    //   XF 3 (forward):  row (n, i, j) of the z grid, channel q = (a, b, c):  lrelu(g[n, 2i + a - 2, 2j + b - 2, c]), 0 outside
    //                    the valid vh x vw region of the previous layer's output g (rows [N, Hs, Ws, Cs]);
    //   XF 4 (backward): row (n, h, w) of [N, Hs, Ws, Cs], channel c: the routed gradient dz[z slot of (h, w)][(a, b, c)] times
    //                    the LeakyReLU gate read from the sign of gate[n, h, w, c] (the forward's source), 0 outside vh x vw.
    // src.C = channels of the packed operand (4 Cs / Cs), P its rows.
    const float* g;
    int64_t g_ld;
    const float* gate;
    int64_t gate_ld;
    int Hs, Ws, Cs, vh, vw, Hz, Wz;
    float slope;
};

template <int XF>
__global__ __launch_bounds__(256) void x3_pack_kernel(X3PackArgs a) {
    __shared__ uint4 stg[768];
    const int C = a.src.C, no = a.nb * 2, tid = threadIdx.x;     // octets per row
    const int64_t total = a.P * no, ntiles = (total + 255) >> 8;
    const float alo = a.src.act == ACT_NONE ? -INFINITY : 0.f, ahi = a.src.act == ACT_RELU6 ? 6.f : INFINITY;
    const bool coef16 = ((reinterpret_cast<uintptr_t>(a.src.scale) | reinterpret_cast<uintptr_t>(a.src.shift)) & 15) == 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t e0 = tile << 8, p0 = e0 / no;
        const int oo = (int)(e0 - p0 * no) + tid, dp = oo / no, o = oo - dp * no;
        const int64_t p = p0 + dp;
        const int c0 = o * 8;
        uint32_t q1[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0}, q3[4] = {0, 0, 0, 0};
        if (XF >= 3 && e0 + tid < total) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (c0 < C) {                                        // (Cs % 8 == 0: an octet never straddles two (a, b) planes)
                if (XF == 3) {
                    const int j = (int)(p % a.Wz), i = (int)((p / a.Wz) % a.Hz);
                    const int64_t n = p / ((int64_t)a.Wz * a.Hz);
                    const int ab = c0 / a.Cs, c = c0 - ab * a.Cs;
                    const int hh = 2 * i + (ab >> 1) - 2, ww = 2 * j + (ab & 1) - 2;
                    if (hh >= 0 && hh < a.vh && ww >= 0 && ww < a.vw) {
                        const float* r = a.g + ((n * a.Hs + hh) * a.Ws + ww) * a.g_ld + c;
                        const float4 x0 = uda_ld4(r), x1 = uda_ld4(r + 4);
                        const float t[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = t[q] > 0.f ? t[q] : t[q] * a.slope;
                    }
                } else {
                    const int w = (int)(p % a.Ws), hh = (int)((p / a.Ws) % a.Hs);
                    const int64_t n = p / ((int64_t)a.Ws * a.Hs);
                    if (hh < a.vh && w < a.vw) {
                        const int64_t pz = (n * a.Hz + ((hh + 2) >> 1)) * a.Wz + ((w + 2) >> 1);
                        const int q0 = ((((hh + 2) & 1) << 1) | ((w + 2) & 1)) * a.Cs + c0;
                        const float* r = a.g + pz * a.g_ld + q0;
                        const float4 x0 = uda_ld4(r), x1 = uda_ld4(r + 4);
                        float t[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        if (a.gate) {
                            const float* gr = a.gate + p * a.gate_ld + c0;
                            const float4 g0 = uda_ld4(gr), g1 = uda_ld4(gr + 4);
                            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                            for (int q = 0; q < 8; ++q) if (!(gv[q] > 0.f)) t[q] *= a.slope;
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = t[q];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) x3_split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
        }
        if (XF < 3 && e0 + tid < total) {
            float v[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c0 + 4 * h;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < C) x = uda_ld4(a.src.x + p * a.src.ldx + c);          // rows hold round4(C) readable floats
                float t[4] = {x.x, x.y, x.z, x.w};
                if (XF >= 1) {
                    uint32_t mk = 0x01010101u;
                    if (XF == 2 && c < C) mk = *reinterpret_cast<const uint32_t*>(a.src.mask + p * a.src.ldm + c);
                    // per-channel coefficients: one 16-byte load each where the four channels exist and the arrays are aligned (every
                    // operand of the engine), scalar loads on a ragged last granule - the loads, not the bytes, set this kernel's rate
                    float scv[4] = {1.f, 1.f, 1.f, 1.f}, shv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.src.scale) {
                        if (c + 4 <= C && coef16) {
                            const float4 q = uda_ld4(a.src.scale + c);
                            scv[0] = q.x; scv[1] = q.y; scv[2] = q.z; scv[3] = q.w;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) if (c + j < C) scv[j] = a.src.scale[c + j];
                        }
                    }
                    if (a.src.shift) {
                        if (c + 4 <= C && coef16) {
                            const float4 q = uda_ld4(a.src.shift + c);
                            shv[0] = q.x; shv[1] = q.y; shv[2] = q.z; shv[3] = q.w;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) if (c + j < C) shv[j] = a.src.shift[c + j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sc = scv[j], sh = shv[j];
                        float u = __builtin_amdgcn_fmed3f(t[j] * sc + sh, alo, ahi);
                        if (XF == 2) u *= ((mk >> (8 * j)) & 0xffu) ? a.src.mask_scale : 0.f;
                        t[j] = u;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * h + j] = (c + j) < C ? t[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) x3_split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
        }
        const int sl = (tid >> 1) * 6 + (tid & 1);               // uint4 slots: block (tid >> 1) holds [piece][half]
        stg[sl] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        stg[sl + 2] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        stg[sl + 4] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
        __syncthreads();
        uint4* dst = reinterpret_cast<uint4*>(a.out) + e0 * 3;
        const int64_t left = (total - e0) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r * 256 + tid < left) dst[r * 256 + tid] = stg[r * 256 + tid];
        __syncthreads();
    }
}

struct X3KArgs {
    const uint32_t* xa;   // packed activations [P][nbA][3][16]
    const uint32_t* xw;   // packed weights [Cout][nchunks][3][16]
    int N, H, W, nbA;
    int stride, Ho, Wo;   // output grid: row (n, oh, ow) is centred on input pixel (n, oh * stride, ow * stride)
    int Cout, ksize, dil, cen, nchunks;
    const float* bias;
    const float* addend;
    int64_t ld_add;
    float* y;
    int64_t ldy;
    double* stats;
    int nMt, nNt;
    int debug;            // diagnostics (UDA_X3_DEBUG): bit0 skip the MFMAs, bit1 skip the loader's global loads, bit2 skip its LDS writes
    // tail launch (launch_x3): this grid covers the tiles [tile_off, tile_off + gridDim.x / ksplit), each by ksplit workgroups that
    // take consecutive ranges of the K chunks and write their fp32 partial tile to partial[blockIdx.x][BM][BN] (x3_tail_reduce_kernel
    // sums them into y); ksplit = 1 / partial = nullptr: the ordinary launch over tiles [0, ntiles_main)
    int tile_off, ksplit, ntiles_main;
    float* partial;
};

template <int KS, int TN, int BM, bool TAIL = false>      // TAIL: the K-split tail launch (fp32 partial tiles, see X3KArgs)
__global__ __launch_bounds__(768) void igemm_conv_x3_kernel(X3KArgs a) {
    constexpr int MW = 8, BN = 64 * TN, TM = 2;
    constexpr int WMM = BM / 64, WNN = MW / WMM, TNW = 2 * TN / WNN;
    static_assert(WMM * WNN == MW && (2 * TN) % WNN == 0, "math-wave grid must tile the workgroup tile");
    constexpr int NTHR = (MW + 4) * 64;
    constexpr int A_U = BM * 6 / 256, B_UNITS = BN * 6, B_U = (B_UNITS + 255) / 256;   // 16-byte units (row, piece, octet) per loader thread and chunk
    static_assert((BM * 6) % 256 == 0, "A units must divide over the 256 loader threads");
    constexpr bool B_FULL = B_UNITS % 256 == 0;                  // 64-column tiles: the last B unit only on the first 128 loader threads
    constexpr int TILE = (BM + BN) * X3_ROW;                     // bf16 elements per buffer
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = __builtin_amdgcn_readfirstlane(wave) >= MW;
    const int lid = TAIL ? a.tile_off + (int)blockIdx.x / a.ksplit : uda_xcd_remap(blockIdx.x, a.ntiles_main);
    const int mt = lid / a.nNt, nt = lid % a.nNt;
    const int H = a.H, W = a.W;
    const int64_t P = (int64_t)a.N * a.Ho * a.Wo;                // output rows (= input pixels at stride 1)
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int T = a.ksize * a.ksize;
    // K chunks of this workgroup: all of them, or the ks-th of ksplit consecutive ranges (tail launch)
    const int ks = TAIL ? (int)blockIdx.x % a.ksplit : 0;
    const int cper = TAIL ? (a.nchunks + a.ksplit - 1) / a.ksplit : a.nchunks, cbeg = ks * cper;
    const int nchunks = TAIL ? max(0, min(a.nchunks, cbeg + cper) - cbeg) : a.nchunks;

    constexpr int OOB = 0x7ffffff0;                              // byte offset beyond every descriptor: loads return 0, stores are dropped
    if (loader) {
    // ------------------------------------------------------------------ loader waves: state (global offsets in 16-byte units), then the
    // staging loop.  Everything of the loader lives inside this branch: state computed before it would stay live (spilled)
    // across the math waves' path.
    const int lt = (tid - MW * 64) & 255;
    // unit u = lt + 256 * i -> (row = u / 6, part = u % 6); its LDS byte offset row * 112 + part * 16 = 16 * (u + row) is recomputed
    // per use, the nine tap-validity bits of three units share one register: the loader's two register sets (96 VGPRs at
    // 256 x 256) leave little room beside them
    auto unit_row = [](int u) { return (u * 43691) >> 18; };    // u / 6 for u < 3072
    int aoff[A_U], boff[B_U];
    unsigned vmask3[(A_U + 2) / 3];
    uint4 areg0[A_U], breg0[B_U], areg1[A_U], breg1[B_U];        // two chunks in flight (two register sets, statically indexed)
    int t_cur = 0, blk = cbeg, half = 0, chunk = cbeg;           // KS == 1: one chunk per 16-channel block
    if (KS == 3 && TAIL) {                                       // chunk c = ((pair * 2 + half) * T + t), pair = 32-channel slice
        const int pair = cbeg / (2 * T), r = cbeg - pair * 2 * T;
        half = r / T;
        t_cur = r - half * T;
        blk = 2 * pair;
    }
    const int rowA16 = a.nbA * 6;                                // uint4 per packed activation row
#pragma unroll
        for (int i = 0; i < (A_U + 2) / 3; ++i) vmask3[i] = 0;
#pragma unroll
        for (int i = 0; i < A_U; ++i) {
            const int u = lt + 256 * i, row = unit_row(u), part = u - row * 6;
            const int64_t p = m0 + row;
            const bool ok = p < P;
            const int qo = ok ? (int)p : 0;
            const int pw = (qo % a.Wo) * a.stride, ph = ((qo / a.Wo) % a.Ho) * a.stride;
            aoff[i] = (((qo / (a.Wo * a.Ho)) * H + ph) * W + pw) * rowA16 + part;
            unsigned vm = 0;
            if (ok) {
                if (KS == 3) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int th = a.ksize == 3 ? t / 3 : t / 2, tw = a.ksize == 3 ? t % 3 : t % 2;
                        const int hh = ph + (th - a.cen) * a.dil, ww = pw + (tw - a.cen) * a.dil;
                        vm |= (t < T && hh >= 0 && hh < H && ww >= 0 && ww < W ? 1u : 0u) << t;
                    }
                } else {
                    vm = 1u;
                }
            }
            vmask3[i / 3] |= vm << (9 * (i % 3));
        }
#pragma unroll
        for (int i = 0; i < B_U; ++i) {
            const int u = min(lt + 256 * i, B_UNITS - 1), row = unit_row(u), part = u - row * 6;
            const int n = min(n0 + row, a.Cout - 1);
            boff[i] = n * (a.nchunks * 6) + part;
        }
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(a.xa), 0, (int)min((int64_t)0x7fffffff, (int64_t)a.N * H * W * rowA16 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(a.xw), 0, (int)min((int64_t)0x7fffffff, (int64_t)a.Cout * a.nchunks * 96), 0x00020000);
    auto issue = [&](uint4 (&ar)[A_U], uint4 (&br)[B_U]) {
        const int t = t_cur, b16 = blk + half;
        const bool kval = b16 < a.nbA;
        int tapoff = 0;
        if (KS == 3) {
            const int th = a.ksize == 3 ? t / 3 : t >> 1;
            tapoff = ((th - a.cen) * W + (t - th * a.ksize - a.cen)) * a.dil;
        }
        const int xoff = tapoff * rowA16 + b16 * 6;
#pragma unroll
        for (int i = 0; i < A_U; ++i) {
            const bool ok = kval && ((vmask3[i / 3] >> (9 * (i % 3) + (KS == 3 ? t : 0))) & 1u);
            ar[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? (aoff[i] + xoff) * 16 : OOB, 0, 0));
        }
        const int woff = (KS == 3 ? (((blk >> 1) * T + t) * 2 + half) : chunk) * 6;
#pragma unroll
        for (int i = 0; i < B_U; ++i)
            br[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wres, (boff[i] + woff) * 16, 0, 0));
        ++chunk;
        if (KS == 3) {
            if (++t_cur == T) {             // all taps of one 16-channel block (its pixels stay in L2 for the T re-reads), then the next
                t_cur = 0;
                if ((half ^= 1) == 0) blk += 2;
            }
        } else {
            blk += 1;
        }
    };

    auto stage = [&](const uint4 (&ar)[A_U], const uint4 (&br)[B_U], __bf16* buf) {
        char* As = reinterpret_cast<char*>(buf);                       // [BM][X3_ROW]
        char* Bs = As + BM * X3_ROW * 2;                               // [BN][X3_ROW]
#pragma unroll
        for (int i = 0; i < A_U; ++i) *reinterpret_cast<uint4*>(As + 16 * (lt + 256 * i + unit_row(lt + 256 * i))) = ar[i];
#pragma unroll
        for (int i = 0; i < B_U; ++i)
            if (B_FULL || lt + 256 * i < B_UNITS) *reinterpret_cast<uint4*>(Bs + 16 * (lt + 256 * i + unit_row(lt + 256 * i))) = br[i];
    };

        // chunk c lives in register set c & 1 and in LDS buffer c & 1; chunks c + 1 and c + 2 are in flight while chunk c is computed
        if (nchunks > 0) {
            issue(areg0, breg0);
            if (nchunks > 1) issue(areg1, breg1);
            stage(areg0, breg0, smem16);
            if (nchunks > 2) issue(areg0, breg0);
        }
        __syncthreads();
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 1 < nchunks) {
                if (!(a.debug & 4)) stage(areg1, breg1, smem16 + TILE);
                if (c + 3 < nchunks && !(a.debug & 2)) issue(areg1, breg1);
            }
            __syncthreads();
            if (c + 1 < nchunks) {
                if (c + 2 < nchunks) {
                    if (!(a.debug & 4)) stage(areg0, breg0, smem16);
                    if (c + 4 < nchunks && !(a.debug & 2)) issue(areg0, breg0);
                }
                __syncthreads();
            }
        }
    } else {
        // -------------------------------------------------------------- math waves
        const int wm = (wave / WNN) % WMM, wn = wave % WNN;
        const int arow = wm * 64 + (lane & 31), brow = wn * (32 * TNW) + (lane & 31);
        const int koff = 8 * (lane >> 5);
        f32x16 acc[TM][TNW];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);
        for (int c = 0; c < nchunks; ++c) {
            if (a.debug & 1) {
                __syncthreads();
                continue;
            }
            const __bf16* As = smem16 + (c & 1) * TILE;
            const __bf16* Bs = As + BM * X3_ROW;
            // one A row-block at a time (12 fragment registers beside the 16 * TM * TNW accumulators: the 256 x 256 tile has 128
            // of them and 3 waves per SIMD leave 168 registers per wave); B fragments are re-read per row-block (LDS reads: 0.3
            // ds_read_b128 per MFMA, far below the 2 per MFMA gap the LDS sustains)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                bf16x8 af[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) af[q] = *reinterpret_cast<const bf16x8*>(&As[(arow + 32 * i) * X3_ROW + q * 16 + koff]);
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    // keep the scheduler from hoisting every block's fragment reads above the MFMAs (it would hold 3 * TNW fragments
                    // live and spill); the exposed LDS latency of one block is covered by the SIMD's other math wave
                    if (BM == 256) asm volatile("" ::: "memory");
                    bf16x8 bq[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const bf16x8*>(&Bs[(brow + 32 * j) * X3_ROW + q * 16 + koff]);
                    // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bq[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bq[1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bq[2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bq[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bq[1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bq[0], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Stores (and the addend's loads) go through
        // buffer descriptors over this tile's rows: an element beyond the matrix (row or column) gets an out-of-range offset and
        // is dropped by the hardware - no per-element branches, one offset register per lane
        // (the per-element 64-bit addresses of plain stores cost ~20 spilled registers per math wave and their scratch traffic)
        if constexpr (TAIL) {          // tail launch: the dense fp32 partial tile of this K range (bias, addend and bounds are the reduce kernel's)
            const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(
                a.partial + (int64_t)blockIdx.x * (BM * BN), 0, BM * BN * 4, 0x00020000);
            const int pv = ((wm * 64 + 4 * (lane >> 5)) * BN + wn * (32 * TNW) + (lane & 31)) * 4;
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), pres, pv + 128 * j,
                                                              (32 * i + (r & 3) + 8 * (r >> 2)) * (BN * 4), 0);
                    }
            return;
        }
        if constexpr (TAIL) return;    // (unreachable; keeps the ordinary epilogue out of the tail instantiation)
        float s1[TNW], s2[TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) s1[j] = s2[j] = 0.f;
        const int rows_left = (int)min((int64_t)BM, P - m0);
        const int ldy4 = (int)a.ldy * 4, lda4 = (int)a.ld_add * 4;
        const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
            a.y + m0 * a.ldy, 0, ((rows_left - 1) * (int)a.ldy + a.Cout) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t adres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.addend ? a.addend + m0 * a.ld_add : a.y), 0, a.addend ? ((rows_left - 1) * (int)a.ld_add + a.Cout) * 4 : 0, 0x00020000);
        const int colb = n0 + wn * (32 * TNW) + (lane & 31);
        const int rbase = wm * 64 + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int col = colb + 32 * j;
            const bool cok = col < a.Cout;
            const float bv = (cok && a.bias) ? a.bias[col] : 0.f;
            const int voff = cok ? rbase * ldy4 + col * 4 : OOB, vadd = cok ? rbase * lda4 + col * 4 : OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * i + (r & 3) + 8 * (r >> 2);          // row inside the wave's 64-row band (compile-time)
                    float v = acc[i][j][r] + bv;
                    const bool ok = cok && (rbase + rl) < rows_left;
                    s1[j] += ok ? v : 0.f;
                    s2[j] += ok ? v * v : 0.f;
                    // (the per-lane offset itself goes out of range for a row beyond the matrix: the descriptor's range check is
                    // not relied upon to include the scalar row offset)
                    if (a.addend) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(adres, ok ? vadd : OOB, rl * lda4, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yres, ok ? voff : OOB, rl * ldy4, 0);
                }
            }
        }
        if (a.stats) {       // (every wave is past the last chunk's barrier: the staging buffers are free)
            float* red = reinterpret_cast<float*>(smem16);   // [WMM][2][BN]
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
    }
    if (!TAIL && a.stats) {  // uniform over the workgroup
        float* red = reinterpret_cast<float*>(smem16);   // [WMM][2][BN]
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

// Sum of the tail launch's partial tiles into y (+ bias, + addend), fixed order: bitwise reproducible.
struct X3TailArgs {
    const float* partial;     // [ntail * ksplit][BM][BN]
    int ksplit, tile_off, ntail, nNt, BM, BN, Cout;
    int64_t P;
    const float* bias;
    const float* addend;
    int64_t ld_add;
    float* y;
    int64_t ldy;
};

__global__ __launch_bounds__(256) void x3_tail_reduce_kernel(X3TailArgs a) {
    const int bn4 = a.BN >> 2, per_tile = a.BM * bn4;
    const int64_t total = (int64_t)a.ntail * per_tile;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int tl = (int)(e / per_tile), q = (int)(e - (int64_t)tl * per_tile), r = q / bn4, c4 = (q - r * bn4) * 4;
        const int lid = a.tile_off + tl, mt = lid / a.nNt, nt = lid - mt * a.nNt;
        const int64_t row = (int64_t)mt * a.BM + r;
        const int col = nt * a.BN + c4;
        if (row >= a.P || col >= a.Cout) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < a.ksplit; ++s) {
            const float4 v = uda_ld4(a.partial + (((int64_t)tl * a.ksplit + s) * a.BM + r) * a.BN + c4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            if (col + jj < a.Cout) {
                float v = o[jj] + (a.bias ? a.bias[col + jj] : 0.f);
                if (a.addend) v += a.addend[row * a.ld_add + col + jj];
                a.y[row * a.ldy + col + jj] = v;
            }
    }
}

// Tail plan of a tile choice: the last, partly filled round of tiles is computed by ksplit workgroups per tile over consecutive K
// ranges (x3_tail_reduce_kernel sums the partial tiles) when the conv has no statistics epilogue and the caller gave a workspace.
struct X3Tail {
    int64_t tiles, full, tail;
    int ksplit;
};

static X3Tail x3_tail_plan(int64_t P, int Cout, int nchunks, int BM, int BN, bool allow) {
    X3Tail t;
    t.tiles = uda_cdiv(P, BM) * uda_cdiv(Cout, BN);
    t.full = (t.tiles / 256) * 256;
    t.tail = t.tiles - t.full;
    t.ksplit = 1;
    // only where it clearly pays: a long K (the fp32 partial tiles are extra traffic - one write and one read per split - and on the
    // short-K layers the split bought 2-3 %) and at least three splits (measured: discriminator L3 / L4 forward 10 % / 16 %)
    static const int mode = getenv("UDA_X3_TAIL_MODE") ? atoi(getenv("UDA_X3_TAIL_MODE")) : 1;      // experiment: 2 = every tail <= 128 tiles
    if (allow && t.tail > 0 && (mode == 2 ? t.tail <= 128 : (t.tail <= 85 && nchunks >= 96))) {
        int s = (int)(256 / t.tail);
        if (s > 8) s = 8;
        if (s > nchunks / 8) s = nchunks / 8;       // at least 8 chunks per workgroup
        if (s >= (mode == 2 ? 2 : 3)) t.ksplit = s;
    }
    return t;
}

static uint64_t x3_tail_bytes(const X3Tail& t, int BM, int BN) {
    return t.ksplit > 1 ? (uint64_t)t.tail * t.ksplit * BM * BN * sizeof(float) : 0;
}

template <int KS, int TN, int BM>
static int launch_x3(X3KArgs& k, int64_t P, hipStream_t st, void* ws = nullptr, uint64_t ws_bytes = 0) {
    constexpr int BN = 64 * TN;
    constexpr size_t lds = 2 * (BM + BN) * X3_ROW * sizeof(__bf16);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool configured_dev[UDA_MAX_DEVICES] = {};       // hipFuncSetAttribute is per device
    bool& configured = configured_dev[uda_device_slot()];
    auto fn = igemm_conv_x3_kernel<KS, TN, BM>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return uda_set_error("igemm_conv_x3: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        configured = true;
    }
    k.nMt = uda_cdiv(P, BM);
    k.nNt = uda_cdiv(k.Cout, BN);
    #ifdef UDA_DIAG          // diagnostic builds only (make DIAG=1): bit0 skips the MFMAs - results are WRONG with it
    static const int dbg = getenv("UDA_X3_DEBUG") ? atoi(getenv("UDA_X3_DEBUG")) : 0;
#else
    const int dbg = 0;
#endif
    k.debug = dbg;
    X3Tail t = x3_tail_plan(P, k.Cout, k.nchunks, BM, BN, k.stats == nullptr && ws != nullptr);
    if (x3_tail_bytes(t, BM, BN) > ws_bytes) t.ksplit = 1;
    k.tile_off = 0; k.ksplit = 1; k.partial = nullptr;
    k.ntiles_main = (int)(t.ksplit > 1 ? t.full : t.tiles);
    if (k.ntiles_main > 0) {
        hipLaunchKernelGGL(fn, dim3(k.ntiles_main), dim3(768), lds, st, k);
        UDA_LAUNCH_CHECK("igemm_conv_x3");
    }
    if (t.ksplit > 1) {
        k.tile_off = (int)t.full; k.ksplit = t.ksplit; k.partial = reinterpret_cast<float*>(ws);
        static bool configured_tail_dev[UDA_MAX_DEVICES] = {};
        bool& configured_tail = configured_tail_dev[uda_device_slot()];
        auto fnt = igemm_conv_x3_kernel<KS, TN, BM, true>;
        if (!configured_tail) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fnt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return uda_set_error("igemm_conv_x3 (tail): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            configured_tail = true;
        }
        hipLaunchKernelGGL(fnt, dim3((int)t.tail * t.ksplit), dim3(768), lds, st, k);
        UDA_LAUNCH_CHECK("igemm_conv_x3 (tail)");
        X3TailArgs r;
        r.partial = k.partial; r.ksplit = t.ksplit; r.tile_off = (int)t.full; r.ntail = (int)t.tail; r.nNt = k.nNt; r.BM = BM; r.BN = BN;
        r.Cout = k.Cout; r.P = P; r.bias = k.bias; r.addend = k.addend; r.ld_add = k.ld_add; r.y = k.y; r.ldy = k.ldy;
        const int64_t work = (int64_t)t.tail * BM * (BN / 4);
        hipLaunchKernelGGL(x3_tail_reduce_kernel, dim3((int)uda_cdiv(work, 256)), dim3(256), 0, st, r);
        UDA_LAUNCH_CHECK("x3_tail_reduce");
    }
    return 0;
}

// Eligibility: the tap-chunked K order (>= 32 channels per tap), and enough MFMA work per packed element to pay for the packing
// pass (2 * Cout * taps FLOPs per activation element).  Measured (tests/bench_x3.py, profiles/r02_bf16x3_vs_f32_conv_microbench.txt):
// 3x3 / 2x2 convs with >= 128 outputs run 1.5-2.0x faster incl. the pass; a 3x3 conv towards 48 channels over K = 2304 1.9x on
// the 256 x 64 tile; 1x1 convs only when very wide (256 -> 2304 tap GEMM of the re-associated decoder conv: 1.6-1.8x; 1280 -> 256
// at 16384 pixels: 1.0x, stays on the fp32 pipe).
bool conv_x3_eligible(const ConvKArgs& k) {
    // 1x1: very wide outputs (the 256 -> 2304 tap GEMM), or a long K towards >= 256 outputs (ResNet-101's bottleneck convs 1024 -> 256 and
    // 2048 -> 512 on the 32x32 maps: their operands are block inputs / gradients that the weight gradient packs anyway).
    // (measured with Cout >= 128 on MobileNetV2's 1x1 convs: they gain nothing, the packing pass eats it)
    static const int long_k = getenv("UDA_X3_LONGK_1X1") ? atoi(getenv("UDA_X3_LONGK_1X1")) : 1;
    if (k.ksize == 1) return k.Kc >= 128 && (k.Cout >= 1024 || (long_k && k.Kc >= 1024 && k.Cout >= 256));
    return k.Kc >= IG_BK && k.Cout * k.ksize * k.ksize >= 432;
}

static inline int x3_nb(int C) { return uda_cdiv(C, X3_BK); }

// bytes of a packed operand with `rows` rows of K values
extern "C" uint64_t uda_x3_packed_bytes(int64_t rows, int K) { return (uint64_t)rows * x3_nb(K) * 96 + 64; }

/* Pack (split) an operand for the bf16x3 kernels: out[rows][ceil(C/16)][3][16] bf16 of the TRANSFORMED values
 * act(x * scale + shift) * mask * mask_scale (rows = N*H*W).  Weight rows: a src with N = H = 1, W = rows, C = K, no transform. */
extern "C" int uda_x3_pack(const uda_src_t* src, void* out, void* stream) {
    UDA_REQUIRE(src && src->x && out && uda_aligned16(out) && uda_aligned16(src->x) && src->ldx % 4 == 0 && src->C > 0,
                "uda_x3_pack: bad args (16-byte aligned rows)");
    hipStream_t st = (hipStream_t)stream;
    X3PackArgs pa = {};
    pa.src = *src; pa.P = (int64_t)src->N * src->H * src->W; pa.nb = x3_nb(src->C); pa.out = reinterpret_cast<uint32_t*>(out);
    UDA_REQUIRE(pa.P > 0, "uda_x3_pack: empty operand");
    const int64_t tot = pa.P * pa.nb * 2;
    const int grid = (int)(uda_cdiv(tot, 256) > 65536 ? 65536 : uda_cdiv(tot, 256));
    if (src->mask) hipLaunchKernelGGL(x3_pack_kernel<2>, dim3(grid), dim3(256), 0, st, pa);
    else if (src->scale || src->act != ACT_NONE) hipLaunchKernelGGL(x3_pack_kernel<1>, dim3(grid), dim3(256), 0, st, pa);
    else hipLaunchKernelGGL(x3_pack_kernel<0>, dim3(grid), dim3(256), 0, st, pa);
    UDA_LAUNCH_CHECK("x3_pack");
    return 0;
}

/* The packed space-to-depth image of a discriminator layer's input, straight from the previous layer's output (rows
 * [N, Hs, Ws, C], valid region valid_h x valid_w, LeakyReLU(slope) fused): what uda_s2d_fwd + uda_x3_pack would produce, without
 * the fp32 image in between.  out: [N*Hz*Wz][4C/16][3][16] bf16 (uda_x3_packed_bytes(N*Hz*Wz, 4C)).  C % 8 == 0. */
extern "C" int uda_x3_pack_s2d_fwd(const float* src, int64_t ld_src, int N, int Hs, int Ws, int C, int valid_h, int valid_w, float slope,
                                   int Hz, int Wz, void* out, void* stream) {
    UDA_REQUIRE(src && out && uda_aligned16(src) && uda_aligned16(out) && ld_src % 4 == 0 && ld_src >= C && C > 0 && C % 8 == 0,
                "uda_x3_pack_s2d_fwd: bad args (C %% 8 == 0, 16-byte aligned rows)");
    UDA_REQUIRE(N > 0 && valid_h > 0 && valid_w > 0 && valid_h <= Hs && valid_w <= Ws && Hz == (valid_h + 5) / 2 && Wz == (valid_w + 5) / 2,
                "uda_x3_pack_s2d_fwd: bad geometry");
    X3PackArgs pa = {};
    pa.src.C = 4 * C; pa.P = (int64_t)N * Hz * Wz; pa.nb = x3_nb(4 * C); pa.out = reinterpret_cast<uint32_t*>(out);
    pa.g = src; pa.g_ld = ld_src; pa.gate = nullptr; pa.gate_ld = 0;
    pa.Hs = Hs; pa.Ws = Ws; pa.Cs = C; pa.vh = valid_h; pa.vw = valid_w; pa.Hz = Hz; pa.Wz = Wz; pa.slope = slope;
    const int64_t tot = pa.P * pa.nb * 2;
    const int grid = (int)(uda_cdiv(tot, 256) > 65536 ? 65536 : uda_cdiv(tot, 256));
    hipLaunchKernelGGL(x3_pack_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pa);
    UDA_LAUNCH_CHECK("x3_pack_s2d_fwd");
    return 0;
}

/* The packed gradient w.r.t. a discriminator layer's INPUT rows [N, Hs, Ws, C] from dz (rows on the z grid, 4C channels): what
 * uda_s2d_bwd + uda_x3_pack would produce.  gate_rows ([N, Hs, Ws, C], may be null): the forward's source, whose sign is the
 * LeakyReLU gate.  out: uda_x3_packed_bytes(N*Hs*Ws, C).  C % 8 == 0. */
extern "C" int uda_x3_pack_s2d_bwd(const float* dz, int64_t ld_z, int Hz, int Wz, const float* gate_rows, int64_t ld_gate, float slope,
                                   int N, int Hs, int Ws, int C, int valid_h, int valid_w, void* out, void* stream) {
    UDA_REQUIRE(dz && out && uda_aligned16(dz) && uda_aligned16(out) && ld_z % 4 == 0 && ld_z >= 4 * C && C > 0 && C % 8 == 0,
                "uda_x3_pack_s2d_bwd: bad args (C %% 8 == 0, 16-byte aligned rows)");
    UDA_REQUIRE(!gate_rows || (uda_aligned16(gate_rows) && ld_gate % 4 == 0 && ld_gate >= C), "uda_x3_pack_s2d_bwd: bad gate rows");
    UDA_REQUIRE(N > 0 && valid_h > 0 && valid_w > 0 && valid_h <= Hs && valid_w <= Ws && Hz == (valid_h + 5) / 2 && Wz == (valid_w + 5) / 2,
                "uda_x3_pack_s2d_bwd: bad geometry");
    X3PackArgs pa = {};
    pa.src.C = C; pa.P = (int64_t)N * Hs * Ws; pa.nb = x3_nb(C); pa.out = reinterpret_cast<uint32_t*>(out);
    pa.g = dz; pa.g_ld = ld_z; pa.gate = gate_rows; pa.gate_ld = ld_gate;
    pa.Hs = Hs; pa.Ws = Ws; pa.Cs = C; pa.vh = valid_h; pa.vw = valid_w; pa.Hz = Hz; pa.Wz = Wz; pa.slope = slope;
    const int64_t tot = pa.P * pa.nb * 2;
    const int grid = (int)(uda_cdiv(tot, 256) > 65536 ? 65536 : uda_cdiv(tot, 256));
    hipLaunchKernelGGL(x3_pack_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pa);
    UDA_LAUNCH_CHECK("x3_pack_s2d_bwd");
    return 0;
}

// Tile choice: the cheapest of 256 x 256, 128 x 256, 256 x 128, 128 x 128 under  rounds x tile area / tile efficiency, a round
// being one tile per CU - padding waste (Cout = 304 fits three 128-wide tiles better than two 256-wide ones) and the partly
// filled last round both count; with a workspace and no statistics epilogue that last round is split over K (x3_tail_plan) and
// costs 1 / ksplit of a round plus the reduce.  Efficiencies fitted to the discriminator layers (tests/bench_x3.py with
// UDA_X3_TILE forcing a tile; the tiles stage 32 / 48 / 48 / 64 B per MFMA clock and CU, two 128 x 128 workgroups can share a CU).
static int x3_pick_tile(int64_t P, int Cout, int nchunks, bool allow_tail) {
    static const int force = getenv("UDA_X3_TILE") ? atoi(getenv("UDA_X3_TILE")) : -1;
    if (force >= 0 && force < 4) return force;
    const int bm[4] = {256, 128, 256, 128}, bn[4] = {256, 256, 128, 128};
    const double eff[4] = {1.0, 0.97, 0.92, 0.84};
    int best = 0;
    double bestc = 1e300;
    for (int t = 0; t < 4; ++t) {
        const X3Tail tp = x3_tail_plan(P, Cout, nchunks, bm[t], bn[t], allow_tail);
        double rounds = (double)(tp.full / 256);
        if (tp.tail > 0) rounds += tp.ksplit > 1 ? 1.0 / tp.ksplit + 0.12 : 1.0;
        const double c = rounds * bm[t] * bn[t] / eff[t];
        if (c < bestc) { bestc = c; best = t; }
    }
    return best;
}

static const int X3_TILE_BM[4] = {256, 128, 256, 128}, X3_TILE_BN[4] = {256, 256, 128, 128};

// workspace the tail split of this conv wants (0: none)
uint64_t conv_x3_workspace_bytes(const ConvKArgs& k, int64_t P) {
    static const bool off = getenv("UDA_X3_NO_TAIL") != nullptr;
    if (off || k.stats || (k.ksize >= 2 && k.Cout <= 64)) return 0;
    const int nch = uda_cdiv(k.Ktot, X3_BK);
    const int t = x3_pick_tile(P, k.Cout, nch, true);
    return x3_tail_bytes(x3_tail_plan(P, k.Cout, nch, X3_TILE_BM[t], X3_TILE_BN[t], true), X3_TILE_BM[t], X3_TILE_BN[t]);
}

int launch_conv_x3(ConvKArgs& k, int64_t P, const void* x3_src, const void* x3_w, hipStream_t st, void* ws, uint64_t ws_bytes) {
    const int64_t lim = (int64_t)1 << 31;
    const int nbA = x3_nb(k.src.C), nch = uda_cdiv(k.Ktot, X3_BK);
    UDA_REQUIRE(x3_src && x3_w && uda_aligned16(x3_src) && uda_aligned16(x3_w),
                "uda_conv_fwd (bf16x3): the packed operands x3_src / x3_w are missing (uda_x3_pack; uda_conv_uses_x3 tells when they are needed)");
    const int64_t Pin = (int64_t)k.src.N * k.src.H * k.src.W;      // (P: output rows)
    UDA_REQUIRE((Pin + 256) * nbA * 6 < lim / 16 && (int64_t)(k.Cout + 320) * nch * 6 < lim / 16,
                "uda_conv_fwd (bf16x3): operand too large for the 32-bit offsets of the wide-tile kernel");
    X3KArgs x;
    x.xa = reinterpret_cast<const uint32_t*>(x3_src); x.xw = reinterpret_cast<const uint32_t*>(x3_w);
    x.N = k.src.N; x.H = k.src.H; x.W = k.src.W; x.nbA = nbA;
    x.stride = k.stride; x.Ho = k.Ho; x.Wo = k.Wo;
    x.Cout = k.Cout; x.ksize = k.ksize; x.dil = k.dil; x.cen = k.cen; x.nchunks = nch;
    x.bias = k.bias; x.addend = k.addend; x.ld_add = k.ld_add; x.y = k.y; x.ldy = k.ldy; x.stats = k.stats;
    if (k.ksize >= 2 && k.Cout <= 64) return launch_x3<3, 1, 256>(x, P, st);      // input gradient towards a narrow tensor (decoder low-level branch)
    const uint64_t want = conv_x3_workspace_bytes(k, P);
    const bool tail_ok = want > 0 && ws != nullptr && ws_bytes >= want && uda_aligned16(ws);
    if (!tail_ok) { ws = nullptr; ws_bytes = 0; }
    const int best = x3_pick_tile(P, k.Cout, nch, tail_ok);
    if (k.ksize >= 2) {
        switch (best) {
            case 0: return launch_x3<3, 4, 256>(x, P, st, ws, ws_bytes);
            case 1: return launch_x3<3, 4, 128>(x, P, st, ws, ws_bytes);
            case 2: return launch_x3<3, 2, 256>(x, P, st, ws, ws_bytes);
            default: return launch_x3<3, 2, 128>(x, P, st, ws, ws_bytes);
        }
    }
    switch (best) {
        case 0: return launch_x3<1, 4, 256>(x, P, st, ws, ws_bytes);
        case 1: return launch_x3<1, 4, 128>(x, P, st, ws, ws_bytes);
        case 2: return launch_x3<1, 2, 256>(x, P, st, ws, ws_bytes);
        default: return launch_x3<1, 2, 128>(x, P, st, ws, ws_bytes);
    }
}

// ==========================================================================================================================
// Weight gradient on the bf16 pipe:  dw[co][j] = sum_p dy[p][co] * u[p + off_t][ci],  j = (t, ci): contraction over PIXELS.
// Both operands are packed pixel-major ([pixel][channel block][3][16]), i.e. K-outer for this GEMM, so the MFMA fragments (8
// consecutive k = pixels of one channel per lane) are read TRANSPOSED from an LDS image [16 pixels][BM channels] per piece:
// ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block (verified on the device:
// tests/bin/tr_probe), two of them make one bf16x8 fragment.  Row stride BM*2 + 64 bytes (= 16 banks mod 64): the 4 rows x 2
// column blocks a 32-lane half reads fall on 64 distinct banks.
// Structure as the forward kernel: 8 math + 4 loader waves, 16-deep chunks (16 pixels), two chunks of loads in flight, one
// barrier per chunk; every split of the pixel range writes its own fp32 slab, wgrad_reduce_kernel sums them (igemm_conv.hip).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct X3WgArgs {
    const uint32_t* xdy;  // packed dy  [P][nbCo][3][16]
    const uint32_t* xs;   // packed (transformed) source [P][nbC][3][16]
    int N, H, W, nbCo, nbC;
    int stride, Ho, Wo;   // grid of dy: its pixel (n, oh, ow) pairs with source pixel (n, oh * stride, ow * stride)
    int Cout, Jtot, Kc, ksize, dil, cen;
    float* slab;          // [S][Cout][Jtot]
    int nCot, nJt, cps, nchunks;
};

template <int BM, int BN>
__global__ __launch_bounds__(768) void igemm_wgrad_x3_kernel(X3WgArgs a) {
    constexpr int MW = 8, NBA = BM / 16, NBB = BN / 16;
    constexpr int UA = NBA * 6 / 16, UB = NBB * 6 / 16; // 16-byte units per loader thread and chunk (its pixel: 16 threads x U)
    constexpr int RSA = BM * 2 + 64, RSB = BN * 2 + 64; // bytes per pixel row of one piece
    constexpr int PLA = 16 * RSA, PLB = 16 * RSB, OPA = 3 * PLA, TILE = OPA + 3 * PLB;      // bytes
    constexpr int TM = BM / 128, TN = BN / 64;          // 32 x 32 blocks per math wave: wave tile (BM/4) x (BN/2)
    extern __shared__ __attribute__((aligned(16))) char smemw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = __builtin_amdgcn_readfirstlane(wave) >= MW;
    const int cot = blockIdx.x / a.nJt, jt = blockIdx.x % a.nJt, split = blockIdx.y;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo, sd = a.stride;
    const int64_t P = (int64_t)a.N * Ho * Wo;                    // pixels of dy
    const int c0 = split * a.cps, c1 = min(a.nchunks, c0 + a.cps), n = c1 - c0;

    constexpr int OOB = 0x7ffffff0;
    if (loader) {
    // ------------------------------------------------------------------ loader waves (all of their state inside this branch, as in
    // the forward kernel)
    const int lt = (tid - MW * 64) & 255, ps = lt >> 4, li = lt & 15;
    int aoff[UA], boff[UB], btap[UB];                   // 16-byte unit offsets relative to the pixel row; packed (dh, dw) of the unit's tap
    unsigned aval = 0, bval = 0;                        // per-unit "block exists" bits
    uint4 ar0[UA], br0[UB], ar1[UA], br1[UB];
    int pcur = c0 * 16 + ps, hcur = 0, wcur = 0, scur = 0;      // dy pixel, its (oh, ow), the source pixel it pairs with
    const int rowDy = a.nbCo * 6, rowS = a.nbC * 6;
    const int rowstep = sd * (W - Wo), imgstep = (H - sd * Ho) * W;      // source-pixel corrections at a row / image wrap (0 at stride 1)
    {
        const int q = (int)(pcur < P ? pcur : 0);
        wcur = q % Wo;
        hcur = (q / Wo) % Ho;
        scur = ((q / (Wo * Ho)) * H + hcur * sd) * W + wcur * sd;
#pragma unroll
        for (int i = 0; i < UA; ++i) {
            const int e = li + 16 * i, b = e / 6, part = e - 6 * b;
            const int gb = cot * NBA + b;
            aoff[i] = gb * 6 + part;
            aval |= (gb < a.nbCo ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int e = li + 16 * i, b = e / 6, part = e - 6 * b;
            const int j0 = (jt * NBB + b) * 16;
            int t = 0, cib = j0 / 16;
            if (a.ksize >= 2) {
                t = j0 / a.Kc;
                cib = (j0 - t * a.Kc) / 16;
            }
            const int th = a.ksize == 3 ? t / 3 : t >> 1;
            const int dh = a.ksize >= 2 ? (th - a.cen) * a.dil : 0, dw = a.ksize >= 2 ? (t - th * a.ksize - a.cen) * a.dil : 0;
            boff[i] = (dh * W + dw) * rowS + cib * 6 + part;
            btap[i] = (dh << 16) | (dw & 0xffff);
            bval |= (j0 < a.Jtot ? 1u : 0u) << i;
        }
    }
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(a.xdy), 0, (int)min((int64_t)0x7fffffff, P * rowDy * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(a.xs), 0, (int)min((int64_t)0x7fffffff, (int64_t)a.N * H * W * rowS * 16), 0x00020000);
    auto issue = [&](uint4 (&ar)[UA], uint4 (&br)[UB]) {
        const bool pin = pcur < (int)P;
#pragma unroll
        for (int i = 0; i < UA; ++i) {
            const bool oka = pin && ((aval >> i) & 1u);
            ar[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yres, oka ? (pcur * rowDy + aoff[i]) * 16 : OOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int hh = hcur * sd + (btap[i] >> 16), ww = wcur * sd + (int)(short)(btap[i] & 0xffff);
            const bool okb = pin && ((bval >> i) & 1u) && hh >= 0 && hh < H && ww >= 0 && ww < W;
            br[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(sres, okb ? (scur * rowS + boff[i]) * 16 : OOB, 0, 0));
        }
        pcur += 16;                 // next chunk: 16 pixels on
        scur += 16 * sd;
        wcur += 16;
        while (wcur >= Wo) {
            wcur -= Wo;
            scur += rowstep;
            if (++hcur >= Ho) {
                hcur = 0;
                scur += imgstep;
            }
        }
    };
    auto stage = [&](const uint4 (&ar)[UA], const uint4 (&br)[UB], char* buf) {
#pragma unroll
        for (int i = 0; i < UA; ++i) {
            const int e = li + 16 * i, b = e / 6, part = e - 6 * b;
            *reinterpret_cast<uint4*>(buf + (part >> 1) * PLA + ps * RSA + (b * 16 + (part & 1) * 8) * 2) = ar[i];
        }
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int e = li + 16 * i, b = e / 6, part = e - 6 * b;
            *reinterpret_cast<uint4*>(buf + OPA + (part >> 1) * PLB + ps * RSB + (b * 16 + (part & 1) * 8) * 2) = br[i];
        }
    };

        if (n > 0) {
            issue(ar0, br0);
            if (n > 1) issue(ar1, br1);
            stage(ar0, br0, smemw);
            if (n > 2) issue(ar0, br0);
        }
        __syncthreads();
        for (int c = 0; c < n; c += 2) {
            if (c + 1 < n) {
                stage(ar1, br1, smemw + TILE);
                if (c + 3 < n) issue(ar1, br1);
            }
            __syncthreads();
            if (c + 1 < n) {
                if (c + 2 < n) {
                    stage(ar0, br0, smemw);
                    if (c + 4 < n) issue(ar0, br0);
                }
                __syncthreads();
            }
        }
    } else {
        const int wm = wave >> 1, wn = wave & 1;
        const int g = lane >> 4, l = lane & 15;
        // transposed-read address of this lane inside a piece plane, for the block at channel offset 0: rows 8*(g>>1) + (l>>2)
        // (+4 for the second half of the k-octet), columns 16*(g&1) + 4*(l&3)
        const int trr = 8 * (g >> 1) + (l >> 2), trc = (16 * (g & 1) + 4 * (l & 3)) * 2;
        const int abase = trr * RSA + trc + wm * (BM / 4) * 2, bbase = trr * RSB + trc + wn * (BN / 2) * 2;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        auto frag = [&](const char* plane, int off, int rs) -> bf16x8 {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off + 4 * rs));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);
        for (int c = 0; c < n; ++c) {
            const char* As = smemw + (c & 1) * TILE;
            const char* Bs = As + OPA;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                bf16x8 af[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) af[q] = frag(As + q * PLA, abase + 64 * i, RSA);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (BM * BN >= 256 * 256) asm volatile("" ::: "memory");
                    bf16x8 bq[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) bq[q] = frag(Bs + q * PLB, bbase + 64 * j, RSB);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bq[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bq[1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bq[2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bq[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bq[1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bq[0], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        __builtin_amdgcn_s_setprio(0);
        // slab tile through a buffer descriptor: rows beyond Cout / columns beyond Jtot get an out-of-range offset (dropped)
        float* slab = a.slab + ((int64_t)split * a.Cout + (int64_t)cot * BM) * a.Jtot;
        const int rows_left = min(BM, a.Cout - cot * BM), J4 = a.Jtot * 4;
        const __amdgpu_buffer_rsrc_t sres2 = __builtin_amdgcn_make_buffer_rsrc(slab, 0, rows_left * J4, 0x00020000);
        const int rbase = wm * (BM / 4) + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = jt * BN + wn * (BN / 2) + 32 * j + (lane & 31);
            const int voff = col < a.Jtot ? rbase * J4 + col * 4 : OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * i + (r & 3) + 8 * (r >> 2);
                    const float v = acc[i][j][r];      // (a __builtin_bit_cast applied to the vector element itself reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), sres2,
                                                          (rbase + rl) < rows_left ? voff : OOB, rl * J4, 0);
                }
        }
    }
}

// Eligibility of the weight gradient: 16-wide channel blocks must not straddle taps (Kc % 16 == 0), enough work to pay for the
// packing of dy (the source's packed form usually exists already from the forward conv).
bool wgrad_x3_eligible(int Cin, int Cout, int ksize, int64_t P) {
    static const int long_k = getenv("UDA_X3_LONGK_1X1") ? atoi(getenv("UDA_X3_LONGK_1X1")) : 1;
    if (ksize == 1) return Cin % 16 == 0 && P >= 4096 && ((Cin >= 128 && Cout >= 1024) || (long_k && Cin >= 1024 && Cout >= 256));
    return Cin % 16 == 0 && Cin >= 32 && Cout >= 96 && P >= 4096;
}

int launch_wgrad_x3(const WgradKArgs& k, int64_t P, int S_max, const void* x3_src, const void* x3_dy, hipStream_t st, int& S_out) {
    UDA_REQUIRE(x3_src && x3_dy && uda_aligned16(x3_src) && uda_aligned16(x3_dy),
                "uda_conv_wgrad (bf16x3): the packed operands x3_src / x3_dy are missing (uda_x3_pack; uda_conv_wgrad_uses_x3)");
    const int64_t lim = (int64_t)1 << 31;
    const int nbCo = x3_nb(k.Cout), nbC = x3_nb(k.src.C);
    const int64_t Pin = (int64_t)k.src.N * k.src.H * k.src.W;      // (P: pixels of dy)
    UDA_REQUIRE((P + 64) * nbCo * 6 < lim / 16 && (Pin + 64 + 4 * k.src.W * k.dil) * nbC * 6 < lim / 16,
                "uda_conv_wgrad (bf16x3): operand too large for the 32-bit offsets of the wide-tile kernel");
    // tiles (Cout x J): 256 x 256 for wide outputs; 128 x 256 otherwise (128 x 128 tiles would stage 64 B per MFMA clock and CU and
    // are load-bound); 128 x 128 only for short J
    const bool big = k.Cout >= 192 && k.Jtot >= 256 && P >= 8192;
    const bool wideJ = !big && k.Jtot >= 256;
    const int BM = big ? 256 : 128, BN = (big || wideJ) ? 256 : 128;
    X3WgArgs x;
    x.xdy = reinterpret_cast<const uint32_t*>(x3_dy); x.xs = reinterpret_cast<const uint32_t*>(x3_src);
    x.N = k.src.N; x.H = k.src.H; x.W = k.src.W; x.nbCo = nbCo; x.nbC = nbC;
    x.stride = k.stride; x.Ho = k.Ho; x.Wo = k.Wo;
    x.Cout = k.Cout; x.Jtot = k.Jtot; x.Kc = k.Kc; x.ksize = k.ksize; x.dil = k.dil; x.cen = k.cen;
    x.slab = k.slab;
    x.nCot = uda_cdiv(k.Cout, BM); x.nJt = uda_cdiv(k.Jtot, BN);
    x.nchunks = uda_cdiv(P, X3_BK);
    int S = (big ? 512 : (wideJ ? 768 : 1024)) / (x.nCot * x.nJt);
    if (S > x.nchunks / 8) S = x.nchunks / 8;
    if (S > S_max) S = S_max;           // the caller's slab holds S_max splits
    if (S < 1) S = 1;
    x.cps = uda_cdiv(x.nchunks, S);
    S = uda_cdiv(x.nchunks, x.cps);
    S_out = S;
    static bool configured_dev[UDA_MAX_DEVICES] = {};       // hipFuncSetAttribute is per device
    bool& configured = configured_dev[uda_device_slot()];
    auto ldsz = [](int bm, int bn) { return (size_t)2 * 3 * 16 * ((bm * 2 + 64) + (bn * 2 + 64)); };
    if (!configured) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_wgrad_x3_kernel<256, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz(256, 256));
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_wgrad_x3_kernel<128, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz(128, 256));
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_wgrad_x3_kernel<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz(128, 128));
        if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) return uda_set_error("igemm_wgrad_x3: cannot reserve LDS");
        configured = true;
    }
    const dim3 grid(x.nCot * x.nJt, S);
    if (big) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<256, 256>), grid, dim3(768), ldsz(256, 256), st, x);
    else if (wideJ) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<128, 256>), grid, dim3(768), ldsz(128, 256), st, x);
    else hipLaunchKernelGGL((igemm_wgrad_x3_kernel<128, 128>), grid, dim3(768), ldsz(128, 128), st, x);
    UDA_LAUNCH_CHECK("igemm_wgrad_x3");
    return 0;
}
