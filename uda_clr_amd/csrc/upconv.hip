// "Upsample, then 3x3 conv" without the high-resolution GEMM (decoder.py:50-53 + last_conv_boundary[0], decoder.py:33).
//
// The decoder's first 3x3 conv reads cat(up4(f), low): 256 bilinearly upsampled ASPP channels and 48 low-level channels.
// Both the conv and the (align_corners) bilinear upsample are linear and the upsample acts per channel, so for the
// upsampled part the channel mixing commutes with the interpolation:
//
//     conv3x3(up(f))[p, co] = sum_t [p + d_t inside]  up(W_t f)(p + d_t)[co],      t = 9 taps, d_t = tap offset,
//
// i.e. nine 256 -> 256 channel GEMMs at the LOW resolution (g = f W_all^T, [P16, 9*256], 1/16 of the pixels: 19 GFLOP
// instead of 309 at B = 16) followed by this file's interpolation kernel, which is byte work: per output pixel and tap it
// blends the four low-resolution neighbours of the tap position with the upsample's own weights and adds the result
// to the conv of the 48 low-level channels (the addend).  Exact reassociation of the same sums - no approximation.
//
//   uda_upconv_fwd   y[p, :] = addend[p % addend_rows, :] + sum_t [tap inside] bilinear(g[:, t*C:(t+1)*C])(p + d_t)
//   uda_upconv_bwd   the adjoint: dG[q, t*C + c] = sum over the high-resolution tap positions that read q (gather: every
//                    low-resolution pixel re-derives, with the forward's own index arithmetic, who touched it -
//                    deterministic, no float atomics)
//
// Both are bound by cache / HBM bandwidth (the strip kernel: 13.5 cached 16-byte reads per 16-byte output at x4).
#include <algorithm>
#include "common.h"
#include "bilinear.h"

__global__ __launch_bounds__(256) void upconv_fwd_kernel(const float* __restrict__ g, int64_t ldg, int N, int h, int w, int C,
                                                         int dil, const float* __restrict__ addend, int64_t ld_add,
                                                         int64_t add_rows, float* __restrict__ y, int64_t ldy, int H, int W,
                                                         float sh, float sw) {
    const int G = C >> 2;
    const int64_t total = (int64_t)N * H * W * G;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const int ow = (int)(p % W), oh = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (addend) acc = uda_ld4(addend + (p % add_rows) * ld_add + cg * 4);
        const float* gb = g + (int64_t)n * h * w * ldg + cg * 4;
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const int yy = oh + (th - 1) * dil;
            if (yy < 0 || yy >= H) continue;
            int h0, h1;
            float lh0, lh1;
            bil_src(yy, sh, h, h0, h1, lh0, lh1);
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                const int xx = ow + (tw - 1) * dil;
                if (xx < 0 || xx >= W) continue;
                int w0, w1;
                float lw0, lw1;
                bil_src(xx, sw, w, w0, w1, lw0, lw1);
                const float* gt = gb + (th * 3 + tw) * C;
                const float4 a00 = uda_ld4(gt + ((int64_t)h0 * w + w0) * ldg), a01 = uda_ld4(gt + ((int64_t)h0 * w + w1) * ldg);
                const float4 a10 = uda_ld4(gt + ((int64_t)h1 * w + w0) * ldg), a11 = uda_ld4(gt + ((int64_t)h1 * w + w1) * ldg);
                acc.x += lh0 * (lw0 * a00.x + lw1 * a01.x) + lh1 * (lw0 * a10.x + lw1 * a11.x);
                acc.y += lh0 * (lw0 * a00.y + lw1 * a01.y) + lh1 * (lw0 * a10.y + lw1 * a11.y);
                acc.z += lh0 * (lw0 * a00.z + lw1 * a01.z) + lh1 * (lw0 * a10.z + lw1 * a11.z);
                acc.w += lh0 * (lw0 * a00.w + lw1 * a01.w) + lh1 * (lw0 * a10.w + lw1 * a11.w);
            }
        }
        uda_st4(y + p * ldy + cg * 4, acc);
    }
}

// Strip variant: one thread computes 4 channels of FOUR consecutive output pixels of a row.  For a fixed tap the four tap
// positions are consecutive, so they read at most NC = 3 (scale <= 1/3) or 4 (scale <= 2/3) distinct low-resolution columns:
// those are loaded once per tap (2 rows x NC columns) and blended from registers - 2.7x / 2x fewer cache reads than the
// pixel-at-a-time kernel.  Optional epilogue: per-channel (sum, sum of squares) of y for the following BatchNorm, accumulated
// per thread over its strips (the grid stride keeps a thread on its channel group), reduced per workgroup and added to one of
// the UDA_STAT_SLOTS replicas with fp64 atomics.
template <int NC>
__global__ __launch_bounds__(256) void upconv_fwd_strip_kernel(const float* __restrict__ g, int64_t ldg, int N, int h, int w, int C,
                                                               int dil, const float* __restrict__ addend, int64_t ld_add,
                                                               int64_t add_rows, float* __restrict__ y, int64_t ldy, int H, int W,
                                                               float sh, float sw, double* __restrict__ stats) {
    __shared__ float red[2][256][4];
    const int G = C >> 2, W4 = W >> 2;
    const int64_t total = (int64_t)N * H * W4 * G;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    int cg_mine = -1;
    // consecutive workgroups share the low-resolution rows they read (every g row serves 4 output rows x 9 taps): the XCD remap keeps
    // each XCD on a contiguous range of output rows, otherwise all eight L2s pull in all of g (302 MB at 32 images)
    // index arithmetic in 32 bits (the host routes here only when every element count fits): the 64-bit divisions and the
    // per-pixel 64-bit modulo of the addend row were a quarter of this kernel's instructions (1.65 k -> 1.4 k vector instructions per
    // strip).  PMC: 59 % of the wave cycles are parked on memory at 3 waves per SIMD (134 VGPRs) - the kernel is latency-bound; batching
    // the 18 loads of a row tap costs the third wave and is slower (886 vs 586 us on 32 images)
    const int wg = uda_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned total32 = (unsigned)total, step32 = gridDim.x * blockDim.x, arows = (unsigned)add_rows;
    const int ldg32 = (int)ldg, lda32 = (int)ld_add;
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, (int)((int64_t)N * h * w * ldg * 4), 0x00020000);
    for (unsigned e = (unsigned)wg * blockDim.x + threadIdx.x; e < total32; e += step32) {
        const unsigned sp = e / (unsigned)G;
        const int cg = (int)(e - sp * (unsigned)G);
        cg_mine = cg;
        // strip order: four vertically adjacent strips are consecutive (with C = 256 the four waves of a workgroup)
        const int k4 = (int)(sp & 3u);
        const unsigned sq = sp >> 2, rq = sq / (unsigned)W4, H4 = (unsigned)(H >> 2);
        const int ow0 = (int)(sq - rq * (unsigned)W4) * 4, n = (int)(rq / H4), oh = (int)(rq - (unsigned)n * H4) * 4 + k4;
        const unsigned p0 = ((unsigned)n * (unsigned)H + (unsigned)oh) * (unsigned)W + (unsigned)ow0;
        float4 acc[4];
        unsigned ar = addend ? p0 % arows : 0u;                 // addend row of the first pixel; the next ones wrap by comparison
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = addend ? uda_ld4(addend + (int64_t)ar * lda32 + cg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (++ar >= arows) ar = 0u;
        }
        const int gb_off = ((n * h * w) * ldg32 + cg * 4) * 4;       // byte offset of this image's / channel group's g (whole g < 2 GiB, host-checked)
        // Column part, once per strip and horizontal tap (it does not depend on the row tap): the leftmost low-resolution column
        // `a` the strip can read and, per output pixel j, the weight with which it reads column a + c (0 for a column it does
        // not read and for a tap position outside the image) - the blend below is then NC multiply-adds per pixel, no selects.
        int acol[3], coff[3][NC];                 // coff: byte offset of low-resolution column a + c (clamped) inside a g row
        float cw[3][4][NC];
        bool tw_on[3];
#pragma unroll
        for (int tw = 0; tw < 3; ++tw) {
            const int xb = ow0 + (tw - 1) * dil;                  // tap position of the strip's first pixel (may be outside)
            tw_on[tw] = !(xb + 3 < 0 || xb >= W);
            int a1;
            float t0, t1;
            bil_src(max(xb, 0), sw, w, acol[tw], a1, t0, t1);
#pragma unroll
            for (int c = 0; c < NC; ++c) coff[tw][c] = min(acol[tw] + c, w - 1) * (ldg32 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = xb + j;
                int w0, w1;
                float lw0, lw1;
                bil_src(min(max(xx, 0), W - 1), sw, w, w0, w1, lw0, lw1);
                const bool in = xx >= 0 && xx < W;
                const int i0 = w0 - acol[tw], i1 = w1 - acol[tw];      // in [0, NC) by construction of NC
#pragma unroll
                for (int c = 0; c < NC; ++c) cw[tw][j][c] = in ? (i0 == c ? lw0 : 0.f) + (i1 == c ? lw1 : 0.f) : 0.f;
            }
        }
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const int yy = oh + (th - 1) * dil;
            if (yy < 0 || yy >= H) continue;
            int h0, h1;
            float lh0, lh1;
            bil_src(yy, sh, h, h0, h1, lh0, lh1);
            const int ro0 = gb_off + h0 * w * (ldg32 * 4), ro1 = gb_off + h1 * w * (ldg32 * 4);
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                if (!tw_on[tw]) continue;
                const int tapoff = (th * 3 + tw) * C * 4;         // scalar: the tap's column block of g
                float4 v[NC];                                     // rows blended first: v[c] = lh0 * g[h0][a+c] + lh1 * g[h1][a+c]
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 r0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gres, ro0 + coff[tw][c], tapoff, 0));
                    const float4 r1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gres, ro1 + coff[tw][c], tapoff, 0));
                    v[c] = make_float4(lh0 * r0.x + lh1 * r1.x, lh0 * r0.y + lh1 * r1.y, lh0 * r0.z + lh1 * r1.z, lh0 * r0.w + lh1 * r1.w);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const float k = cw[tw][j][c];
                        acc[j].x += k * v[c].x;
                        acc[j].y += k * v[c].y;
                        acc[j].z += k * v[c].z;
                        acc[j].w += k * v[c].w;
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uda_st4(y + (int64_t)(p0 + j) * ldy + cg * 4, acc[j]);
            s1[0] += acc[j].x; s1[1] += acc[j].y; s1[2] += acc[j].z; s1[3] += acc[j].w;
            s2[0] += acc[j].x * acc[j].x; s2[1] += acc[j].y * acc[j].y; s2[2] += acc[j].z * acc[j].z; s2[3] += acc[j].w * acc[j].w;
        }
    }
    if (stats) {            // uniform over the grid; requires 256 % G == 0 (host-checked), so thread t always owns channel group t % G
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][threadIdx.x][j] = s1[j];
            red[1][threadIdx.x][j] = s2[j];
        }
        __syncthreads();
        double* dst = stats + (int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 2 * C;
        for (int e = threadIdx.x; e < 2 * C; e += 256) {
            const int qd = e / C, c = e % C, cgc = c >> 2, j = c & 3;
            float t = 0.f;
            for (int k = cgc; k < 256; k += G) t += red[qd][k][j];
            atomicAdd(&dst[qd * C + c], (double)t);
        }
    }
    (void)cg_mine;
}

// Tile variant (x4-like upsampling, the decoder's shape): one workgroup computes a 16 x 16 output tile of 32 channels.
//   * The low-resolution footprint of the tile - every g pixel one of its tap positions blends, at most R x RC pixels x 9 taps - is
//     fetched ONCE into LDS with back-to-back 16-byte loads; the blends read it from there.
//   * The strip kernel turned out to be bound by its VECTOR ARITHMETIC, not by its cached reads: ~1.4 k vector instructions per
//     strip (index / weight arithmetic of 15 bil_src per strip and 72 multiply-adds per tap), 4 cycles each on a wave64 - 170 of its
//     300 us at 16 images.  Here the interpolation weights are tabulated once per workgroup (18 tap rows, 4 strip columns x 3
//     horizontal taps), the three vertical taps are summed BEFORE the horizontal blend
//         V[tw][c] = sum_th lh0(th) * g_{th,tw}[h0(th)][a + c] + lh1(th) * g_{th,tw}[h1(th)][a + c],   y[j] += cw[tw][j][c] * V[tw][c]
//     (90 instead of 162 four-channel multiply-adds per strip), on packed pairs (v_pk_fma_f32).
//   Same sums as the strip kernel in another association: results agree to fp32 rounding, not bit for bit.
#define UPT_TH 16
#define UPT_TW 16
#define UPT_CS 32            // channels per workgroup (8 granules of 4)
__global__ __launch_bounds__(256) void upconv_fwd_tile_kernel(const float* __restrict__ g, int64_t ldg, int N, int h, int w, int C,
                                                              const float* __restrict__ addend, int64_t ld_add, int64_t add_rows,
                                                              float* __restrict__ y, int64_t ldy, int H, int W, float sh, float sw,
                                                              double* __restrict__ stats, int R, int RC) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float4 foot[];            // [R][RC][9 taps][8 granules]
    __shared__ float red[2][256][4];
    __shared__ float4 rowtab[UPT_TH + 2];        // tap row yy = oh0 - 1 + i: (footprint row of h0) * RC, (.. of h1) * RC [as ints], lh0, lh1 (0, 0 outside the image)
    __shared__ float coltab[4][3][16];           // strip column sc, horizontal tap tw: 3 footprint columns [as ints], then cw[4 pixels][3 columns]
    constexpr int NC = 3, dil = 1;
    const int tid = threadIdx.x;
    const int nCs = C / UPT_CS, nTx = W / UPT_TW, nTy = H / UPT_TH;
    // consecutive workgroups: the channel slices of one tile (adjacent 128-byte segments of the same output rows), then the tiles of
    // a row of tiles; each XCD keeps a contiguous range
    int b = uda_xcd_remap(blockIdx.x, gridDim.x);
    const int cs = b % nCs;
    b /= nCs;
    const int tx = b % nTx;
    b /= nTx;
    const int ty = b % nTy, n = b / nTy;
    const int oh0 = ty * UPT_TH, ow0t = tx * UPT_TW;
    // footprint origin: the first low-resolution row / column the tile's topmost / leftmost tap position reads
    int hlo, wlo, i1;
    float t0, t1;
    bil_src(max(oh0 - dil, 0), sh, h, hlo, i1, t0, t1);
    bil_src(max(ow0t - dil, 0), sw, w, wlo, i1, t0, t1);
    {
        const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, (int)((int64_t)N * h * w * ldg * 4), 0x00020000);
        const int ldg4 = (int)ldg * 4, nload = R * RC * 72;
        const int base = ((n * h) * w) * ldg4 + (cs * UPT_CS) * 4;
        // six loads in flight per thread (a load followed by its own LDS store in one loop body keeps ONE in flight)
        for (int i0 = tid; i0 < nload; i0 += 256 * 6) {
            float4 v[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int i = min(i0 + 256 * u, nload - 1);
                const int cg = i & 7, q = i >> 3, pc = q / 9, tap = q - pc * 9, r = pc / RC, c = pc - r * RC;
                const int hh = min(hlo + r, h - 1), ww = min(wlo + c, w - 1);        // (rows / columns beyond the image: never blended)
                v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gres, base + (hh * w + ww) * ldg4 + cg * 16, tap * C * 4, 0));
            }
#pragma unroll
            for (int u = 0; u < 6; ++u)
                if (i0 + 256 * u < nload) foot[i0 + 256 * u] = v[u];
        }
    }
    if (tid < UPT_TH + 2) {                      // row table
        const int yy = oh0 - dil + tid;
        int h0 = hlo, h1 = hlo;
        float lh0 = 0.f, lh1 = 0.f;
        if (yy >= 0 && yy < H) bil_src(yy, sh, h, h0, h1, lh0, lh1);
        rowtab[tid] = make_float4(__int_as_float((h0 - hlo) * RC), __int_as_float((h1 - hlo) * RC), lh0, lh1);
    } else if (tid >= 64 && tid < 64 + 12) {     // column tables (as the strip kernel's column part)
        const int sc = (tid - 64) / 3, tw = (tid - 64) % 3;
        const int xb = ow0t + sc * 4 + (tw - 1) * dil;
        int acol, a1;
        bil_src(max(xb, 0), sw, w, acol, a1, t0, t1);
        float* ct = coltab[sc][tw];
        // footprint column of low-resolution column a + c, clamped into the loaded range (its weight is 0 beyond: finite data, no NaN)
        for (int c = 0; c < NC; ++c) ct[c] = __int_as_float(max(min(min(acol + c, w - 1) - wlo, RC - 1), 0));
        for (int jx = 0; jx < 4; ++jx) {
            const int xx = xb + jx;
            int w0, w1;
            float lw0, lw1;
            bil_src(min(max(xx, 0), W - 1), sw, w, w0, w1, lw0, lw1);
            const bool in = xx >= 0 && xx < W;
            const int i0 = w0 - acol, i1c = w1 - acol;
            for (int c = 0; c < NC; ++c) ct[3 + jx * 3 + c] = in ? (i0 == c ? lw0 : 0.f) + (i1c == c ? lw1 : 0.f) : 0.f;
        }
    }
    // the addend rows of this thread's first strip travel with the footprint loads, those of its second strip under the first's blends
    const int cg = tid & 7, s = tid >> 3, sc = s & 3, ow0 = ow0t + sc * 4, srow = s >> 2;
    const unsigned arows = (unsigned)add_rows;
    const int lda32 = (int)ld_add, ch = cs * UPT_CS + cg * 4;
    float4 ad[2][4];
    auto load_addend = [&](int k) {
        const unsigned p0 = ((unsigned)n * (unsigned)H + (unsigned)(oh0 + srow + 8 * k)) * (unsigned)W + (unsigned)ow0;
        unsigned ar = addend ? p0 % arows : 0u;
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            ad[k][jx] = addend ? uda_ld4(addend + (int64_t)ar * lda32 + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (++ar >= arows) ar = 0u;
        }
    };
    load_addend(0);
    __syncthreads();
    load_addend(1);
    int lc[3][NC];
    float cw[3][4][NC];
#pragma unroll
    for (int tw = 0; tw < 3; ++tw) {
        const float* ct = coltab[sc][tw];
#pragma unroll
        for (int c = 0; c < NC; ++c) lc[tw][c] = __float_as_int(ct[c]) * 72 + cg;
#pragma unroll
        for (int jx = 0; jx < 4; ++jx)
#pragma unroll
            for (int c = 0; c < NC; ++c) cw[tw][jx][c] = ct[3 + jx * 3 + c];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = srow + 8 * k, oh = oh0 + row;
        const unsigned p0 = ((unsigned)n * (unsigned)H + (unsigned)oh) * (unsigned)W + (unsigned)ow0;
        f32x2 acc[4][2];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            acc[jx][0] = f32x2{ad[k][jx].x, ad[k][jx].y};
            acc[jx][1] = f32x2{ad[k][jx].z, ad[k][jx].w};
        }
        int r0[3], r1[3];
        float lh0[3], lh1[3];
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const float4 rt = rowtab[row + th];
            r0[th] = __float_as_int(rt.x) * 72 + th * 24;        // + tap * 8 with tap = th * 3 + tw
            r1[th] = __float_as_int(rt.y) * 72 + th * 24;
            lh0[th] = rt.z;
            lh1[th] = rt.w;
        }
#pragma unroll
        for (int tw = 0; tw < 3; ++tw) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
#pragma unroll
                for (int th = 0; th < 3; ++th) {
                    const float4 a0 = foot[r0[th] + lc[tw][c] + tw * 8], a1v = foot[r1[th] + lc[tw][c] + tw * 8];
                    v0 += f32x2{a0.x, a0.y} * lh0[th];
                    v0 += f32x2{a1v.x, a1v.y} * lh1[th];
                    v1 += f32x2{a0.z, a0.w} * lh0[th];
                    v1 += f32x2{a1v.z, a1v.w} * lh1[th];
                }
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    acc[jx][0] += v0 * cw[tw][jx][c];
                    acc[jx][1] += v1 * cw[tw][jx][c];
                }
            }
        }
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const float4 o = make_float4(acc[jx][0].x, acc[jx][0].y, acc[jx][1].x, acc[jx][1].y);
            uda_st4(y + (int64_t)(p0 + jx) * ldy + ch, o);
            s1[0] += o.x; s1[1] += o.y; s1[2] += o.z; s1[3] += o.w;
            s2[0] += o.x * o.x; s2[1] += o.y * o.y; s2[2] += o.z * o.z; s2[3] += o.w * o.w;
        }
    }
    if (stats) {            // thread t owns granule t & 7 of this workgroup's 32 channels
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            red[0][tid][jx] = s1[jx];
            red[1][tid][jx] = s2[jx];
        }
        __syncthreads();
        if (tid < 2 * UPT_CS) {
            const int qd = tid / UPT_CS, c = tid % UPT_CS, cgc = c >> 2, jx = c & 3;
            float t = 0.f;
            for (int k = cgc; k < 256; k += 8) t += red[qd][k][jx];
            double* dst = stats + (int64_t)(blockIdx.x % UDA_STAT_SLOTS) * 2 * C;
            atomicAdd(&dst[qd * C + cs * UPT_CS + c], (double)t);
        }
    }
}

// rows (columns) of the low-resolution footprint of a tile whose tap positions span [o_min, o_max] (clipped to the image)
static int upconv_foot(int o_min, int o_max, float scale, int n_in, int n_out) {
    const float lo = scale * (float)(o_min < 0 ? 0 : o_min), hi = scale * (float)(o_max > n_out - 1 ? n_out - 1 : o_max);
    int i_lo = (int)lo, i_hi = (int)hi;
    if (i_lo > n_in - 1) i_lo = n_in - 1;
    if (i_hi > n_in - 1) i_hi = n_in - 1;
    i_hi += i_hi < n_in - 1 ? 1 : 0;
    return i_hi - i_lo + 1;
}

static bool upconv_strip_ok(int w, int W, int C, int dil, int H = 4) {
    if (H % 4) return false;
    // the four tap positions of a strip span 3*sw low-resolution columns: floor(frac + 3*sw) + 1 <= NC - 1 with NC <= 4
    // (margins of 2 %: the column indices come from fp32 products scale * x, whose rounding must not push a strip over its last cached column)
    return W % 4 == 0 && dil == 1 && C % 4 == 0 && 3.f * bil_scale(w, W) < 1.96f && 256 % (C / 4) == 0;       // (+ H % 4 == 0, checked by the caller)
}

extern "C" int uda_upconv_fused_stats(int h, int w, int H, int W, int C, int dil) {
    (void)h; (void)H;
    return upconv_strip_ok(w, W, C, dil, H) ? 1 : 0;
}

extern "C" int uda_upconv_fwd(const float* g, int64_t ldg, int N, int h, int w, int C, int dil, const float* addend,
                              int64_t ld_add, int64_t addend_rows, float* y, int64_t ldy, int H, int W, double* stats,
                              void* stream) {
    UDA_REQUIRE(g && y && uda_aligned16(g) && uda_aligned16(y) && C > 0 && C % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 &&
                    ldg >= 9 * (int64_t)C && ldy >= C, "uda_upconv_fwd: g must be [N*h*w, >= 9*C], C and lds multiples of 4, 16-byte aligned");
    UDA_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && dil >= 1, "uda_upconv_fwd: bad geometry");
    UDA_REQUIRE(!addend || (uda_aligned16(addend) && ld_add % 4 == 0 && ld_add >= C && addend_rows > 0 &&
                            ((int64_t)N * H * W) % addend_rows == 0),
                "uda_upconv_fwd: addend must be [addend_rows, >= C] with addend_rows dividing N*H*W");
    hipStream_t st = (hipStream_t)stream;
    const float sh = bil_scale(h, H), sw = bil_scale(w, W);
    const int G = C / 4;
    const int64_t lim32 = (int64_t)1 << 31;          // the strip kernel indexes in 32 bits
    if (upconv_strip_ok(w, W, C, dil, H) && (int64_t)N * H * W < lim32 && (int64_t)N * H * (W / 4) * G < lim32 - 65536 * 256 &&
        (int64_t)N * h * w * ldg * 4 < lim32 && (!addend || addend_rows * ld_add < lim32 * 4)) {
        // the LDS-tiled kernel where the tiling fits (x4-like upsampling of a map whose sides are multiples of 16)
        static const int tile_env = getenv("UDA_UPCONV_TILE") ? atoi(getenv("UDA_UPCONV_TILE")) : 1;      // A/B switch
        if (tile_env && 3.f * sw < 0.98f && H % UPT_TH == 0 && W % UPT_TW == 0 && C % UPT_CS == 0) {
            int R = 1, RC = 1;
            for (int ty = 0; ty < H / UPT_TH; ++ty) R = std::max(R, upconv_foot(ty * UPT_TH - 1, ty * UPT_TH + UPT_TH, sh, h, H));
            for (int tx = 0; tx < W / UPT_TW; ++tx) RC = std::max(RC, upconv_foot(tx * UPT_TW - 1, tx * UPT_TW + UPT_TW, sw, w, W));
            const size_t lds = (size_t)R * RC * 72 * sizeof(float4);
            const int64_t nwg = (int64_t)N * (H / UPT_TH) * (W / UPT_TW) * (C / UPT_CS);
            if (lds <= 64 * 1024 - 8192 && nwg < lim32) {
                hipLaunchKernelGGL(upconv_fwd_tile_kernel, dim3((unsigned)nwg), dim3(256), lds, st, g, ldg, N, h, w, C, addend, ld_add,
                                   addend ? addend_rows : 1, y, ldy, H, W, sh, sw, stats, R, RC);
                UDA_LAUNCH_CHECK("upconv_fwd_tile");
                return 0;
            }
        }
        const int64_t total = (int64_t)N * H * (W / 4) * G;
        int grid = uda_cdiv(total, 256);
        if (grid > 4096) grid = 4096;           // bounded: the statistics epilogue issues 2*C atomics per workgroup
        if (3.f * sw < 0.98f)
            hipLaunchKernelGGL(upconv_fwd_strip_kernel<3>, dim3(grid), dim3(256), 0, st, g, ldg, N, h, w, C, dil, addend, ld_add,
                               addend ? addend_rows : 1, y, ldy, H, W, sh, sw, stats);
        else
            hipLaunchKernelGGL(upconv_fwd_strip_kernel<4>, dim3(grid), dim3(256), 0, st, g, ldg, N, h, w, C, dil, addend, ld_add,
                               addend ? addend_rows : 1, y, ldy, H, W, sh, sw, stats);
        UDA_LAUNCH_CHECK("upconv_fwd_strip");
        return 0;
    }
    UDA_REQUIRE(!stats, "uda_upconv_fwd: the fused statistics need W %% 4 == 0, dil == 1, an upsampling factor >= 1.5 and C/4 dividing 256; "
                        "accumulate them with uda_colstats instead");
    const int64_t total = (int64_t)N * H * W * G;
    int grid = uda_cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(upconv_fwd_kernel, dim3(grid), dim3(256), 0, st, g, ldg, N, h, w, C, dil, addend, ld_add,
                       addend ? addend_rows : 1, y, ldy, H, W, sh, sw);
    UDA_LAUNCH_CHECK("upconv_fwd");
    return 0;
}

// one thread: 4 channels of one low-resolution pixel, ALL nine taps: every dy value of the neighbourhood is loaded once and
// feeds the nine accumulators with the product of its row weight (per vertical tap) and column weight (per horizontal tap) -
// 16 cached 16-byte reads per 16-byte output at x4 instead of 81 for a thread per (pixel, tap)
__global__ __launch_bounds__(256) void upconv_bwd_kernel(const float* __restrict__ dy, int64_t ldy, int N, int H, int W, int C,
                                                         int dil, float* __restrict__ dg, int64_t ldg, int h, int w, float sh,
                                                         float sw) {
    const int G = C >> 2;
    const int64_t total = (int64_t)N * h * w * G;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t q = e / G;
        const int iw = (int)(q % w), ih = (int)((q / w) % h), n = (int)(q / ((int64_t)w * h));
        int hlo, hhi, wlo, whi;
        bil_range(ih, sh, H, hlo, hhi);         // tap positions (inside the image) that can read low-resolution row ih / column iw
        bil_range(iw, sw, W, wlo, whi);
        float4 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oh = max(hlo - dil, 0); oh <= min(hhi + dil, H - 1); ++oh) {
            float a3[3];
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                const int yy = oh + (th - 1) * dil;              // where output row oh's tap th sits
                a3[th] = (yy >= 0 && yy < H) ? bil_weight(yy, sh, h, ih) : 0.f;
            }
            if (a3[0] == 0.f && a3[1] == 0.f && a3[2] == 0.f) continue;
            const float* row = dy + (((int64_t)n * H + oh) * W) * ldy + cg * 4;
            for (int ow = max(wlo - dil, 0); ow <= min(whi + dil, W - 1); ++ow) {
                float b3[3];
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    const int xx = ow + (tw - 1) * dil;
                    b3[tw] = (xx >= 0 && xx < W) ? bil_weight(xx, sw, w, iw) : 0.f;
                }
                if (b3[0] == 0.f && b3[1] == 0.f && b3[2] == 0.f) continue;
                const float4 v = uda_ld4(row + (int64_t)ow * ldy);
#pragma unroll
                for (int th = 0; th < 3; ++th)
#pragma unroll
                    for (int tw = 0; tw < 3; ++tw) {
                        const float k = a3[th] * b3[tw];
                        float4& r = acc[th * 3 + tw];
                        r.x += k * v.x; r.y += k * v.y; r.z += k * v.z; r.w += k * v.w;
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) uda_st4(dg + q * ldg + t * C + cg * 4, acc[t]);
    }
}

// Wave-per-pixel variant (C = 256: the 64 lanes of a wave are the 64 channel granules of ONE low-resolution pixel).  The thread-per-
// (pixel, granule) kernel above is bound by its vector arithmetic: every lane re-derives the same interpolation weights (a bil_src
// per (row, tap) and per (row, column, tap)) and spends 36 multiply-adds per dy value.  Here
//   * the weights are wave-uniform: lane l < 32 evaluates the row weight A(hlo + l) = weight of tap ROW position hlo + l on source row ih,
//     lane 32 + l the column weight B(wlo + l), once; the loops fetch them with v_readlane into scalar registers;
//   * the column sums are formed first, tmp[tw] = sum_ow B(ow + tw - 1) dy[oh][ow] (12 multiply-adds per dy value), then
//     dG[(th, tw)] += A(oh + th - 1) tmp[tw] once per row.
__global__ __launch_bounds__(256) void upconv_bwd_wave_kernel(const float* __restrict__ dy, int64_t ldy, int N, int H, int W, int C,
                                                              float* __restrict__ dg, int64_t ldg, int h, int w, float sh, float sw) {
    const int lane = threadIdx.x & 63;
    const int64_t npix = (int64_t)N * h * w;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t qv = wave0; qv < npix; qv += nwaves) {
        const int64_t q = __builtin_amdgcn_readfirstlane((int)qv);          // (npix < 2^31, host-checked)
        const int iw = (int)(q % w), ih = (int)((q / w) % h), n = (int)(q / ((int64_t)w * h));
        int hlo, hhi, wlo, whi;
        bil_range(ih, sh, H, hlo, hhi);         // tap positions (inside the image) that can read low-resolution row ih / column iw
        bil_range(iw, sw, W, wlo, whi);
        // (host-checked: both ranges are at most 32 wide)
        const int l31 = lane & 31;
        const float tab = lane < 32 ? (hlo + l31 <= hhi ? bil_weight(hlo + l31, sh, h, ih) : 0.f)
                                    : (wlo + l31 <= whi ? bil_weight(wlo + l31, sw, w, iw) : 0.f);
        auto A = [&](int yy) { return (yy >= hlo && yy <= hhi) ? __builtin_amdgcn_readlane(__float_as_int(tab), yy - hlo) : 0; };
        auto B = [&](int xx) { return (xx >= wlo && xx <= whi) ? __builtin_amdgcn_readlane(__float_as_int(tab), 32 + xx - wlo) : 0; };
        float4 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int oh0 = max(hlo - 1, 0), oh1 = min(hhi + 1, H - 1), ow0 = max(wlo - 1, 0), ow1 = min(whi + 1, W - 1);
        for (int oh = oh0; oh <= oh1; ++oh) {
            const float a0 = __int_as_float(A(oh - 1)), a1 = __int_as_float(A(oh)), a2 = __int_as_float(A(oh + 1));
            if (a0 == 0.f && a1 == 0.f && a2 == 0.f) continue;
            const float* row = dy + (((int64_t)n * H + oh) * W) * ldy + lane * 4;
            float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0, t2 = t0;
            float bm = __int_as_float(B(ow0 - 1)), bc = __int_as_float(B(ow0));
            for (int owb = ow0; owb <= ow1; owb += 4) {           // four columns' loads in flight (clamped; a column beyond ow1 has weights 0)
                float4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v4[u] = uda_ld4(row + (int64_t)min(owb + u, ow1) * ldy);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ow = owb + u;
                    const float bp = __int_as_float(B(ow + 1));
                    const float4 v = v4[u];
                    const float km = ow <= ow1 ? bm : 0.f, kc = ow <= ow1 ? bc : 0.f, kp = ow <= ow1 ? bp : 0.f;
                    t0.x += km * v.x; t0.y += km * v.y; t0.z += km * v.z; t0.w += km * v.w;      // tap tw = 0 sits at ow - 1
                    t1.x += kc * v.x; t1.y += kc * v.y; t1.z += kc * v.z; t1.w += kc * v.w;
                    t2.x += kp * v.x; t2.y += kp * v.y; t2.z += kp * v.z; t2.w += kp * v.w;
                    bm = bc;
                    bc = bp;
                }
            }
            const float a3[3] = {a0, a1, a2};
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                float4& r0 = acc[th * 3 + 0];
                float4& r1 = acc[th * 3 + 1];
                float4& r2 = acc[th * 3 + 2];
                const float k = a3[th];
                r0.x += k * t0.x; r0.y += k * t0.y; r0.z += k * t0.z; r0.w += k * t0.w;
                r1.x += k * t1.x; r1.y += k * t1.y; r1.z += k * t1.z; r1.w += k * t1.w;
                r2.x += k * t2.x; r2.y += k * t2.y; r2.z += k * t2.z; r2.w += k * t2.w;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) uda_st4(dg + q * ldg + t * C + lane * 4, acc[t]);
    }
}

extern "C" int uda_upconv_bwd(const float* dy, int64_t ldy, int N, int H, int W, int C, int dil, float* dg, int64_t ldg, int h,
                              int w, void* stream) {
    UDA_REQUIRE(dy && dg && uda_aligned16(dy) && uda_aligned16(dg) && C > 0 && C % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 &&
                    ldg >= 9 * (int64_t)C && ldy >= C, "uda_upconv_bwd: dg must be [N*h*w, >= 9*C], C and lds multiples of 4, 16-byte aligned");
    UDA_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && dil >= 1, "uda_upconv_bwd: bad geometry");
    const int64_t total = (int64_t)N * h * w * (C / 4);
    int grid = uda_cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    const float sh = bil_scale(h, H), sw = bil_scale(w, W);
    // one wave per low-resolution pixel when its 64 lanes are exactly the channel granules and the tap ranges fit the 32-entry weight tables
    static const int wave_env = getenv("UDA_UPCONV_BWD_WAVE") ? atoi(getenv("UDA_UPCONV_BWD_WAVE")) : 1;      // A/B switch
    if (wave_env && C == 256 && dil == 1 && sh > 0.f && sw > 0.f && 2.f / sh + 6.f <= 32.f && 2.f / sw + 6.f <= 32.f && (int64_t)N * h * w < ((int64_t)1 << 31)) {
        const int64_t npix = (int64_t)N * h * w;
        int gw = (int)uda_cdiv(npix, 4);
        if (gw > 65536) gw = 65536;
        hipLaunchKernelGGL(upconv_bwd_wave_kernel, dim3(gw), dim3(256), 0, (hipStream_t)stream, dy, ldy, N, H, W, C, dg, ldg, h, w, sh, sw);
        UDA_LAUNCH_CHECK("upconv_bwd_wave");
        return 0;
    }
    hipLaunchKernelGGL(upconv_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, ldy, N, H, W, C, dil, dg, ldg, h, w,
                       sh, sw);
    UDA_LAUNCH_CHECK("upconv_bwd");
    return 0;
}
