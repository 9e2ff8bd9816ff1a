// Device-side tail of the input pipeline: what the reference's dataloader workers do per sample on
// the CPU after the random augmentations, done per BATCH on the GPU from the uint8 data instead.
//
//   uda_normalize_tf   dataloaders/custom_transforms.py:432-466 (Normalize_tf) + :414-429 (GetBoundary) + :504-507
//                      (ToTensor): image/127.5 - 1, grey-coded mask -> 2-channel map, boundary = Gaussian(sigma 3) of the
//                      |dilate5 - erode5| ring of both channels.  The reference calls scipy.ndimage for the morphology and
//                      the blur; the kernels below restate those algorithms exactly (tests compare with scipy bit for bit):
//                        * binary_dilation / binary_erosion with the default cross, iterations = 5, border_value = 0
//                          == dilation / erosion by the L1 ball of radius 5 with zeros outside the image;
//                        * gaussian_filter on a uint8 array keeps uint8: axis 0 first, each pass accumulates in double in
//                          correlate1d's symmetric order (centre, then the pairs from the farthest inwards), truncates to
//                          uint8, mode 'reflect'.
//   uda_elastic_warp   custom_transforms.py:95-147 (elastic_transform): bilinear map_coordinates of the image (constant 0
//                      outside) and of the label (nearest edge outside) along a displacement field, rounded to uint8.
//   uda_field_smooth   the separable float64 Gaussian (mode 'constant') that turns the uniform noise into that displacement
//                      field, in scipy's summation order (bit-identical field).
//
// All kernels are HBM-bound byte work: one coalesced pass per plane, halo rows/columns through LDS or the L2.
#include "common.h"

// scipy's C loops are compiled without fused multiply-add: a*b + c rounds twice.  The truncations / roundings to uint8 below
// must see the same doubles (255 * sum(w) is 254.99999999999997 or 255.0 depending on it), so no contraction in this file.
#pragma clang fp contract(off)

#define NTF_MAXR 16

struct GaussW {
    double w[NTF_MAXR + 1];     // w[0] centre, w[k] the two taps at distance k
};

// ------------------------------------------------------------------------------------------ decode
__global__ __launch_bounds__(256) void ntf_decode_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lab,
                                                         int64_t HW, float* __restrict__ image, float* __restrict__ map) {
    const int b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const uint8_t* px = img + ((int64_t)b * HW + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) image[((int64_t)b * 3 + c) * HW + p] = (float)px[c] / 127.5f - 1.0f;
    const int g = lab[(int64_t)b * HW + p];          // > 200 background, 51..200 disc rim, <= 50 cup (inside the disc)
    map[((int64_t)b * 2 + 0) * HW + p] = g <= 50 ? 1.f : 0.f;
    map[((int64_t)b * 2 + 1) * HW + p] = g <= 200 ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------ ring
#define RING_R 5
#define RING_TH 16
#define RING_TW 64
__global__ __launch_bounds__(256) void ntf_ring_kernel(const uint8_t* __restrict__ lab, int H, int W, uint8_t* __restrict__ ring) {
    __shared__ uint8_t t[RING_TH + 2 * RING_R][RING_TW + 2 * RING_R + 2];       // bit0 cup, bit1 disc; 0 outside the image
    const int b = blockIdx.z, h0 = blockIdx.y * RING_TH, w0 = blockIdx.x * RING_TW;
    const uint8_t* L = lab + (int64_t)b * H * W;
    for (int e = threadIdx.x; e < (RING_TH + 2 * RING_R) * (RING_TW + 2 * RING_R); e += 256) {
        const int r = e / (RING_TW + 2 * RING_R), c = e % (RING_TW + 2 * RING_R);
        const int h = h0 + r - RING_R, w = w0 + c - RING_R;
        uint8_t v = 0;
        if (h >= 0 && h < H && w >= 0 && w < W) {
            const int g = L[(int64_t)h * W + w];
            v = (g <= 50 ? 1 : 0) | (g <= 200 ? 2 : 0);
        }
        t[r][c] = v;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < RING_TH; rr += 4) {
        const int h = h0 + rr, w = w0 + tx;
        if (h >= H || w >= W) continue;
        uint8_t any = 0, all = 3;
#pragma unroll
        for (int dy = -RING_R; dy <= RING_R; ++dy) {
            const int span = RING_R - (dy < 0 ? -dy : dy);
#pragma unroll
            for (int dx = -RING_R; dx <= RING_R; ++dx) {
                if (dx < -span || dx > span) continue;
                const uint8_t v = t[rr + RING_R + dy][tx + RING_R + dx];
                any |= v;
                all &= v;
            }
        }
        ring[((int64_t)b * H + h) * W + w] = (any ^ all) ? 255 : 0;         // dilated but not eroded, in either channel
    }
}

// ------------------------------------------------------------------------------------------ uint8 Gaussian, one axis
__device__ __forceinline__ int reflect_idx(int i, int n) {      // scipy 'reflect': d c b a | a b c d | d c b a
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
    return i;
}

template <int AXIS, bool LAST>
__global__ __launch_bounds__(256) void ntf_gauss_kernel(const uint8_t* __restrict__ in, int H, int W, GaussW gw, int R,
                                                        uint8_t* __restrict__ out8, float* __restrict__ outf) {
    const int b = blockIdx.z, h = blockIdx.y, w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    const uint8_t* I = in + (int64_t)b * H * W;
    const int n = AXIS == 0 ? H : W, c = AXIS == 0 ? h : w;
    auto at = [&](int i) -> double {
        const int j = reflect_idx(i, n);
        return (double)(AXIS == 0 ? I[(int64_t)j * W + w] : I[(int64_t)h * W + j]);
    };
    double tmp = at(c) * gw.w[0];
    for (int k = R; k >= 1; --k) tmp += (at(c - k) + at(c + k)) * gw.w[k];
    const uint8_t v = (uint8_t)tmp;                                   // truncation, like the cast into scipy's uint8 output
    const int64_t o = ((int64_t)b * H + h) * W + w;
    if (LAST) outf[o] = (float)((double)v / 255.0);
    else out8[o] = v;
}

extern "C" size_t uda_normalize_tf_workspace_bytes(int B, int H, int W) { return (size_t)2 * B * H * W + 32; }

extern "C" int uda_normalize_tf(const uint8_t* image_hwc, const uint8_t* label, int B, int H, int W, const double* gauss_w,
                                int radius, float* image, float* map, float* boundary, void* workspace,
                                size_t workspace_bytes, void* stream) {
    UDA_REQUIRE(image_hwc && label && gauss_w && image && map && boundary && workspace, "uda_normalize_tf: null argument");
    UDA_REQUIRE(B > 0 && radius >= 1 && radius <= NTF_MAXR && H > radius && W > radius,
                "uda_normalize_tf: need B > 0, 1 <= radius <= %d and H, W > radius (got B=%d, %dx%d, radius %d)", NTF_MAXR, B, H, W, radius);
    UDA_REQUIRE(workspace_bytes >= uda_normalize_tf_workspace_bytes(B, H, W), "uda_normalize_tf: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    uint8_t* ring = (uint8_t*)workspace;
    uint8_t* pass1 = ring + (size_t)B * HW;
    GaussW gw;
    for (int k = 0; k <= NTF_MAXR; ++k) gw.w[k] = k <= radius ? gauss_w[k] : 0.0;
    hipLaunchKernelGGL(ntf_decode_kernel, dim3(uda_cdiv(HW, 256), B), dim3(256), 0, st, image_hwc, label, HW, image, map);
    UDA_LAUNCH_CHECK("ntf_decode");
    hipLaunchKernelGGL(ntf_ring_kernel, dim3(uda_cdiv(W, RING_TW), uda_cdiv(H, RING_TH), B), dim3(256), 0, st, label, H, W, ring);
    UDA_LAUNCH_CHECK("ntf_ring");
    hipLaunchKernelGGL((ntf_gauss_kernel<0, false>), dim3(uda_cdiv(W, 256), H, B), dim3(256), 0, st, ring, H, W, gw, radius, pass1,
                       (float*)nullptr);
    UDA_LAUNCH_CHECK("ntf_gauss axis 0");
    hipLaunchKernelGGL((ntf_gauss_kernel<1, true>), dim3(uda_cdiv(W, 256), H, B), dim3(256), 0, st, pass1, H, W, gw, radius,
                       (uint8_t*)nullptr, boundary);
    UDA_LAUNCH_CHECK("ntf_gauss axis 1");
    return 0;
}

// ------------------------------------------------------------------------------------------ elastic deformation
// One separable pass of scipy.ndimage.gaussian_filter(field, sigma, mode='constant', cval=0) along AXIS on float64 planes,
// EXACTLY as scipy computes it (the reference feeds float64 noise, custom_transforms.py:116-117, and the warp below rounds the
// interpolated greys to uint8: a field that differs in its last bits flips ~1e-3 of the output bytes, so fp32 or another
// summation order is not byte-identical to the reference).  correlate1d's symmetric branch (ni_filters.c):
//     tmp = x[c] * w[0];  for d = R .. 1:  tmp += (x[c - d] + x[c + d]) * w[d]        (no fused multiply-add)
// The radius is large (0.32 * side: 164 taps each way at 512), so the operand is staged in LDS once per workgroup:
//   tile = FS_RUN * FS_GROUPS outputs along the smoothed axis x FS_LINES lines across it, LDS image [position][line] with a
//   line stride of FS_LD doubles; a thread owns FS_RUN consecutive outputs of its line and walks d from R down to 1 with two
//   sliding register windows (the left taps shift right, the right taps shift left: two new LDS operands and one weight per d
//   feed FS_RUN pair-sums), so every accumulator sees its terms in scipy's order.
#define FS_RUN 8
#define FS_GROUPS 16
#define FS_TILE (FS_RUN * FS_GROUPS)       // 128 outputs along the axis per workgroup
#define FS_LINES 32
#define FS_LD 33

template <int AXIS>
__global__ __launch_bounds__(FS_GROUPS * FS_LINES) void field_smooth_kernel(const double* __restrict__ in, int H, int W,
                                                                            const double* __restrict__ wts, int R, double scale,
                                                                            int apply_scale, double* __restrict__ out) {
    extern __shared__ double fs_lds[];            // [FS_TILE + 2R][FS_LD] values, then [R + 1] weights
    const int n = AXIS == 0 ? H : W, m = AXIS == 0 ? W : H;             // smoothed extent, number of lines
    const int span = FS_TILE + 2 * R;
    double* wl = fs_lds + (size_t)span * FS_LD;
    const int b = blockIdx.z, a0 = blockIdx.x * FS_TILE, l0 = blockIdx.y * FS_LINES;
    const double* I = in + (int64_t)b * H * W;
    const int tid = threadIdx.x;
    for (int k = tid; k <= R; k += blockDim.x) wl[k] = wts[k];
    // fill: element (position a0 - R + i, line l0 + l); zeros outside the plane ('constant' mode, cval 0)
    if (AXIS == 0) {        // lines are columns: consecutive threads take consecutive columns of one row (coalesced)
        for (int e = tid; e < span * FS_LINES; e += blockDim.x) {
            const int i = e / FS_LINES, l = e % FS_LINES;
            const int pos = a0 - R + i, line = l0 + l;
            fs_lds[i * FS_LD + l] = (pos >= 0 && pos < n && line < m) ? I[(int64_t)pos * W + line] : 0.0;
        }
    } else {                // lines are rows: consecutive threads walk along a row (coalesced), transposed into [position][line]
        for (int e = tid; e < span * FS_LINES; e += blockDim.x) {
            const int l = e / span, i = e % span;
            const int pos = a0 - R + i, line = l0 + l;
            fs_lds[i * FS_LD + l] = (pos >= 0 && pos < n && line < m) ? I[(int64_t)line * W + pos] : 0.0;
        }
    }
    __syncthreads();
    const int l = tid % FS_LINES, grp = tid / FS_LINES;
    const double* col = fs_lds + (size_t)(grp * FS_RUN + R) * FS_LD + l;        // col[i * FS_LD]: operand i positions from the run's first output
    double acc[FS_RUN], lo[FS_RUN], hi[FS_RUN];
    const double w0 = wl[0];
#pragma unroll
    for (int j = 0; j < FS_RUN; ++j) {
        acc[j] = col[j * FS_LD] * w0;
        lo[j] = col[(j - R) * FS_LD];           // x[c_j - R]
        hi[j] = col[(j + R) * FS_LD];           // x[c_j + R]
    }
    for (int d = R; d >= 1; --d) {
        const double wd = wl[d];
#pragma unroll
        for (int j = 0; j < FS_RUN; ++j) acc[j] += (lo[j] + hi[j]) * wd;
        // d -> d - 1: x[c_j - (d-1)] = x[c_{j+1} - d], x[c_j + (d-1)] = x[c_{j-1} + d]
#pragma unroll
        for (int j = 0; j < FS_RUN - 1; ++j) lo[j] = lo[j + 1];
        lo[FS_RUN - 1] = col[(FS_RUN - d) * FS_LD];
#pragma unroll
        for (int j = FS_RUN - 1; j > 0; --j) hi[j] = hi[j - 1];
        hi[0] = col[(d - 1) * FS_LD];
    }
    const int line = l0 + l;
    if (line < m) {
#pragma unroll
        for (int j = 0; j < FS_RUN; ++j) {
            const int pos = a0 + grp * FS_RUN + j;
            if (pos < n)
                out[(int64_t)b * H * W + (AXIS == 0 ? (int64_t)pos * W + line : (int64_t)line * W + pos)] =
                    apply_scale ? acc[j] * scale : acc[j];
        }
    }
}

extern "C" int uda_field_smooth(const double* noise, int B, int H, int W, const double* weights_dev, int radius, double alpha,
                                double* tmp, double* out, void* stream) {
    UDA_REQUIRE(noise && weights_dev && tmp && out && B > 0 && H > 0 && W > 0 && radius >= 1, "uda_field_smooth: bad args");
    const size_t lds = ((size_t)(FS_TILE + 2 * radius) * FS_LD + radius + 1) * sizeof(double);
    UDA_REQUIRE(lds <= 160 * 1024, "uda_field_smooth: radius %d needs %zu B of LDS (limit 160 KiB: sigma up to ~61, sides up to ~768)", radius, lds);
    hipStream_t st = (hipStream_t)stream;
    static bool configured_dev[UDA_MAX_DEVICES] = {};       // hipFuncSetAttribute is per device
    bool& configured = configured_dev[uda_device_slot()];
    if (!configured) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(field_smooth_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(field_smooth_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess || e1 != hipSuccess) return uda_set_error("uda_field_smooth: cannot reserve LDS: %s", hipGetErrorString(e0 != hipSuccess ? e0 : e1));
        configured = true;
    }
    hipLaunchKernelGGL(field_smooth_kernel<0>, dim3(uda_cdiv(H, FS_TILE), uda_cdiv(W, FS_LINES), B), dim3(FS_GROUPS * FS_LINES), lds, st,
                       noise, H, W, weights_dev, radius, 1.0, 0, tmp);
    UDA_LAUNCH_CHECK("field_smooth axis 0");
    hipLaunchKernelGGL(field_smooth_kernel<1>, dim3(uda_cdiv(W, FS_TILE), uda_cdiv(H, FS_LINES), B), dim3(FS_GROUPS * FS_LINES), lds, st,
                       tmp, H, W, weights_dev, radius, alpha, 1, out);
    UDA_LAUNCH_CHECK("field_smooth axis 1");
    return 0;
}

// map_coordinates(order=1): bilinear sample at (h + dx[h,w], w + dy[h,w]); the image takes 0 outside ('constant'), the
// label its nearest edge value ('nearest'); results are rounded to uint8 as scipy does for integer outputs.  `apply[b]` = 0 copies the
// sample through (the transform fires with p = 0.5 per sample).
__global__ __launch_bounds__(256) void elastic_warp_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lab,
                                                           const double* __restrict__ dx, const double* __restrict__ dy,
                                                           const uint8_t* __restrict__ apply, int H, int W,
                                                           uint8_t* __restrict__ img_out, uint8_t* __restrict__ lab_out) {
    const int b = blockIdx.z, h = blockIdx.y, w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    const int64_t o = ((int64_t)b * H + h) * W + w;
    const uint8_t* I = img + (int64_t)b * H * W * 3;
    const uint8_t* L = lab + (int64_t)b * H * W;
    if (apply && !apply[b]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) img_out[o * 3 + c] = I[((int64_t)h * W + w) * 3 + c];
        lab_out[o] = L[(int64_t)h * W + w];
        return;
    }
    const double y = (double)h + dx[o], x = (double)w + dy[o];      // the reference adds dx to the ROW index
    const double fy = floor(y), fx = floor(x);
    const int y0 = (int)fy, x0 = (int)fx;
    const double ty = y - fy, tx = x - fx;
    const double wy[2] = {1.0 - ty, ty}, wx[2] = {1.0 - tx, tx};
    auto pix = [&](int yy, int xx, int c) -> double {
        return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (double)I[((int64_t)yy * W + xx) * 3 + c] : 0.0;
    };
    // scipy's 'constant' mode: a coordinate outside [0, n-1] gives cval (no blending with the edge); every tap is
    // value * row weight * column weight, summed in (0,0) (0,1) (1,0) (1,1) order, and integer outputs are ROUNDED (+0.5)
    const bool inside = y >= 0.0 && y <= (double)(H - 1) && x >= 0.0 && x <= (double)(W - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = 0.0;
        if (inside) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) v += pix(y0 + i, x0 + j, c) * wy[i] * wx[j];
        }
        img_out[o * 3 + c] = (uint8_t)(v + 0.5);
    }
    auto labv = [&](int yy, int xx) -> double {
        yy = min(max(yy, 0), H - 1);
        xx = min(max(xx, 0), W - 1);
        return (double)L[(int64_t)yy * W + xx];
    };
    const double yc = fmin(fmax(y, 0.0), (double)(H - 1)), xc = fmin(fmax(x, 0.0), (double)(W - 1));    // 'nearest': clamp the coordinate
    const double gy = floor(yc), gx = floor(xc);
    const int ly = (int)gy, lx = (int)gx;
    const double sy = yc - gy, sx = xc - gx;
    const double vy[2] = {1.0 - sy, sy}, vx[2] = {1.0 - sx, sx};
    double lv = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) lv += labv(ly + i, lx + j) * vy[i] * vx[j];
    lab_out[o] = (uint8_t)(lv + 0.5);
}

extern "C" int uda_elastic_warp(const uint8_t* image_hwc, const uint8_t* label, const double* dx, const double* dy,
                                const uint8_t* apply, int B, int H, int W, uint8_t* image_out, uint8_t* label_out, void* stream) {
    UDA_REQUIRE(image_hwc && label && dx && dy && image_out && label_out && B > 0 && H > 1 && W > 1, "uda_elastic_warp: bad args");
    UDA_REQUIRE(image_hwc != image_out && label != label_out, "uda_elastic_warp: cannot run in place");
    hipLaunchKernelGGL(elastic_warp_kernel, dim3(uda_cdiv(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, image_hwc, label, dx, dy,
                       apply, H, W, image_out, label_out);
    UDA_LAUNCH_CHECK("elastic_warp");
    return 0;
}

// ------------------------------------------------------------------------------------------ photometric transforms on uint8 batches
// custom_transforms.py:150-250 of the reference (add_salt_pepper_noise, adjust_light, eraser), applied in that order with the
// per-sample random parameters the dataloader workers drew: the positions and the value (1 = "salt", 0 = "pepper": the
// reference writes 1, not 255) of the noisy pixels, the 256-entry gamma table (identity when the transform did not fire), the
// erased box and its grey level.
__global__ __launch_bounds__(256) void salt_pepper_kernel(uint8_t* __restrict__ img, int H, int W, const int* __restrict__ pos,
                                                          const int* __restrict__ cnt, const int* __restrict__ value, int maxn) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt[b] || i >= maxn) return;
    const int y = pos[((int64_t)b * maxn + i) * 2], x = pos[((int64_t)b * maxn + i) * 2 + 1];
    if (y < 0 || y >= H || x < 0 || x >= W) return;
    uint8_t* p = img + (((int64_t)b * H + y) * W + x) * 3;
    const uint8_t v = (uint8_t)value[b];
    p[0] = v; p[1] = v; p[2] = v;
}

__global__ __launch_bounds__(256) void lut_erase_kernel(uint8_t* __restrict__ img, int H, int W, const uint8_t* __restrict__ lut,
                                                        const int* __restrict__ box) {
    __shared__ uint8_t t[256];
    const int b = blockIdx.y;
    t[threadIdx.x] = lut[b * 256 + threadIdx.x];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    const int y = (int)(p / W), x = (int)(p % W);
    const int top = box[b * 5], left = box[b * 5 + 1], bh = box[b * 5 + 2], bw = box[b * 5 + 3];
    uint8_t* px = img + ((int64_t)b * H * W + p) * 3;
    if (bh > 0 && y >= top && y < top + bh && x >= left && x < left + bw) {
        const uint8_t v = (uint8_t)box[b * 5 + 4];
        px[0] = v; px[1] = v; px[2] = v;
    } else {
        px[0] = t[px[0]]; px[1] = t[px[1]]; px[2] = t[px[2]];
    }
}

extern "C" int uda_photometric_u8(uint8_t* image_hwc, int B, int H, int W, const int* sp_pos, const int* sp_count,
                                  const int* sp_value, int sp_max, const uint8_t* lut, const int* erase_box, void* stream) {
    UDA_REQUIRE(image_hwc && sp_pos && sp_count && sp_value && lut && erase_box && B > 0 && H > 0 && W > 0 && sp_max >= 0,
                "uda_photometric_u8: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (sp_max > 0) {
        hipLaunchKernelGGL(salt_pepper_kernel, dim3(uda_cdiv(sp_max, 256), B), dim3(256), 0, st, image_hwc, H, W, sp_pos, sp_count, sp_value, sp_max);
        UDA_LAUNCH_CHECK("salt_pepper");
    }
    hipLaunchKernelGGL(lut_erase_kernel, dim3(uda_cdiv((int64_t)H * W, 256), B), dim3(256), 0, st, image_hwc, H, W, lut, erase_box);
    UDA_LAUNCH_CHECK("lut_erase");
    return 0;
}
