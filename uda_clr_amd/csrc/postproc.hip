// Evaluation post-processing of the predicted cup / disc probability maps on the device.
//
//   uda_postprocess   utils/Utils.py:427-463 of the reference (postprocessing + get_largest_fillhole), per image and channel:
//       threshold -> 5 x scipy.signal.medfilt2d(mask, 7) -> skimage binary_erosion(diamond(7)) -> largest connected
//       component (skimage.measure.label, 8-connectivity; regionprops areas, first maximum) -> scipy binary_fill_holes.
//   Restated semantics (tests compare with the scipy calls bit for bit):
//     * medfilt2d zero-pads, so on a 0/1 image the 7x7 median is 1 exactly when at least 25 of the 49 (in-image) values are 1;
//     * skimage's erosion treats pixels outside the image as set (border_value = True): erosion by the L1 ball of radius 7;
//     * components: every set pixel starts with its own raster index + 1 and repeatedly takes the minimum over its 3x3
//       neighbourhood; at the fixed point a component carries the index of its first pixel in raster order, which is also the
//       order in which skimage numbers components, so "first maximum of the areas" = smallest such index among the largest;
//     * hole filling: background not 4-connected to the image border becomes foreground.
//   Both propagations run TILE x TILE blocks to their local fixed point in LDS per launch (a launch moves information across
//   whole tiles, not single pixels) for a bounded number of launches; the last launch reports whether anything still changed,
//   and the entry point returns that count in *not_converged (device int) so a caller can ask for more sweeps.
//
// Latency-bound byte work on small planes (validation only, not on the training path).
#include "common.h"

#define PP_T 32

__global__ __launch_bounds__(256) void pp_threshold_kernel(const float* __restrict__ pred, int64_t HW, float thr_cup, float thr_disc,
                                                           uint8_t* __restrict__ mask) {
    const int plane = blockIdx.y;                    // b * 2 + channel (0 cup, 1 disc)
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    mask[plane * HW + p] = pred[plane * HW + p] > ((plane & 1) ? thr_disc : thr_cup) ? 1 : 0;
}

// window sum over (2R+1)^2 with zeros outside the image; out = sum >= need
__global__ __launch_bounds__(256) void pp_median7_kernel(const uint8_t* __restrict__ in, int H, int W, uint8_t* __restrict__ out) {
    __shared__ uint8_t t[PP_T + 6][PP_T + 6 + 2];
    const int plane = blockIdx.z, h0 = blockIdx.y * PP_T, w0 = blockIdx.x * PP_T;
    const uint8_t* I = in + (int64_t)plane * H * W;
    for (int e = threadIdx.x; e < (PP_T + 6) * (PP_T + 6); e += 256) {
        const int r = e / (PP_T + 6), c = e % (PP_T + 6);
        const int h = h0 + r - 3, w = w0 + c - 3;
        t[r][c] = (h >= 0 && h < H && w >= 0 && w < W) ? I[(int64_t)h * W + w] : 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PP_T * PP_T; e += 256) {
        const int r = e / PP_T, c = e % PP_T;
        const int h = h0 + r, w = w0 + c;
        if (h >= H || w >= W) continue;
        int s = 0;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) s += t[r + dy][c + dx];
        out[((int64_t)plane * H + h) * W + w] = s >= 25 ? 1 : 0;
    }
}

// erosion by the L1 ball of radius R, pixels outside the image count as set
#define PP_ER 7
__global__ __launch_bounds__(256) void pp_erode_kernel(const uint8_t* __restrict__ in, int H, int W, uint8_t* __restrict__ out) {
    __shared__ uint8_t t[PP_T + 2 * PP_ER][PP_T + 2 * PP_ER + 2];
    const int plane = blockIdx.z, h0 = blockIdx.y * PP_T, w0 = blockIdx.x * PP_T;
    const uint8_t* I = in + (int64_t)plane * H * W;
    for (int e = threadIdx.x; e < (PP_T + 2 * PP_ER) * (PP_T + 2 * PP_ER); e += 256) {
        const int r = e / (PP_T + 2 * PP_ER), c = e % (PP_T + 2 * PP_ER);
        const int h = h0 + r - PP_ER, w = w0 + c - PP_ER;
        t[r][c] = (h >= 0 && h < H && w >= 0 && w < W) ? I[(int64_t)h * W + w] : 1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PP_T * PP_T; e += 256) {
        const int r = e / PP_T, c = e % PP_T;
        const int h = h0 + r, w = w0 + c;
        if (h >= H || w >= W) continue;
        uint8_t all = 1;
        for (int dy = -PP_ER; dy <= PP_ER; ++dy) {
            const int span = PP_ER - (dy < 0 ? -dy : dy);
            for (int dx = -span; dx <= span; ++dx) all &= t[r + PP_ER + dy][c + PP_ER + dx];
        }
        out[((int64_t)plane * H + h) * W + w] = all;
    }
}

__global__ __launch_bounds__(256) void pp_label_init_kernel(const uint8_t* __restrict__ mask, int64_t HW, int* __restrict__ lab) {
    const int plane = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    lab[plane * HW + p] = mask[plane * HW + p] ? (int)p + 1 : 0;
}

// One sweep of the min-label propagation (CONN8 = 1: 3x3 neighbourhood, 0: 4-neighbourhood).  Labels 0 = not part of the
// propagating set.  The tile (with a 1-pixel halo of the INPUT labels) iterates in LDS until nothing in it changes.
template <int CONN8>
__global__ __launch_bounds__(256) void pp_propagate_kernel(const int* __restrict__ in, int H, int W, int* __restrict__ out,
                                                           int* __restrict__ changed) {
    __shared__ int t[PP_T + 2][PP_T + 2 + 1];
    __shared__ int again;
    const int plane = blockIdx.z, h0 = blockIdx.y * PP_T, w0 = blockIdx.x * PP_T;
    const int* I = in + (int64_t)plane * H * W;
    for (int e = threadIdx.x; e < (PP_T + 2) * (PP_T + 2); e += 256) {
        const int r = e / (PP_T + 2), c = e % (PP_T + 2);
        const int h = h0 + r - 1, w = w0 + c - 1;
        t[r][c] = (h >= 0 && h < H && w >= 0 && w < W) ? I[(int64_t)h * W + w] : 0;
    }
    __syncthreads();
    bool any = false;
    for (int sweep = 0; sweep < 4 * PP_T; ++sweep) {          // bounded: a tile's longest in-tile path is < 4 * PP_T steps for blob-like sets
        if (threadIdx.x == 0) again = 0;
        __syncthreads();
        bool mine = false;
        for (int e = threadIdx.x; e < PP_T * PP_T; e += 256) {
            const int r = e / PP_T + 1, c = e % PP_T + 1;
            const int v = t[r][c];
            if (v == 0) continue;
            int m = v;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!CONN8 && dy != 0 && dx != 0) continue;
                    const int n = t[r + dy][c + dx];
                    if (n > 0 && n < m) m = n;
                }
            if (m < v) {
                t[r][c] = m;            // in-place (Gauss-Seidel): only ever lowers a label towards its component minimum
                mine = true;
            }
        }
        if (mine) again = 1;
        __syncthreads();
        const bool more = again != 0;
        any = any || more;
        __syncthreads();
        if (!more) break;
    }
    for (int e = threadIdx.x; e < PP_T * PP_T; e += 256) {
        const int r = e / PP_T, c = e % PP_T;
        const int h = h0 + r, w = w0 + c;
        if (h < H && w < W) out[((int64_t)plane * H + h) * W + w] = t[r + 1][c + 1];
    }
    if (any && threadIdx.x == 0) atomicAdd(changed, 1);
}

__global__ __launch_bounds__(256) void pp_area_kernel(const int* __restrict__ lab, int64_t HW, int* __restrict__ cnt) {
    const int plane = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int v = lab[plane * HW + p];
    if (v > 0) atomicAdd(&cnt[plane * HW + v - 1], 1);
}

// per plane: the label (component minimum index + 1) with the largest area, smallest label on ties; 0 if the plane is empty
__global__ __launch_bounds__(1024) void pp_argmax_kernel(const int* __restrict__ cnt, int64_t HW, int* __restrict__ best) {
    __shared__ long long red[1024];
    const int plane = blockIdx.x;
    long long key = 0;                              // (area << 32) | (0xffffffff - index): larger area wins, then smaller index
    for (int64_t p = threadIdx.x; p < HW; p += 1024) {
        const int a = cnt[plane * HW + p];
        if (a > 0) {
            const long long k = ((long long)a << 32) | (long long)(0xffffffffu - (unsigned)p);
            if (k > key) key = k;
        }
    }
    red[threadIdx.x] = key;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) best[plane] = red[0] == 0 ? 0 : (int)(0xffffffffu - (unsigned)(red[0] & 0xffffffffll)) + 1;
}

// keep the largest component; start the border flood of its complement: reach = 1 on background pixels of the image border
__global__ __launch_bounds__(256) void pp_keep_kernel(const int* __restrict__ lab, const int* __restrict__ best, int H, int W,
                                                      uint8_t* __restrict__ keep, int* __restrict__ reach) {
    const int plane = blockIdx.y;
    const int64_t HW = (int64_t)H * W, p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int b = best[plane];
    const uint8_t k = (b > 0 && lab[plane * HW + p] == b) ? 1 : 0;
    keep[plane * HW + p] = k;
    const int h = (int)(p / W), w = (int)(p % W);
    // flood labels: 1 = reached from the border, 2 = background not (yet) reached, 0 = foreground (does not propagate)
    reach[plane * HW + p] = k ? 0 : ((h == 0 || w == 0 || h == H - 1 || w == W - 1) ? 1 : 2);
}

__global__ __launch_bounds__(256) void pp_fill_kernel(const uint8_t* __restrict__ keep, const int* __restrict__ reach, int64_t HW,
                                                      uint8_t* __restrict__ out) {
    const int plane = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    out[plane * HW + p] = (keep[plane * HW + p] || reach[plane * HW + p] == 2) ? 1 : 0;      // holes: background never reached
}

extern "C" size_t uda_postprocess_workspace_bytes(int B, int H, int W) {
    const size_t planes = (size_t)2 * B, HW = (size_t)H * W;
    return planes * HW * (2 * sizeof(uint8_t) + 3 * sizeof(int)) + planes * sizeof(int) + 256;
}

extern "C" int uda_postprocess(const float* pred, int B, int H, int W, float thr_cup, float thr_disc, int sweeps,
                               uint8_t* out, int* not_converged, void* workspace, size_t workspace_bytes, void* stream) {
    UDA_REQUIRE(pred && out && not_converged && workspace && B > 0 && H > 0 && W > 0 && sweeps >= 1, "uda_postprocess: bad args");
    UDA_REQUIRE((int64_t)H * W < ((int64_t)1 << 30), "uda_postprocess: plane too large for 32-bit labels");
    UDA_REQUIRE(workspace_bytes >= uda_postprocess_workspace_bytes(B, H, W), "uda_postprocess: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int planes = 2 * B;
    const int64_t HW = (int64_t)H * W;
    // workspace: labA, labB, cnt (int planes), best (int per plane), mA, mB (byte planes)
    int* labA = (int*)workspace;
    int* labB = labA + planes * HW;
    int* cnt = labB + planes * HW;
    int* best = cnt + planes * HW;
    uint8_t* mA = (uint8_t*)(best + planes + 16);
    uint8_t* mB = mA + planes * HW;
    const dim3 flat(uda_cdiv(HW, 256), planes), tiles(uda_cdiv(W, PP_T), uda_cdiv(H, PP_T), planes);
    hipLaunchKernelGGL(pp_threshold_kernel, flat, dim3(256), 0, st, pred, HW, thr_cup, thr_disc, mA);
    for (int i = 0; i < 5; ++i) {
        hipLaunchKernelGGL(pp_median7_kernel, tiles, dim3(256), 0, st, mA, H, W, mB);
        uint8_t* t = mA; mA = mB; mB = t;
    }
    hipLaunchKernelGGL(pp_erode_kernel, tiles, dim3(256), 0, st, mA, H, W, mB);          // eroded mask in mB
    hipLaunchKernelGGL(pp_label_init_kernel, flat, dim3(256), 0, st, mB, HW, labA);
    UDA_LAUNCH_CHECK("postprocess (masks)");
    if (hipMemsetAsync(not_converged, 0, 2 * sizeof(int), st) != hipSuccess) return uda_set_error("uda_postprocess: memset failed");
    int* scratch_flag = best + planes;             // absorbs the flags of all but the last sweep
    for (int i = 0; i <= sweeps; ++i) {            // the extra sweep only reports: anything it still changes = not converged
        hipLaunchKernelGGL(pp_propagate_kernel<1>, tiles, dim3(256), 0, st, labA, H, W, labB, i == sweeps ? not_converged : scratch_flag);
        int* t = labA; labA = labB; labB = t;
    }
    if (hipMemsetAsync(cnt, 0, planes * HW * sizeof(int), st) != hipSuccess) return uda_set_error("uda_postprocess: memset failed");
    hipLaunchKernelGGL(pp_area_kernel, flat, dim3(256), 0, st, labA, HW, cnt);
    hipLaunchKernelGGL(pp_argmax_kernel, dim3(planes), dim3(1024), 0, st, cnt, HW, best);
    hipLaunchKernelGGL(pp_keep_kernel, flat, dim3(256), 0, st, labA, best, H, W, mA, labB);   // keep mask in mA, flood labels in labB
    UDA_LAUNCH_CHECK("postprocess (components)");
    int* fa = labB;
    int* fb = labA;
    for (int i = 0; i <= sweeps; ++i) {
        hipLaunchKernelGGL(pp_propagate_kernel<0>, tiles, dim3(256), 0, st, fa, H, W, fb, i == sweeps ? not_converged + 1 : scratch_flag);
        int* t = fa; fa = fb; fb = t;
    }
    hipLaunchKernelGGL(pp_fill_kernel, flat, dim3(256), 0, st, mA, fa, HW, out);
    UDA_LAUNCH_CHECK("postprocess (fill)");
    return 0;
}
