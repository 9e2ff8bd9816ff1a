// Probe (TEST INFRASTRUCTURE): raw buffer stores with a scalar row offset in soffset, as the bf16x3 epilogues issue them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float* out, int rows, int ld, int mode) {
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, rows * ld * 4, 0x00020000);
    const int rbase = 4 * (lane >> 5), col = lane & 31;
    const int voff = (rbase * ld + col) * 4, ld4 = ld * 4;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int rl = (q & 3) + 8 * (q >> 2);
        const float v = 100.f * (rbase + rl) + col;
        if (mode == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, rl * ld4, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff + rl * ld4, 0, 0);
    }
}
int main() {
    const int rows = 32, ld = 2304;
    float* d; hipMalloc(&d, rows * ld * 4);
    std::vector<float> h(rows * ld);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, rows * ld * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rows, ld, mode);
        hipMemcpy(h.data(), d, rows * ld * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int rr = 0; rr < rows; ++rr) for (int c = 0; c < 32; ++c) if (h[rr * ld + c] != 100.f * rr + c) { if (bad < 6) printf("mode %d row %d col %d got %g\n", mode, rr, c, h[rr * ld + c]); ++bad; }
        printf("mode %d: %d wrong of %d\n", mode, bad, rows * 32);
    }
    return 0;
}
