// Calibration of rocprofv3's WRITE_SIZE counter on MI355X (TEST INFRASTRUCTURE):  three kernels that write a known number of bytes,
//   A  4 B per lane, a wave-instruction covers 256 contiguous bytes
//   B  16 B per lane, a wave-instruction covers 1024 contiguous bytes
//   C  the conv epilogue's pattern: 4 B per lane, lanes 0-31 one 128-byte segment of a row, lanes 32-63 the same columns 4 rows on,
//      rows 1 KiB apart (Cout = 256), every row segment written exactly once
// each over the same 256 MiB buffer.  Build: hipcc --offload-arch=gfx950 -O3 tests/tools/write_probe.hip -o tests/bin/write_probe
// Run on the GPU box:  rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -o w -- tests/bin/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void write_a(float* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1.0f;
}
__global__ void write_b(float4* p, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// one wave per 32 x 32 block of a [rows][256] matrix: 16 store instructions, instruction r writes rows (r&3) + 8*(r>>2) + 4*(lane>>5)
__global__ void write_c(float* p, int rows) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int rb = wave / 8, cb = wave % 8;                  // 8 column blocks of 32
    if (rb * 32 >= rows) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        p[(size_t)row * 256 + cb * 32 + (lane & 31)] = (float)r;
    }
}

int main() {
    const size_t bytes = 256u << 20, n = bytes / 4;
    float* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipMemset(d, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(write_a, dim3((n + 255) / 256), dim3(256), 0, 0, d, n);
        hipLaunchKernelGGL(write_b, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (float4*)d, n / 4);
        const int rows = (int)(n / 256);
        hipLaunchKernelGGL(write_c, dim3((rows / 32) * 8 * 64 / 256), dim3(256), 0, 0, d, rows);
    }
    hipDeviceSynchronize();
    printf("each kernel wrote %zu bytes\n", bytes);
    hipFree(d);
    return 0;
}
