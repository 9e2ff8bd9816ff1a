// Which bf16 MFMA shape should the bf16x3 conv kernel's math waves use?  (TEST TOOL, GPU box; not part of the product.)
//
// The six piece products of one fp32 product block can be issued as
//   A: six v_mfma_f32_32x32x16_bf16 per 32 x 32 block and 16-deep chunk (what igemm_x3.hip does: 8 math waves of 64 x 128 inside a
//      12-wave workgroup, A fragments of one row block held, B fragments re-read per row block), or
//   B: three v_mfma_f32_16x16x32_bf16 per 16 x 16 block: K = 32 holds TWO 16-deep piece products ([a_i | a_j] . [b_m | b_n]);
//      8 waves of 64 x 128 (no loader waves: 256 VGPRs per wave), the A fragments of all four row blocks held (48 VGPRs), B streamed.
// Both run the same number of matrix-pipe cycles per chunk (1536 per wave).  This probe runs ONLY the math waves' loop (fragment
// reads from a resident LDS tile image + MFMAs + one barrier per chunk) on random and on zero operands and prints the rate each
// shape sustains - the ceiling a rewrite of the kernel around shape B could reach, before any of it is written.
//
//   hipcc --offload-arch=gfx950 -O3 -o tests/bin/mfma_shape_probe tests/tools/mfma_shape_probe.hip && tests/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ROWB 112              // bytes per LDS row: 3 pieces x 16 bf16 + 16 bytes of padding (igemm_x3.hip's image)
#define TILE_ROWS 512         // 256 A rows + 256 B rows

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// fill the two tile images with bf16 values in [-2, 2) (random) or zeros
__device__ void fill_lds(char* smem, int nthreads, bool zero) {
    uint16_t* s = reinterpret_cast<uint16_t*>(smem);
    const int n = 2 * TILE_ROWS * ROWB / 2;
    for (int i = threadIdx.x; i < n; i += nthreads) {
        const uint32_t h = hash32(i * 2654435761u + blockIdx.x * 97u);
        // sign | exponent 126..128 | 7 random mantissa bits
        const uint16_t v = (uint16_t)(((h & 1u) << 15) | ((126u + (h >> 1) % 3u) << 7) | ((h >> 8) & 0x7fu));
        s[i] = zero ? 0 : v;
    }
}

// ---------------------------------------------------------------- shape A: 32x32x16, 12-wave workgroup (4 waves only wait at the barriers)
__global__ __launch_bounds__(768) void probe_a(float* out, int chunks, int zero) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill_lds(smem, 768, zero != 0);
    __syncthreads();
    if (wave >= 8) {
        for (int c = 0; c < chunks; ++c) __syncthreads();
        return;
    }
    const int wm = wave >> 1, wn = wave & 1;
    const int arow = wm * 64 + (lane & 31), brow = wn * 128 + (lane & 31), koff = 8 * (lane >> 5);
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __builtin_amdgcn_s_setprio(1);
    for (int c = 0; c < chunks; ++c) {
        const __bf16* As = reinterpret_cast<const __bf16*>(smem + (c & 1) * TILE_ROWS * ROWB);
        const __bf16* Bs = As + 256 * (ROWB / 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 af[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) af[q] = *reinterpret_cast<const bf16x8*>(&As[(arow + 32 * i) * (ROWB / 2) + q * 16 + koff]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("" ::: "memory");
                bf16x8 bq[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const bf16x8*>(&Bs[(brow + 32 * j) * (ROWB / 2) + q * 16 + koff]);
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
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

// ---------------------------------------------------------------- shape B: 16x16x32, 8-wave workgroup
// lane l of a 16x16x32 operand: row (or column) l & 15, k group g = l >> 4 (8 consecutive k each).  Groups 0, 1 take the first piece
// of the pair (k 0..15 of the chunk), groups 2, 3 the second: byte offset inside the row = piece * 32 + (g & 1) * 16.
__global__ __launch_bounds__(512) void probe_b(float* out, int chunks, int zero) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill_lds(smem, 512, zero != 0);
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4, oct = (g & 1) * 16, second = g >> 1;
    // piece pairs: A' = [a2|a1], [a0|a1], [a0|a0];  B' = [b0|b1], [b2|b0], [b1|b0]   ->  a2b0 + a1b1, a0b2 + a1b0, a0b1 + a0b0
    const int pa0 = (second ? 1 : 2) * 32 + oct, pa1 = (second ? 1 : 0) * 32 + oct, pa2 = 0 * 32 + oct;
    const int pb0 = (second ? 1 : 0) * 32 + oct, pb1 = (second ? 0 : 2) * 32 + oct, pb2 = (second ? 0 : 1) * 32 + oct;
    const int abase = (wm * 64 + l15) * ROWB, bbase = (256 + wn * 128 + l15) * ROWB;
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j)
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    __builtin_amdgcn_s_setprio(1);
    for (int c = 0; c < chunks; ++c) {
        const char* T = smem + (c & 1) * TILE_ROWS * ROWB;
        bf16x8 af[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const bf16x8*>(T + abase + 16 * i * ROWB + pa0);
            af[i][1] = *reinterpret_cast<const bf16x8*>(T + abase + 16 * i * ROWB + pa1);
            af[i][2] = *reinterpret_cast<const bf16x8*>(T + abase + 16 * i * ROWB + pa2);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("" ::: "memory");
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(T + bbase + 16 * j * ROWB + pb0);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(T + bbase + 16 * j * ROWB + pb1);
            const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(T + bbase + 16 * j * ROWB + pb2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], b0, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], b1, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], b2, acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j)
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F>
static int run(const char* name, F fn, int threads, int zero, float* out) {
    const int wgs = 2048, chunks = 144;           // conv4 at B = 16: 1024 tiles x 2 column tiles ... 8 rounds on 256 CUs
    const size_t lds = 2 * TILE_ROWS * ROWB;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fn, dim3(wgs), dim3(threads), lds, 0, out, chunks, zero);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3(wgs), dim3(threads), lds, 0, out, chunks, zero);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops_eq = 2.0 * wgs * 256.0 * 256.0 * 16.0 * chunks;       // fp32-equivalent FLOPs (one of the six products counted)
    const double cyc = (wgs / 256.0) * chunks * 3072.0;                      // matrix-pipe cycles per SIMD
    printf("%-34s %-7s %.3f ms  %6.1f TF-eq  %6.0f bf16 TF  clock implied by the cycle floor %.2f GHz\n", name, zero ? "zeros" : "random", ms,
           flops_eq / ms / 1e9, 6 * flops_eq / ms / 1e9, cyc / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    float* out;
    CK(hipMalloc(&out, (size_t)2048 * 768 * sizeof(float)));
    printf("math-wave loop only (LDS-resident tile images, no global loads): 2048 workgroups x 144 chunks of a 256 x 256 tile\n");
    for (int zero = 0; zero < 2; ++zero) {
        if (run("A  6 x mfma_32x32x16 (12-wave WG)", probe_a, 768, zero, out)) return 1;
        if (run("B  3 x mfma_16x16x32 ( 8-wave WG)", probe_b, 512, zero, out)) return 1;
    }
    CK(hipFree(out));
    return 0;
}
