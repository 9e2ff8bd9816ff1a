// Shared device/host helpers for the gfx950 kernels of libuda_clr_hip.so.
// Target: MI355X (CDNA4, wave64, 256 CUs in 8 XCDs).  No other architecture is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/uda_clr_hip.h"

// ------------------------------------------------------------------------------------ errors
int uda_set_error(const char* fmt, ...);

#define UDA_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return uda_set_error(__VA_ARGS__);         \
    } while (0)

#define UDA_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess)                                                        \
            return uda_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline bool uda_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int uda_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------ device
#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_RELU6 2

// none / ReLU / ReLU6 as one clamp (v_med3_f32) between bounds that depend only on the launch's activation code
__device__ __forceinline__ float uda_act(float v, int act) {
    const float lo = act == ACT_NONE ? -INFINITY : 0.f, hi = act == ACT_RELU6 ? 6.f : INFINITY;
    return __builtin_amdgcn_fmed3f(v, lo, hi);
}
// derivative gate of the activation at pre-activation value a (PyTorch: relu a>0, hardtanh 0<a<6)
__device__ __forceinline__ float uda_act_gate(float a, int act) {
    if (act == ACT_RELU) return a > 0.f ? 1.f : 0.f;
    if (act == ACT_RELU6) return (a > 0.f && a < 6.f) ? 1.f : 0.f;
    return 1.f;
}

// Host side: slot of the current device for per-device one-time setup (hipFuncSetAttribute is per device; one process may drive
// several, although the trainers use one process per GPU).
#define UDA_MAX_DEVICES 16
static inline int uda_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d % UDA_MAX_DEVICES;
}

// Blocks b and b+8 share an XCD (round-robin dispatch).  Give every XCD one contiguous chunk of
// the logical tile order so neighbouring tiles (shared halo rows / shared weight panels) hit the
// same 4 MiB L2.  Bijective for any nwg (cdna_hip_programming.md, 8-phase template notes).
__device__ __forceinline__ int uda_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Per-channel transform descriptor prepared per thread for 4 consecutive channels.
struct Xf4 {
    float sc[4], sh[4];
};

__device__ __forceinline__ void uda_load_xf4(Xf4& t, const float* scale, const float* shift, int c0, int C) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = scale != nullptr && (c0 + j) < C;
        t.sc[j] = ok ? scale[c0 + j] : 1.f;
        t.sh[j] = ok ? shift[c0 + j] : 0.f;
    }
}

__device__ __forceinline__ bool uda_aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ float4 uda_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void uda_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
