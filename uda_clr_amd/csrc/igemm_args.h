// Kernel-argument blocks shared by igemm_conv.hip (single-role kernels, narrow tiles) and
// igemm_ws.hip (warp-specialised kernels, wide tiles).
#pragma once
#include "common.h"

struct ConvKArgs {
    uda_src_t src;
    const float* w;
    int Cout, ksize, dil, Kc, Ktot;
    int cen;       // tap (kh, kw) reads pixel offset ((kh - cen) * dil, (kw - cen) * dil): 1 for 3x3, 0|1 for 2x2
    const float* bias;
    const float* addend;
    int64_t ld_add;
    float* y;
    int64_t ldy;
    double* stats; // [UDA_STAT_SLOTS][2][Cout] or null (fp64 atomics)
    int nMt, nNt;
    int debug;     // diagnostics only (UDA_WS_DEBUG): bit0 skip MFMAs, bit1 skip loader work
    int x3;        // UDA_MFMA_BF16X3: wide tiles on the bf16 pipe by exact 3-way splitting (igemm_x3.hip)
    int stride, Ho, Wo;   // output grid: pixel (n, oh, ow) reads the taps around input pixel (n, oh * stride, ow * stride); wide-tile kernels only
};

struct WgradKArgs {
    uda_src_t src;
    const float* dy;
    int64_t lddy;
    int Cout, ksize, dil, Kc, Jtot;
    int cen;
    float* slab;      // [S][Cout][Jtot]
    int nCot, nJt, chunks_per_split, nchunks;
    int x3;
    int stride, Ho, Wo;   // grid of dy (see ConvKArgs); wide-tile kernels only
};

#define IG_BK 32
#define IG_LD 36
#define WG_BKP 32

int launch_conv_ws(ConvKArgs& k, int64_t P, hipStream_t st);          // 128 x {128,256} tiles
int launch_wgrad_ws(WgradKArgs& k, int S, bool big, hipStream_t st);  // 128 x 128 or 256 x 256 tiles
bool conv_x3_eligible(const ConvKArgs& k);
bool wgrad_x3_eligible(int Cin, int Cout, int ksize, int64_t P);
int launch_wgrad_x3(const WgradKArgs& k, int64_t P, int S_max, const void* x3_src, const void* x3_dy, hipStream_t st, int& S_out);   // bf16x3 weight gradient
int launch_conv_x3(ConvKArgs& k, int64_t P, const void* x3_src, const void* x3_w, hipStream_t st, void* ws, uint64_t ws_bytes);
uint64_t conv_x3_workspace_bytes(const ConvKArgs& k, int64_t P);      // bytes the tail split of this conv wants (0: none)   // bf16x3 forward / dgrad (igemm_x3.hip)
