/* libuda_clr_hip.so - C ABI of the MI355X (gfx950) kernels behind the UDA_CLR per-step hot path.
 *
 * The reference (fengweie/UDA_CLR) is pure Python/PyTorch: it has no FFI of its own.  Each entry
 * point below replaces the ATen operator(s) that the cited reference line executes; the
 * reference-side binding a maintainer would add is the ctypes stub in INTEGRATION.md
 * (uda_clr_amd/kernels.py is that stub, grown into a class).
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, a hipStream_t passed as void*; no C++/torch types.
 *   - every function returns 0 on success, a negative code on failure and never throws;
 *     uda_last_error() returns a thread-local message for the last failure.
 *   - kernels never allocate: outputs and workspaces are caller-owned (PyTorch caching allocator);
 *     workspace sizes come from the *_workspace_bytes queries.
 *   - launches are asynchronous on `stream`; functions are re-entrant (no global mutable state).
 *   - activations are NHWC fp32 matrices [P = N*H*W, ld] (channels fastest, ld % 4 == 0, base
 *     16-byte aligned, at least round4(C) readable floats per row).  uda_src_t adds the pending
 *     per-channel transform the consumer applies on load:
 *         u[p,c] = act(x[p,c]*scale[c] + shift[c]) * (mask[p,c] * mask_scale)
 */
#ifndef UDA_CLR_HIP_H
#define UDA_CLR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-channel statistics are accumulated with fp64 atomics into UDA_STAT_SLOTS replicas
 * (slot = workgroup index mod UDA_STAT_SLOTS, spreads contention); consumers sum the replicas. */
#define UDA_STAT_SLOTS 16

#define UDA_ACT_NONE 0
#define UDA_ACT_RELU 1
#define UDA_ACT_RELU6 2

typedef struct uda_src {
    const float* x;       /* [P, ldx] */
    int64_t ldx;
    int32_t N, H, W, C;
    const float* scale;   /* [C] or NULL (identity) */
    const float* shift;   /* [C] or NULL */
    int32_t act;          /* UDA_ACT_* */
    int32_t _pad;
    const uint8_t* mask;  /* [P, ldm] keep-mask (1 = keep) or NULL */
    int64_t ldm;
    float mask_scale;     /* 1/(1-p) */
    float _pad2;
} uda_src_t;

const char* uda_last_error(void);
int uda_version(void);

/* ---- weight re-layouts (tiny; once per step).  torch layout OIHW in. */
/* K order of one weight row: k = 1: [round4(I)];  k > 1 ("tap-chunked"): [nCC][k*k][32] with nCC = ceil(round4(I)/32),
 * i.e. element (tap t, channel c) at ((c/32)*k*k + t)*32 + c%32, zero padded: all taps of one 32-channel slice are
 * consecutive K-chunks of the implicit GEMM, so a workgroup re-reads its pixel strip while it is L2-resident;
 * with round4(I) < 32 the row stays tap-major and unpadded, [k*k][round4(I)].
 * out[O][row as above]                              - operand of uda_conv_fwd              */
int uda_relayout_ohwi(const float* w, int O, int I, int k, float* out, void* stream);
/* out[I][row over O as above], taps flipped        - operand of uda_conv_fwd used as dgrad */
int uda_relayout_dgrad(const float* w, int O, int I, int k, float* out, void* stream);
/* depthwise [C][1][3][3] -> [9][C] */
int uda_relayout_dw(const float* w, int C, float* out, void* stream);

/* ---- dense convolution (stride 1; stride 2 on the wide tiles) as MFMA implicit GEMM (1x1, 3x3 with any dilation, and 2x2).
 * Replaces F.conv2d at mobilenet.py:43,49,57, aspp.py:50-53,56,59, decoder.py:20,32,33,37,41 and,
 * with uda_relayout_dgrad weights, their input-gradient.
 *   y[p,co] = bias[co] + addend[p,co] + sum_{t,ci} u(p+off_t, ci) * w[co][t][ci]
 * stats (optional, double[UDA_STAT_SLOTS][2][Cout], ADDED into): sum and sum of squares of y
 * before addend. */
/* UDA_MFMA_F32: v_mfma_f32_32x32x2_f32, the exact fp32 fma chain.  UDA_MFMA_BF16X3: fp32 emulated on the bf16 matrix pipe:
 * every operand is split exactly into three bf16 pieces (a = a1 + a2 + a3) and the six piece products down to 2^-24 relative
 * are accumulated in fp32 (v_mfma_f32_32x32x16_bf16) - same fp32-level error as the fma chain (measured per call against
 * float64 by tests/noise_report.py), 2.67x its matrix-pipe rate.  Inputs, outputs, accumulators and statistics stay fp32/fp64. */
#define UDA_MFMA_F32 0
#define UDA_MFMA_BF16X3 1
typedef struct uda_conv_args {
    uda_src_t src;
    const float* w;        /* [Cout][row], the layout uda_relayout_ohwi writes */
    int32_t Cout, ksize, dil;
    int32_t origin;        /* ksize 2 only: tap (kh,kw) reads pixel (h + kh - origin, w + kw - origin); 0 = the
                              space-to-depth form of the 4x4 stride-2 convs of GAN.py:90-101, 1 = its input gradient.
                              ksize 3 is centred (pad = dil), ksize 1 has one tap. */
    const float* bias;     /* [Cout] or NULL */
    const float* addend;   /* [P, ld_add] or NULL (may alias y) */
    int64_t ld_add;
    float* y;              /* [P, ldy] */
    int64_t ldy;
    double* stats;         /* [UDA_STAT_SLOTS][2][Cout] or NULL */
    int32_t mfma;          /* UDA_MFMA_*: which matrix instructions the wide (MFMA-bound) tiles use; narrow kernels ignore it */
    int32_t stride;        /* 0 | 1: stride 1.  2 (resnet.py:66,93; wide-tile kernels only: Cout > 96, K > 192, else an error):
                              y, addend, stats live on the grid ((H-1)/2+1) x ((W-1)/2+1), row (n,oh,ow) is centred on src pixel (n,2oh,2ow) */
    const void* x3_src;    /* UDA_MFMA_BF16X3, when uda_conv_uses_x3(a): src packed by uda_x3_pack (transform already applied) ... */
    const void* x3_w;      /* ... and the weight rows w ([Cout] rows of the relayouted row length) packed by uda_x3_pack */
    void* workspace;       /* optional, 16-byte aligned: uda_conv_fwd_workspace_bytes(a) bytes let the bf16x3 kernel split the last, */
    uint64_t workspace_bytes; /* partly filled round of tiles over K (fp32 partial tiles, summed in a fixed order); NULL / too small: not split */
} uda_conv_args_t;
/* 1 when uda_conv_fwd will run these arguments on the bf16x3 wide-tile kernel and therefore needs x3_src / x3_w */
int uda_conv_uses_x3(const uda_conv_args_t* a);
/* bytes of workspace these arguments can make use of (0: none; the conv runs the same without it) */
uint64_t uda_conv_fwd_workspace_bytes(const uda_conv_args_t* a);
int uda_conv_fwd(const uda_conv_args_t* a, void* stream);
/* Operand packing for UDA_MFMA_BF16X3: every fp32 value as its three bf16 pieces, out[rows][ceil(C/16)][3][16] (96 contiguous
 * bytes per row and 16-wide block), rows = N*H*W of src, values = the TRANSFORMED ones act(x*scale+shift)*mask*mask_scale, so a
 * tensor is split once however many taps, workgroups or convolutions read it.  Weight rows: src with N = H = 1, W = rows,
 * C = row length, no transform.  uda_x3_packed_bytes(rows, C): size of out. */
uint64_t uda_x3_packed_bytes(int64_t rows, int K);
int uda_x3_pack(const uda_src_t* src, void* out, void* stream);

/* weight gradient of the same convolution: dw[co][ci][kh][kw] = sum_p dy[p,co]*u(p+off_t,ci) */
typedef struct uda_wgrad_args {
    uda_src_t src;
    const float* dy;       /* [P, lddy] */
    int64_t lddy;
    int32_t Cout, ksize, dil;
    int32_t origin;        /* as in uda_conv_args_t */
    float* dw;             /* OIHW, contiguous */
    float* workspace;
    uint64_t workspace_bytes;
    int32_t mfma;          /* UDA_MFMA_*, as in uda_conv_args_t */
    int32_t stride;        /* as in uda_conv_args_t: 2 = dy lives on the strided output grid (wide-tile kernels only) */
    const void* x3_src;    /* UDA_MFMA_BF16X3, when uda_conv_wgrad_uses_x3(a): src packed by uda_x3_pack (the forward conv's packed form) ... */
    const void* x3_dy;     /* ... and dy ([P] rows of Cout values) packed by uda_x3_pack */
} uda_wgrad_args_t;
uint64_t uda_conv_wgrad_workspace_bytes(int64_t P, int Cout, int Cin, int ksize);
int uda_conv_wgrad_uses_x3(const uda_wgrad_args_t* a);
int uda_conv_wgrad(const uda_wgrad_args_t* a, void* stream);

/* ---- depthwise 3x3 (mobilenet.py:39,53): stride 1|2, dilation 1|2, "pad 0 on a padded input".
 * border_mode 0: out-of-image taps read 0; 1: they read act(shift[c]) (quirk Q1). */
uint64_t uda_dwconv_workspace_bytes(int64_t Pout, int C);
int uda_dwconv_fwd(const uda_src_t* src, const float* w9c, int stride, int dil, int border_mode,
                   float* y, int64_t ldy, double* stats /* [SLOTS][2][C] or NULL */, void* stream);
int uda_dwconv_dgrad(const float* dy, int64_t lddy, const float* w9c, int C, int stride, int dil,
                     int N, int H, int W, float* dx, int64_t lddx, void* stream);
int uda_dwconv_wgrad(const uda_src_t* src, const float* dy, int64_t lddy, int stride, int dil,
                     int border_mode, float* dw /* [C][9] */, float* workspace,
                     uint64_t workspace_bytes, void* stream);

/* ---- stem conv 3x3 stride 2 pad 1, 3 -> 32, NCHW image in, NHWC out (mobilenet.py:10) */
uint64_t uda_stem_workspace_bytes(int64_t Pout);
int uda_stem_fwd(const float* x, int N, int H, int W, const float* w, float* y, int64_t ldy,
                 double* stats /* [SLOTS][2][32] or NULL */, void* stream);
int uda_stem_wgrad(const float* x, int N, int H, int W, const float* dy, int64_t lddy, float* dw,
                   float* workspace, uint64_t workspace_bytes, void* stream);

/* ---- ResNet-101 variant (BASELINE.json configs[4]; networks/backbone/resnet.py)
 * conv1 7x7 stride 2 pad 3, 3 -> 64, NCHW image in, NHWC out (resnet.py:59,114); W % 8 == 0 */
uint64_t uda_stem7_workspace_bytes(int64_t Pout);
int uda_stem7_fwd(const float* x, int N, int H, int W, const float* w, float* y, int64_t ldy,
                  double* stats /* [SLOTS][2][64] or NULL */, void* stream);
int uda_stem7_wgrad(const float* x, int N, int H, int W, const float* dy, int64_t lddy, float* dw,
                    float* workspace, uint64_t workspace_bytes, void* stream);
/* MaxPool2d(3, 2, 1) of the pending transform of src (resnet.py:63,117).  idx[p,c] = winning tap
 * kh*3+kw (first maximum in scan order, as ATen); the backward gathers dz through idx. */
int uda_maxpool_fwd(const uda_src_t* src, float* out, int64_t ldo, uint8_t* idx, int64_t ldi, void* stream);
int uda_maxpool_bwd(const float* dz, int64_t lddz, const uint8_t* idx, int64_t ldi, int N, int H, int W,
                    int C, float* du, int64_t ldu, void* stream);
/* every stride-th pixel of an [N,H,W,C] matrix (scatter = 0) or the transpose, zero insertion
 * (scatter = 1, dst is the H x W side): the stride-2 convs of resnet.py:13,76-79 around the stride-1 kernels */
int uda_rows_stride(const float* src, int64_t ld_src, int N, int H, int W, int C, int stride, int scatter,
                    float* dst, int64_t ld_dst, void* stream);
/* Bottleneck tail (resnet.py:37-41): out = relu(transform(a) + transform(b)) */
int uda_bn_add_relu(const uda_src_t* a, const uda_src_t* b, float* out, int64_t ldo, void* stream);
/* its backward gate: out = dz where z > 0 else 0 */
int uda_relu_gate(const float* dz, int64_t lddz, const float* z, int64_t ldz, int64_t P, int C, float* out,
                  int64_t ldo, void* stream);

/* ---- patch discriminators (networks/GAN.py:86-148): Conv2d(4, stride 2, pad 2) = ksize-2
 * uda_conv_fwd on the space-to-depth image z[n,i,j,(a,b,c)] = x[n, 2i+a-2, 2j+b-2, c] (zero outside the valid
 * region valid_h x valid_w of the [N,Hs,Ws,C] source grid; Hz = (valid_h+5)/2).  uda_s2d_fwd also applies the
 * previous layer's LeakyReLU (slope; 1 = none); uda_s2d_bwd routes dz back to the source grid (zeros outside
 * the valid region), times the LeakyReLU gate read from the sign of z (z_sign NULL = no gate).
 * nchw_*: the source side is an NCHW tensor (the discriminator input / its gradient). */
/* weight [O, C, 4, 4] -> operand of the ksize-2 uda_conv_fwd on z: dgrad = 0: [O][row over (a,b,c)], wz[o,(u,v),(a,b,c)] =
 * w[o,c,2u+a,2v+b]; dgrad = 1: [4C][row over o] with flipped taps (its input gradient).  Rows as uda_relayout_ohwi writes them. */
int uda_relayout_s2d(const float* w, int O, int C, int dgrad, float* out, void* stream);
int uda_s2d_fwd(const float* src, int64_t ld_src, int nchw_in, int N, int Hs, int Ws, int C, int valid_h,
                int valid_w, float slope, float* z, int64_t ld_z, int Hz, int Wz, void* stream);
int uda_s2d_bwd(const float* dz, const float* z_sign, int64_t ld_z, int Hz, int Wz, float slope, int N, int Hs,
                int Ws, int C, int valid_h, int valid_w, float* dst, int64_t ld_dst, int nchw_out, void* stream);
/* uda_s2d_bwd (rows out) with the LeakyReLU gate read from the sign of the forward's SOURCE rows [N, Hs, Ws, C] (the previous
 * layer's raw output: z = lrelu(source), same signs) - for when the fp32 z image was never written, see below. */
int uda_s2d_bwd_gate(const float* dz, int64_t ld_z, int Hz, int Wz, const float* gate_rows, int64_t ld_gate, float slope,
                     int N, int Hs, int Ws, int C, int valid_h, int valid_w, float* dst, int64_t ld_dst, void* stream);
/* bf16x3 mode (GAN.py:102-107 / :135-140, layers 2-4): the space-to-depth operands in PACKED form without their fp32 images.
 * uda_x3_pack_s2d_fwd = uda_s2d_fwd + uda_x3_pack of z (out: uda_x3_packed_bytes(N*Hz*Wz, 4C) bytes); uda_x3_pack_s2d_bwd =
 * uda_s2d_bwd_gate + uda_x3_pack of the routed gradient (out: uda_x3_packed_bytes(N*Hs*Ws, C) bytes).  C % 8 == 0.  The results
 * are bit-identical to the two-pass forms (the split is exact); one read of the source and one packed write instead of a read,
 * an fp32 write, an fp32 read and the packed write. */
int uda_x3_pack_s2d_fwd(const float* src, int64_t ld_src, int N, int Hs, int Ws, int C, int valid_h, int valid_w, float slope,
                        int Hz, int Wz, void* out, void* stream);
int uda_x3_pack_s2d_bwd(const float* dz, int64_t ld_z, int Hz, int Wz, const float* gate_rows, int64_t ld_gate, float slope,
                        int N, int Hs, int Ws, int C, int valid_h, int valid_w, void* out, void* stream);

/* ---- batch-norm pieces (F.batch_norm, training and eval) */
/* stats: double[UDA_STAT_SLOTS][2][C] */
int uda_bn_finalize(const double* stats, int C, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* scale,
                    float* shift, float* mean, float* invstd, void* stream);
/* k more momentum updates of the running statistics with unchanged batch statistics over `count`
 * elements (the 4 stochastic passes of Trainer_prototype_full.py:364-368 repeat the deterministic,
 * pre-dropout part of the network on the same batch) */
int uda_bn_running_replay(const float* mean, const float* invstd, int C, double count, int k, float momentum,
                          float eps, float* running_mean, float* running_var, void* stream);
int uda_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, int C, float eps, float* scale, float* shift,
                       void* stream);
/* TransNorm, the --use_TN normalisation (networks/sync_batchnorm/batchnorm.py:436-520, training branch :445-495,
 * eval branch :497-520).  Training: a batch is normalised per domain half (first N/2 images "source", the rest
 * "target"): the caller runs uda_bn_finalize once per half (its own statistics accumulator, count and running
 * buffers), then uda_tn_gain turns the two accumulators into gain[c] = 1 + C p_c / sum(p),
 * p_c = 1 / (1 + |mu_s/sqrt(var_s+eps) - mu_t/sqrt(var_t+eps)|) (unbiased variances, :474-483) and multiplies
 * both halves' scale/shift by it in place (the reference's z * (1 + alpha.detach()), :495).  Eval: the target
 * running statistics normalise and the gain comes from the two pairs of running statistics (:501-520). */
int uda_tn_gain(const double* stats0, const double* stats1, int C, double count0, double count1, float eps,
                float* scale0, float* shift0, float* scale1, float* shift1, float* gain, void* stream);
int uda_tn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean_source,
                       const float* running_var_source, const float* running_mean_target,
                       const float* running_var_target, int C, float eps, float* scale, float* shift, void* stream);
/* out = transform(src) + residual */
int uda_bn_apply(const uda_src_t* src, const float* residual, int64_t ldr, float* out, int64_t ldo,
                 void* stream);
/* out (double[UDA_STAT_SLOTS][nq][C], ADDED into): nq=1 sum, nq=2 sum and sum of squares of x */
int uda_colstats(const float* x, int64_t ldx, int64_t P, int C, int nq, double* out, void* stream);
/* the same into a channel WINDOW of a wider accumulator double[UDA_STAT_SLOTS][nq][out_C]: out points at the window's first
 * channel (slot 0, quantity 0).  The statistics of decoder.py:23's BatchNorm(305) over cat(up(x), low-level, boundary) are
 * gathered this way: the upsampled channels by uda_upsample_fwd_stats, the other 49 by one window pass. */
int uda_colstats_window(const float* x, int64_t ldx, int64_t P, int C, int nq, double* out, int out_C, void* stream);
/* g = dU*mask*act'(a);  sums (double[UDA_STAT_SLOTS][3][C], ADDED into) = (sum g, sum g*xhat, sum dU) */
int uda_bnbwd_reduce(const float* dU, int64_t ldu, const uda_src_t* y, const float* mean,
                     const float* invstd, double* sums, void* stream);
/* sums: double[UDA_STAT_SLOTS][3][C].  count = elements per channel of the batch statistics; +infinity for a FROZEN
 * (eval-mode) BatchNorm, DeepLab.freeze_bn() of deeplabv3.py:43-50: c1 = c2 = 0, dx = scale * g, dgamma / dbeta as usual with
 * mean / invstd = the running statistics.  q1_total ([C] or NULL = 0): sum of the upstream gradient over ALL positions of the
 * zero-padded block input (quirk Q1) - needed when the depthwise BatchNorm behind this one is frozen. */
int uda_bnbwd_finalize(const double* sums, int C, double count, int q1_border, int act,
                       const float* shift, const float* mean, const float* invstd, const float* q1_total,
                       float* c1, float* c2, float* dgamma, float* dbeta, void* stream);
/* out = addend + scale*(g - c1 - xhat*c2) */
int uda_bnbwd_apply(const float* dU, int64_t ldu, const uda_src_t* y, const float* mean,
                    const float* invstd, const float* c1, const float* c2, const float* addend,
                    int64_t ld_add, float* out, int64_t ldo, void* stream);
/* The same two passes with the upstream gradient in LOW-RANK form dU[p, c] = sum_{o < k} d[p, o] * w[o, c] (k = 1 or 2, w row-major
 * [k][C]): the input gradient of a 1x1 conv to k outputs - the decoder's heads, decoder.py:32 (305 -> 2) and :41 (256 -> 1) - is
 * formed inside the passes instead of being written by a conv and read back twice. */
int uda_bnbwd_reduce_lowrank(const float* d, int64_t ldd, int k, const float* w, const uda_src_t* y, const float* mean,
                             const float* invstd, double* sums, void* stream);
int uda_bnbwd_apply_lowrank(const float* d, int64_t ldd, int k, const float* w, const uda_src_t* y, const float* mean,
                            const float* invstd, const float* c1, const float* c2, const float* addend, int64_t ld_add,
                            float* out, int64_t ldo, void* stream);

/* ---- resampling / pooling (F.interpolate bilinear align_corners=True, adaptive_avg_pool2d) */
int uda_upsample_fwd(const float* x, int64_t ldx, int N, int h, int w, int C, float* out,
                     int64_t ldo, int H, int W, void* stream);
/* uda_upsample_fwd + per-channel (sum, sum of squares) of the output ADDED into channels [0, C) of
 * stats = double[UDA_STAT_SLOTS][2][stat_C]; needs 256 % (C / 4) == 0 */
int uda_upsample_fwd_stats(const float* x, int64_t ldx, int N, int h, int w, int C, float* out, int64_t ldo, int H, int W,
                           double* stats, int stat_C, void* stream);
/* (out = NULL: statistics only, the upsampled tensor is not written.)
 * uda_mc_seg_head: the segmentation head of a no-grad stochastic pass (Trainer_prototype_full.py:358-368; decoder.py:23-32,51-53):
 *     x1b[p, o] = bias[o] + sum_c w[o][c] * mask[p, c] * mask_scale * act(scale[c] * xf[p, c] + shift[c]),  o = 0, 1,
 *     xf[p, :] = cat(bilinear_up(feature)[p, 0:Cf], low[p mod P_low, 0:Cl], boundary[p])
 * WITHOUT the [P, Cf + Cl + 1] x_feature matrix: the upsampled channels are interpolated on the fly from feature [N*h*w, Cf], the
 * low-level channels come from the un-repeated rows low [P_low, Cl] (x.repeat(2) shares them), the boundary logit from its own
 * column.  weight: two rows of ldw floats (uda_relayout_ohwi of decoder.last_conv.3.weight); mask: uint8 [P][ldm] or NULL. */
int uda_mc_seg_head(const float* feature, int64_t ld_feat, int N, int h, int w, int Cf, const float* low, int64_t ld_low,
                    int Cl, int64_t P_low, const float* boundary, int64_t ld_bnd, int H, int W, const float* scale,
                    const float* shift, int act, const uint8_t* mask, int64_t ldm, float mask_scale, const float* weight,
                    int64_t ldw, const float* bias, float* out, int64_t ldo, void* stream);
int uda_upsample_bwd(const float* dout, int64_t ldo, int N, int H, int W, int C, float* dx,
                     int64_t ldx, int h, int w, void* stream);
/* NHWC [N*h*w, C<=4] -> contiguous NCHW [N][C][H][W] and its adjoint */
int uda_head_upsample_fwd(const float* x, int64_t ldx, int N, int h, int w, int C, float* out,
                          int H, int W, void* stream);
int uda_head_upsample_bwd(const float* dout, int N, int C, int H, int W, float* dx, int64_t ldx,
                          int h, int w, int accumulate, void* stream);
/* out[n,c] = scale * sum_{p in image n} x[p,c] */
int uda_gap_fwd(const float* x, int64_t ldx, int N, int HW, int C, float scale, float* out,
                int64_t ldo, void* stream);
/* out[p,c] = addend[p,c] + scale*g[n(p),c] */
int uda_broadcast_rows(const float* g, int64_t ldg, int N, int HW, int C, float scale,
                       const float* addend, int64_t ld_add, float* out, int64_t ldo, void* stream);

/* ---- dropout keep-mask, Philox4x32-10 counter stream (nn.Dropout at aspp.py:62, decoder.py:31,36,40) */
int uda_dropout_mask(uint8_t* mask, int64_t ldm, int64_t P, int C, float p, uint64_t seed,
                     uint64_t offset, void* stream);

/* ---- segmentation loss: BCELoss(sigmoid(o), map) + MSELoss(sigmoid(b), boundary), both 'mean'
 * (Trainer_prototype_full.py:18-19,292-294; Trainer_baseline.py:206-208; log clamp -100).
 * loss3 (device float[3]) = (total, bce, mse); workspace2 = device double[2].
 * bwd: d_o / d_b = gradient of the total times the device scalar *gscale. */
int uda_seg_loss_fwd(const float* o, const float* map, int64_t n_o, const float* b, const float* boundary,
                     int64_t n_b, float* loss3, double* workspace2, void* stream);
int uda_seg_loss_bwd(const float* o, const float* map, int64_t n_o, const float* b, const float* boundary,
                     int64_t n_b, const float* gscale, float* d_o, float* d_b, void* stream);
/* counts[c][3] = (intersection, predicted, ground-truth) pixels for sigmoid(logit) > thr
 * (utils/metrics.py:118-132,149-168: Dice, pixel accuracy and IoU follow from them) */
int uda_seg_counts(const float* logits, const float* target, int B, int C, int64_t HW, float thr,
                   uint64_t* counts, void* stream);

/* ---- category prototypes (utils/Utils.py:108-131, 159-225) */
/* preds [T][n] logits -> unbiased std over T of sigmoid(x/2), mean over T of sigmoid(x)  (T = 8) */
int uda_mc_stats(const float* preds, int T, int64_t n, float* std_map, float* mean_map, void* stream);
/* wts [P][4] for the classes (cup obj, disc obj, cup bck, disc bck);
 * mode 0: from a [B,2,H,W] map, nearest-resized to h x w;  mode 1: from sigmoid of [P,2] logits;
 * mode 2: retrified pseudo labels (sigmoid>0.75) x (bilinear(std)<0.04) x bilinear(mean) */
int uda_proto_weights(int mode, int B, int h, int w, int H, int W, const float* map, const float* logits,
                      int64_t ldl, const float* std_map, const float* mean_map, float* wts, float* mask0,
                      float* mask1, void* stream);
uint64_t uda_proto_workspace_bytes(int64_t P, int C);
/* sums (double[4][C+1], ADDED into): weighted channel sums, then the weight total, per class */
int uda_proto_reduce(const float* feat, int64_t ldf, int64_t P, int C, const float* wts, double* sums,
                     float* workspace, uint64_t workspace_bytes, void* stream);
int uda_proto_finalize(const double* sums, int C, float* centroids /* [4][C] */, void* stream);
/* adjoint of centroid = sum/count: d_feat[P,C] (optional, optionally accumulated) and d_w[P,4] (optional) */
int uda_proto_bwd(const float* feat, int64_t ldf, int64_t P, int C, const float* wts, const double* sums,
                  const float* dC, float* coef_ws /* float[4][C+1] */, float* d_feat, int64_t ldd,
                  int accumulate, float* d_w, void* stream);

/* out[p][k] = sum_c feat[p,c]*coef[k][c] + coef[k][C], k < 4 (coef: float[4][C+1]) and its adjoint
 * d_feat[p,c] (+)= sum_k wts[p][k]*coef[k][c].  Used by the prototype-guided discriminative loss
 * (no shipped source - parity unpinned). */
int uda_feat_dot4(const float* feat, int64_t ldf, int64_t P, int C, const float* coef, float* out, void* stream);
int uda_feat_rank4(const float* wts, const float* coef, int64_t P, int C, float* d_feat, int64_t ldd,
                   int accumulate, void* stream);

/* ---- torch.optim.Adam update (no weight decay / amsgrad) over one flat fp32 buffer
 * (train_use_fix_initial.py:210-214).  The hyper-parameters arrive as doubles: 1 - beta, lr / (1 - beta1^step) and
 * sqrt(1 - beta2^step) are formed in double and rounded to fp32 where torch rounds them.  Used by uda_clr_amd.optim.FlatAdam. */
int uda_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                  double beta1, double beta2, double eps, int64_t step, void* stream);

/* ---- "bilinear upsample, then 3x3 conv" without the high-resolution GEMM (networks/decoder.py:50-53 feeding
 * last_conv_boundary[0], decoder.py:33): the channel mixing of the upsampled part commutes with the interpolation, so the
 * caller runs the nine tap GEMMs at LOW resolution (g = f W_all^T, [N*h*w, 9*C], tap-major columns) and these kernels do
 * the interpolation.  fwd: y[p,:] = addend[p % addend_rows,:] + sum_t [p + d_t inside H x W] bilinear(g_t)(p + d_t),
 * align_corners=True, d_t = ((t/3) - 1, (t%3) - 1) * dil.  bwd: dg = adjoint of that sum applied to dy (gather form).
 * stats (double[UDA_STAT_SLOTS][2][C], ADDED into, or null): per-channel sum / sum of squares of y for the next BatchNorm;
 * uda_upconv_fused_stats tells whether a geometry supports them (otherwise accumulate with uda_colstats). */
int uda_upconv_fused_stats(int h, int w, int H, int W, int C, int dil);
int uda_upconv_fwd(const float* g, int64_t ldg, int N, int h, int w, int C, int dil, const float* addend, int64_t ld_add,
                   int64_t addend_rows, float* y, int64_t ldy, int H, int W, double* stats, void* stream);
int uda_upconv_bwd(const float* dy, int64_t ldy, int N, int H, int W, int C, int dil, float* dg, int64_t ldg, int h, int w,
                   void* stream);

/* ---- evaluation post-processing of the predicted probability maps (utils/Utils.py:427-463: postprocessing +
 * get_largest_fillhole), per image and channel (0 cup, 1 disc): threshold -> 5 x 7x7 median (zero padded) -> erosion by the
 * L1 ball of radius 7 (outside = set, skimage's convention) -> largest 8-connected component (first maximum in raster order of
 * the components' first pixels) -> holes filled (background not 4-connected to the border).
 * pred f32 [B,2,H,W] -> out uint8 [B,2,H,W] in {0,1}.  sweeps: launches of each of the two tile-wise propagations (a launch
 * carries labels across whole 32x32 tiles); not_converged: DEVICE int[2], set to the number of tiles an extra sweep still
 * changed (components, flood) - 0 means the result is final, otherwise call again with more sweeps. */
size_t uda_postprocess_workspace_bytes(int B, int H, int W);
int uda_postprocess(const float* pred, int B, int H, int W, float thr_cup, float thr_disc, int sweeps, uint8_t* out,
                    int* not_converged, void* workspace, size_t workspace_bytes, void* stream);

/* ---- device-side tail of the input pipeline.  The reference's dataloader workers run these per sample
 * on the CPU with scipy.ndimage; here they run per uint8 BATCH on the GPU, bit-identical to the scipy calls.
 * uda_normalize_tf: dataloaders/custom_transforms.py:432-466 (Normalize_tf), :414-429 (GetBoundary), :504-507 (ToTensor).
 *   image_hwc uint8 [B,H,W,3], label uint8 [B,H,W] (grey code: > 200 background, 51..200 disc rim, <= 50 cup) ->
 *   image f32 [B,3,H,W] = v/127.5 - 1, map f32 [B,2,H,W] (cup, disc), boundary f32 [B,1,H,W] = uint8 Gaussian (sigma 3,
 *   scipy semantics: axis 0 then axis 1, each truncated to uint8, 'reflect') of the |dilate5 - erode5| ring, / 255.
 *   gauss_w: HOST pointer to radius+1 doubles, w[0] the centre tap, w[k] the taps at distance k (as numpy computes them). */
size_t uda_normalize_tf_workspace_bytes(int B, int H, int W);
int uda_normalize_tf(const uint8_t* image_hwc, const uint8_t* label, int B, int H, int W, const double* gauss_w, int radius,
                     float* image, float* map, float* boundary, void* workspace, size_t workspace_bytes, void* stream);
/* custom_transforms.py:95-147 (elastic_transform).  uda_field_smooth: out = alpha * gaussian_filter(noise, sigma,
 * mode='constant') on [B,H,W] float64 planes, in scipy's own summation order (bit-identical to the reference's field;
 * weights_dev: DEVICE pointer to radius+1 doubles, centre first; tmp: [B,H,W] doubles).
 * uda_elastic_warp: map_coordinates(order=1) of image (mode 'constant', 0 outside) and label (mode 'nearest') at
 * (h + dx, w + dy), rounded to uint8; apply (uint8 [B] or null) = 0 copies a sample through. */
int uda_field_smooth(const double* noise, int B, int H, int W, const double* weights_dev, int radius, double alpha, double* tmp,
                     double* out, void* stream);
/* custom_transforms.py:150-250 (add_salt_pepper_noise, adjust_light, eraser) on a uint8 batch IN PLACE, in that order, with the
 * per-sample parameters the dataloader workers drew: sp_pos int32 [B, sp_max, 2] (row, column), sp_count int32 [B], sp_value
 * int32 [B] (1 salt / 0 pepper, as the reference writes them); lut uint8 [B, 256] (identity when adjust_light did not fire);
 * erase_box int32 [B, 5] = (top, left, height, width, grey level), height 0 = no erasing. */
int uda_photometric_u8(uint8_t* image_hwc, int B, int H, int W, const int* sp_pos, const int* sp_count, const int* sp_value,
                       int sp_max, const uint8_t* lut, const int* erase_box, void* stream);
int uda_elastic_warp(const uint8_t* image_hwc, const uint8_t* label, const double* dx, const double* dy, const uint8_t* apply,
                     int B, int H, int W, uint8_t* image_out, uint8_t* label_out, void* stream);

/* Trainer_prototype_full.py:335-355, 378-398 (EMA of the eight centroids, gradient through the current term only) + :428-444
 * (intra = sum_k MSE(src_k, tgt_k), inter = MSE(src_1, src_3) + MSE(src_0, src_2)) on two [4][C] centroid matrices in one
 * launch: new = prev ? keep * prev + decay * cur : cur (keep = 1 - decay as the caller's double expression rounds it);
 * losses2 = (intra, inter).  prev_* may be null (first use).
 * uda_proto_align_bwd: d intra / d cur_src = g * w_src * 2 (new_src - new_tgt) / C, d cur_tgt = -g * w_tgt * (...). */
int uda_proto_align_fwd(const float* cur_src, const float* cur_tgt, const float* prev_src, const float* prev_tgt, float keep,
                        float decay, int C, float* new_src, float* new_tgt, float* losses2, void* stream);
int uda_proto_align_bwd(const float* new_src, const float* new_tgt, const float* g_intra, float w_src, float w_tgt, int C,
                        float* d_cur_src, float* d_cur_tgt, void* stream);
/* Trainer_prototype_full.py:456-458, 479-513: scale * (BCEWithLogitsLoss(d1, label) + BCEWithLogitsLoss(d2, label)) on the two
 * patch-discriminator outputs (n1, n2 elements), forward in one launch, both gradients in one launch (g: device scalar). */
int uda_adv_loss_fwd(const float* d1, int n1, const float* d2, int n2, float label, float scale, float* loss, void* stream);
int uda_adv_loss_bwd(const float* d1, int n1, const float* d2, int n2, float label, float scale, const float* g, float* g1,
                     float* g2, void* stream);
/* Trainer_prototype_full.py:452-454: the first discriminator layer fed by generator LOGITS (NCHW): z = s2d(pre(x)),
 * pre_op 1 = sigmoid(x) (boundary branch), 2 = -sigmoid(x) * log(sigmoid(x) + 1e-7) (uncertainty map); the adjoint routes
 * dz back to the logits and multiplies by pre'(x).  No full-resolution intermediate map is materialised. */
int uda_adv_s2d_fwd(const float* logits_nchw, int N, int C, int H, int W, int pre_op, float* z, int64_t ld_z, int Hz, int Wz,
                    void* stream);
int uda_adv_s2d_bwd(const float* dz, int64_t ld_z, int Hz, int Wz, const float* logits_nchw, int N, int C, int H, int W,
                    int pre_op, float* d_logits_nchw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UDA_CLR_HIP_H */
