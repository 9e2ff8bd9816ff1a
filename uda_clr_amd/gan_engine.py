"""Execution plan of the patch discriminators (networks/GAN.py:86-148 of the reference) on the HIP kernels.

Every layer is Conv2d(k=4, stride 2, pad 2, bias=False) followed (layers 1-4) by LeakyReLU(0.2).  The 4x4
stride-2 convolution runs as a 2x2 stride-1 implicit GEMM (``uda_conv_fwd`` with ksize 2) on the space-to-depth
image of the zero-padded input; ``uda_s2d_fwd`` builds that image and fuses the previous layer's LeakyReLU and the
crop of its output grid, ``uda_s2d_bwd`` routes gradients back and applies the LeakyReLU gate.  Activations are
NHWC matrices on the z grids (Hz = (H + 5) // 2: 258, 131, 67, 35, 19 for a 512 x 512 input); a layer's output
lives on its own z grid with Ho = H // 2 + 1 valid rows/columns (257, 129, 65, 33, 17).

``kernels`` is the binding object (HipKernels in the product; the tests' torch statement on CPU).
"""
from __future__ import annotations

import os

import torch

from .acts import Act, round4

SLOPE = 0.2
_S2D_PACK = os.environ.get("UDA_CLR_S2D_PACK", "1") != "0"      # A/B switch of the one-pass packed space-to-depth operands


def _zgrid(v):
    return (v + 5) // 2


class PatchDiscriminatorEngine:
    def __init__(self, kernels):
        self.K = kernels

    def forward(self, x, weights, need_grad, pre_op=0, w_share=None):
        """x: [N, C, H, W] (NCHW, as the reference feeds it).  ``pre_op`` 1 / 2: x holds generator LOGITS and the first layer
        reads sigmoid(x) / the uncertainty map -sigmoid(x) log(sigmoid(x) + 1e-7) (Trainer_prototype_full.py:452-454) - applied
        inside the space-to-depth pass, no full-resolution map is written.  Returns ([N, 1, Ho, Wo] logits, ctx)."""
        K = self.K
        N, Cc, H, W = x.shape
        src, nchw, Hs, Ws, vh, vw, slope = x, True, H, W, H, W, 1.0
        layers = []
        for li, w in enumerate(weights):
            O = w.shape[0]
            # z-space operands, built per forward (torch's fused optimizer steps do not bump a parameter's version,
            # so a cross-forward cache could go stale unnoticed) - unless the caller hands over a dict and with it the promise
            # that the weights do not change while it does (the passes of one training step: _PatchDiscriminator.shared_weight_layouts)
            if w_share is None:
                wf = K.relayout_s2d(w, False)
                wd = K.relayout_s2d(w, True) if need_grad else None
            else:
                if (li, 0) not in w_share:
                    w_share[(li, 0)] = K.relayout_s2d(w, False)
                wf = w_share[(li, 0)]
                wd = None
                if need_grad:
                    if (li, 1) not in w_share:
                        w_share[(li, 1)] = K.relayout_s2d(w, True)
                    wd = w_share[(li, 1)]
            Hz, Wz = _zgrid(vh), _zgrid(vw)
            # bf16x3 mode, layers whose conv (and weight gradient) run on the packed operands: the z image is written in packed form
            # only, straight from the previous layer's output (uda_x3_pack_s2d_fwd) - no fp32 image, no separate packing pass; the
            # backward then takes the LeakyReLU gate from the sign of that source (z = lrelu(source)), which is kept instead
            packed = (li > 0 and Cc % 8 == 0 and hasattr(K, "s2d_pack_fwd") and _S2D_PACK and K.conv_route_x3(N, Hz, Wz, 4 * Cc, O, 2)
                      and (not need_grad or K.wgrad_route_x3(N, Hz, Wz, 4 * Cc, O, 2)))
            if packed:
                z = K.s2d_pack_fwd(src, N, Hs, Ws, Cc, vh, vw, slope)
            else:
                z = torch.empty((N * Hz * Wz, 4 * Cc), dtype=torch.float32, device=x.device)
                if li == 0 and pre_op:
                    K.adv_s2d_fwd(src, pre_op, z)
                else:
                    K.s2d_fwd(src, nchw, N, Hs, Ws, Cc, vh, vw, slope, z)
            y = torch.empty((N * Hz * Wz, round4(O)), dtype=torch.float32, device=x.device)[:, :O]
            K.conv(Act(z, N, Hz, Wz), wf, 2, 1, y, origin=0)
            # (bf16x3 mode keeps the source rows of every layer but the first as the backward's gate: the gradient below a layer
            # can then be written in packed form whether or not this layer's own z image was)
            keep_gate = need_grad and li > 0 and (packed or getattr(K, "mfma", None) == getattr(K, "MFMA_BF16X3", -1))
            layers.append((z if need_grad else None, Cc, Hs, Ws, vh, vw, Hz, Wz, nchw, wd, src if keep_gate else None))
            src, nchw, Hs, Ws, Cc, slope = y, False, Hz, Wz, O, SLOPE
            vh, vw = vh // 2 + 1, vw // 2 + 1
        out = src.reshape(N, Hs, Ws, Cc)[:, :vh, :vw].permute(0, 3, 1, 2)
        return out, ((layers, N, tuple(x.shape), (x if pre_op else None), pre_op) if need_grad else None)

    def backward(self, ctx, gout, weights, need_x, need_w):
        """gout: gradient of the [N, 1, Ho, Wo] logits.  Returns (dx NCHW or None, [dw OIHW 4x4] or None)."""
        K = self.K
        layers, N, xshape, x_logits, pre_op = ctx
        z5, C5, Hs5, Ws5, vh5, vw5, Hz, Wz, _, _, _ = layers[-1]
        O = weights[-1].shape[0]
        vh, vw = vh5 // 2 + 1, vw5 // 2 + 1
        dy = torch.zeros((N, Hz, Wz, round4(O)), dtype=torch.float32, device=gout.device)
        dy[:, :vh, :vw, :O] = gout.permute(0, 2, 3, 1)
        dy = dy.reshape(N * Hz * Wz, round4(O))[:, :O]
        dws = [None] * len(weights)
        dx = None
        for l in range(len(weights) - 1, -1, -1):
            z, Cc, Hs, Ws, vh, vw, Hz, Wz, nchw, wd, gate = layers[l]
            w = weights[l]
            O = w.shape[0]
            if need_w:
                dwz = torch.empty((O, 4 * Cc, 2, 2), dtype=torch.float32, device=z.device)
                K.conv_wgrad(Act(z, N, Hz, Wz), dy, 2, 1, dwz, origin=0)
                # [o, (a,b,c), u, v] -> [o, c, 2u+a, 2v+b]
                dws[l] = dwz.reshape(O, 2, 2, Cc, 2, 2).permute(0, 3, 4, 1, 5, 2).reshape(O, Cc, 4, 4)
            if l == 0 and not need_x:
                break
            dz = torch.empty_like(z)
            K.conv(Act(dy, N, Hz, Wz), wd, 2, 1, dz, origin=1)
            if nchw:
                dx = torch.empty(xshape, dtype=torch.float32, device=dz.device)
                if pre_op:
                    K.adv_s2d_bwd(dz, x_logits, pre_op, dx)
                else:
                    K.s2d_bwd(dz, None, 1.0, N, Hs, Ws, Cc, vh, vw, dx, True)
            else:
                # the gradient w.r.t. layer l-1's output: read by that layer's weight gradient and input-gradient conv.  When both run
                # on packed operands it is written in packed form only (uda_x3_pack_s2d_bwd)
                below = layers[l - 1]
                need_dx_below = (l - 1 > 0) or need_x
                pack_dy = (gate is not None and hasattr(K, "s2d_pack_bwd") and _S2D_PACK and Cc % 8 == 0
                           and (not need_w or K.wgrad_route_x3(N, Hs, Ws, 4 * below[1], Cc, 2))
                           and (not need_dx_below or K.conv_route_x3(N, Hs, Ws, Cc, 4 * below[1], 2)))
                if pack_dy:
                    dy = K.s2d_pack_bwd(dz, gate, SLOPE, N, Hs, Ws, Cc, vh, vw)
                else:
                    dyn = torch.empty((N * Hs * Ws, round4(Cc)), dtype=torch.float32, device=dz.device)[:, :Cc]
                    if gate is not None:
                        K.s2d_bwd(dz, None, SLOPE, N, Hs, Ws, Cc, vh, vw, dyn, False, gate=gate)
                    else:
                        K.s2d_bwd(dz, z, SLOPE, N, Hs, Ws, Cc, vh, vw, dyn, False)
                    dy = dyn
            del dz
        return dx, (dws if need_w else None)
