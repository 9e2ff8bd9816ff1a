"""Plain-PyTorch fp32 statement of every kernel entry point (TEST INFRASTRUCTURE).

``SpecKernels`` has the same methods as ``uda_clr_amd.kernels.HipKernels`` and states, with
ordinary torch ops, exactly what each HIP kernel must compute.  It is used
  * by the ``-m gpu`` tests as the per-kernel fp32 reference, and
  * by the CPU tests to run the engine's orchestration (``uda_clr_amd.engine``) end to end
    against the oracle without a GPU (the engine accepts any object with these methods; the
    product only ever constructs ``HipKernels``).
It is never imported from ``uda_clr_amd``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from uda_clr_amd.acts import ACT_NONE, ACT_RELU, ACT_RELU6, K_CHUNK, Act, round4, tap_chunked


def _apply_act(a, act):
    if act == ACT_RELU:
        return torch.relu(a)
    if act == ACT_RELU6:
        return torch.clamp(a, 0.0, 6.0)
    return a


def _act_grad(a, act):
    if act == ACT_RELU:
        return (a > 0).to(a.dtype)
    if act == ACT_RELU6:
        return ((a > 0) & (a < 6)).to(a.dtype)
    return torch.ones_like(a)


def transform(src: Act):
    """[P, C] activated values u = act(x*scale+shift) * mask*mask_scale."""
    u = src.x
    if src.scale is not None:
        u = u * src.scale + src.shift
    u = _apply_act(u, src.act)
    if src.mask is not None:
        u = u * (src.mask.to(u.dtype) * src.mask_scale)
    return u


def _nchw(m, N, H, W):
    return m.reshape(N, H, W, m.shape[1]).permute(0, 3, 1, 2)


def _rows(t):
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def _src_index(o, scale, n_in):
    real = scale * o.to(torch.float32)
    i0 = real.to(torch.int64)
    i1 = i0 + (i0 < n_in - 1).to(torch.int64)
    l1 = real - i0.to(torch.float32)
    return i0, i1, 1.0 - l1, l1


def _bilinear_matrix(n_in, n_out, device, dtype=torch.float32):
    """[n_out, n_in] interpolation matrix of align_corners=True bilinear resize."""
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    o = torch.arange(n_out, device=device)
    i0, i1, l0, l1 = _src_index(o, torch.tensor(scale, dtype=torch.float32, device=device), n_in)
    m = torch.zeros(n_out, n_in, device=device, dtype=dtype)      # lerp weights in fp32 like the kernels, whatever the value type
    m[o, i0] += l0.to(dtype)
    m[o, i1] += l1.to(dtype)
    return m


class SpecKernels:
    name = "spec"

    # ------------------------------------------------------------------ weight layouts
    def relayout_ohwi(self, w):
        """[O, I, kh, kw] -> [O, 1, round4(I)] for 1x1, else the tap-chunked K order [O, nCC*T, 32] (acts.tap_chunked)."""
        O, I, kh, kw = w.shape
        rtc = w.permute(0, 2, 3, 1).reshape(O, kh * kw, I)
        if kh * kw > 1:
            return tap_chunked(rtc)
        out = w.new_zeros(O, 1, round4(I))
        out[:, :, :I] = rtc
        return out

    def relayout_dgrad(self, w):
        """rows = input channels, taps flipped, channels = output channels (operand of conv used as dgrad)."""
        O, I, kh, kw = w.shape
        rtc = w.flip(2, 3).permute(1, 2, 3, 0).reshape(I, kh * kw, O)
        if kh * kw > 1:
            return tap_chunked(rtc)
        out = w.new_zeros(I, 1, round4(O))
        out[:, :, :O] = rtc
        return out

    @staticmethod
    def _taps_channels(w, T, Cin):
        """inverse of the weight layout: [R, T, Cin]"""
        R = w.shape[0]
        if T == 1 or round4(Cin) < K_CHUNK:
            return w[:, :, :Cin]
        ncc = w.shape[1] // T
        return w.reshape(R, ncc, T, K_CHUNK).permute(0, 2, 1, 3).reshape(R, T, ncc * K_CHUNK)[:, :, :Cin]

    def relayout_dw(self, w):
        return w.reshape(w.shape[0], 9).t().contiguous()

    # ------------------------------------------------------------------ dense conv (stride 1)
    @staticmethod
    def _pad_taps(u, ksize, dil, origin):
        """Zero padding that realises the tap geometry: ksize 3 centred (pad dil each side), ksize 2 with taps
        (kh - origin, kw - origin): origin 0 reaches +1 (pad right/bottom), origin 1 reaches -1 (pad left/top)."""
        if ksize == 3:
            return F.pad(u, (dil, dil, dil, dil))
        if ksize == 2:
            return F.pad(u, (1, 0, 1, 0)) if origin else F.pad(u, (0, 1, 0, 1))
        return u

    def conv(self, src: Act, w, ksize, dil, out, bias=None, addend=None, stats=None, origin=0, stride=1):
        """out[p, co] = bias[co] + addend[p, co] + sum_{t, ci} u(stride * p + off_t, ci) * w[co, t, ci];
        ``w`` is the relayout_ohwi operand; stats (fp64 [2, Cout]) receives sum and sum of squares of the
        value before ``addend``."""
        Cin, Cout = src.C, out.shape[1]
        u = self._pad_taps(_nchw(transform(src), src.N, src.H, src.W), ksize, dil, origin)
        w4 = self._taps_channels(w, ksize * ksize, Cin).reshape(Cout, ksize, ksize, Cin).permute(0, 3, 1, 2)
        y = _rows(F.conv2d(u, w4, bias, stride, 0, dil if ksize == 3 else 1))
        if stats is not None:
            stats[0, 0] += y.double().sum(0)
            stats[0, 1] += (y.double() ** 2).sum(0)
        if addend is not None:
            y = y + addend
        out.copy_(y)

    def conv_wgrad(self, src: Act, dy, ksize, dil, dw, origin=0, stride=1):
        """dw[co, ci, kh, kw] = sum_p dy[p, co] * u(stride * p + off_t, ci)   (OIHW result)."""
        u = self._pad_taps(_nchw(transform(src), src.N, src.H, src.W), ksize, dil, origin)
        g = _nchw(dy, src.N, (src.H - 1) // stride + 1, (src.W - 1) // stride + 1)
        res = torch.nn.grad.conv2d_weight(u, dw.shape, g, stride, 0, dil if ksize == 3 else 1)
        dw.copy_(res)

    # ------------------------------------------------------------------ depthwise 3x3
    def _dw_input(self, src, dil, border_mode):
        u = _nchw(transform(src), src.N, src.H, src.W)
        up = F.pad(u, (dil, dil, dil, dil))
        if border_mode == 1:          # quirk Q1: border holds act(shift) instead of 0
            b = _apply_act(src.shift, src.act).view(1, -1, 1, 1)
            m = F.pad(torch.zeros_like(u[:, :1]), (dil, dil, dil, dil), value=1.0)
            up = up + b * m
        return up

    def dwconv_fwd(self, src: Act, w9c, stride, dil, border_mode, out, stats=None):
        C = src.C
        up = self._dw_input(src, dil, border_mode)
        y = _rows(F.conv2d(up, w9c.t().reshape(C, 1, 3, 3), None, stride, 0, dil, C))
        if stats is not None:
            stats[0, 0] += y.double().sum(0)
            stats[0, 1] += (y.double() ** 2).sum(0)
        out.copy_(y)

    def dwconv_dgrad(self, dy, w9c, stride, dil, N, H, W, out):
        """Gradient w.r.t. the INTERIOR H x W positions of the padded depthwise input."""
        C = dy.shape[1]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        g = _nchw(dy, N, Ho, Wo)
        full = torch.nn.grad.conv2d_input((N, C, H + 2 * dil, W + 2 * dil), w9c.t().reshape(C, 1, 3, 3),
                                          g, stride, 0, dil, C)
        out.copy_(_rows(full[:, :, dil:dil + H, dil:dil + W]))

    def dwconv_wgrad(self, src: Act, dy, stride, dil, border_mode, dw):
        C = src.C
        Ho, Wo = (src.H - 1) // stride + 1, (src.W - 1) // stride + 1
        up = self._dw_input(src, dil, border_mode)
        g = _nchw(dy, src.N, Ho, Wo)
        dw.copy_(torch.nn.grad.conv2d_weight(up, (C, 1, 3, 3), g, stride, 0, dil, C))

    # ------------------------------------------------------------------ stem 3x3 s2 p1, NCHW in
    def stem_fwd(self, x, w, out, stats=None):
        y = _rows(F.conv2d(x, w, None, 2, 1))
        if stats is not None:
            stats[0, 0] += y.double().sum(0)
            stats[0, 1] += (y.double() ** 2).sum(0)
        out.copy_(y)

    def stem_wgrad(self, x, dy, dw):
        N, _, H, W = x.shape
        g = _nchw(dy, N, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
        dw.copy_(torch.nn.grad.conv2d_weight(x, dw.shape, g, 2, 1))

    # ------------------------------------------------------------------ ResNet-101 pieces
    def stem7_fwd(self, x, w, out, stats=None):
        y = _rows(F.conv2d(x, w, None, 2, 3))
        if stats is not None:
            stats[0, 0] += y.double().sum(0)
            stats[0, 1] += (y.double() ** 2).sum(0)
        out.copy_(y)

    def stem7_wgrad(self, x, dy, dw):
        N, _, H, W = x.shape
        g = _nchw(dy, N, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
        dw.copy_(torch.nn.grad.conv2d_weight(x, dw.shape, g, 2, 3))

    def maxpool_fwd(self, src: Act, out, idx):
        """MaxPool2d(3, 2, 1) of the transformed src; idx = winning tap kh*3+kw (first maximum)."""
        u = _nchw(transform(src), src.N, src.H, src.W)
        cols = F.unfold(F.pad(u, (1, 1, 1, 1), value=float("-inf")), 3, stride=2)      # [N, C*9, L]
        cols = cols.reshape(src.N, src.C, 9, -1)
        best = cols.max(2, keepdim=True).values
        first = (cols == best).to(torch.uint8).argmax(2)                                # first index of the maximum
        out.copy_(best[:, :, 0].permute(0, 2, 1).reshape(-1, src.C))
        idx.copy_(first.permute(0, 2, 1).reshape(-1, src.C).to(torch.uint8))

    def maxpool_bwd(self, dz, idx, N, H, W, out):
        Cc = dz.shape[1]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        g = dz.reshape(N, Ho * Wo, Cc).permute(0, 2, 1)                                 # [N, C, L]
        onehot = (idx.reshape(N, Ho * Wo, Cc).permute(0, 2, 1).unsqueeze(2).long() ==
                  torch.arange(9, device=dz.device).view(1, 1, 9, 1)).to(dz.dtype)
        cols = (onehot * g.unsqueeze(2)).reshape(N, Cc * 9, Ho * Wo)
        full = F.fold(cols, (H + 2, W + 2), 3, stride=2)[:, :, 1:H + 1, 1:W + 1]
        out.copy_(_rows(full))

    def rows_stride(self, src, N, H, W, stride, out, scatter=False):
        Cc = src.shape[1]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        if not scatter:
            out.copy_(src.reshape(N, H, W, Cc)[:, ::stride, ::stride].reshape(-1, Cc))
        else:
            full = src.new_zeros(N, H, W, Cc)
            full[:, ::stride, ::stride] = src.reshape(N, Ho, Wo, Cc)
            out.copy_(full.reshape(-1, Cc))

    def bn_add_relu(self, a: Act, b: Act, out):
        out.copy_(torch.relu(transform(a) + transform(b)))

    def relu_gate(self, dz, z, out):
        out.copy_(torch.where(z > 0, dz, torch.zeros_like(dz)))

    # ------------------------------------------------------------------ patch-discriminator geometry
    @staticmethod
    def _s2d_grid(src, nchw, N, Hs, Ws, Cc):
        return src if nchw else src.reshape(N, Hs, Ws, Cc).permute(0, 3, 1, 2)

    def relayout_s2d(self, w, dgrad):
        """z-space operands of a [O, C, 4, 4] weight: wz[o, (u,v), (a,b,c)] = w[o, c, 2u+a, 2v+b] (tap-chunked rows); dgrad:
        rows (a,b,c), taps flipped, channels o."""
        O, Cc = w.shape[0], w.shape[1]
        wz = w.detach().reshape(O, Cc, 2, 2, 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(O, 4, 4 * Cc)
        return tap_chunked(wz.flip(1).permute(2, 1, 0)) if dgrad else tap_chunked(wz)

    def s2d_fwd(self, src, nchw, N, Hs, Ws, Cc, vh, vw, slope, z):
        """z[n,i,j,(a,b,c)] = leaky(x[n, 2i+a-2, 2j+b-2, c]) inside the valid vh x vw region, else 0."""
        Hz, Wz = (vh + 5) // 2, (vw + 5) // 2
        x = self._s2d_grid(src, nchw, N, Hs, Ws, Cc)[:, :, :vh, :vw]
        x = torch.where(x > 0, x, x * slope)
        xp = F.pad(x, (2, 2 * Wz - vw - 2, 2, 2 * Hz - vh - 2))                       # [N, C, 2Hz, 2Wz]
        zz = xp.reshape(N, Cc, Hz, 2, Wz, 2).permute(0, 2, 4, 3, 5, 1)               # n, i, j, a, b, c
        z.copy_(zz.reshape(N * Hz * Wz, 4 * Cc))

    def s2d_bwd(self, dz, z_sign, slope, N, Hs, Ws, Cc, vh, vw, dst, nchw):
        Hz, Wz = (vh + 5) // 2, (vw + 5) // 2
        g = dz
        if z_sign is not None:
            g = torch.where(z_sign > 0, dz, dz * slope)
        gp = g.reshape(N, Hz, Wz, 2, 2, Cc).permute(0, 5, 1, 3, 2, 4).reshape(N, Cc, 2 * Hz, 2 * Wz)
        out = torch.zeros(N, Cc, Hs, Ws, dtype=dz.dtype, device=dz.device)
        out[:, :, :vh, :vw] = gp[:, :, 2:2 + vh, 2:2 + vw]
        dst.copy_(out if nchw else out.permute(0, 2, 3, 1).reshape(N * Hs * Ws, Cc))

    @staticmethod
    def _adv_pre(x, op):
        s = torch.sigmoid(x)
        return s if op == 1 else -1.0 * s * torch.log(s + 1e-7)

    def adv_s2d_fwd(self, logits, pre_op, z):
        N, Cc, H, W = logits.shape
        self.s2d_fwd(self._adv_pre(logits, pre_op), True, N, H, W, Cc, H, W, 1.0, z)

    def adv_s2d_bwd(self, dz, logits, pre_op, d_logits):
        N, Cc, H, W = logits.shape
        g = torch.empty_like(logits)
        self.s2d_bwd(dz, None, 1.0, N, H, W, Cc, H, W, g, True)
        with torch.enable_grad():
            x = logits.detach().clone().requires_grad_(True)
            self._adv_pre(x, pre_op).backward(g)
        d_logits.copy_(x.grad)

    # ------------------------------------------------------------------ batch norm pieces
    def bn_finalize(self, stats, count, gamma, beta, rmean, rvar, momentum, eps,
                    scale, shift, mean, invstd):
        """Training BN coefficients from (sum, sumsq) over ``count`` elements; updates the running
        statistics exactly like F.batch_norm (unbiased variance into running_var)."""
        st = stats.sum(0)
        m = st[0] / count
        var = (st[1] / count - m * m).clamp_min(0.0)
        istd = 1.0 / torch.sqrt(var + eps)
        mean.copy_(m.float())
        invstd.copy_(istd.float())
        sc = gamma.double() * istd
        scale.copy_(sc.float())
        shift.copy_((beta.double() - m * sc).float())
        rmean.mul_(1 - momentum).add_(momentum * m.float())
        rvar.mul_(1 - momentum).add_(momentum * (var * (count / max(count - 1.0, 1.0))).float())

    def bn_running_replay(self, mean, invstd, count, k, momentum, eps, rmean, rvar):
        var = (1.0 / invstd.double() ** 2 - eps).clamp_min(0.0) * (count / max(count - 1.0, 1.0))
        for _ in range(k):
            rmean.mul_(1 - momentum).add_(momentum * mean)
            rvar.mul_(1 - momentum).add_(momentum * var.float())

    def bn_eval_coeffs(self, gamma, beta, rmean, rvar, eps, scale, shift):
        sc = gamma.double() / torch.sqrt(rvar.double() + eps)
        scale.copy_(sc.float())
        shift.copy_((beta.double() - rmean.double() * sc).float())

    def tn_gain(self, stats0, stats1, count0, count1, eps, scale0, shift0, scale1, shift1, gain):
        """TransNorm gain 1 + alpha from the two halves' (sum, sumsq) accumulators (unbiased variances), folded into
        both halves' BN coefficients (batchnorm.py:474-495)."""
        ratio = []
        for st, n in ((stats0, count0), (stats1, count1)):
            t = st.sum(0)
            m = t[0] / n
            var = (t[1] / n - m * m).clamp_min(0.0) * (n / (n - 1.0))
            ratio.append(m / torch.sqrt(var + eps))
        prob = 1.0 / (1.0 + (ratio[0] - ratio[1]).abs())
        g = (1.0 + prob.numel() * prob / prob.sum()).float()
        gain.copy_(g)
        for t in (scale0, shift0, scale1, shift1):
            t.mul_(g)

    def tn_eval_coeffs(self, gamma, beta, rmean_s, rvar_s, rmean_t, rvar_t, eps, scale, shift):
        rs = rmean_s.double() / torch.sqrt(rvar_s.double() + eps)
        rt = rmean_t.double() / torch.sqrt(rvar_t.double() + eps)
        prob = 1.0 / (1.0 + (rs - rt).abs())
        g = 1.0 + prob.numel() * prob / prob.sum()
        sc = gamma.double() / torch.sqrt(rvar_t.double() + eps)
        scale.copy_((sc * g).float())
        shift.copy_(((beta.double() - rmean_t.double() * sc) * g).float())

    def bn_apply(self, src: Act, out, residual=None):
        u = transform(src)
        if residual is not None:
            u = u + residual
        out.copy_(u)

    def colstats(self, x, stats):
        """stats: fp64 [SLOTS, nq, C], ADDED into (any slot; consumers sum the slots)."""
        stats[0, 0] += x.double().sum(0)
        if stats.shape[1] > 1:
            stats[0, 1] += (x.double() ** 2).sum(0)

    def colstats_window(self, x, stats, c_off):
        Cc = x.shape[1]
        stats[0, 0, c_off:c_off + Cc] += x.double().sum(0)
        if stats.shape[1] > 1:
            stats[0, 1, c_off:c_off + Cc] += (x.double() ** 2).sum(0)

    def bnbwd_reduce(self, dU, y: Act, sums, lowrank=None):
        """y carries (x, scale, shift, act, mask, bn.mean/invstd).  g = dU * mask*ms * act'(a);
        sums (fp64 [3, C]) = (sum g, sum g*xhat, sum dU).  ``lowrank`` = (d [P, k], w [k, C]): dU = d @ w."""
        if lowrank is not None:
            dU = lowrank[0] @ lowrank[1]
        a = y.x * y.scale + y.shift
        g = dU * _act_grad(a, y.act)
        if y.mask is not None:
            g = g * (y.mask.to(g.dtype) * y.mask_scale)
        xhat = (y.x - y.bn.mean) * y.bn.invstd
        sums[0, 0] += g.double().sum(0)
        sums[0, 1] += (g.double() * xhat.double()).sum(0)
        sums[0, 2] += dU.double().sum(0)

    def bnbwd_finalize(self, sums, y: Act, c1, c2, dgamma, dbeta, q1_total=None):
        """Adds the border term of quirk Q1 when y.bn.q1_border, then c1 = sum g / n,
        c2 = sum g*xhat / n, dgamma = sum g*xhat, dbeta = sum g."""
        sums = sums.sum(0)
        sg, sgx = sums[0].clone(), sums[1].clone()
        if y.bn.q1_border:
            sh = y.shift.double()
            gate = torch.ones_like(sh) if y.act == ACT_NONE else (
                (sh > 0).double() if y.act == ACT_RELU else ((sh > 0) & (sh < 6)).double())
            gb = ((0.0 if q1_total is None else q1_total.double()) - sums[2]) * gate      # q1_total: frozen depthwise BN behind
            sg = sg + gb
            sgx = sgx + gb * (-y.bn.mean.double() * y.bn.invstd.double())
        c1.copy_((sg / y.bn.count).float())
        c2.copy_((sgx / y.bn.count).float())
        dgamma.copy_(sgx.float())
        dbeta.copy_(sg.float())

    def bnbwd_apply(self, dU, y: Act, c1, c2, out, addend=None, lowrank=None):
        """out = addend + scale * (g - c1 - xhat * c2)   (may run in place over dU / addend)."""
        if lowrank is not None:
            dU = lowrank[0] @ lowrank[1]
        a = y.x * y.scale + y.shift
        g = dU * _act_grad(a, y.act)
        if y.mask is not None:
            g = g * (y.mask.to(g.dtype) * y.mask_scale)
        xhat = (y.x - y.bn.mean) * y.bn.invstd
        r = y.scale * (g - c1 - xhat * c2)
        if addend is not None:
            r = r + addend
        out.copy_(r)

    def act_bwd(self, dU, y: Act, out):
        """Backward of the pending transform when BN is frozen/absent: out = dU*mask*act'(a)*scale."""
        a = y.x if y.scale is None else y.x * y.scale + y.shift
        g = dU * _act_grad(a, y.act)
        if y.mask is not None:
            g = g * (y.mask.to(g.dtype) * y.mask_scale)
        if y.scale is not None:
            g = g * y.scale
        out.copy_(g)

    # ------------------------------------------------------------------ resampling / pooling
    def upsample_fwd(self, x, N, h, w, out, H, W, stats=None):
        """Bilinear align_corners=True, NHWC [N*h*w, C] -> NHWC [N*H*W, C]; ``stats``: the output's per-channel (sum, sum of
        squares) ADDED into channels [0, C) of the fp64 [SLOTS, 2, Cs] accumulator."""
        mh, mw = _bilinear_matrix(h, H, x.device, x.dtype), _bilinear_matrix(w, W, x.device, x.dtype)
        t = x.reshape(N, h, w, -1)
        t = torch.einsum("Hh,nhwc->nHwc", mh, t)
        t = torch.einsum("Ww,nHwc->nHWc", mw, t)
        out.copy_(t.reshape(N * H * W, -1))
        if stats is not None:
            self.colstats_window(out, stats, 0)

    def upsample_stats(self, x, N, h, w, H, W, stats):
        tmp = torch.empty(N * H * W, x.shape[1], dtype=x.dtype, device=x.device)
        self.upsample_fwd(x, N, h, w, tmp, H, W, stats=stats)

    def mc_seg_head(self, feature, N, h, w, low, bnd, H, W, scale, shift, act, mask, mask_scale, wgt, bias, out):
        """x1b = W . (mask * ms * act(scale * xf + shift)) + bias on xf = cat(up(feature), low repeated over the batch, boundary)"""
        P, Cf, Cl = N * H * W, feature.shape[1], low.shape[1]
        Cc = Cf + Cl + 1
        xf = torch.empty(P, Cc, dtype=feature.dtype, device=feature.device)
        self.upsample_fwd(feature, N, h, w, xf[:, :Cf], H, W)
        xf[:, Cf:Cf + Cl] = low.repeat(P // low.shape[0], 1)
        xf[:, Cf + Cl:] = bnd
        u = transform(Act(xf, N, H, W, scale, shift, act, mask, mask_scale))
        wm = wgt.reshape(2, -1)[:, :Cc]
        out.copy_(u @ wm.t() + bias)

    def upsample_bwd(self, dout, N, H, W, dx, h, w):
        mh, mw = _bilinear_matrix(h, H, dout.device, dout.dtype), _bilinear_matrix(w, W, dout.device, dout.dtype)
        t = dout.reshape(N, H, W, -1)
        t = torch.einsum("Hh,nHWc->nhWc", mh, t)
        t = torch.einsum("Ww,nhWc->nhwc", mw, t)
        dx.copy_(t.reshape(N * h * w, -1))

    @staticmethod
    def _upconv(g, N, h, w, H, W, C, dil):
        """sum over the 9 taps of shift_t(upsample(g_t)): upsample each tap plane, read it at p + d_t, zero outside."""
        mh, mw = _bilinear_matrix(h, H, g.device, g.dtype), _bilinear_matrix(w, W, g.device, g.dtype)
        t = g.reshape(N, h, w, 9, C)
        up = torch.einsum("Hh,Ww,nhwtc->ntHWc", mh.to(g.dtype), mw.to(g.dtype), t)
        out = torch.zeros(N, H, W, C, dtype=g.dtype, device=g.device)
        for tap in range(9):
            dh, dw = (tap // 3 - 1) * dil, (tap % 3 - 1) * dil
            oh0, oh1 = max(0, -dh), min(H, H - dh)
            ow0, ow1 = max(0, -dw), min(W, W - dw)
            if oh0 < oh1 and ow0 < ow1:
                out[:, oh0:oh1, ow0:ow1] += up[:, tap, oh0 + dh:oh1 + dh, ow0 + dw:ow1 + dw]
        return out.reshape(N * H * W, C)

    def upconv_fwd(self, g, N, h, w, out, H, W, addend=None, dil=1, stats=None):
        y = self._upconv(g, N, h, w, H, W, out.shape[1], dil)
        if addend is not None:
            y = y + addend.repeat(y.shape[0] // addend.shape[0], 1)
        out.copy_(y)
        if stats is not None:
            self.colstats(out, stats)

    def upconv_bwd(self, dy, N, H, W, dg, h, w, dil=1):
        with torch.enable_grad():
            gv = torch.zeros(dg.shape, dtype=dy.dtype, device=dy.device, requires_grad=True)
            self._upconv(gv, N, h, w, H, W, dy.shape[1], dil).backward(dy)
        dg.copy_(gv.grad)

    def head_upsample_fwd(self, x, N, h, w, out):
        """NHWC [N*h*w, C<=2] -> contiguous NCHW [N, C, H, W] (bilinear, align_corners)."""
        H, W = out.shape[2], out.shape[3]
        mh, mw = _bilinear_matrix(h, H, x.device, x.dtype), _bilinear_matrix(w, W, x.device, x.dtype)
        t = x.reshape(N, h, w, -1)
        out.copy_(torch.einsum("Hh,Ww,nhwc->ncHW", mh, mw, t))

    def head_upsample_bwd(self, dout, dx, N, h, w, accumulate=False):
        H, W = dout.shape[2], dout.shape[3]
        mh, mw = _bilinear_matrix(h, H, dout.device, dout.dtype), _bilinear_matrix(w, W, dout.device, dout.dtype)
        t = torch.einsum("Hh,Ww,ncHW->nhwc", mh, mw, dout).reshape(N * h * w, -1)
        if accumulate:
            dx.add_(t)
        else:
            dx.copy_(t)

    def gap_fwd(self, x, N, out, scale):
        """out[n, c] = scale * sum over the image's pixels of x."""
        out.copy_(x.reshape(N, -1, x.shape[1]).sum(1) * scale)

    def broadcast_rows(self, g, N, out, scale, addend=None):
        """out[p, c] = addend[p, c] + scale * g[n(p), c]."""
        P = out.shape[0]
        t = (g * scale).repeat_interleave(P // N, dim=0)
        if addend is not None:
            t = t + addend
        out.copy_(t)

    def colsum(self, x, out):
        out.copy_(x.double().sum(0).float())

    STAT_SLOTS = 16

    def dropout_mask(self, mask, p, seed, offset):
        """Bernoulli(1-p) keep-mask (uint8).  The bit stream is implementation-defined (Philox on
        the device); parity tests inject masks instead of comparing streams."""
        g = torch.Generator(device="cpu").manual_seed(int(seed) * 1000003 + int(offset))
        mask.copy_((torch.rand(mask.shape, generator=g) >= p).to(torch.uint8).to(mask.device))

    # ------------------------------------------------------------------ losses / metrics
    def seg_loss_fwd(self, o, tmap, b, tbd):
        bce = F.binary_cross_entropy(torch.sigmoid(o), tmap)
        mse = F.mse_loss(torch.sigmoid(b), tbd)
        return torch.stack([bce + mse, bce, mse])

    def seg_loss_bwd(self, o, tmap, b, tbd, gscale):
        with torch.enable_grad():          # also callable from inside an autograd backward (grad mode is off there)
            o2, b2 = o.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
            (F.binary_cross_entropy(torch.sigmoid(o2), tmap) + F.mse_loss(torch.sigmoid(b2), tbd)).backward()
        return o2.grad * gscale, b2.grad * gscale

    def seg_counts(self, logits, target, thr):
        pr = torch.sigmoid(logits) > thr
        gt = target != 0
        return torch.stack([(pr & gt).sum((0, 2, 3)), pr.sum((0, 2, 3)), gt.sum((0, 2, 3))], 1).to(torch.int64)

    # ------------------------------------------------------------------ prototypes
    def mc_stats(self, preds, T):
        p = preds.reshape((T, preds.shape[0] // T) + tuple(preds.shape[1:]))
        return torch.std(torch.sigmoid(p / 2.0), dim=0), torch.mean(torch.sigmoid(p), dim=0)

    def proto_weights(self, mode, B, h, w, map_=None, logits=None, std_map=None, mean_map=None):
        m0 = m1 = None
        if mode == 0:
            m = F.interpolate(map_, size=(h, w), mode="nearest")
            a, b = m[:, 0], m[:, 1]
            ws = (a, b, 1 - a, 1 - b)
        else:
            p = torch.sigmoid(logits).reshape(B, h, w, 2)
            if mode == 1:
                ws = (p[..., 0], p[..., 1], 1 - p[..., 0], 1 - p[..., 1])
            else:
                pl = (p > 0.75).float()
                sd = F.interpolate(std_map, size=(h, w), mode="bilinear", align_corners=True)
                q = F.interpolate(mean_map, size=(h, w), mode="bilinear", align_corners=True)
                mk = (sd < 0.04).float()
                ws = (mk[:, 0] * pl[..., 0] * q[:, 0], mk[:, 1] * pl[..., 1] * q[:, 1],
                      mk[:, 0] * (1 - pl[..., 0]) * (1 - q[:, 0]), mk[:, 1] * (1 - pl[..., 1]) * (1 - q[:, 1]))
                m0, m1 = (2 * mk[:, 0]).reshape(-1), (2 * mk[:, 1]).reshape(-1)
        return torch.stack([t.reshape(-1) for t in ws], 1).contiguous(), m0, m1

    def proto_reduce(self, feat, wts, sums):
        Cc = feat.shape[1]
        sums[:, :Cc] += wts.double().t() @ feat.double()
        sums[:, Cc] += wts.double().sum(0)

    def proto_finalize(self, sums):
        Cc = sums.shape[1] - 1
        return (sums[:, :Cc] / sums[:, Cc:]).float()

    def proto_bwd(self, feat, wts, sums, dC, d_feat=None, accumulate=False, want_dw=False):
        Cc = feat.shape[1]
        cnt = sums[:, Cc:]
        coef = dC.double() / cnt                                           # [4, C]
        extra = -(dC.double() * sums[:, :Cc]).sum(1) / cnt[:, 0] ** 2      # [4]
        if d_feat is not None:
            g = (wts.double() @ coef).float()
            if accumulate:
                d_feat.add_(g)
            else:
                d_feat.copy_(g)
        if want_dw:
            return (feat.double() @ coef.t() + extra).float()
        return None

    def proto_align_fwd(self, cur_src, cur_tgt, prev_src, prev_tgt, decay):
        new_src = cur_src.clone() if prev_src is None else (1 - decay) * prev_src + decay * cur_src
        new_tgt = cur_tgt.clone() if prev_tgt is None else (1 - decay) * prev_tgt + decay * cur_tgt
        intra = sum(F.mse_loss(new_src[k], new_tgt[k]) for k in range(4))
        inter = F.mse_loss(new_src[1], new_src[3]) + F.mse_loss(new_src[0], new_src[2])
        return new_src, new_tgt, torch.stack([intra, inter])

    def proto_align_bwd(self, new_src, new_tgt, g, w_src, w_tgt):
        v = g.reshape(()) * 2.0 * (new_src - new_tgt) / new_src.shape[1]
        return w_src * v, -w_tgt * v

    def adv_loss_fwd(self, d1, d2, label, scale):
        f = F.binary_cross_entropy_with_logits
        return (scale * (f(d1, torch.full_like(d1, label)) + f(d2, torch.full_like(d2, label)))).reshape(1)

    def adv_loss_bwd(self, d1, d2, label, scale, g):
        gs = g.reshape(()) * scale
        return (torch.sigmoid(d1) - label) * gs / d1.numel(), (torch.sigmoid(d2) - label) * gs / d2.numel()

    def feat_dot4(self, feat, coef):
        Cc = feat.shape[1]
        return (feat.double() @ coef[:, :Cc].double().t() + coef[:, Cc].double()).float()

    def feat_rank4(self, wts, coef, d_feat, accumulate=False):
        g = (wts.double() @ coef[:, :d_feat.shape[1]].double()).float()
        if accumulate:
            d_feat.add_(g)
        else:
            d_feat.copy_(g)

    def adam_step(self, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
        exp_avg.lerp_(grads, 1 - beta1)
        exp_avg_sq.mul_(beta2).addcmul_(grads, grads, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        params.addcdiv_(exp_avg, exp_avg_sq.sqrt() / (bc2 ** 0.5) + eps, value=-lr / bc1)
