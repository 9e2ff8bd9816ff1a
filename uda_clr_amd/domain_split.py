"""TransNorm execution (``--use_TN``): every launch that touches a per-domain normalisation runs once per domain half.

The reference's TransNorm layer (networks/sync_batchnorm/batchnorm.py:445-495) normalises the first N//2 images of a
training batch with their own batch statistics and the rest with theirs.  In this engine a normalisation is a
pair of per-channel coefficients that the CONSUMER kernel applies while loading, and batch statistics are
accumulated in the PRODUCER's epilogue, so "two statistics, two coefficient sets" maps onto the existing kernels
without touching them: a domain half is a contiguous row range of every NHWC matrix, and

  * a kernel that accumulates statistics gets one accumulator per half ([2, SLOTS, nq, C]) and is launched per half,
  * a kernel that consumes a split activation (``Act.split > 0``, coefficient rows [2, C]) is launched per half,
  * weight gradients of the two halves are added, the shared gamma / beta gradients are gain-weighted sums.

``DomainSplit`` wraps the kernel binding object (HipKernels, or the tests' torch statement) with exactly that
rule; the engine's launch sequence stays the one of the plain-BN network.  Launches on unsplit operands with no
statistics (input gradients, resampling, dropout masks) pass through untouched, on the whole batch.
"""
from __future__ import annotations

import torch

from .acts import Act


def _rows(t, N, n0, n1):
    ppi = t.shape[0] // N
    return t[n0 * ppi:n1 * ppi]


class DomainSplit:
    def __init__(self, kernels):
        self.K = kernels
        self.name = getattr(kernels, "name", "kernels") + "+domain-split"

    def __getattr__(self, name):          # everything that needs no split
        return getattr(self.K, name)

    @staticmethod
    def _halves(N):
        n0 = N // 2                        # batchnorm.py:446: batch_size = input.size()[0] // 2
        return ((0, 0, n0), (1, n0, N))

    @staticmethod
    def _sub(a: Act, h, n0, n1):
        if a.split:
            assert a.split == n0 or a.split == n1, "domain halves of operand and launch disagree"
            return a.half(h)
        rows = _rows(a.x, a.N, n0, n1)
        return Act(rows, n1 - n0, a.H, a.W, a.scale, a.shift, a.act,
                   None if a.mask is None else _rows(a.mask, a.N, n0, n1), a.mask_scale, a.bn, a.meta)

    # ------------------------------------------------------------------ forward producers
    def conv(self, src: Act, w, ksize, dil, out, bias=None, addend=None, stats=None, origin=0, stride=1):
        if not src.split and (stats is None or stats.dim() == 3):
            return self.K.conv(src, w, ksize, dil, out, bias, addend, stats, origin=origin, stride=stride)
        for h, n0, n1 in self._halves(src.N):
            self.K.conv(self._sub(src, h, n0, n1), w, ksize, dil, _rows(out, src.N, n0, n1), bias,
                        None if addend is None else _rows(addend, src.N, n0, n1),
                        None if stats is None else stats[h], origin=origin, stride=stride)

    def dwconv_fwd(self, src: Act, w9c, stride, dil, border_mode, out, stats=None):
        if not src.split and (stats is None or stats.dim() == 3):
            return self.K.dwconv_fwd(src, w9c, stride, dil, border_mode, out, stats)
        for h, n0, n1 in self._halves(src.N):
            self.K.dwconv_fwd(self._sub(src, h, n0, n1), w9c, stride, dil, border_mode, _rows(out, src.N, n0, n1),
                              None if stats is None else stats[h])

    def stem_fwd(self, x, w, out, stats=None):
        if stats is None or stats.dim() == 3:
            return self.K.stem_fwd(x, w, out, stats)
        N = x.shape[0]
        for h, n0, n1 in self._halves(N):
            self.K.stem_fwd(x[n0:n1], w, _rows(out, N, n0, n1), stats[h])

    def stem7_fwd(self, x, w, out, stats=None):
        if stats is None or stats.dim() == 3:
            return self.K.stem7_fwd(x, w, out, stats)
        N = x.shape[0]
        for h, n0, n1 in self._halves(N):
            self.K.stem7_fwd(x[n0:n1], w, _rows(out, N, n0, n1), stats[h])

    def maxpool_fwd(self, src: Act, out, idx):
        if not src.split:
            return self.K.maxpool_fwd(src, out, idx)
        for h, n0, n1 in self._halves(src.N):
            self.K.maxpool_fwd(src.half(h), _rows(out, src.N, n0, n1), _rows(idx, src.N, n0, n1))

    def bn_add_relu(self, a: Act, b: Act, out):
        if not a.split and not b.split:
            return self.K.bn_add_relu(a, b, out)
        N = a.N
        for h, n0, n1 in self._halves(N):
            self.K.bn_add_relu(self._sub(a, h, n0, n1), self._sub(b, h, n0, n1), _rows(out, N, n0, n1))

    def colstats(self, x, stats, N=None):
        if stats.dim() == 3:
            return self.K.colstats(x, stats)
        for h, n0, n1 in self._halves(N):
            self.K.colstats(_rows(x, N, n0, n1), stats[h])

    def colstats_window(self, x, stats, c_off, N=None):
        if stats.dim() == 3:
            return self.K.colstats_window(x, stats, c_off)
        for h, n0, n1 in self._halves(N):
            self.K.colstats_window(_rows(x, N, n0, n1), stats[h], c_off)

    def upsample_fwd(self, x, N, h, w, out, H, W, stats=None):
        if stats is None or stats.dim() == 3:
            return self.K.upsample_fwd(x, N, h, w, out, H, W, stats) if stats is not None else self.K.upsample_fwd(x, N, h, w, out, H, W)
        for hf, n0, n1 in self._halves(N):
            self.K.upsample_fwd(_rows(x, N, n0, n1), n1 - n0, h, w, _rows(out, N, n0, n1), H, W, stats[hf])

    def upsample_stats(self, x, N, h, w, H, W, stats):
        if stats.dim() == 3:
            return self.K.upsample_stats(x, N, h, w, H, W, stats)
        for hf, n0, n1 in self._halves(N):
            self.K.upsample_stats(_rows(x, N, n0, n1), n1 - n0, h, w, H, W, stats[hf])

    def mc_seg_head(self, feature, N, h, w, low, bnd, H, W, scale, shift, act, mask, mask_scale, wgt, bias, out):
        if scale.dim() == 1:
            return self.K.mc_seg_head(feature, N, h, w, low, bnd, H, W, scale, shift, act, mask, mask_scale, wgt, bias, out)
        for hf, n0, n1 in self._halves(N):          # per-half coefficients [2, C]; the low-level rows are shared
            self.K.mc_seg_head(_rows(feature, N, n0, n1), n1 - n0, h, w, low, _rows(bnd, N, n0, n1), H, W, scale[hf], shift[hf], act,
                               None if mask is None else _rows(mask, N, n0, n1), mask_scale, wgt, bias, _rows(out, N, n0, n1))

    def bn_apply(self, src: Act, out, residual=None):
        if not src.split:
            return self.K.bn_apply(src, out, residual)
        for h, n0, n1 in self._halves(src.N):
            self.K.bn_apply(src.half(h), _rows(out, src.N, n0, n1),
                            None if residual is None else _rows(residual, src.N, n0, n1))

    # ------------------------------------------------------------------ backward
    def conv_wgrad(self, src: Act, dy, ksize, dil, dw, origin=0, stride=1):
        if not src.split:
            return self.K.conv_wgrad(src, dy, ksize, dil, dw, origin=origin, stride=stride)
        d1 = torch.empty_like(dw)
        for (h, n0, n1), dst in zip(self._halves(src.N), (dw, d1)):
            self.K.conv_wgrad(src.half(h), _rows(dy, src.N, n0, n1), ksize, dil, dst, origin=origin, stride=stride)
        dw.add_(d1)

    def dwconv_wgrad(self, src: Act, dy, stride, dil, border_mode, dw):
        if not src.split:
            return self.K.dwconv_wgrad(src, dy, stride, dil, border_mode, dw)
        d1 = torch.empty_like(dw)
        for (h, n0, n1), dst in zip(self._halves(src.N), (dw, d1)):
            self.K.dwconv_wgrad(src.half(h), _rows(dy, src.N, n0, n1), stride, dil, border_mode, dst)
        dw.add_(d1)

    @staticmethod
    def _lr(lowrank, N, n0, n1):
        return None if lowrank is None else (_rows(lowrank[0], N, n0, n1), lowrank[1])

    def bnbwd_reduce(self, dU, y: Act, sums, lowrank=None):
        kw = {} if lowrank is None else {"lowrank": lowrank}
        if not y.split:
            return self.K.bnbwd_reduce(dU, y, sums, **kw)
        for h, n0, n1 in self._halves(y.N):
            kh = {} if lowrank is None else {"lowrank": self._lr(lowrank, y.N, n0, n1)}
            self.K.bnbwd_reduce(None if dU is None else _rows(dU, y.N, n0, n1), y.half(h), sums[h], **kh)

    def bnbwd_finalize(self, sums, y: Act, c1, c2, dgamma, dbeta, q1_total=None):
        if not y.split:
            return self.K.bnbwd_finalize(sums, y, c1, c2, dgamma, dbeta, q1_total)
        for h in (0, 1):
            self.K.bnbwd_finalize(sums[h], y.half(h), c1[h], c2[h], dgamma[h], dbeta[h])

    def bnbwd_apply(self, dU, y: Act, c1, c2, out, addend=None, lowrank=None):
        kw = {} if lowrank is None else {"lowrank": lowrank}
        if not y.split:
            return self.K.bnbwd_apply(dU, y, c1, c2, out, addend, **kw)
        for h, n0, n1 in self._halves(y.N):
            kh = {} if lowrank is None else {"lowrank": self._lr(lowrank, y.N, n0, n1)}
            self.K.bnbwd_apply(None if dU is None else _rows(dU, y.N, n0, n1), y.half(h), c1[h], c2[h], _rows(out, y.N, n0, n1),
                               None if addend is None else _rows(addend, y.N, n0, n1), **kh)
