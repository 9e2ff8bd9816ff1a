"""Execution plan of the DeepLabV3+ generator (MobileNetV2 or ResNet-101 backbone) on the HIP kernels.

The whole generator is ONE autograd node (``networks.deeplabv3._GeneratorFn``): this module runs
its forward as a fixed sequence of kernel launches on NHWC buffers and its backward as the
hand-written reverse sequence, so no per-op autograd bookkeeping, no ``torch.cat`` and no
separate BN / ReLU / dropout passes exist on the hot path.

Reference behaviour reproduced (file:line in /root/reference):
  networks/deeplabv3.py:32-41            7-tuple outputs
  networks/backbone/mobilenet.py:61-67   quirk Q1 - the block input is zero-padded BEFORE the
                                         1x1 expand conv, so the expand BN's statistics run over
                                         (H+2d)(W+2d) positions and the depthwise conv sees
                                         relu6(shift) on its border
  networks/backbone/resnet.py:23-43,113-124   Bottleneck / ResNet.forward (BASELINE.json configs[4])
  networks/aspp.py:65-78, networks/decoder.py:45-56

Fusion scheme ("kernels"): every conv writes its raw output once and accumulates the
per-channel (sum, sum^2) for BN in its epilogue; the BN affine + ReLU/ReLU6 + dropout mask are
applied by the CONSUMER while it loads its operand tile.

``kernels`` is the binding object (``uda_clr_amd.kernels.HipKernels`` in the product; the tests
also drive this file with their fp32 torch statement of the same entry points).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import os as _os

_MC_VIRTUAL = _os.environ.get("UDA_CLR_MC_VIRTUAL", "1") != "0"      # A/B switch: stochastic passes without the x_feature matrix
_NATIVE_S2 = _os.environ.get("UDA_CLR_NATIVE_STRIDE2", "1") != "0"    # A/B switch: ResNet's stride-2 3x3 convs on the strided grid (0: stride 1 + subsample)
POISON_BUFFERS = False      # tests/test_generator_gpu.py sets it: every fp32 work matrix starts as NaN / Inf / huge values

from .acts import ACT_NONE, ACT_RELU, ACT_RELU6, Act, BNRec, nchw_view, round4
from .domain_split import DomainSplit

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (t, c, n, s) rows of the MobileNetV2 table (mobilenet.py:77-86)
_MBV2 = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1),
         (6, 160, 3, 2), (6, 320, 1, 1))

DROPOUT = {"aspp.dropout": 0.5, "decoder.last_conv_boundary.3": 0.5,
           "decoder.last_conv_boundary.7": 0.1, "decoder.last_conv.2": 0.1}


def block_plan(output_stride: int = 16):
    """[(inp, oup, stride, dilation, expand)] per inverted-residual block (mobilenet.py:88-111)."""
    plan, inp, cur, rate = [], 32, 2, 1
    for t, c, n, s in _MBV2:
        if cur == output_stride:
            stride, dil = 1, rate
            rate *= s
        else:
            stride, dil = s, 1
            cur *= s
        for i in range(n):
            plan.append((inp, c, stride if i == 0 else 1, dil, t))
            inp = c
    return plan


def _rows(g: torch.Tensor) -> torch.Tensor:
    """Logical NCHW gradient -> [P, C] rows (zero-copy when it already is channels-last)."""
    n, c, h, w = g.shape
    return g.permute(0, 2, 3, 1).reshape(n * h * w, c)


STAT_SLOTS = 16     # UDA_STAT_SLOTS: replicas of every per-channel statistics accumulator


class _Arena:
    """One zero-filled fp64 buffer per pass for ALL per-channel statistics accumulators (a single
    memset instead of one torch.zeros launch per BatchNorm)."""

    def __init__(self, like, n_doubles):
        self.buf = torch.zeros(n_doubles, dtype=torch.float64, device=like.device)
        self.off = 0

    def take(self, nq, C, halves=1):
        """[SLOTS, nq, C], or [2, SLOTS, nq, C] (one accumulator per domain half, TransNorm)."""
        n = halves * STAT_SLOTS * nq * C
        if self.off + n > self.buf.numel():
            raise RuntimeError("statistics arena exhausted")
        v = self.buf[self.off:self.off + n]
        v = v.view(STAT_SLOTS, nq, C) if halves == 1 else v.view(halves, STAT_SLOTS, nq, C)
        self.off += n
        return v


class _Ctx:
    __slots__ = ("S", "N", "dims", "w_cache", "params", "x", "arena", "nbt", "bnlog", "tn_repeat", "need_grad")

    def __init__(self):
        self.S = {}
        self.w_cache = {}
        self.arena = None
        self.nbt = []
        self.bnlog = []          # (prefix, BNRec) of every training-mode BN of this forward
        self.tn_repeat = False   # TransNorm on a batch that is x repeated twice: both halves ARE x (see forward(repeat_prefix=True))


def resnet_plan(output_stride: int = 16, layers=(3, 4, 23)):
    """[(prefix, inplanes, planes, stride, dilation, has_downsample)] per Bottleneck (resnet.py:47-111)."""
    if output_stride == 16:
        strides, dils = (1, 2, 2, 1), (1, 1, 1, 2)
    elif output_stride == 8:
        strides, dils = (1, 2, 1, 1), (1, 1, 2, 4)
    else:
        raise NotImplementedError
    plan, inp = [], 64
    for li, (planes, n) in enumerate(zip((64, 128, 256), layers), start=1):
        for b in range(n):
            st = strides[li - 1] if b == 0 else 1
            plan.append(("backbone.layer%d.%d" % (li, b), inp, planes, st, dils[li - 1],
                         b == 0 and (st != 1 or inp != 4 * planes)))
            inp = 4 * planes
    for b, mg in enumerate((1, 2, 4)):
        st = strides[3] if b == 0 else 1
        plan.append(("backbone.layer4.%d" % b, inp, 512, st, mg * dils[3], b == 0 and (st != 1 or inp != 2048)))
        inp = 2048
    return plan


class GeneratorEngine:
    C_FEAT = 256          # channels of the ASPP output that the decoder upsamples (aspp.py:59), the rest of conv0's input is low-level

    def __init__(self, kernels, output_stride: int = 16, seed: int = 1337, backbone: str = "mobilenet",
                 transnorm: bool = False):
        # TransNorm (--use_TN, networks/sync_batchnorm/batchnorm.py:436-520): training batches are normalised per
        # domain half; the same launch sequence runs through DomainSplit, which issues per-half launches
        self.tn = bool(transnorm)
        self.K = DomainSplit(kernels) if self.tn else kernels
        self.os = output_stride
        self.backbone = backbone
        self.dils = (1, 6, 12, 18) if output_stride == 16 else (1, 12, 24, 36)
        self.seed = seed
        self.rng_offset = 0
        # device-side "something was NaN / Inf" flag of the passes since the last pop_nonfinite(): the fused BN + activation
        # prologues clamp with v_med3_f32, which returns a finite bound for a NaN operand (relu(NaN) = 0 where torch propagates
        # NaN), so a diverged activation would not reach the loss.  Every conv output and every BN-backward operand passes
        # through a per-channel sum in the statistics arena, so ONE reduction over that arena per pass sees any NaN / Inf;
        # the trainers fetch the flag with their single host sync and raise like the reference's NaN checks
        # (Trainer_prototype_full.py:296-299).
        self.nonfinite = None
        # channels that receive BN statistics in one forward (stem, blocks, ASPP, decoder)
        if backbone == "mobilenet":
            self.blocks = block_plan(output_stride)
            self.c_high, self.c_low = 320, 24
            n = 32
            for inp, oup, stride, dil, t in self.blocks:
                n += (inp * t if t != 1 else 0) + inp * t + oup
        elif backbone == "resnet":
            self.rblocks = resnet_plan(output_stride)
            self.c_high, self.c_low = 2048, 256
            n = 64
            for pre, inp, planes, stride, dil, has_ds in self.rblocks:
                n += 2 * planes + 4 * planes * (2 if has_ds else 1)
        else:
            raise NotImplementedError("backbone %r" % (backbone,))
        self.bn_channels = n + 5 * 256 + 256 + 48 + 256 + 256 + 305

    # ------------------------------------------------------------------ small helpers
    def _check_arena(self, ctx):
        if ctx.arena is not None:
            bad = ~torch.isfinite(ctx.arena.buf.sum())
            self.nonfinite = bad if self.nonfinite is None else (self.nonfinite | bad)

    def pop_nonfinite(self):
        """0-dim bool tensor (or None): a NaN / Inf went through a BN statistic since the last call."""
        f, self.nonfinite = self.nonfinite, None
        return f

    @staticmethod
    def _empty(x, *shape, dtype=torch.float32):
        t = torch.empty(shape, dtype=dtype, device=x.device)
        if POISON_BUFFERS and dtype == torch.float32 and t.dim() == 2:
            # tests: what an uninitialised buffer may hold (the padding columns of a [P, round4(C)] matrix are never written)
            t.copy_(torch.tensor([float("nan"), float("inf"), -float("inf"), 3.0e38], device=x.device)[
                torch.arange(t.shape[0], device=x.device) % 4].unsqueeze(1).expand_as(t))
        return t

    def _buf(self, x, P, C):
        """[P, C] view of a fresh [P, round4(C)] buffer."""
        return self._empty(x, P, round4(C))[:, :C]

    def _stats(self, ctx, C, training):
        return ctx.arena.take(2, C, 2 if self._split(ctx) else 1) if training else None

    def _split(self, ctx):
        """True when a training batch is normalised per domain half with DIFFERENT statistics (TransNorm on a genuine batch)."""
        return self.tn and not ctx.tn_repeat

    def _w(self, ctx, key, kind):
        """Kernel-side layout of a weight, built once per forward context.  (Not cached across forwards: the fused
        multi-tensor optimizer steps of torch do not bump a parameter's autograd version, so staleness could not be
        detected.)"""
        ck = (key, kind)
        if ck not in ctx.w_cache:
            w = ctx.params[key]
            K, CF = self.K, self.C_FEAT

            def taps(t):            # [O, CF, 3, 3] -> the nine tap matrices stacked along the rows, [(t, o), CF, 1, 1]
                return t[:, :CF].permute(2, 3, 0, 1).reshape(9 * t.shape[0], CF, 1, 1).contiguous()
            ctx.w_cache[ck] = {"ohwi": K.relayout_ohwi, "dgrad": K.relayout_dgrad,
                               "dw": K.relayout_dw,
                               "dwflip": lambda t: K.relayout_dw(t).flip(0).contiguous(),
                               # decoder conv0 split into its upsampled-feature part (tap GEMMs at low resolution) and its low-level part
                               "up_taps": lambda t: K.relayout_ohwi(taps(t)),
                               "up_taps_dgrad": lambda t: K.relayout_dgrad(taps(t)),
                               "low_ohwi": lambda t: K.relayout_ohwi(t[:, CF:].contiguous()),
                               "low_dgrad": lambda t: K.relayout_dgrad(t[:, CF:].contiguous())}[kind](w)
        return ctx.w_cache[ck]

    def _bn(self, ctx, prefix, stats, count, training, scale, shift, mean=None, invstd=None,
            q1=False, N=None) -> Optional[BNRec]:
        p = ctx.params
        g, b = p[prefix + ".weight"], p[prefix + ".bias"]
        if self.tn:
            return self._tn(ctx, prefix, stats, count, training, scale, shift, mean, invstd, q1, ctx.N if N is None else N)
        rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
        if training:
            if count <= 1:
                raise ValueError("Expected more than 1 value per channel when training (%s)" % prefix)
            self.K.bn_finalize(stats, float(count), g, b, rm, rv, BN_MOMENTUM, BN_EPS,
                               scale, shift, mean, invstd)
            nbt = p.get(prefix + ".num_batches_tracked")
            if nbt is not None:
                ctx.nbt.append(nbt)
            rec = BNRec(prefix, mean, invstd, float(count), q1)
            ctx.bnlog.append((prefix, rec))
            return rec
        self.K.bn_eval_coeffs(g, b, rm, rv, BN_EPS, scale, shift)
        if not ctx.need_grad:
            return None
        # frozen BatchNorm inside a training pass (DeepLab.freeze_bn, deeplabv3.py:43-50): the backward needs xhat against the
        # running statistics; count = inf switches the batch-statistics terms off (uda_bnbwd_finalize)
        istd = torch.rsqrt(rv + BN_EPS)
        if mean is not None:             # callers that keep (mean, invstd) in their coefficient block (the ASPP's shared one)
            mean.copy_(rm)
            invstd.copy_(istd)
            return BNRec(prefix, mean, invstd, float("inf"), q1, frozen=True)
        return BNRec(prefix, rm, istd, float("inf"), q1, frozen=True)

    def _tn(self, ctx, prefix, stats, count, training, scale, shift, mean, invstd, q1, N):
        """TransNorm coefficients.  Training: scale / shift / mean / invstd are [2, C] (row h = domain half h), the
        halves' counts split ``count`` like the images (N//2 first); eval: plain [C] coefficients."""
        p, K = ctx.params, self.K
        g, b = p[prefix + ".weight"], p[prefix + ".bias"]
        rms, rvs = p[prefix + ".running_mean_source"], p[prefix + ".running_var_source"]
        rmt, rvt = p[prefix + ".running_mean_target"], p[prefix + ".running_var_target"]
        if not training:
            K.tn_eval_coeffs(g, b, rms, rvs, rmt, rvt, BN_EPS, scale, shift)
            if not ctx.need_grad:
                return None
            # frozen TransNorm inside a training pass (DeepLab.freeze_bn evals both BN kinds, deeplabv3.py:47-50; eval branch
            # batchnorm.py:497-520): z = ((x - mu_t) / sigma_t * gamma + beta) * gain with gain = 1 + alpha from the RUNNING
            # statistics of both domains, a constant of the pass.  scale / shift above already carry gain, so dx = scale * g is
            # right as it stands; dgamma / dbeta are gain * the plain sums (as for the per-half training records).
            rs, rt = rms / torch.sqrt(rvs + BN_EPS), rmt / torch.sqrt(rvt + BN_EPS)
            prob = 1.0 / (1.0 + (rs - rt).abs())
            gain = 1.0 + prob.numel() * prob / prob.sum()
            istd = torch.rsqrt(rvt + BN_EPS)
            if mean is not None:
                mean.copy_(rmt)
                invstd.copy_(istd)
                return BNRec(prefix, mean, invstd, float("inf"), q1, gain, frozen=True)
            return BNRec(prefix, rmt, istd, float("inf"), q1, gain, frozen=True)
        if ctx.tn_repeat:
            # the batch is x repeated twice, so both halves have x's statistics: distance 0, alpha = 1, every channel scaled
            # by exactly 2.  Plain coefficients over x, doubled; the running buffers of both domains get their updates from
            # mc_forward's replay (one per stochastic pass), not from this prefix forward.
            if count <= 1:
                raise ValueError("Expected more than 1 value per channel when training (%s)" % prefix)
            K.bn_finalize(stats, float(count), g, b, rms.clone(), rvs.clone(), BN_MOMENTUM, BN_EPS, scale, shift, mean, invstd)
            scale.mul_(2.0)
            shift.mul_(2.0)
            rec = BNRec(prefix, mean, invstd, float(count), q1)
            ctx.bnlog.append((prefix, rec))
            return rec
        n0 = N // 2
        per_image = count // N
        counts = (float(n0 * per_image), float((N - n0) * per_image))
        if min(counts) <= 1:
            raise ValueError("Expected more than 1 value per channel when training (%s, TransNorm halves of %d and %d "
                             "images)" % (prefix, n0, N - n0))
        for h, (rm, rv) in enumerate(((rms, rvs), (rmt, rvt))):
            K.bn_finalize(stats[h], counts[h], g, b, rm, rv, BN_MOMENTUM, BN_EPS, scale[h], shift[h], mean[h], invstd[h])
        gain = torch.empty_like(g)
        K.tn_gain(stats[0], stats[1], counts[0], counts[1], BN_EPS, scale[0], shift[0], scale[1], shift[1], gain)
        ctx.nbt.append(p[prefix + ".num_batches_tracked"])
        rec = BNRec(prefix, mean, invstd, counts, q1, gain)
        ctx.bnlog.append((prefix, rec))
        return rec

    def _coef(self, ctx, x, C, training):
        """[4, C] (scale, shift, mean, invstd), or [4, 2, C] when a training batch is normalised per domain half."""
        return self._empty(x, 4, 2, C) if (self._split(ctx) and training) else self._empty(x, 4, C)

    def _bn_act(self, ctx, prefix, y, N, H, W, stats, count, training, act, mask=None,
                mask_scale=1.0, q1=False) -> Act:
        C = y.shape[1]
        coef = self._coef(ctx, y, C, training)
        rec = self._bn(ctx, prefix, stats, count, training, coef[0], coef[1], coef[2], coef[3], q1, N)
        return Act(y, N, H, W, coef[0], coef[1], act, mask, mask_scale, rec,
                   split=N // 2 if (self._split(ctx) and training) else 0)

    def _mask(self, x, name, P, C, N, H, W, training, masks):
        if not training:
            return None, 1.0
        p = DROPOUT[name]
        m = torch.empty((P, round4(C)), dtype=torch.uint8, device=x.device)[:, :C]
        if masks is not None:
            m.copy_(_rows(masks[name].to(x.device)))
        else:
            self.K.dropout_mask(m, p, self.seed, self.rng_offset)
            self.rng_offset += 1
        return m, 1.0 / (1.0 - p)

    def _conv0(self, ctx, feature, xf, N, H16, W16, H4, W4, out, stats, y0=None):
        """decoder.last_conv_boundary[0] on cat(up(feature), low) (decoder.py:33,50-53) WITHOUT the high-resolution GEMM over
        the upsampled channels: conv and bilinear upsample are linear and the upsample acts per channel, so
        conv3x3(up(f)) = sum_t shift_t(up(f W_t^T)): nine 256 -> 256 GEMMs at 1/16 of the pixels (one 1x1 conv with 9*256
        outputs) + an interpolation pass (uda_upconv_fwd) that adds them onto the conv of the 48 low-level channels (y0).
        The same sums re-associated: 19 + 53 GFLOP instead of 367 at B = 16.  ``feature``: [N*H16*W16, 256] (N may be a
        multiple of the batch y0 was computed on: the MC passes repeat the batch and share y0).  Returns y0."""
        K, key = self.K, "decoder.last_conv_boundary.0.weight"
        CF = self.C_FEAT
        Cout = out.shape[1]
        if y0 is None:
            y0 = self._empty(out, out.shape[0], Cout)
            K.conv(Act(xf[:, CF:CF + 48], N, H4, W4), self._w(ctx, key, "low_ohwi"), 3, 1, y0)
        g = self._empty(out, feature.shape[0], 9 * Cout)
        K.conv(Act(feature, N, H16, W16), self._w(ctx, key, "up_taps"), 1, 1, g)
        if self.tn and stats is not None:          # one accumulator per domain half: a separate statistics pass over the halves' rows
            K.upconv_fwd(g, N, H16, W16, out, H4, W4, addend=y0)
            K.colstats(out, stats, N=N)
        else:
            K.upconv_fwd(g, N, H16, W16, out, H4, W4, addend=y0, stats=stats)
        return y0

    # ------------------------------------------------------------------ MobileNetV2 backbone
    def _mobilenet_forward(self, ctx, x, training):
        K, S, params = self.K, ctx.S, ctx.params
        N, _, Hin, Win = x.shape
        # ---- stem (mobilenet.py:8-13)
        H, W = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
        y0 = self._buf(x, N * H * W, 32)
        st = self._stats(ctx, 32, training)
        K.stem_fwd(x, params["backbone.features.0.0.weight"], y0, st)
        a = self._bn_act(ctx, "backbone.features.0.1", y0, N, H, W, st, N * H * W, training, ACT_RELU6)
        S["stem"] = a
        # ---- inverted residual blocks (mobilenet.py:25-67)
        recs = []
        low = None
        for i, (inp, oup, stride, dil, t) in enumerate(self.blocks, start=1):
            pre = "backbone.features.%d" % i
            zin, H, W = a, a.H, a.W
            hid = inp * t
            if t != 1:
                ye = self._buf(x, N * H * W, hid)
                st = self._stats(ctx, hid, training)
                K.conv(zin, self._w(ctx, pre + ".conv.0.weight", "ohwi"), 1, 1, ye, stats=st)
                cnt = N * (H + 2 * dil) * (W + 2 * dil)          # quirk Q1
                e = self._bn_act(ctx, pre + ".conv.1", ye, N, H, W, st, cnt, training, ACT_RELU6, q1=True)
                border, kd, kdb, kp, kpb = 1, ".conv.3", ".conv.4", ".conv.6", ".conv.7"
            else:
                e, border, kd, kdb, kp, kpb = zin, 0, ".conv.0", ".conv.1", ".conv.3", ".conv.4"
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            Po = N * Ho * Wo
            yd = self._buf(x, Po, hid)
            st = self._stats(ctx, hid, training)
            K.dwconv_fwd(e, self._w(ctx, pre + kd + ".weight", "dw"), stride, dil, border, yd, st)
            d = self._bn_act(ctx, pre + kdb, yd, N, Ho, Wo, st, Po, training, ACT_RELU6)
            yp = self._buf(x, Po, oup)
            st = self._stats(ctx, oup, training)
            K.conv(d, self._w(ctx, pre + kp + ".weight", "ohwi"), 1, 1, yp, stats=st)
            pb = self._bn_act(ctx, pre + kpb, yp, N, Ho, Wo, st, Po, training, ACT_NONE)
            use_res = stride == 1 and inp == oup
            z = self._buf(x, Po, oup)
            K.bn_apply(pb, z, zin.x if use_res else None)
            a = Act(z, N, Ho, Wo)
            recs.append(dict(pre=pre, t=t, stride=stride, dil=dil, zin=zin, e=e, d=d, pb=pb,
                             use_res=use_res, border=border, keys=(kd, kdb, kp, kpb)))
            if i == 3:
                low = a
        S["blocks"] = recs
        return a, low

    # ------------------------------------------------------------------ ResNet-101 backbone
    def _resnet_forward(self, ctx, x, training):
        """resnet.py:113-124.  The 3x3 convs of the two stride-2 bottlenecks (layer2.0, layer3.0) and their weight gradients walk
        the strided output grid in the wide-tile kernels' loaders; only their input gradient is a stride-1 conv of the
        zero-stuffed gradient."""
        K, S, params = self.K, ctx.S, ctx.params
        N, _, Hin, Win = x.shape
        H, W = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
        y0 = self._buf(x, N * H * W, 64)
        st = self._stats(ctx, 64, training)
        K.stem7_fwd(x, params["backbone.conv1.weight"], y0, st)
        a0 = self._bn_act(ctx, "backbone.bn1", y0, N, H, W, st, N * H * W, training, ACT_RELU)
        Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        z = self._buf(x, N * Hp * Wp, 64)
        idx = torch.empty((N * Hp * Wp, 64), dtype=torch.uint8, device=x.device)
        K.maxpool_fwd(a0, z, idx)
        S["stem"] = dict(a0=a0, idx=idx)
        a = Act(z, N, Hp, Wp)
        recs, low = [], None
        for pre, inp, planes, stride, dil, has_ds in self.rblocks:
            zin, H, W = a, a.H, a.W
            P = N * H * W
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            Po = N * Ho * Wo
            y1 = self._buf(x, P, planes)
            st = self._stats(ctx, planes, training)
            K.conv(zin, self._w(ctx, pre + ".conv1.weight", "ohwi"), 1, 1, y1, stats=st)
            a1 = self._bn_act(ctx, pre + ".bn1", y1, N, H, W, st, P, training, ACT_RELU)
            y2 = self._buf(x, Po, planes)
            st = self._stats(ctx, planes, training)
            if stride == 1 or _NATIVE_S2:
                K.conv(a1, self._w(ctx, pre + ".conv2.weight", "ohwi"), 3, dil, y2, stats=st, **({"stride": stride} if stride != 1 else {}))
            else:
                yfull = self._buf(x, P, planes)
                K.conv(a1, self._w(ctx, pre + ".conv2.weight", "ohwi"), 3, dil, yfull)
                K.rows_stride(yfull, N, H, W, stride, y2)
                if training:
                    K.colstats(y2, st, **({"N": N} if self.tn else {}))
                del yfull
            a2 = self._bn_act(ctx, pre + ".bn2", y2, N, Ho, Wo, st, Po, training, ACT_RELU)
            y3 = self._buf(x, Po, 4 * planes)
            st = self._stats(ctx, 4 * planes, training)
            K.conv(a2, self._w(ctx, pre + ".conv3.weight", "ohwi"), 1, 1, y3, stats=st)
            a3 = self._bn_act(ctx, pre + ".bn3", y3, N, Ho, Wo, st, Po, training, ACT_NONE)
            zs = ad = None
            if has_ds:
                zs = zin
                if stride != 1:
                    zsb = self._buf(x, Po, inp)
                    K.rows_stride(zin.x, N, H, W, stride, zsb)
                    zs = Act(zsb, N, Ho, Wo)
                yd = self._buf(x, Po, 4 * planes)
                st = self._stats(ctx, 4 * planes, training)
                K.conv(zs, self._w(ctx, pre + ".downsample.0.weight", "ohwi"), 1, 1, yd, stats=st)
                ad = self._bn_act(ctx, pre + ".downsample.1", yd, N, Ho, Wo, st, Po, training, ACT_NONE)
            zo = self._buf(x, Po, 4 * planes)
            K.bn_add_relu(a3, ad if has_ds else zin, zo)
            a = Act(zo, N, Ho, Wo)
            recs.append(dict(pre=pre, stride=stride, dil=dil, zin=zin, a1=a1, a2=a2, a3=a3, zs=zs, ad=ad, zo=a))
            if pre.endswith("layer1.2"):
                low = a
        S["rblocks"] = recs
        return a, low

    def _resnet_backward(self, ctx, G, d_z, d_low):
        """d_z: gradient w.r.t. the [P16, 2048] backbone output, d_low: w.r.t. the layer1 output."""
        K, S, x = self.K, ctx.S, ctx.x
        N = ctx.N
        for r in reversed(S["rblocks"]):
            pre, stride, dil = r["pre"], r["stride"], r["dil"]
            zin, a1, a2, a3, zs, ad, zo = r["zin"], r["a1"], r["a2"], r["a3"], r["zs"], r["ad"], r["zo"]
            H, W, Ho, Wo = zin.H, zin.W, zo.H, zo.W
            if pre.endswith("layer1.2"):
                d_z.add_(d_low)
            g = self._buf(x, zo.P, zo.C)
            K.relu_gate(d_z, zo.x, g)
            del d_z
            dy3 = self._buf(x, zo.P, zo.C)
            self._bn_backward(ctx, G, a3, g, out=dy3)
            self._wgrad(ctx, G, pre + ".conv3.weight", a2, dy3, 1, 1)
            dU2 = self._buf(x, a2.P, a2.C)
            self._dgrad(ctx, pre + ".conv3.weight", dy3, N, Ho, Wo, 1, 1, dU2)
            del dy3
            dy2 = self._bn_backward(ctx, G, a2, dU2)
            if _NATIVE_S2:
                self._wgrad(ctx, G, pre + ".conv2.weight", a1, dy2, 3, dil, stride)
            if stride != 1:                   # input gradient: stride-1 conv of the zero-stuffed gradient
                full = self._buf(x, a1.P, a2.C)
                K.rows_stride(dy2, N, H, W, stride, full, scatter=True)
                dy2 = full
            if not _NATIVE_S2:
                self._wgrad(ctx, G, pre + ".conv2.weight", a1, dy2, 3, dil)
            dU1 = self._buf(x, a1.P, a1.C)
            self._dgrad(ctx, pre + ".conv2.weight", dy2, N, H, W, 3, dil, dU1)
            del dU2, dy2
            dy1 = self._bn_backward(ctx, G, a1, dU1)
            self._wgrad(ctx, G, pre + ".conv1.weight", zin, dy1, 1, 1)
            d_zin = self._buf(x, zin.P, zin.C)
            if ad is not None:
                dyd = self._bn_backward(ctx, G, ad, g)
                self._wgrad(ctx, G, pre + ".downsample.0.weight", zs, dyd, 1, 1)
                if stride == 1:
                    self._dgrad(ctx, pre + ".downsample.0.weight", dyd, N, H, W, 1, 1, d_zin)
                else:
                    d_zs = self._buf(x, zs.P, zs.C)
                    self._dgrad(ctx, pre + ".downsample.0.weight", dyd, N, Ho, Wo, 1, 1, d_zs)
                    K.rows_stride(d_zs, N, H, W, stride, d_zin, scatter=True)
                    del d_zs
                self._dgrad(ctx, pre + ".conv1.weight", dy1, N, H, W, 1, 1, d_zin, addend=d_zin)
            else:
                self._dgrad(ctx, pre + ".conv1.weight", dy1, N, H, W, 1, 1, d_zin, addend=g)
            del g, dU1, dy1
            d_z = d_zin
        st = S["stem"]
        a0 = st["a0"]
        dU0 = self._buf(x, a0.P, 64)
        K.maxpool_bwd(d_z, st["idx"], N, a0.H, a0.W, dU0)
        dy0 = self._bn_backward(ctx, G, a0, dU0)
        dw0 = torch.empty_like(ctx.params["backbone.conv1.weight"])
        K.stem7_wgrad(x, dy0, dw0)
        G["backbone.conv1.weight"] = dw0

    def _mobilenet_backward(self, ctx, G, d_a, d_low):
        K, S, x = self.K, ctx.S, ctx.x
        N = ctx.N
        # ---- backbone, last block first (mobilenet.py:61-67)
        d_z = d_a
        dU_stem = None
        for i in range(len(S["blocks"]), 0, -1):
            r = S["blocks"][i - 1]
            pre, t, stride, dil = r["pre"], r["t"], r["stride"], r["dil"]
            kd, kdb, kp, kpb = r["keys"]
            zin, e, d, pb = r["zin"], r["e"], r["d"], r["pb"]
            No, Ho, Wo = d.N, d.H, d.W
            Hi, Wi = zin.H, zin.W
            dyp = self._buf(x, d.P, pb.C)
            self._bn_backward(ctx, G, pb, d_z, out=dyp)
            self._wgrad(ctx, G, pre + kp + ".weight", d, dyp, 1, 1)
            dUd = self._buf(x, d.P, d.C)
            self._dgrad(ctx, pre + kp + ".weight", dyp, No, Ho, Wo, 1, 1, dUd)
            dyd = self._bn_backward(ctx, G, d, dUd)
            dwg = torch.empty_like(ctx.params[pre + kd + ".weight"])
            K.dwconv_wgrad(e, dyd, stride, dil, r["border"], dwg)
            G[pre + kd + ".weight"] = dwg
            dUe = self._buf(x, zin.P, d.C)
            if stride == 1:
                # the input gradient of a stride-1 depthwise conv IS a depthwise conv of dy with the taps reversed:
                # runs on the LDS-tiled forward kernel (the flat gather kernel stays for the four stride-2 blocks)
                K.dwconv_fwd(Act(dyd, N, Hi, Wi), self._w(ctx, pre + kd + ".weight", "dwflip"), 1, dil, 0, dUe, None)
            else:
                K.dwconv_dgrad(dyd, self._w(ctx, pre + kd + ".weight", "dw"), stride, dil, N, Hi, Wi, dUe)
            del dUd, dyd, dyp
            if t != 1:
                q1_total = None
                if e.bn.frozen:
                    # quirk Q1 with a frozen depthwise BN behind: the gradient summed over ALL padded positions of the block input
                    # is colsum(dy_dw) * sum of the depthwise taps = scale_dw * dbeta_dw * sum_t w (engine docstring)
                    dbeta = G[pre + kdb + ".bias"]
                    if d.bn.gain is not None:          # frozen TransNorm: dbeta carries the gain, colsum(dy_dw) = scale * sum(g) does not need it twice
                        dbeta = dbeta / d.bn.gain
                    q1_total = (d.scale * dbeta * ctx.params[pre + kd + ".weight"].sum((1, 2, 3))).contiguous()
                dye = self._bn_backward(ctx, G, e, dUe, q1_total=q1_total)
                self._wgrad(ctx, G, pre + ".conv.0.weight", zin, dye, 1, 1)
                d_zin = self._buf(x, zin.P, zin.C)
                addend = d_z if r["use_res"] else (d_low if i == 4 else None)
                self._dgrad(ctx, pre + ".conv.0.weight", dye, N, Hi, Wi, 1, 1, d_zin, addend=addend)
                d_z = d_zin
                del dUe, dye
            else:
                dU_stem = dUe
        # ---- stem (mobilenet.py:8-13); the image itself needs no gradient
        dy0 = self._bn_backward(ctx, G, S["stem"], dU_stem)
        dw0 = torch.empty_like(ctx.params["backbone.features.0.0.weight"])
        K.stem_wgrad(x, dy0, dw0)
        G["backbone.features.0.0.weight"] = dw0

    # ------------------------------------------------------------------ forward
    def forward(self, params: Dict[str, torch.Tensor], x: torch.Tensor, training: bool,
                need_grad: bool, masks=None, repeat_prefix: bool = False, bn_training: Optional[bool] = None,
                w_share: Optional[dict] = None):
        """``repeat_prefix`` (TransNorm only): run the deterministic part of the network (up to the ASPP output before its
        dropout and the decoder's low-level branch) as it comes out for the batch ``x.repeat(2, 1, 1, 1)`` - both TransNorm halves
        are x - without touching running statistics; returns (None, ctx) for ``mc_forward``."""
        K = self.K
        ctx = _Ctx()
        if w_share is not None:          # kernel-side weight layouts (and their packed forms) shared by the passes of one step: the caller
            ctx.w_cache = w_share        # promises that no parameter changes while it hands over the same dict (DeepLab.shared_weight_layouts)
        ctx.tn_repeat = bool(repeat_prefix and self.tn)
        ctx.params, ctx.x = params, x
        ctx.need_grad = bool(need_grad)
        # ``training`` drives the dropout layers; the BatchNorm layers follow ``bn_training`` (default: the same), which
        # DeepLab.freeze_bn() turns off while the model keeps training (deeplabv3.py:43-50)
        drop_tr = training
        training = training if bn_training is None else bool(bn_training)
        S = ctx.S
        N, _, Hin, Win = x.shape
        if Hin % 16 or Win % 16:
            raise ValueError("input height/width must be multiples of 16, got %dx%d" % (Hin, Win))
        ctx.N = N
        if training:
            ctx.arena = _Arena(x, STAT_SLOTS * 2 * self.bn_channels * (2 if self.tn else 1))
        if ctx.tn_repeat and not training:
            raise ValueError("repeat_prefix describes a training-mode (batch statistics) forward")
        if self.backbone == "mobilenet":
            a, low = self._mobilenet_forward(ctx, x, training)
        else:
            a, low = self._resnet_forward(ctx, x, training)
        # ---- ASPP (aspp.py:65-78): branches write channel windows of one [P, 1280] buffer
        a17, H16, W16 = a, a.H, a.W
        P16 = N * H16 * W16
        cat = self._empty(x, P16, 1280)
        coef = self._coef(ctx, x, 1280, training)   # channel windows below: coef[q][..., sl]
        split = N // 2 if (self._split(ctx) and training) else 0
        brecs = []
        for j, dl in enumerate(self.dils, start=1):
            sl = slice(256 * (j - 1), 256 * j)
            st = self._stats(ctx, 256, training)
            key = "aspp.aspp%d" % j
            K.conv(a17, self._w(ctx, key + ".atrous_conv.weight", "ohwi"), 1 if j == 1 else 3, dl,
                   cat[:, sl], stats=st)
            brecs.append(self._bn(ctx, key + ".bn", st, P16, training, coef[0][..., sl], coef[1][..., sl],
                                  coef[2][..., sl], coef[3][..., sl]))
        gp = self._empty(x, N, self.c_high)
        K.gap_fwd(a17.x, N, gp, 1.0 / (H16 * W16))
        yg = self._empty(x, N, 256)
        st = self._stats(ctx, 256, training)
        gpa = Act(gp, N, 1, 1)
        K.conv(gpa, self._w(ctx, "aspp.global_avg_pool.1.weight", "ohwi"), 1, 1, yg, stats=st)
        sl = slice(1024, 1280)
        grec = self._bn(ctx, "aspp.global_avg_pool.2", st, N, training, coef[0][..., sl], coef[1][..., sl],
                        coef[2][..., sl], coef[3][..., sl])
        K.broadcast_rows(yg, N, cat[:, sl], 1.0)
        catA = Act(cat, N, H16, W16, coef[0], coef[1], ACT_RELU, split=split)
        y1 = self._empty(x, P16, 256)
        st = self._stats(ctx, 256, training)
        K.conv(catA, self._w(ctx, "aspp.conv1.weight", "ohwi"), 1, 1, y1, stats=st)
        m, ms = self._mask(x, "aspp.dropout", P16, 256, N, H16, W16, drop_tr, masks)
        fa = self._bn_act(ctx, "aspp.bn1", y1, N, H16, W16, st, P16, training, ACT_RELU, m, ms)
        feature = self._empty(x, P16, 256)
        K.bn_apply(fa, feature, None)
        S["aspp"] = dict(a17=a17, cat=cat, coef=coef, brecs=brecs, grec=grec, gpa=gpa, yg=yg,
                         catA=catA, fa=fa)
        # ---- decoder (decoder.py:45-56): x_bu_feature / x_feature / boundary share one buffer
        H4, W4 = low.H, low.W
        P4 = N * H4 * W4
        xf = self._empty(x, P4, 308)
        ylo = self._empty(x, P4, 48)
        st = self._stats(ctx, 48, training)
        K.conv(low, self._w(ctx, "decoder.conv1.weight", "ohwi"), 1, 1, ylo, stats=st)
        lo = self._bn_act(ctx, "decoder.bn1", ylo, N, H4, W4, st, P4, training, ACT_RELU)
        if ctx.tn_repeat:
            S["dec"] = dict(lo=lo, y0=None)
            ctx.dims = (N, Hin, Win, H16, W16, H4, W4)
            self._check_arena(ctx)
            ctx.arena, ctx.nbt = None, []
            return None, ctx
        K.bn_apply(lo, xf[:, 256:304], None)
        # statistics of decoder.last_conv's BatchNorm(305) over cat(up(feature), low, boundary) (decoder.py:23,51-53): the 256
        # upsampled channels' sums come out of the pass that writes them, the other 49 from one window pass below - no pass over
        # the whole 305-channel buffer
        st305 = self._stats(ctx, 305, training)
        fused_up = training and 256 % (feature.shape[1] // 4) == 0
        K.upsample_fwd(feature, N, H16, W16, xf[:, 0:256], H4, W4, **({"stats": st305} if fused_up else {}))
        xbu = Act(xf[:, :304], N, H4, W4)
        yb1 = self._empty(x, P4, 256)
        st = self._stats(ctx, 256, training)
        y0 = self._conv0(ctx, feature, xf, N, H16, W16, H4, W4, yb1, st)
        m, ms = self._mask(x, "decoder.last_conv_boundary.3", P4, 256, N, H4, W4, drop_tr, masks)
        b1 = self._bn_act(ctx, "decoder.last_conv_boundary.1", yb1, N, H4, W4, st, P4, training,
                          ACT_RELU, m, ms)
        yb2 = self._empty(x, P4, 256)
        st = self._stats(ctx, 256, training)
        K.conv(b1, self._w(ctx, "decoder.last_conv_boundary.4.weight", "ohwi"), 3, 1, yb2, stats=st)
        m, ms = self._mask(x, "decoder.last_conv_boundary.7", P4, 256, N, H4, W4, drop_tr, masks)
        b2 = self._bn_act(ctx, "decoder.last_conv_boundary.5", yb2, N, H4, W4, st, P4, training,
                          ACT_RELU, m, ms)
        K.conv(b2, self._w(ctx, "decoder.last_conv_boundary.8.weight", "ohwi"), 1, 1,
               xf[:, 304:305], bias=params["decoder.last_conv_boundary.8.bias"])
        st = st305
        if training:
            if fused_up:
                K.colstats_window(xf[:, 256:305], st, 256, **({"N": N} if self.tn else {}))
            else:
                K.colstats(xf[:, :305], st, **({"N": N} if self.tn else {}))
        m, ms = self._mask(x, "decoder.last_conv.2", P4, 305, N, H4, W4, drop_tr, masks)
        sa = self._bn_act(ctx, "decoder.last_conv.0", xf[:, :305], N, H4, W4, st, P4, training,
                          ACT_RELU, m, ms)
        x1b = self._buf(x, P4, 2)
        K.conv(sa, self._w(ctx, "decoder.last_conv.3.weight", "ohwi"), 1, 1, x1b,
               bias=params["decoder.last_conv.3.bias"])
        x1 = self._empty(x, N, 2, Hin, Win)
        x2 = self._empty(x, N, 1, Hin, Win)
        K.head_upsample_fwd(x1b, N, H4, W4, x1)
        K.head_upsample_fwd(xf[:, 304:305], N, H4, W4, x2)
        S["dec"] = dict(low=low, xf=xf, lo=lo, xbu=xbu, b1=b1, b2=b2, sa=sa, x1b=x1b, feature=feature, y0=y0)
        ctx.dims = (N, Hin, Win, H16, W16, H4, W4)
        if ctx.nbt:
            torch._foreach_add_(ctx.nbt, 1)          # num_batches_tracked of all 61 BNs in one launch
            ctx.nbt = []
        self._check_arena(ctx)
        ctx.arena = None
        outs = (x1, x2, nchw_view(feature, N, H16, W16), nchw_view(xf[:, :304], N, H4, W4),
                nchw_view(xf[:, :305], N, H4, W4), nchw_view(x1b, N, H4, W4),
                nchw_view(xf[:, 304:305], N, H4, W4))
        return outs, (ctx if need_grad else None)

    # ------------------------------------------------------------------ stochastic (MC-dropout) passes
    _STOCHASTIC_BN = ("decoder.last_conv_boundary.1", "decoder.last_conv_boundary.5", "decoder.last_conv.0")

    def mc_forward(self, ctx, reps: int, passes: int, out=None, masks=None):
        """The ``passes`` no-grad training-mode forwards on ``x.repeat(reps,1,1,1)`` of
        Trainer_prototype_full.py:358-368, given the context of the (grad-mode) training forward on
        ``x`` itself.  Everything up to the first dropout (backbone, ASPP convs + BNs, the decoder's
        low-level branch) is deterministic and its training-mode BN statistics on the repeated batch
        equal those on ``x``, so those activations are REUSED from ``ctx``; only the dropout-dependent
        tail (ASPP dropout -> bilinear x4 -> two 3x3 convs -> heads) is recomputed per pass on the
        reps*N batch, and the running statistics of the reused BNs receive the ``passes`` momentum
        updates the reference's full forwards would have applied (uda_bn_running_replay).
        Returns logits x1 of all passes, [passes*reps*N, 2, H, W] (pass-major, as ``preds_trg``)."""
        K, S, params, x = self.K, ctx.S, ctx.params, ctx.x
        N, Hin, Win, H16, W16, H4, W4 = ctx.dims
        N2 = reps * N
        P16, P4 = N * H16 * W16, N * H4 * W4
        A, D = S["aspp"], S["dec"]
        fa, lo = A["fa"], D["lo"]
        if out is None:
            out = self._empty(x, passes * N2, 2, Hin, Win)
        repeated_tn = ctx.tn_repeat           # ctx comes from forward(repeat_prefix=True): its BN records describe x itself
        if repeated_tn and reps != 2:
            raise ValueError("the TransNorm fast path needs the batch repeated exactly twice (its halves are the two copies)")
        ctx.tn_repeat = False                 # the dropout-dependent tail is a genuine TransNorm batch: the two copies differ
        if D.get("y0") is None:               # low-level part of conv0, shared by all passes and repetitions
            tmp = self._empty(x, P4, 48)
            K.bn_apply(lo, tmp, None)
            D["y0"] = self._empty(x, P4, 256)
            K.conv(Act(tmp, N, H4, W4), self._w(ctx, "decoder.last_conv_boundary.0.weight", "low_ohwi"), 3, 1, D["y0"])
            del tmp
        p05 = DROPOUT["aspp.dropout"]
        # The 305-channel x_feature matrix of a stochastic pass feeds only the BatchNorm(305) statistics and the 305 -> 2 head.  It is
        # not written (``virtual``): the statistics of its 256 upsampled channels come from an interpolation pass without a store
        # (uda_upsample_fwd_stats, out = NULL), those of the 48 low-level channels from the un-repeated [P4, 48] rows (once per copy),
        # the boundary channel's from its own column, and the head interpolates the upsampled channels on the fly (uda_mc_seg_head).
        virtual = hasattr(K, "mc_seg_head") and 256 % (self.C_FEAT // 4) == 0 and _MC_VIRTUAL
        if virtual:
            lo_rows = self._empty(x, P4, 48)
            K.bn_apply(lo, lo_rows, None)
            xf = None
        else:
            # one x_feature buffer for all passes: its 48 low-level channels do not depend on a dropout mask and are written once
            xf = self._empty(x, reps * P4, 308)
            for r in range(reps):
                K.bn_apply(lo, xf[r * P4:(r + 1) * P4, 256:304], None)
        for ps in range(passes):
            mk = None if masks is None else masks[ps]
            ctx.arena = _Arena(x, STAT_SLOTS * 2 * (256 + 256 + 305) * (2 if self.tn else 1))
            feature = self._empty(x, reps * P16, 256)
            for r in range(reps):
                m = torch.empty((P16, 256), dtype=torch.uint8, device=x.device)
                if mk is not None:
                    m.copy_(_rows(mk["aspp.dropout"][r * N:(r + 1) * N].to(x.device)))
                else:
                    K.dropout_mask(m, p05, self.seed, self.rng_offset)
                    self.rng_offset += 1
                K.bn_apply(Act(fa.x, N, H16, W16, fa.scale, fa.shift, ACT_RELU, m, 1.0 / (1.0 - p05)),
                           feature[r * P16:(r + 1) * P16], None)
            st305 = self._stats(ctx, 305, True)
            fused_up = 256 % (feature.shape[1] // 4) == 0
            if virtual:
                K.upsample_stats(feature, N2, H16, W16, H4, W4, st305)
            else:
                K.upsample_fwd(feature, N2, H16, W16, xf[:, 0:256], H4, W4, **({"stats": st305} if fused_up else {}))
            yb1 = self._empty(x, reps * P4, 256)
            st = self._stats(ctx, 256, True)
            # the low-level part of conv0 (y0) does not depend on a dropout mask: shared by all passes and repetitions
            self._conv0(ctx, feature, xf, N2, H16, W16, H4, W4, yb1, st, y0=D["y0"])
            m, ms = self._mask(x, "decoder.last_conv_boundary.3", reps * P4, 256, N2, H4, W4, True, mk)
            b1 = self._bn_act(ctx, "decoder.last_conv_boundary.1", yb1, N2, H4, W4, st, reps * P4, True, ACT_RELU, m, ms)
            yb2 = self._empty(x, reps * P4, 256)
            st = self._stats(ctx, 256, True)
            K.conv(b1, self._w(ctx, "decoder.last_conv_boundary.4.weight", "ohwi"), 3, 1, yb2, stats=st)
            m, ms = self._mask(x, "decoder.last_conv_boundary.7", reps * P4, 256, N2, H4, W4, True, mk)
            b2 = self._bn_act(ctx, "decoder.last_conv_boundary.5", yb2, N2, H4, W4, st, reps * P4, True, ACT_RELU, m, ms)
            bnd = self._buf(x, reps * P4, 1) if virtual else xf[:, 304:305]
            K.conv(b2, self._w(ctx, "decoder.last_conv_boundary.8.weight", "ohwi"), 1, 1, bnd,
                   bias=params["decoder.last_conv_boundary.8.bias"])
            st = st305
            m, ms = self._mask(x, "decoder.last_conv.2", reps * P4, 305, N2, H4, W4, True, mk)
            x1b = self._buf(x, reps * P4, 2)
            if virtual:
                split = self._split(ctx)
                for r in range(reps):            # the low-level channels: every copy of the batch adds the same sums (TransNorm: its half's)
                    K.colstats_window(lo_rows, st[r] if split else st, 256)
                K.colstats_window(bnd, st, 304, **({"N": N2} if self.tn else {}))
                coef = self._coef(ctx, x, 305, True)
                self._bn(ctx, "decoder.last_conv.0", st, reps * P4, True, coef[0], coef[1], coef[2], coef[3], False, N2)
                K.mc_seg_head(feature, N2, H16, W16, lo_rows, bnd, H4, W4, coef[0], coef[1], ACT_RELU, m, ms,
                              self._w(ctx, "decoder.last_conv.3.weight", "ohwi"), params["decoder.last_conv.3.bias"], x1b)
            else:
                if fused_up:
                    K.colstats_window(xf[:, 256:305], st, 256, **({"N": N2} if self.tn else {}))
                else:
                    K.colstats(xf[:, :305], st, **({"N": N2} if self.tn else {}))
                sa = self._bn_act(ctx, "decoder.last_conv.0", xf[:, :305], N2, H4, W4, st, reps * P4, True, ACT_RELU, m, ms)
                K.conv(sa, self._w(ctx, "decoder.last_conv.3.weight", "ohwi"), 1, 1, x1b, bias=params["decoder.last_conv.3.bias"])
            K.head_upsample_fwd(x1b, N2, H4, W4, out[ps * N2:(ps + 1) * N2])
            self._check_arena(ctx)
            ctx.arena = None
            if ctx.nbt:      # the three stochastic BNs of this pass
                torch._foreach_add_(ctx.nbt, 1)
                ctx.nbt = []
        # running statistics of the reused (deterministic) BNs: `passes` more updates on the repeated batch
        stoch = set(self._STOCHASTIC_BN)
        det = [(prefix, rec) for prefix, rec in ctx.bnlog if prefix not in stoch]
        for prefix, rec in det:
            if repeated_tn:      # each TransNorm half is one copy of x: both domains' buffers move towards x's statistics
                for dom in ("source", "target"):
                    K.bn_running_replay(rec.mean, rec.invstd, rec.count, passes, BN_MOMENTUM, BN_EPS,
                                        params[prefix + ".running_mean_" + dom], params[prefix + ".running_var_" + dom])
            else:
                K.bn_running_replay(rec.mean, rec.invstd, rec.count * reps, passes, BN_MOMENTUM, BN_EPS,
                                    params[prefix + ".running_mean"], params[prefix + ".running_var"])
        nbt = [params[p + ".num_batches_tracked"] for p, _ in det if (p + ".num_batches_tracked") in params]
        if nbt:
            torch._foreach_add_(nbt, passes)
        # drop the per-pass records of the stochastic BNs again (bnlog describes the grad-mode forward)
        ctx.bnlog = ctx.bnlog[:len(ctx.bnlog) - 3 * passes]
        return out

    # ------------------------------------------------------------------ backward pieces
    def _bn_backward(self, ctx, G, y: Act, dU, out=None, addend=None, keys=None, q1_total=None, lowrank=None):
        """dU: gradient w.r.t. the activated values of ``y``.  Writes the gradient w.r.t. the raw
        tensor y.x into ``out`` (default: in place over dU) and the BN parameter gradients into G.
        ``lowrank`` = (d [P, k], w [k, C]) instead of dU: the gradient is the outer product d @ w (the input gradient of a 1x1 conv
        to one or two outputs) and is formed inside the two passes, never written (``out`` is then required)."""
        K = self.K
        lr = {} if lowrank is None else {"lowrank": lowrank}
        if y.bn is None:
            raise RuntimeError("this activation's BatchNorm kept no backward record (forward ran without gradient bookkeeping)")
        C = y.C
        if y.split:
            # TransNorm: z = (xhat*gamma + beta) * gain per domain half, gain detached (batchnorm.py:495).  The per-half
            # kernels see gamma*gain as the scale, so dx is already right; the shared gamma / beta collect gain * (per-half sums)
            sums = ctx.arena.take(3, C, 2)
            cg = self._empty(y.x, 4, 2, C)
        else:
            sums = ctx.arena.take(3, C)
            cg = self._empty(y.x, 4, C)
        K.bnbwd_reduce(dU, y, sums, **lr)
        K.bnbwd_finalize(sums, y, cg[0], cg[1], cg[2], cg[3], **({} if q1_total is None else {"q1_total": q1_total}))
        out = dU if out is None else out
        K.bnbwd_apply(dU, y, cg[0], cg[1], out, addend, **lr)
        if y.split:
            dg, db = (cg[2] * y.bn.gain).sum(0), (cg[3] * y.bn.gain).sum(0)
        elif y.bn.gain is not None and y.bn.frozen:      # frozen TransNorm: one coefficient set, the constant gain on the affine gradients
            dg, db = cg[2] * y.bn.gain, cg[3] * y.bn.gain
        else:
            dg, db = cg[2], cg[3]
        if keys is None:
            keys = [(y.bn.key, slice(0, C))]
        for key, sl in keys:
            G[key + ".weight"] = dg[sl]
            G[key + ".bias"] = db[sl]
        return out

    def _wgrad(self, ctx, G, key, src: Act, dy, ksize, dil, stride=1):
        dw = torch.empty_like(ctx.params[key])
        self.K.conv_wgrad(src, dy, ksize, dil, dw, **({"stride": stride} if stride != 1 else {}))
        G[key] = dw

    def _dgrad(self, ctx, key, dy, N, H, W, ksize, dil, out, addend=None):
        self.K.conv(Act(dy, N, H, W), self._w(ctx, key, "dgrad"), ksize, dil, out, addend=addend)
        return out

    def _bias_grad(self, G, key, dy):
        g = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
        self.K.colsum(dy, g)
        G[key] = g

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, grads):
        """grads: 7 optional tensors shaped like the forward outputs.  Returns {param key: grad}."""
        K = self.K
        gx1, gx2, gfeat, gxbu, gxf, gx1b, gx2b = grads
        S, x = ctx.S, ctx.x
        N, Hin, Win, H16, W16, H4, W4 = ctx.dims
        P4, P16 = N * H4 * W4, N * H16 * W16
        G: Dict[str, torch.Tensor] = {}
        ctx.arena = _Arena(x, STAT_SLOTS * 3 * (self.bn_channels + 64) * (2 if self.tn else 1))
        D = S["dec"]
        xf = D["xf"]
        # ---- heads (deeplabv3.py:39-40)
        d_x1b = self._buf(x, P4, 2)
        if gx1 is not None:
            K.head_upsample_bwd(gx1.contiguous(), d_x1b, N, H4, W4, False)
        else:
            d_x1b.zero_()
        if gx1b is not None:
            d_x1b.add_(_rows(gx1b))
        if gxf is not None:           # (the usual case: the prototype losses read x_feature) - its gradient starts the buffer, no zero fill + add
            d_xf = torch.empty(P4, 308, dtype=torch.float32, device=x.device)
            d_xf[:, :305].copy_(_rows(gxf))
            d_xf[:, 305:].zero_()
        else:
            d_xf = torch.zeros(P4, 308, dtype=torch.float32, device=x.device)
        if gxbu is not None:
            d_xf[:, :304].add_(_rows(gxbu))
        if gx2b is not None:
            d_xf[:, 304:305].add_(_rows(gx2b))
        if gx2 is not None:
            K.head_upsample_bwd(gx2.contiguous(), d_xf[:, 304:305], N, H4, W4, True)
        # ---- decoder.last_conv: BN(305) -> ReLU -> Dropout -> 1x1 (decoder.py:23-32)
        sa = D["sa"]
        self._wgrad(ctx, G, "decoder.last_conv.3.weight", sa, d_x1b, 1, 1)
        self._bias_grad(G, "decoder.last_conv.3.bias", d_x1b)
        # the input gradient of the 305 -> 2 head is the outer product d_x1b @ W: formed inside the BN-backward passes
        w_head = ctx.params["decoder.last_conv.3.weight"].reshape(2, 305).contiguous()
        self._bn_backward(ctx, G, sa, None, out=d_xf[:, :305], addend=d_xf[:, :305],
                          lowrank=(d_x1b, w_head))
        # ---- boundary head (decoder.py:33-41)
        b1, b2 = D["b1"], D["b2"]
        d_x2b = d_xf[:, 304:305]
        self._wgrad(ctx, G, "decoder.last_conv_boundary.8.weight", b2, d_x2b, 1, 1)
        self._bias_grad(G, "decoder.last_conv_boundary.8.bias", d_x2b)
        dy2 = self._empty(x, P4, 256)
        w_bnd = ctx.params["decoder.last_conv_boundary.8.weight"].reshape(1, 256).contiguous()
        self._bn_backward(ctx, G, b2, None, out=dy2, lowrank=(d_x2b, w_bnd))       # 256 -> 1 head: rank one
        self._wgrad(ctx, G, "decoder.last_conv_boundary.4.weight", b1, dy2, 3, 1)
        dU1 = self._empty(x, P4, 256)
        self._dgrad(ctx, "decoder.last_conv_boundary.4.weight", dy2, N, H4, W4, 3, 1, dU1)
        del dy2
        dy1 = self._bn_backward(ctx, G, b1, dU1)
        # conv0, split as in _conv0: low-level part on the high-resolution kernels, upsampled part through the adjoint
        # interpolation (dG) and low-resolution GEMMs
        key0, CF = "decoder.last_conv_boundary.0.weight", self.C_FEAT
        w0 = ctx.params[key0]
        dw_low = self._empty(x, w0.shape[0], w0.shape[1] - CF, 3, 3)
        K.conv_wgrad(Act(xf[:, CF:304], N, H4, W4), dy1, 3, 1, dw_low)
        K.conv(Act(dy1, N, H4, W4), self._w(ctx, key0, "low_dgrad"), 3, 1, d_xf[:, CF:304], addend=d_xf[:, CF:304])
        dG = self._empty(x, P16, 9 * w0.shape[0])
        K.upconv_bwd(dy1, N, H4, W4, dG, H16, W16)
        dw_taps = self._empty(x, 9 * w0.shape[0], CF, 1, 1)
        K.conv_wgrad(Act(D["feature"], N, H16, W16), dG, 1, 1, dw_taps)
        dw0 = torch.empty_like(w0)
        dw0[:, :CF] = dw_taps.view(3, 3, w0.shape[0], CF).permute(2, 3, 0, 1)
        dw0[:, CF:] = dw_low
        G[key0] = dw0
        del dU1, dy1, dw_low, dw_taps
        # ---- low-level branch (decoder.py:46-48)
        lo, low = D["lo"], D["low"]
        dylo = self._empty(x, P4, 48)
        self._bn_backward(ctx, G, lo, d_xf[:, 256:304], out=dylo)
        self._wgrad(ctx, G, "decoder.conv1.weight", low, dylo, 1, 1)
        d_low = self._buf(x, P4, self.c_low)
        self._dgrad(ctx, "decoder.conv1.weight", dylo, N, H4, W4, 1, 1, d_low)
        # ---- bilinear x4 of the ASPP output (decoder.py:50)
        d_feat = self._empty(x, P16, 256)
        K.upsample_bwd(d_xf[:, 0:256], N, H4, W4, d_feat, H16, W16)
        del d_xf
        K.conv(Act(dG, N, H16, W16), self._w(ctx, key0, "up_taps_dgrad"), 1, 1, d_feat, addend=d_feat)
        del dG
        if gfeat is not None:
            d_feat.add_(_rows(gfeat))
        # ---- ASPP (aspp.py:65-78)
        A = S["aspp"]
        dy1a = self._bn_backward(ctx, G, A["fa"], d_feat)
        self._wgrad(ctx, G, "aspp.conv1.weight", A["catA"], dy1a, 1, 1)
        dUc = self._empty(x, P16, 1280)
        self._dgrad(ctx, "aspp.conv1.weight", dy1a, N, H16, W16, 1, 1, dUc)
        coef, cat = A["coef"], A["cat"]
        catA = A["catA"]
        w4, wg = slice(0, 1024), slice(1024, 1280)
        if catA.split:
            n0 = catA.split
            cnt4, cntg = (float(n0 * H16 * W16), float((N - n0) * H16 * W16)), (float(n0), float(N - n0))
            gain4 = torch.cat([r.gain for r in A["brecs"]])
            gaing = A["grec"].gain
        else:
            cnt4, cntg, gain4, gaing = A["brecs"][0].count, A["grec"].count, None, None      # P16 and N, or inf when frozen
            if A["brecs"][0].gain is not None:      # frozen TransNorm: the constant gains of the four branches / the pooling branch
                gain4, gaing = torch.cat([r.gain for r in A["brecs"]]), A["grec"].gain
        c4 = Act(cat[:, :1024], N, H16, W16, coef[0][..., w4], coef[1][..., w4], ACT_RELU, None, 1.0,
                 BNRec("aspp", coef[2][..., w4], coef[3][..., w4], cnt4, False, gain4, frozen=A["brecs"][0].frozen), split=catA.split)
        keys = [("aspp.aspp%d.bn" % j, slice(256 * (j - 1), 256 * j)) for j in (1, 2, 3, 4)]
        dyc = self._bn_backward(ctx, G, c4, dUc[:, :1024], keys=keys)
        dUg = self._empty(x, N, 256)
        K.gap_fwd(dUc[:, 1024:1280], N, dUg, 1.0)
        yga = Act(A["yg"], N, 1, 1, coef[0][..., wg], coef[1][..., wg], ACT_RELU, None, 1.0,
                  BNRec("aspp.global_avg_pool.2", coef[2][..., wg], coef[3][..., wg], cntg, False, gaing, frozen=A["grec"].frozen),
                  split=catA.split)
        dyg = self._bn_backward(ctx, G, yga, dUg)
        self._wgrad(ctx, G, "aspp.global_avg_pool.1.weight", A["gpa"], dyg, 1, 1)
        d_gp = self._empty(x, N, self.c_high)
        self._dgrad(ctx, "aspp.global_avg_pool.1.weight", dyg, N, 1, 1, 1, 1, d_gp)
        d_a = self._empty(x, P16, self.c_high)
        K.broadcast_rows(d_gp, N, d_a, 1.0 / (H16 * W16), None)
        a17 = A["a17"]
        for j, dl in enumerate(self.dils, start=1):
            key = "aspp.aspp%d.atrous_conv.weight" % j
            sl = slice(256 * (j - 1), 256 * j)
            k = 1 if j == 1 else 3
            self._wgrad(ctx, G, key, a17, dyc[:, sl], k, dl)
            self._dgrad(ctx, key, dyc[:, sl], N, H16, W16, k, dl, d_a, addend=d_a)
        del dUc, dyc
        if self.backbone == "mobilenet":
            self._mobilenet_backward(ctx, G, d_a, d_low)
        else:
            self._resnet_backward(ctx, G, d_a, d_low)
        self._check_arena(ctx)
        ctx.arena = None
        return G
