"""Per-kernel parity cases: HIP kernel (through the C ABI) vs. the fp32 torch statement in
tests/kernel_spec.py, on seeded inputs.  ``CASES`` is a list of (name, fn); fn(dev) returns the
worst relative error (max |a-b| / max |b|) over the outputs of that case and the tolerance.
Used by tests/test_kernels_gpu.py (asserting) and tests/gpu_report.py (printing everything)."""
from __future__ import annotations

import torch

from kernel_spec import SpecKernels
from uda_clr_amd.acts import ACT_NONE, ACT_RELU, ACT_RELU6, Act, BNRec, round4

SPEC = SpecKernels()
_HIP = None
DETAIL = {}          # last case's per-output errors (for the report)


def hip():
    global _HIP
    if _HIP is None:
        from uda_clr_amd.kernels import HipKernels
        _HIP = HipKernels()
    return _HIP


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if not torch.isfinite(a).all():
        return float("inf")
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def _poison(buf):
    """Fill a float matrix with what an uninitialised buffer may hold: NaN, +-Inf and huge finite values by row (the fused
    clamps swallow a NaN, so NaN alone would not show a padding lane that takes part in a sum; Inf * 0 does)."""
    vals = torch.tensor([float("nan"), float("inf"), -float("inf"), 3.0e38], dtype=buf.dtype, device=buf.device)
    buf.copy_(vals[torch.arange(buf.shape[0], device=buf.device) % 4].unsqueeze(1).expand_as(buf))
    return buf


def padded(P, C, g, dev=None, scale=1.0, dtype=torch.float32):
    """[P, C] view of a [P, round4(C)+4] buffer whose padding holds NaN / Inf / huge values (must never leak)."""
    buf = _poison(torch.empty((P, round4(C) + 4), dtype=dtype))
    buf[:, :C] = torch.randn(P, C, generator=g) * scale
    if dev is not None:
        buf = buf.to(dev)
    return buf[:, :C]


def to_dev(t, dev):
    """Move a [P, C] view keeping its row stride (so padding / alignment are preserved)."""
    if t is None:
        return None
    if t.dim() == 2 and t.stride(0) != t.shape[1]:
        base = torch.empty(t.shape[0], t.stride(0), dtype=t.dtype, device=dev)
        if t.dtype.is_floating_point:
            _poison(base)
        elif t.dtype == torch.uint8:
            base.copy_((torch.arange(base.numel(), device=dev) % 251).to(torch.uint8).view_as(base))      # keep-mask padding: arbitrary bytes
        v = base[:, :t.shape[1]]
        v.copy_(t)
        return v
    return t.to(dev)


def act_to(a: Act, dev) -> Act:
    bn = None
    if a.bn is not None:
        bn = BNRec(a.bn.key, a.bn.mean.to(dev), a.bn.invstd.to(dev), a.bn.count, a.bn.q1_border)
    return Act(to_dev(a.x, dev), a.N, a.H, a.W, None if a.scale is None else a.scale.to(dev),
               None if a.shift is None else a.shift.to(dev), a.act, to_dev(a.mask, dev), a.mask_scale, bn)


def make_src(N, H, W, C, g, lazy=True, act=ACT_RELU, mask=False, bn=False, q1=False):
    P = N * H * W
    x = padded(P, C, g)
    scale = shift = None
    if lazy:
        scale = 0.5 + torch.rand(C, generator=g)
        shift = 0.3 * torch.randn(C, generator=g)
    m, ms = None, 1.0
    if mask:
        mb = torch.zeros(P, round4(C), dtype=torch.uint8)
        mb[:, :C] = (torch.rand(P, C, generator=g) > 0.3).to(torch.uint8)
        m, ms = mb[:, :C], 1.0 / 0.7
    rec = None
    if bn:
        rec = BNRec("t", 0.2 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g), float(P * (1.3 if q1 else 1.0)), q1)
    return Act(x, N, H, W, scale, shift, act if lazy else ACT_NONE, m, ms, rec)


# ------------------------------------------------------------------------------------- cases
def case_conv(N, H, W, Cin, Cout, k, dil, lazy=True, mask=False, bias=False, addend=False, stats=True, seed=0, origin=0, stride=1):
    def run(dev):
        g = gen(seed)
        src = make_src(N, H, W, Cin, g, lazy, ACT_RELU6 if Cin % 8 else ACT_RELU, mask)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b = torch.randn(Cout, generator=g) if bias else None
        P = N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1)
        kw = {"stride": stride} if stride != 1 else {}
        ad = padded(P, Cout, g) if addend else None
        out_r = padded(P, Cout, g)
        st_r = torch.zeros(16, 2, Cout, dtype=torch.float64) if stats else None
        SPEC.conv(src, SPEC.relayout_ohwi(w), k, dil, out_r, b, ad, st_r, origin=origin, **kw)
        K = hip()
        out_h = to_dev(padded(P, Cout, g), dev)
        st_h = torch.zeros(16, 2, Cout, dtype=torch.float64, device=dev) if stats else None
        wl = K.relayout_ohwi(w.to(dev))
        errs = [rel(wl, SPEC.relayout_ohwi(w))]
        K.conv(act_to(src, dev), wl, k, dil, out_h, None if b is None else b.to(dev), to_dev(ad, dev), st_h, origin=origin, **kw)
        errs.append(rel(out_h, out_r))
        if stats:
            errs.append(rel(st_h.sum(0), st_r.sum(0)))
        return max(errs), 2e-5
    return run


def case_dgrad(N, H, W, Cin, Cout, k, dil, accumulate=False, seed=1):
    """input-gradient of a conv = conv of dy with the dgrad weight layout"""
    def run(dev):
        g = gen(seed)
        P = N * H * W
        dy = padded(P, Cout, g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        ad = padded(P, Cin, g) if accumulate else None
        ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy.reshape(N, H, W, Cout).permute(0, 3, 1, 2), 1,
                                         dil * (k // 2), dil).permute(0, 2, 3, 1).reshape(P, Cin)
        if ad is not None:
            ref = ref + ad
        K = hip()
        wd = K.relayout_dgrad(w.to(dev))
        e0 = rel(wd, SPEC.relayout_dgrad(w))
        out = to_dev(padded(P, Cin, g), dev)
        adh = to_dev(ad, dev)
        if accumulate:      # in-place accumulate, as the engine uses it
            out.copy_(adh)
            adh = out
        K.conv(Act(to_dev(dy, dev), N, H, W), wd, k, dil, out, addend=adh)
        return max(e0, rel(out, ref)), 2e-5
    return run


def case_wgrad(N, H, W, Cin, Cout, k, dil, lazy=True, mask=False, seed=2, origin=0, stride=1):
    def run(dev):
        g = gen(seed)
        src = make_src(N, H, W, Cin, g, lazy, ACT_RELU, mask)
        dy = padded(N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1), Cout, g)
        ref = torch.empty(Cout, Cin, k, k)
        kw = {"stride": stride} if stride != 1 else {}
        SPEC.conv_wgrad(src, dy, k, dil, ref, origin=origin, **kw)
        out = torch.empty(Cout, Cin, k, k, device=dev)
        hip().conv_wgrad(act_to(src, dev), to_dev(dy, dev), k, dil, out, origin=origin, **kw)
        return rel(out, ref), 3e-5
    return run


def case_dw(N, H, W, C, stride, dil, border, seed=3):
    def run(dev):
        g = gen(seed)
        src = make_src(N, H, W, C, g, True, ACT_RELU6)
        w = torch.randn(C, 1, 3, 3, generator=g) / 3.0
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        Po = N * Ho * Wo
        K = hip()
        w9 = SPEC.relayout_dw(w)
        w9h = K.relayout_dw(w.to(dev))
        errs = [rel(w9h, w9)]
        y_r, st_r = padded(Po, C, g), torch.zeros(16, 2, C, dtype=torch.float64)
        SPEC.dwconv_fwd(src, w9, stride, dil, border, y_r, st_r)
        y_h, st_h = to_dev(padded(Po, C, g), dev), torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
        sh = act_to(src, dev)
        K.dwconv_fwd(sh, w9h, stride, dil, border, y_h, st_h)
        errs += [rel(y_h, y_r), rel(st_h.sum(0), st_r.sum(0))]
        dy = padded(Po, C, g)
        dx_r = padded(N * H * W, C, g)
        SPEC.dwconv_dgrad(dy, w9, stride, dil, N, H, W, dx_r)
        dx_h = to_dev(padded(N * H * W, C, g), dev)
        K.dwconv_dgrad(to_dev(dy, dev), w9h, stride, dil, N, H, W, dx_h)
        errs.append(rel(dx_h, dx_r))
        dw_r = torch.empty(C, 1, 3, 3)
        SPEC.dwconv_wgrad(src, dy, stride, dil, border, dw_r)
        dw_h = torch.empty(C, 1, 3, 3, device=dev)
        K.dwconv_wgrad(sh, to_dev(dy, dev), stride, dil, border, dw_h)
        errs.append(rel(dw_h, dw_r))
        return max(errs), 3e-5
    return run


def case_stem(N, H, W, seed=4):
    def run(dev):
        g = gen(seed)
        x = torch.randn(N, 3, H, W, generator=g)
        w = torch.randn(32, 3, 3, 3, generator=g) / 5.0
        Po = N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        y_r, st_r = padded(Po, 32, g), torch.zeros(16, 2, 32, dtype=torch.float64)
        SPEC.stem_fwd(x, w, y_r, st_r)
        K = hip()
        y_h, st_h = to_dev(padded(Po, 32, g), dev), torch.zeros(16, 2, 32, dtype=torch.float64, device=dev)
        K.stem_fwd(x.to(dev), w.to(dev), y_h, st_h)
        dy = padded(Po, 32, g)
        dw_r, dw_h = torch.empty(32, 3, 3, 3), torch.empty(32, 3, 3, 3, device=dev)
        SPEC.stem_wgrad(x, dy, dw_r)
        K.stem_wgrad(x.to(dev), to_dev(dy, dev), dw_h)
        return max(rel(y_h, y_r), rel(st_h.sum(0), st_r.sum(0)), rel(dw_h, dw_r)), 3e-5
    return run


def case_transnorm(C, n0=300, n1=500, seed=9):
    """uda_tn_gain / uda_tn_eval_coeffs against their torch statement (batchnorm.py:474-520)."""
    def run(dev):
        g = gen(seed)
        K = hip()
        xs = [2.0 * torch.randn(n0, C, generator=g) + 0.5, 0.7 * torch.randn(n1, C, generator=g) - 0.2]
        st = torch.zeros(2, 16, 2, C, dtype=torch.float64)
        for h in (0, 1):
            SPEC.colstats(xs[h], st[h])
        st = st[:, torch.randperm(16, generator=g)]                  # any slot may hold the sums
        gamma, beta = 0.5 + torch.rand(C, generator=g), torch.randn(C, generator=g)
        cr = torch.randn(2, 2, C, generator=g)                        # [scale | shift][half][C], pre-filled as after bn_finalize
        ch = cr.to(dev)
        gr, gh = torch.empty(C), torch.empty(C, device=dev)
        SPEC.tn_gain(st[0], st[1], float(n0), float(n1), 1e-5, cr[0, 0], cr[1, 0], cr[0, 1], cr[1, 1], gr)
        sth = st.to(dev)
        K.tn_gain(sth[0], sth[1], float(n0), float(n1), 1e-5, ch[0, 0], ch[1, 0], ch[0, 1], ch[1, 1], gh)
        errs = [rel(gh, gr), rel(ch, cr)]
        rms, rmt = torch.randn(C, generator=g), torch.randn(C, generator=g)
        rvs, rvt = 0.5 + torch.rand(C, generator=g), 0.5 + torch.rand(C, generator=g)
        er, eh = torch.empty(2, C), torch.empty(2, C, device=dev)
        SPEC.tn_eval_coeffs(gamma, beta, rms, rvs, rmt, rvt, 1e-5, er[0], er[1])
        K.tn_eval_coeffs(gamma.to(dev), beta.to(dev), rms.to(dev), rvs.to(dev), rmt.to(dev), rvt.to(dev), 1e-5, eh[0], eh[1])
        errs.append(rel(eh, er))
        return max(errs), 2e-6
    return run


def case_bn(P, C, q1=False, mask=False, training=True, seed=5, frozen=False):
    def run(dev):
        g = gen(seed)
        K = hip()
        errs = []
        x = padded(P, C, g, scale=2.0)
        st_r = torch.zeros(16, 2, C, dtype=torch.float64)
        SPEC.colstats(x, st_r)
        st_h = torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
        K.colstats(to_dev(x, dev), st_h)
        errs.append(rel(st_h.sum(0), st_r.sum(0)))
        gamma, beta = 0.5 + torch.rand(C, generator=g), torch.randn(C, generator=g)
        rm, rv = torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
        cr = torch.empty(4, C)
        ch = torch.empty(4, C, device=dev)
        rmh, rvh = rm.to(dev), rv.to(dev)
        cnt = float(P * (1.25 if q1 else 1.0))
        if training:
            SPEC.bn_finalize(st_r, cnt, gamma, beta, rm, rv, 0.1, 1e-5, cr[0], cr[1], cr[2], cr[3])
            K.bn_finalize(st_h, cnt, gamma.to(dev), beta.to(dev), rmh, rvh, 0.1, 1e-5, ch[0], ch[1], ch[2], ch[3])
            errs += [rel(ch, cr), rel(rmh, rm), rel(rvh, rv)]
        else:
            SPEC.bn_eval_coeffs(gamma, beta, rm, rv, 1e-5, cr[0], cr[1])
            K.bn_eval_coeffs(gamma.to(dev), beta.to(dev), rmh, rvh, 1e-5, ch[0], ch[1])
            cr[2:], ch[2:] = 0.0, 0.0
            errs.append(rel(ch[:2], cr[:2]))
        # keep pre-activations away from the ReLU/ReLU6 kinks: there the gate legitimately depends
        # on fma-vs-(mul,add) rounding, which is not what this case tests
        a0 = x * cr[0] + cr[1]
        x[(a0.abs() < 1e-4) | ((a0 - 6).abs() < 1e-4)] += 0.01
        if C % 4 == 0:
            m = None
            if mask:
                m = (torch.rand(P, C, generator=g) > 0.5).to(torch.uint8)
            src = Act(x, 1, 1, P, cr[0].clone(), cr[1].clone(), ACT_RELU, m, 2.0 if mask else 1.0)
            res = padded(P, C, g)
            o_r = padded(P, C, g)
            SPEC.bn_apply(src, o_r, res)
            o_h = to_dev(padded(P, C, g), dev)
            K.bn_apply(act_to(src, dev), o_h, to_dev(res, dev))
            errs.append(rel(o_h, o_r))
        if training or frozen:
            q1t = None
            if frozen:       # eval-mode BN inside a training pass (freeze_bn): xhat against the running statistics, count = inf,
                cr[2], cr[3] = rm, torch.rsqrt(rv + 1e-5)      # and the quirk-Q1 total handed in
                cnt = float("inf")
                q1t = torch.randn(C, generator=g) if q1 else None
            mb = None
            if mask:
                mbuf = torch.zeros(P, round4(C), dtype=torch.uint8)
                mbuf[:, :C] = (torch.rand(P, C, generator=g) > 0.5).to(torch.uint8)
                mb = mbuf[:, :C]
            y = Act(x, 1, 1, P, cr[0].clone(), cr[1].clone(), ACT_RELU6 if q1 else ACT_RELU, mb, 2.0 if mask else 1.0,
                    BNRec("t", cr[2].clone(), cr[3].clone(), cnt, q1))
            dU = padded(P, C, g)
            s_r = torch.zeros(16, 3, C, dtype=torch.float64)
            SPEC.bnbwd_reduce(dU, y, s_r)
            yh = act_to(y, dev)
            dUh = to_dev(dU, dev)
            s_h = torch.zeros(16, 3, C, dtype=torch.float64, device=dev)
            K.bnbwd_reduce(dUh, yh, s_h)
            errs.append(rel(s_h.sum(0), s_r.sum(0)))
            gr, gh = torch.empty(4, C), torch.empty(4, C, device=dev)
            SPEC.bnbwd_finalize(s_r, y, gr[0], gr[1], gr[2], gr[3], q1_total=q1t)
            K.bnbwd_finalize(s_h, yh, gh[0], gh[1], gh[2], gh[3], q1_total=None if q1t is None else q1t.to(dev))
            errs.append(rel(gh[2:], gr[2:]))
            errs.append(float((gh[:2].cpu() - gr[:2]).abs().max()) if frozen else rel(gh[:2], gr[:2]))     # frozen: c1 = c2 = 0 exactly
            ad = padded(P, C, g)
            o_r = padded(P, C, g)
            SPEC.bnbwd_apply(dU, y, gr[0], gr[1], o_r, ad)
            adh = to_dev(ad, dev)
            K.bnbwd_apply(dUh, yh, gh[0], gh[1], adh, adh)      # in place over the addend
            errs.append(rel(adh, o_r))
        DETAIL.clear()
        DETAIL.update({"errs": ["%.1e" % e for e in errs]})
        return max(errs), 3e-5
    return run


def case_bnbwd_lowrank(P, C, k, mask, seed=9):
    """uda_bnbwd_reduce_lowrank / uda_bnbwd_apply_lowrank: the BN-backward passes on an upstream gradient given as the outer product
    d [P, k] @ w [k, C] (the input gradient of the decoder's 1x1 heads, k = 2 / 1) against (a) the torch statement and (b) the
    ordinary passes fed the materialised matrix."""
    def run(dev):
        g = gen(seed)
        K = hip()
        x = padded(P, C, g)
        sc, sh = 0.5 + torch.rand(C, generator=g), 0.3 * torch.randn(C, generator=g)
        a0 = x * sc + sh
        x[a0.abs() < 1e-4] += 0.01
        mean, istd = 0.1 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
        mb = None
        if mask:
            mbuf = torch.zeros(P, round4(C), dtype=torch.uint8)
            mbuf[:, :C] = (torch.rand(P, C, generator=g) > 0.1).to(torch.uint8)
            mb = mbuf[:, :C]
        y = Act(x, 1, 1, P, sc, sh, ACT_RELU, mb, 1.0 / 0.9 if mask else 1.0, BNRec("t", mean, istd, float(P), False))
        wide = padded(P, 8, g)                                  # d is a column window of a wider matrix, as in the engine
        d = wide[:, 4:4 + k]
        w = torch.randn(k, C, generator=g) / C ** 0.5
        s_r = torch.zeros(16, 3, C, dtype=torch.float64)
        SPEC.bnbwd_reduce(None, y, s_r, lowrank=(d, w))
        yh, dh, wh = act_to(y, dev), to_dev(wide, dev)[:, 4:4 + k], w.to(dev)
        s_h = torch.zeros(16, 3, C, dtype=torch.float64, device=dev)
        K.bnbwd_reduce(None, yh, s_h, lowrank=(dh, wh))
        s_m = torch.zeros(16, 3, C, dtype=torch.float64, device=dev)
        dU = to_dev(padded(P, C, g), dev)
        dU.copy_(dh @ wh)
        K.bnbwd_reduce(dU, yh, s_m)
        errs = [rel(s_h.sum(0), s_r.sum(0)), rel(s_h.sum(0), s_m.sum(0))]
        gr = torch.empty(4, C)
        SPEC.bnbwd_finalize(s_r, y, gr[0], gr[1], gr[2], gr[3])
        gh = gr.to(dev)
        ad = padded(P, C, g)
        o_r = padded(P, C, g)
        SPEC.bnbwd_apply(None, y, gr[0], gr[1], o_r, ad, lowrank=(d, w))
        adh = to_dev(ad, dev)
        o_m = to_dev(padded(P, C, g), dev)
        K.bnbwd_apply(dU, yh, gh[0], gh[1], o_m, adh)
        K.bnbwd_apply(None, yh, gh[0], gh[1], adh, adh, lowrank=(dh, wh))      # in place over the addend, as the engine does
        errs += [rel(adh, o_r), rel(adh, o_m)]
        return max(errs), 3e-5
    return run


def case_mc_seg_head(N, h, w, H, W, Cf, Cl, reps, mask, seed=10):
    """uda_mc_seg_head (the 305 -> 2 head of a no-grad stochastic pass on the virtual x_feature = cat(up(feature), low rows shared by
    the repeated batch, boundary)) and the store-less statistics pass uda_upsample_fwd_stats(out = NULL), against the torch statement
    that materialises the matrix."""
    def run(dev):
        g = gen(seed)
        K = hip()
        P, Cc = N * H * W, Cf + Cl + 1
        feat = padded(N * h * w, Cf, g)
        low = padded(P // reps, Cl, g)
        bnd = padded(P, 1, g)
        sc, sh = 0.5 + torch.rand(Cc, generator=g), 0.3 * torch.randn(Cc, generator=g)
        mk = None
        if mask:
            mbuf = torch.randint(0, 256, (P, round4(Cc)), generator=g, dtype=torch.uint8)     # padding bytes arbitrary
            mbuf[:, :Cc] = (torch.rand(P, Cc, generator=g) > 0.1).to(torch.uint8)
            mk = mbuf[:, :Cc]
        w4 = torch.randn(2, Cc, 1, 1, generator=g) / Cc ** 0.5
        bias = torch.randn(2, generator=g)
        o_r, o_h = padded(P, 2, g), to_dev(padded(P, 2, g), dev)
        SPEC.mc_seg_head(feat, N, h, w, low, bnd, H, W, sc, sh, ACT_RELU, mk, 1.0 / 0.9, SPEC.relayout_ohwi(w4), bias, o_r)
        K.mc_seg_head(to_dev(feat, dev), N, h, w, to_dev(low, dev), to_dev(bnd, dev), H, W, sc.to(dev), sh.to(dev), ACT_RELU,
                      None if mk is None else to_dev(mbuf, dev)[:, :Cc], 1.0 / 0.9, K.relayout_ohwi(w4.to(dev)), bias.to(dev), o_h)
        st_r = torch.zeros(16, 2, Cc, dtype=torch.float64)
        st_h = torch.zeros(16, 2, Cc, dtype=torch.float64, device=dev)
        SPEC.upsample_stats(feat, N, h, w, H, W, st_r)
        K.upsample_stats(to_dev(feat, dev), N, h, w, H, W, st_h)
        return max(rel(o_h, o_r), rel(st_h.sum(0), st_r.sum(0))), 2e-5
    return run


def case_upsample_stats(N, h, w, H, W, C, Cs, seed=8):
    """uda_upsample_fwd_stats: the upsampled tensor AND its per-channel (sum, sum of squares) in channels [0, C) of a wider
    accumulator; uda_colstats_window: the remaining channels of the wide buffer (a strided column window) into the same accumulator
    (round 3: the BatchNorm(305) statistics of decoder.py:23 without a pass over the whole 305-channel buffer)."""
    def run(dev):
        g = gen(seed)
        K = hip()
        x = padded(N * h * w, C, g)
        wide_r, wide_h = padded(N * H * W, Cs, g), to_dev(padded(N * H * W, Cs, g), dev)
        wide_r[:, C:] = torch.randn(N * H * W, Cs - C, generator=g)
        wide_h[:, C:] = wide_r[:, C:].to(dev)
        st_r = torch.zeros(16, 2, Cs, dtype=torch.float64)
        st_h = torch.zeros(16, 2, Cs, dtype=torch.float64, device=dev)
        SPEC.upsample_fwd(x, N, h, w, wide_r[:, :C], H, W, stats=st_r)
        K.upsample_fwd(to_dev(x, dev), N, h, w, wide_h[:, :C], H, W, stats=st_h)
        SPEC.colstats_window(wide_r[:, C:], st_r, C)
        K.colstats_window(wide_h[:, C:], st_h, C)
        full = torch.zeros(16, 2, Cs, dtype=torch.float64)
        SPEC.colstats(wide_r, full)                       # the one-pass statement over the whole buffer
        return max(rel(wide_h, wide_r), rel(st_h.sum(0), st_r.sum(0)), rel(st_h.sum(0), full.sum(0))), 2e-5
    return run


def case_resample(N, h, w, H, W, C, seed=6):
    def run(dev):
        g = gen(seed)
        K = hip()
        x = padded(N * h * w, C, g)
        o_r, o_h = padded(N * H * W, C, g), to_dev(padded(N * H * W, C, g), dev)
        SPEC.upsample_fwd(x, N, h, w, o_r, H, W)
        K.upsample_fwd(to_dev(x, dev), N, h, w, o_h, H, W)
        ref_t = torch.nn.functional.interpolate(x.reshape(N, h, w, C).permute(0, 3, 1, 2), size=(H, W), mode="bilinear",
                                                align_corners=True).permute(0, 2, 3, 1).reshape(N * H * W, C)
        errs = [rel(o_h, o_r), rel(o_h, ref_t)]
        d = padded(N * H * W, C, g)
        dx_r, dx_h = padded(N * h * w, C, g), to_dev(padded(N * h * w, C, g), dev)
        SPEC.upsample_bwd(d, N, H, W, dx_r, h, w)
        K.upsample_bwd(to_dev(d, dev), N, H, W, dx_h, h, w)
        errs.append(rel(dx_h, dx_r))
        return max(errs), 2e-5
    return run


def case_head(N, h, w, H, W, C, seed=7):
    def run(dev):
        g = gen(seed)
        K = hip()
        x = padded(N * h * w, C, g)
        o_r, o_h = torch.empty(N, C, H, W), torch.empty(N, C, H, W, device=dev)
        SPEC.head_upsample_fwd(x, N, h, w, o_r)
        K.head_upsample_fwd(to_dev(x, dev), N, h, w, o_h)
        ref_t = torch.nn.functional.interpolate(x.reshape(N, h, w, C).permute(0, 3, 1, 2), size=(H, W), mode="bilinear",
                                                align_corners=True)
        errs = [rel(o_h, o_r), rel(o_h, ref_t)]
        d = torch.randn(N, C, H, W, generator=g)
        base = padded(N * h * w, C, g)
        dx_r = base.clone()
        SPEC.head_upsample_bwd(d, dx_r, N, h, w, True)
        dx_h = to_dev(base, dev)
        K.head_upsample_bwd(d.to(dev), dx_h, N, h, w, True)
        errs.append(rel(dx_h, dx_r))
        dx_h2 = to_dev(base, dev)
        K.head_upsample_bwd(d.to(dev), dx_h2, N, h, w, False)
        errs.append(rel(dx_h2, dx_r - base))
        return max(errs), 2e-5
    return run


def case_gap(N, HW, C, seed=8):
    def run(dev):
        g = gen(seed)
        K = hip()
        x = padded(N * HW, C, g)
        o_r, o_h = torch.empty(N, C), torch.empty(N, C, device=dev)
        SPEC.gap_fwd(x, N, o_r, 1.0 / HW)
        K.gap_fwd(to_dev(x, dev), N, o_h, 1.0 / HW)
        errs = [rel(o_h, o_r)]
        ad = padded(N * HW, C, g)
        b_r, b_h = padded(N * HW, C, g), to_dev(padded(N * HW, C, g), dev)
        SPEC.broadcast_rows(o_r, N, b_r, 0.25, ad)
        K.broadcast_rows(o_r.to(dev), N, b_h, 0.25, to_dev(ad, dev))
        errs.append(rel(b_h, b_r))
        return max(errs), 2e-5
    return run


def case_dropout(P, C, p, seed=9):
    def run(dev):
        K = hip()
        m = torch.zeros(P, round4(C), dtype=torch.uint8, device=dev)[:, :C]
        K.dropout_mask(m, p, 1234, 7)
        m2 = torch.zeros(P, round4(C), dtype=torch.uint8, device=dev)[:, :C]
        K.dropout_mask(m2, p, 1234, 7)
        m3 = torch.zeros(P, round4(C), dtype=torch.uint8, device=dev)[:, :C]
        K.dropout_mask(m3, p, 1234, 8)
        assert torch.equal(m, m2), "same (seed, offset) must reproduce the mask"
        assert not torch.equal(m, m3), "different offsets must give different masks"
        assert int(m.max()) <= 1
        keep = m.float().mean().item()
        col = m.float().mean(0)
        # binomial tolerance (5 sigma) on the overall rate, loose bound per column
        sig = ((1 - p) * p / (P * C)) ** 0.5
        err = abs(keep - (1 - p)) / (5 * sig)
        assert (col - (1 - p)).abs().max().item() < 8 * ((1 - p) * p / P) ** 0.5
        return err, 1.0
    return run


CASES = [
    # 1x1 convs of the backbone (narrow N configs, small K, Q1-style sizes)
    ("conv1x1 16->96 relu6", case_conv(2, 24, 20, 16, 96, 1, 1)),
    ("conv1x1 96->24 none-lazy", case_conv(2, 17, 13, 96, 24, 1, 1, lazy=False)),
    ("conv1x1 144->32", case_conv(1, 16, 16, 144, 32, 1, 1)),
    ("conv1x1 32->192 (128-wide tiles)", case_conv(2, 12, 12, 32, 192, 1, 1)),
    ("conv1x1 960->320", case_conv(2, 8, 8, 960, 320, 1, 1, stats=True)),
    ("conv1x1 1280->256 relu", case_conv(2, 8, 8, 1280, 256, 1, 1)),
    ("conv1x1 305->2 bias mask (C%4!=0)", case_conv(2, 16, 16, 305, 2, 1, 1, mask=True, bias=True, stats=False)),
    ("conv1x1 256->1 bias mask", case_conv(2, 16, 16, 256, 1, 1, 1, mask=True, bias=True, stats=False)),
    ("conv1x1 100->2 addend, ragged P (heads kernel, raw operand)", case_conv(2, 17, 13, 100, 2, 1, 1, lazy=False, addend=True, stats=False)),
    ("conv1x1 68->1 lazy, P < 128 (heads kernel)", case_conv(1, 9, 7, 68, 1, 1, 1, bias=True, stats=False)),
    ("conv1x1 320->256 P=N (gap branch)", case_conv(4, 1, 1, 320, 256, 1, 1, lazy=False)),
    ("conv1x1 24->48 addend", case_conv(2, 16, 16, 24, 48, 1, 1, addend=True, stats=False)),
    # 3x3 convs
    ("conv3x3 304->256 p1 mask", case_conv(2, 16, 16, 304, 256, 3, 1, lazy=False)),
    ("conv3x3 256->256 relu mask", case_conv(2, 16, 12, 256, 256, 3, 1, mask=True)),
    ("conv3x3 320->256 dil6", case_conv(2, 8, 8, 320, 256, 3, 6, lazy=False)),
    ("conv3x3 320->256 dil12 (mostly padding)", case_conv(2, 8, 8, 320, 256, 3, 12, lazy=False)),
    ("conv3x3 64->40 dil2 odd sizes", case_conv(1, 11, 9, 64, 40, 3, 2)),
    ("conv3x3 256->48 (long K, narrow tile)", case_conv(2, 16, 16, 256, 48, 3, 1, lazy=False)),
    ("conv3x3 128->64 relu mask stats (long K, narrow tile)", case_conv(2, 12, 12, 128, 64, 3, 1, mask=True)),
    ("dgrad3x3 48<-256 addend (the decoder's low-level input gradient)", case_dgrad(2, 16, 16, 48, 256, 3, 1, accumulate=True)),
    # dgrad through the same kernel
    ("dgrad1x1 96<-16", case_dgrad(2, 12, 12, 96, 16, 1, 1)),
    ("dgrad1x1 305<-2", case_dgrad(2, 16, 16, 305, 2, 1, 1)),
    ("dgrad1x1 256<-1", case_dgrad(2, 16, 16, 256, 1, 1, 1)),
    ("dgrad3x3 304<-256 accumulate", case_dgrad(2, 12, 12, 304, 256, 3, 1, accumulate=True)),
    ("dgrad3x3 320<-256 dil6 accumulate", case_dgrad(2, 8, 8, 320, 256, 3, 6, accumulate=True)),
    # wgrad
    ("wgrad1x1 16->96", case_wgrad(2, 24, 20, 16, 96, 1, 1)),
    ("wgrad1x1 96->24", case_wgrad(2, 17, 13, 96, 24, 1, 1)),
    ("wgrad1x1 32->16 (64x64 tiles)", case_wgrad(2, 16, 16, 32, 16, 1, 1)),
    ("wgrad1x1 960->320", case_wgrad(2, 8, 8, 960, 320, 1, 1)),
    ("wgrad1x1 305->2 mask", case_wgrad(2, 16, 16, 305, 2, 1, 1, mask=True)),
    ("wgrad1x1 256->1 mask", case_wgrad(2, 16, 16, 256, 1, 1, 1, mask=True)),
    ("wgrad1x1 320->256 P=4", case_wgrad(4, 1, 1, 320, 256, 1, 1, lazy=False)),
    ("wgrad3x3 304->256", case_wgrad(2, 16, 16, 304, 256, 3, 1, lazy=False)),
    ("wgrad3x3 256->256 mask", case_wgrad(2, 16, 12, 256, 256, 3, 1, mask=True)),
    ("wgrad3x3 320->256 dil6", case_wgrad(2, 8, 8, 320, 256, 3, 6, lazy=False)),
    ("wgrad3x3 64->40 dil2", case_wgrad(1, 11, 9, 64, 40, 3, 2)),
    # depthwise
    ("dw 32 s1 d1 border0", case_dw(2, 16, 16, 32, 1, 1, 0)),
    ("dw 96 s2 d1 border1", case_dw(2, 16, 16, 96, 2, 1, 1)),
    ("dw 144 s1 d1 border1 odd", case_dw(1, 13, 11, 144, 1, 1, 1)),
    ("dw 960 s1 d2 border1", case_dw(2, 8, 8, 960, 1, 2, 1)),
    ("dw 576 s1 d1 border1", case_dw(2, 8, 8, 576, 1, 1, 1)),
    ("stem 2x3x32x32", case_stem(2, 32, 32)),
    ("stem 1x3x48x80", case_stem(1, 48, 80)),
    ("stem 2x3x20x512 (row-staged kernels: Wo = 256)", case_stem(2, 20, 512)),
    ("stem 1x3x7x1024 (row-staged kernels: two segments per row)", case_stem(1, 7, 1024)),
    # batch norm pieces
    ("bn C=32 P=3000", case_bn(3000, 32)),
    ("bn C=96 q1", case_bn(1500, 96, q1=True)),
    ("bn C=256 mask", case_bn(700, 256, mask=True)),
    ("bn C=305 mask (C%4!=0)", case_bn(600, 305, mask=True)),
    ("bn frozen (eval-mode backward) C=96 q1 total", case_bn(1500, 96, q1=True, training=False, frozen=True)),
    ("bn frozen (eval-mode backward) C=256 mask", case_bn(700, 256, mask=True, training=False, frozen=True)),
    ("transnorm gain / eval coefficients C=305", case_transnorm(305)),
    ("transnorm gain / eval coefficients C=1280 (> one pass of the workgroup)", case_transnorm(1280, 40, 24)),
    ("bn C=1024 (concat)", case_bn(300, 1024)),
    ("bn eval C=144", case_bn(500, 144, training=False)),
    ("bn C=256 P=4", case_bn(4, 256)),
    # resampling
    ("upsample 4x4->16x16 C=256", case_resample(2, 4, 4, 16, 16, 256)),
    ("upsample 8x6->32x24 C=64", case_resample(1, 8, 6, 32, 24, 64)),
    ("upsample + stats 8x8->32x32 C=256 into 305 (+ 49-channel window)", case_upsample_stats(2, 8, 8, 32, 32, 256, 305)),
    ("mc seg head 4 x (8x8 -> 32x32) 256 + 48 + 1, batch repeated twice, mask", case_mc_seg_head(4, 8, 8, 32, 32, 256, 48, 2, True)),
    ("mc seg head 3 x (5x7 -> 20x28) 64 + 8 + 1, no repeat, no mask", case_mc_seg_head(3, 5, 7, 20, 28, 64, 8, 1, False)),
    ("mc seg head 2 x (5x9 -> 17x33) exact ratio 1/4 (last source row / column reached exactly)", case_mc_seg_head(2, 5, 9, 17, 33, 64, 8, 1, False)),
    ("mc seg head 2 x (6x6 -> 6x6) ratio 1, (3x4 -> 13x7) mixed", case_mc_seg_head(2, 6, 6, 6, 6, 64, 8, 1, True)),
    ("mc seg head 2 x (3x4 -> 13x7) 64 + 8 + 1", case_mc_seg_head(2, 3, 4, 13, 7, 64, 8, 1, False)),
    ("bn backward, low-rank dU: C=305 k=2 mask (seg head)", case_bnbwd_lowrank(3000, 305, 2, True)),
    ("bn backward, low-rank dU: C=256 k=1 mask (boundary head)", case_bnbwd_lowrank(2500, 256, 1, True)),
    ("bn backward, low-rank dU: C=40 k=2 raw", case_bnbwd_lowrank(777, 40, 2, False)),
    ("upsample + stats 5x7->20x28 C=64 into 72 (+ 8-channel window)", case_upsample_stats(3, 5, 7, 20, 28, 64, 72)),
    ("head 16x16->64x64 C=2", case_head(2, 16, 16, 64, 64, 2)),
    ("head 12x10->48x40 C=1", case_head(1, 12, 10, 48, 40, 1)),
    ("gap C=320", case_gap(3, 64, 320)),
    ("gap C=256 HW=16", case_gap(4, 16, 256)),
    ("dropout p=.5", case_dropout(4096, 256, 0.5)),
    ("dropout p=.1 C=305", case_dropout(4096, 305, 0.1)),
]


# ------------------------------------------------------------------------------------- losses / prototypes
def case_seg_loss(B, S, seed=10):
    def run(dev):
        g = gen(seed)
        K = hip()
        o = 3 * torch.randn(B, 2, S, S, generator=g)
        o[0, 0, 0, :8] = torch.tensor([40., -40., 110., -110., 17., -17., 90., -90.])    # saturating logits (log clamp)
        b = 2 * torch.randn(B, 1, S, S, generator=g)
        tm = (torch.rand(B, 2, S, S, generator=g) > 0.5).float()
        tb = torch.rand(B, 1, S, S, generator=g)
        l_r = SPEC.seg_loss_fwd(o, tm, b, tb)
        l_h = K.seg_loss_fwd(o.to(dev), tm.to(dev), b.to(dev), tb.to(dev))
        gs = torch.tensor([0.37])
        do_r, db_r = SPEC.seg_loss_bwd(o, tm, b, tb, gs)
        do_h, db_h = K.seg_loss_bwd(o.to(dev), tm.to(dev), b.to(dev), tb.to(dev), gs.to(dev))
        return max(rel(l_h, l_r), rel(do_h, do_r), rel(db_h, db_r)), 2e-5
    return run


def case_seg_counts(B, S, seed=11):
    def run(dev):
        g = gen(seed)
        lg = 2 * torch.randn(B, 2, S, S, generator=g) + 1.0
        tg = (torch.rand(B, 2, S, S, generator=g) > 0.6).float()
        c_r = SPEC.seg_counts(lg, tg, 0.75)
        c_h = hip().seg_counts(lg.to(dev), tg.to(dev), 0.75)
        # a logit within one ulp of the threshold may flip: allow a handful of pixels
        return float((c_h.cpu() - c_r).abs().max()) / 4.0, 1.0
    return run


def case_proto(B, h, C, mode, seed=12):
    def run(dev):
        g = gen(seed)
        K = hip()
        P = B * h * h
        feat = padded(P, C, g)
        H = 4 * h
        errs = []
        if mode == 0:
            mp = (torch.rand(B, 2, H, H, generator=g) > 0.6).float()
            w_r, _, _ = SPEC.proto_weights(0, B, h, h, map_=mp)
            w_h, _, _ = K.proto_weights(0, B, h, h, map_=mp.to(dev))
        elif mode == 1:
            lg = padded(P, 2, g, scale=2.0)
            w_r, _, _ = SPEC.proto_weights(1, B, h, h, logits=lg)
            w_h, _, _ = K.proto_weights(1, B, h, h, logits=to_dev(lg, dev))
        else:
            T = 8
            base = torch.nn.functional.avg_pool2d(2.0 * torch.randn(B, 2, H, H, generator=g), 9, 1, 4) * 6.0
            preds = base.repeat(T, 1, 1, 1) + 0.35 * torch.randn(T * B, 2, H, H, generator=g) * \
                (torch.rand(1, 2, H, H, generator=g) > 0.5).float()
            sd_r, mn_r = SPEC.mc_stats(preds, T)
            sd_h, mn_h = K.mc_stats(preds.to(dev), T)
            errs += [rel(sd_h, sd_r), rel(mn_h, mn_r)]
            lgn = torch.nn.functional.interpolate(base, size=(h, h), mode="bilinear", align_corners=True)
            lg = padded(P, 2, g)
            lg.copy_(lgn.permute(0, 2, 3, 1).reshape(P, 2))
            w_r, m0_r, m1_r = SPEC.proto_weights(2, B, h, h, logits=lg, std_map=sd_r, mean_map=mn_r)
            # feed the HIP weights kernel the reference std/mean so that threshold flips cannot differ
            w_h, m0_h, m1_h = K.proto_weights(2, B, h, h, logits=to_dev(lg, dev), std_map=sd_r.to(dev), mean_map=mn_r.to(dev))
            flips = int((m0_h.cpu() != m0_r).sum() + (m1_h.cpu() != m1_r).sum())
            assert flips <= 4, "reliability mask differs on %d pixels" % flips
            if flips:
                w_h = w_r.to(dev)
            assert 0.05 < float(m0_r.mean()) / 2 < 0.999, "degenerate mask in the test input"
        errs.append(rel(w_h, w_r))
        s_r = torch.zeros(4, C + 1, dtype=torch.float64)
        SPEC.proto_reduce(feat, w_r, s_r)
        s_h = torch.zeros(4, C + 1, dtype=torch.float64, device=dev)
        fh, wh = to_dev(feat, dev), w_r.to(dev)
        K.proto_reduce(fh, wh, s_h)
        errs += [rel(s_h, s_r), rel(K.proto_finalize(s_h), SPEC.proto_finalize(s_r))]
        dC = torch.randn(4, C, generator=g)
        base = padded(P, C, g)
        df_r = base.clone()
        dw_r = SPEC.proto_bwd(feat, w_r, s_r, dC, df_r, True, True)
        df_h = to_dev(base, dev)
        dw_h = K.proto_bwd(fh, wh, s_h, dC.to(dev), df_h, True, True)
        errs += [rel(df_h, df_r), rel(dw_h, dw_r)]
        DETAIL.clear()
        DETAIL.update({"errs": ["%.1e" % e for e in errs]})
        return max(errs), 3e-5
    return run


def case_adam(n, seed=13):
    def run(dev):
        g = gen(seed)
        p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
        m, v = 0.1 * torch.randn(n, generator=g), torch.rand(n, generator=g) * 0.01
        ph, mh, vh = p.to(dev), m.to(dev), v.to(dev)
        for step in (1, 2, 7):
            SPEC.adam_step(p, gr, m, v, 1e-3, 0.9, 0.99, 1e-8, step)
            hip().adam_step(ph, gr.to(dev), mh, vh, 1e-3, 0.9, 0.99, 1e-8, step)
        ref = torch.nn.Parameter(torch.randn(n, generator=gen(seed)))
        return max(rel(ph, p), rel(mh, m), rel(vh, v)), 1e-5
    return run


def case_feat4(P, C, seed=14):
    def run(dev):
        g = gen(seed)
        K = hip()
        feat = padded(P, C, g)
        coef = torch.randn(4, C + 1, generator=g)
        o_r = SPEC.feat_dot4(feat, coef)
        o_h = K.feat_dot4(to_dev(feat, dev), coef.to(dev))
        w = torch.randn(P, 4, generator=g)
        base = padded(P, C, g)
        d_r = base.clone()
        SPEC.feat_rank4(w, coef, d_r, True)
        d_h = to_dev(base, dev)
        K.feat_rank4(w.to(dev), coef.to(dev), d_h, True)
        return max(rel(o_h, o_r), rel(d_h, d_r)), 2e-5
    return run


CASES += [
    ("feat dot4/rank4 C=305", case_feat4(3000, 305)),
    # wide warp-specialised tiles (BN = 256 needs >= 512 workgroups: P >= 65536)
    ("conv3x3 16->256 P=65536 (BN=256 tiles) relu mask", case_conv(4, 128, 128, 16, 256, 3, 1, mask=True)),
    ("conv1x1 64->200 P=65536 (BN=256 tiles) bias addend", case_conv(4, 128, 128, 64, 200, 1, 1, bias=True, addend=True)),
    ("dgrad3x3 304<-32 P=65536 accumulate", case_dgrad(4, 128, 128, 304, 32, 3, 1, accumulate=True)),
    ("wgrad3x3 16->256 P=65536 mask", case_wgrad(4, 128, 128, 16, 256, 3, 1, mask=True)),
    ("seg loss 2x64", case_seg_loss(2, 64)),
    ("seg counts 3x96", case_seg_counts(3, 96)),
    ("proto hard C=305 h=32", case_proto(2, 32, 305, 0)),
    ("proto soft C=305 h=16", case_proto(2, 16, 305, 1)),
    ("proto retrify C=305 h=32", case_proto(1, 32, 305, 2)),
    ("proto hard C=64 h=8", case_proto(3, 8, 64, 0)),
    ("adam 100003", case_adam(100003)),
]


def case_discriminative(B, h, C, seed=15):
    """ops.discriminative_loss (HIP) against oracle/losses_ref.py (parity unpinned: our reading of Appendix B)."""
    def run(dev):
        from oracle import losses_ref
        from uda_clr_amd import ops
        g = gen(seed)
        feat = torch.randn(B, C, h, h, generator=g)
        cents = tuple(torch.randn(1, C, 1, 1, generator=g) for _ in range(4))
        lab = (torch.rand(B, 2, h, h, generator=g) > 0.5).float()
        f_r = feat.clone().requires_grad_(True)
        l_r = losses_ref.discriminative_loss(f_r, cents, lab)
        l_r.backward()
        f_h = feat.to(dev).requires_grad_(True)
        l_h = ops.discriminative_loss(f_h, tuple(c.to(dev) for c in cents), lab.to(dev))
        l_h.backward()
        return max(rel(l_h, l_r), rel(f_h.grad, f_r.grad)), 5e-5
    return run


CASES += [("discriminative loss (unpinned) C=305", case_discriminative(2, 16, 305))]


# ------------------------------------------------------------------ ResNet-101 pieces
def case_stem7(N, H, W, seed=16):
    def run(dev):
        g = gen(seed)
        x = torch.randn(N, 3, H, W, generator=g)
        w = torch.randn(64, 3, 7, 7, generator=g) / 12.0
        Po = N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        y_r, st_r = padded(Po, 64, g), torch.zeros(16, 2, 64, dtype=torch.float64)
        SPEC.stem7_fwd(x, w, y_r, st_r)
        K = hip()
        y_h, st_h = to_dev(padded(Po, 64, g), dev), torch.zeros(16, 2, 64, dtype=torch.float64, device=dev)
        K.stem7_fwd(x.to(dev), w.to(dev), y_h, st_h)
        dy = padded(Po, 64, g)
        dw_r, dw_h = torch.empty(64, 3, 7, 7), torch.empty(64, 3, 7, 7, device=dev)
        SPEC.stem7_wgrad(x, dy, dw_r)
        K.stem7_wgrad(x.to(dev), to_dev(dy, dev), dw_h)
        return max(rel(y_h, y_r), rel(st_h.sum(0), st_r.sum(0)), rel(dw_h, dw_r)), 3e-5
    return run


def case_maxpool(N, H, W, C, seed=17):
    def run(dev):
        g = gen(seed)
        K = hip()
        src = make_src(N, H, W, C, g, lazy=True, act=ACT_RELU)
        Po = N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        z_r, i_r = padded(Po, C, g), torch.zeros(Po, round4(C), dtype=torch.uint8)[:, :C]
        SPEC.maxpool_fwd(src, z_r, i_r)
        z_h, i_h = to_dev(padded(Po, C, g), dev), torch.zeros(Po, round4(C), dtype=torch.uint8, device=dev)[:, :C]
        K.maxpool_fwd(act_to(src, dev), z_h, i_h)
        dz = padded(Po, C, g)
        du_r, du_h = padded(N * H * W, C, g), to_dev(padded(N * H * W, C, g), dev)
        SPEC.maxpool_bwd(dz, i_r, N, H, W, du_r)
        K.maxpool_bwd(to_dev(dz, dev), i_h, N, H, W, du_h)
        return max(rel(z_h, z_r), float((i_h.cpu() != i_r).sum()), rel(du_h, du_r)), 1e-6
    return run


def case_rows_stride(N, H, W, C, s, seed=18):
    def run(dev):
        g = gen(seed)
        K = hip()
        Po = N * ((H - 1) // s + 1) * ((W - 1) // s + 1)
        big, small = padded(N * H * W, C, g), padded(Po, C, g)
        o_r, o_h = padded(Po, C, g), to_dev(padded(Po, C, g), dev)
        SPEC.rows_stride(big, N, H, W, s, o_r)
        K.rows_stride(to_dev(big, dev), N, H, W, s, o_h)
        f_r, f_h = padded(N * H * W, C, g), to_dev(padded(N * H * W, C, g), dev)
        SPEC.rows_stride(small, N, H, W, s, f_r, scatter=True)
        K.rows_stride(to_dev(small, dev), N, H, W, s, f_h, scatter=True)
        return max(rel(o_h, o_r), rel(f_h, f_r)), 0.0
    return run


def case_bottleneck_tail(N, H, W, C, seed=19):
    def run(dev):
        g = gen(seed)
        K = hip()
        a = make_src(N, H, W, C, g, lazy=True, act=ACT_NONE)
        errs = []
        for lazy_b in (False, True):
            b = make_src(N, H, W, C, g, lazy=lazy_b, act=ACT_NONE)
            z_r, z_h = padded(a.P, C, g), to_dev(padded(a.P, C, g), dev)
            SPEC.bn_add_relu(a, b, z_r)
            K.bn_add_relu(act_to(a, dev), act_to(b, dev), z_h)
            errs.append(rel(z_h, z_r))
        dz = padded(a.P, C, g)
        g_r, g_h = padded(a.P, C, g), to_dev(padded(a.P, C, g), dev)
        SPEC.relu_gate(dz, z_r, g_r)
        K.relu_gate(to_dev(dz, dev), to_dev(z_r, dev), g_h)
        errs.append(rel(g_h, g_r))
        return max(errs), 1e-6
    return run


CASES += [
    ("stem7 2x3x32x32", case_stem7(2, 32, 32)),
    ("stem7 1x3x48x80", case_stem7(1, 48, 80)),
    ("maxpool 2x16x16 C=64", case_maxpool(2, 16, 16, 64)),
    ("maxpool 1x24x40 C=64", case_maxpool(1, 24, 40, 64)),
    ("rows stride 2 C=256", case_rows_stride(2, 16, 16, 256, 2)),
    ("rows stride 2 C=128 8x12", case_rows_stride(1, 8, 12, 128, 2)),
    ("bottleneck tail C=256", case_bottleneck_tail(2, 16, 16, 256)),
    ("bottleneck tail C=2048", case_bottleneck_tail(2, 4, 4, 2048)),
    ("conv1x1 2048->512 (ResNet layer4)", case_conv(2, 8, 8, 2048, 512, 1, 1)),
    ("conv3x3 512->512 dil4 (ResNet layer4)", case_conv(2, 8, 8, 512, 512, 3, 4)),
    ("conv3x3 2048->256 dil6 (ResNet ASPP)", case_conv(2, 8, 8, 2048, 256, 3, 6, lazy=False)),
    ("dgrad1x1 1024<-256", case_dgrad(2, 8, 8, 1024, 256, 1, 1, accumulate=True)),
    ("wgrad1x1 1024->2048", case_wgrad(2, 8, 8, 1024, 2048, 1, 1)),
    ("wgrad3x3 2048->256 dil6", case_wgrad(2, 8, 8, 2048, 256, 3, 6, lazy=False)),
]

# 256-pixel workgroup tiles of the wide-tile kernel (taken when >= 512 of them exist: P >= 131072 at Cout = 256)
CASES += [
    ("conv3x3 32->256 P=131580 (256x256 tiles, ragged) relu mask", case_conv(2, 255, 258, 32, 256, 3, 1, mask=True)),
    ("conv1x1 200->256 P=131072 (256x256 tiles) bias addend", case_conv(2, 256, 256, 200, 256, 1, 1, bias=True, addend=True)),
    ("conv3x3 24->250 P=131072 dil2 (256x256 tiles) raw", case_conv(2, 256, 256, 24, 250, 3, 2, lazy=False)),
]


# 2x2 taps (the space-to-depth form of the discriminators' 4x4 stride-2 convs, GAN.py:90-101)
CASES += [
    ("conv2x2 o0 8->64 (narrow) raw", case_conv(2, 18, 18, 8, 64, 2, 1, lazy=False, origin=0)),
    ("conv2x2 o0 256->128 (wide)", case_conv(2, 19, 17, 256, 128, 2, 1, lazy=False, origin=0)),
    ("conv2x2 o1 128->256 (wide, dgrad form) addend", case_conv(2, 19, 17, 128, 256, 2, 1, lazy=False, addend=True, origin=1)),
    ("conv2x2 o0 2048->1", case_conv(2, 9, 9, 2048, 1, 2, 1, lazy=False, origin=0)),
    ("wgrad2x2 o0 8->64", case_wgrad(2, 18, 18, 8, 64, 2, 1, lazy=False, origin=0)),
    ("wgrad2x2 o0 256->128", case_wgrad(2, 19, 17, 256, 128, 2, 1, lazy=False, origin=0)),
    ("wgrad2x2 o0 2048->1", case_wgrad(2, 9, 9, 2048, 1, 2, 1, lazy=False, origin=0)),
]


def case_s2d(N, H, W, C, nchw, vh=None, vw=None, seed=20):
    def run(dev):
        g = gen(seed)
        K = hip()
        vh_, vw_ = vh or H, vw or W
        Hz, Wz = (vh_ + 5) // 2, (vw_ + 5) // 2
        src = torch.randn(N, C, H, W, generator=g) if nchw else padded(N * H * W, C, g)
        z_r, z_h = padded(N * Hz * Wz, 4 * C, g), to_dev(padded(N * Hz * Wz, 4 * C, g), dev)
        SPEC.s2d_fwd(src, nchw, N, H, W, C, vh_, vw_, 0.2, z_r)
        K.s2d_fwd(src.to(dev) if nchw else to_dev(src, dev), nchw, N, H, W, C, vh_, vw_, 0.2, z_h)
        errs = [rel(z_h, z_r)]
        dz = torch.randn(N * Hz * Wz, 4 * C, generator=g)
        zs = torch.randn(N * Hz * Wz, 4 * C, generator=g)
        for sign in (None, zs):
            if nchw:
                d_r, d_h = torch.empty(N, C, H, W), torch.empty(N, C, H, W, device=dev)
            else:
                d_r, d_h = padded(N * H * W, C, g), to_dev(padded(N * H * W, C, g), dev)
            SPEC.s2d_bwd(dz, sign, 0.2, N, H, W, C, vh_, vw_, d_r, nchw)
            K.s2d_bwd(dz.to(dev), None if sign is None else sign.to(dev), 0.2, N, H, W, C, vh_, vw_, d_h, nchw)
            errs.append(rel(d_h, d_r))
        return max(errs), 0.0
    return run


CASES += [
    ("s2d nchw C=1 32x32", case_s2d(2, 32, 32, 1, True)),
    ("s2d nchw C=2 30x34", case_s2d(2, 30, 34, 2, True)),
    ("s2d rows C=64 grid 18x18 valid 17x17", case_s2d(2, 18, 18, 64, False, 17, 17)),
    ("s2d rows C=6 grid 11x9 valid 9x8 (scalar path)", case_s2d(1, 11, 9, 6, False, 9, 8)),
]


def case_s2d_packed(N, H, W, C, vh=None, vw=None, seed=22):
    """bf16x3 mode: the packed space-to-depth operands written in ONE pass (uda_x3_pack_s2d_fwd / _bwd, round 3) are bit-identical
    to the two-pass forms they replace (uda_s2d_fwd / uda_s2d_bwd + uda_x3_pack: the split is exact), and the gate read from the
    sign of the forward's source rows equals the gate read from the sign of z = lrelu(source)."""
    def run(dev):
        g = gen(seed)
        K = hip()
        vh_, vw_ = vh or H, vw or W
        Hz, Wz = (vh_ + 5) // 2, (vw_ + 5) // 2
        src = to_dev(padded(N * H * W, C, g), dev)
        src[::7] = 0.0                                   # exact zeros: lrelu(0) = 0 is "not positive" for both kinds of gate
        z = to_dev(padded(N * Hz * Wz, 4 * C, g), dev)
        K.s2d_fwd(src, False, N, H, W, C, vh_, vw_, 0.2, z)
        za = Act(z, N, Hz, Wz)
        want = K.x3_pack(K._src(za), za.P, za.C, dev)
        got = K.s2d_pack_fwd(src, N, H, W, C, vh_, vw_, 0.2)
        body = lambda t, rows, ch: t[:rows * ((ch + 15) // 16) * 96]          # (uda_x3_packed_bytes adds 64 bytes that nobody writes)
        bad = int((body(got._x3, za.P, za.C) != body(want, za.P, za.C)).sum())
        dz = to_dev(padded(N * Hz * Wz, 4 * C, g), dev)
        d_z, d_g = to_dev(padded(N * H * W, C, g), dev), to_dev(padded(N * H * W, C, g), dev)
        K.s2d_bwd(dz, z, 0.2, N, H, W, C, vh_, vw_, d_z, False)
        K.s2d_bwd(dz, None, 0.2, N, H, W, C, vh_, vw_, d_g, False, gate=src)
        bad += int((d_z != d_g).sum())
        da = Act(d_g, N, H, W)
        want_b = K.x3_pack(K._src(da), da.P, da.C, dev)
        got_b = K.s2d_pack_bwd(dz, src, 0.2, N, H, W, C, vh_, vw_)
        bad += int((body(got_b._x3, da.P, da.C) != body(want_b, da.P, da.C)).sum())
        got_n = K.s2d_pack_bwd(dz, None, 0.2, N, H, W, C, vh_, vw_)          # no gate (slope irrelevant)
        K.s2d_bwd(dz, None, 1.0, N, H, W, C, vh_, vw_, d_z, False)
        dn = Act(d_z, N, H, W)
        bad += int((body(got_n._x3, dn.P, dn.C) != body(K.x3_pack(K._src(dn), dn.P, dn.C, dev), dn.P, dn.C)).sum())
        return float(bad), 0.0
    return run


CASES += [
    ("s2d packed C=64 grid 18x18 valid 17x17", case_s2d_packed(2, 18, 18, 64, 17, 17)),
    ("s2d packed C=128 grid 35x35 valid 33x33", case_s2d_packed(3, 35, 35, 128, 33, 33)),
    ("s2d packed C=8 grid 11x9 valid 9x8 (half-filled 16-blocks)", case_s2d_packed(1, 11, 9, 8, 9, 8)),
    ("s2d packed C=72 grid 10x12 (4C = 288, ragged tiles)", case_s2d_packed(2, 10, 12, 72)),
]


def case_relayout_s2d(O, C, seed=21):
    def run(dev):
        w = torch.randn(O, C, 4, 4, generator=gen(seed))
        K = hip()
        return max(rel(K.relayout_s2d(w.to(dev), False), SPEC.relayout_s2d(w, False)),
                   rel(K.relayout_s2d(w.to(dev), True), SPEC.relayout_s2d(w, True))), 0.0
    return run


CASES += [("relayout s2d 64<-2", case_relayout_s2d(64, 2)), ("relayout s2d 128<-64", case_relayout_s2d(128, 64)),
          ("relayout s2d 1<-512", case_relayout_s2d(1, 512))]

# 256 x 256 weight-gradient tiles (Cout >= 192, J >= 256, >= 4096 pixel chunks = 131072 pixels)
CASES += [
    ("wgrad3x3 32->256 P=131072 (256x256 tiles)", case_wgrad(2, 256, 256, 32, 256, 3, 1, lazy=False)),
    ("wgrad3x3 40->200 P=131325 mask (256x256 tiles, ragged)", case_wgrad(1, 255, 515, 40, 200, 3, 2, mask=True)),
    ("wgrad1x1 300->320 P=131072 lazy (256x256 tiles)", case_wgrad(2, 256, 256, 300, 320, 1, 1)),
    ("wgrad2x2 o0 64->256 P=131841 (256x256 tiles)", case_wgrad(1, 363, 363, 64, 256, 2, 1, lazy=False)),
]

# weight gradients the bf16x3 mode routes to its own kernel (Cin % 16 == 0, Cout >= 96, k >= 2, P >= 4096; transposed LDS reads)
CASES += [
    ("wgrad3x3 256->256 P=4608 mask (x3 128 tiles)", case_wgrad(2, 48, 48, 256, 256, 3, 1, mask=True)),
    ("wgrad3x3 304->256 P=8192 (x3 128 tiles)", case_wgrad(2, 64, 64, 304, 256, 3, 1, lazy=False)),
    ("wgrad3x3 320->256 dil12 P=4608 (x3)", case_wgrad(2, 48, 48, 320, 256, 3, 12)),
    ("wgrad3x3 48->100 P=4700 ragged (x3)", case_wgrad(2, 50, 47, 48, 100, 3, 1)),
    ("wgrad2x2 o0 256->128 P=8978 (x3)", case_wgrad(2, 67, 67, 256, 128, 2, 1, lazy=False, origin=0)),
    ("wgrad2x2 o0 64->200 P=70225 (x3 256 tiles, ragged)", case_wgrad(1, 265, 265, 64, 200, 2, 1, lazy=False, origin=0)),
    ("wgrad1x1 256->2304 P=4608 (x3, tap GEMM of the re-associated decoder conv)", case_wgrad(2, 48, 48, 256, 2304, 1, 1, lazy=False)),
    ("wgrad2x2 o0 1024->512 P=9248 (x3 256 tiles on few pixels)", case_wgrad(2, 68, 68, 1024, 512, 2, 1, lazy=False, origin=0)),
]
# bf16x3: the last, partly filled round of tiles split over K (no statistics epilogue; fp32 partial tiles + x3_tail_reduce_kernel)
CASES += [
    ("conv3x3 64->128 P=37965 bias addend, no stats (x3 tail split, ragged rows)", case_conv(1, 195, 195 - 0, 64, 128, 3, 1, bias=True, addend=True, stats=False)),
    ("conv2x2 o1 256->250 P=38000 raw, no stats (x3 tail split, ragged columns)", case_conv(1, 200, 190, 256, 250, 2, 1, lazy=False, stats=False, origin=1)),
    ("conv1x1 256->2304 P=4864 no stats (x3 tail split, wide 1x1)", case_conv(1, 38, 128, 256, 2304, 1, 1, lazy=False, stats=False)),
    ("dgrad3x3 304<-256 P=37888 accumulate (x3 tail split)", case_dgrad(2, 148, 128, 304, 256, 3, 1, accumulate=True)),
]
# bf16x3 routes added with the 256 x 64 tile and the wide 1x1 route
CASES += [
    ("conv1x1 256->2304 stats (x3 wide 1x1)", case_conv(2, 40, 36, 256, 2304, 1, 1, lazy=False)),
    ("conv1x1 144->1030 relu6 ragged (x3 wide 1x1)", case_conv(1, 37, 29, 144, 1030, 1, 1)),
    ("conv3x3 256->48 P=2442 ragged (x3 256 x 64 tile)", case_conv(2, 33, 37, 256, 48, 3, 1, lazy=False)),
    ("conv3x3 160->60 dil2 mask stats (x3 256 x 64 tile)", case_conv(1, 45, 41, 160, 60, 3, 2, mask=True)),
    ("conv2x2 o1 256->56 addend (x3 256 x 64 tile)", case_conv(2, 30, 30, 256, 56, 2, 1, lazy=False, addend=True, origin=1)),
]
# narrow 1x1 convs: 64-pixel tiles when 128-pixel tiles would leave half of the CUs without a workgroup (the 32x32-map layers at
# B = 16: P = 16384), and the 128-pixel tiles of the same routes on more pixels (round 3)
CASES += [
    ("conv1x1 384->64 P=16384 relu6 stats (64-pixel tiles)", case_conv(16, 32, 32, 384, 64, 1, 1)),
    ("conv1x1 576->96 P=16384 addend no stats (64-pixel tiles, 128 columns)", case_conv(16, 32, 32, 576, 96, 1, 1, addend=True, stats=False)),
    ("conv1x1 192->30 P=5655 ragged mask bias (64-pixel tiles)", case_conv(3, 65, 29, 192, 30, 1, 1, mask=True, bias=True)),
    ("dgrad1x1 64<-384 P=16384 accumulate (64-pixel tiles)", case_dgrad(16, 32, 32, 64, 384, 1, 1, accumulate=True)),
    ("conv1x1 384->64 P=32000 ragged stats (128-pixel tiles)", case_conv(2, 125, 128, 384, 64, 1, 1)),
    ("conv1x1 96->24 P=40000 raw addend (128-pixel tiles)", case_conv(1, 200, 200, 96, 24, 1, 1, lazy=False, addend=True)),
    ("conv1x1 64->96 P=28900 relu6 mask (128-pixel tiles)", case_conv(1, 170, 170, 64, 96, 1, 1, mask=True)),
]
# bf16x3 on long-K 1x1 convs towards >= 256 outputs (ResNet-101's bottleneck convs on the 32x32 maps, round 3)
CASES += [
    ("conv1x1 1024->256 P=8192 raw stats (x3 long-K 1x1)", case_conv(8, 32, 32, 1024, 256, 1, 1, lazy=False)),
    ("conv1x1 2048->512 P=2312 relu ragged (x3 long-K 1x1)", case_conv(2, 34, 34, 2048, 512, 1, 1)),
    ("dgrad1x1 1024<-256 P=8192 accumulate (x3 wide 1x1)", case_dgrad(8, 32, 32, 1024, 256, 1, 1, accumulate=True)),
    ("wgrad1x1 1024->256 P=8192 raw (x3 long-K 1x1)", case_wgrad(8, 32, 32, 1024, 256, 1, 1, lazy=False)),
    ("wgrad1x1 2048->512 P=4624 relu (x3 long-K 1x1)", case_wgrad(4, 34, 34, 2048, 512, 1, 1)),
]
# short-K 1x1 convs over >= 32768 pixels: the barrier-free one-wave-per-32-pixels kernel (conv1x1_stream_kernel)
CASES += [
    ("conv1x1 16->96 P=33800 relu6 stats (stream kernel, ragged last tile)", case_conv(2, 130, 130, 16, 96, 1, 1)),
    ("conv1x1 16->90 P=33800 raw addend no stats (stream kernel, ragged columns)", case_conv(2, 130, 130, 16, 90, 1, 1, lazy=False, addend=True, stats=False)),
    ("conv1x1 24->144 P=40000 raw addend (stream kernel, two column groups)", case_conv(1, 200, 200, 24, 144, 1, 1, lazy=False, addend=True)),
    ("conv1x1 24->48 P=36100 relu stats (stream kernel)", case_conv(1, 190, 190, 24, 48, 1, 1)),
    ("conv1x1 32->192 P=36300 relu stats addend (stream kernel, two column groups)", case_conv(3, 110, 110, 32, 192, 1, 1, addend=True)),
    ("conv1x1 32->130 P=32768 raw (stream kernel, second group ragged)", case_conv(2, 128, 128, 32, 130, 1, 1, lazy=False)),
    ("conv1x1 32->192 P=262144 relu6 stats (stream kernel, two column groups)", case_conv(4, 256, 256, 32, 192, 1, 1)),
    ("conv1x1 16->32 P=33800 raw addend (stream kernel, one block)", case_conv(2, 130, 130, 16, 32, 1, 1, lazy=False, addend=True, stats=False)),
    ("conv1x1 24->96 P=40000 raw addend (stream kernel, three blocks)", case_conv(1, 200, 200, 24, 96, 1, 1, lazy=False, addend=True, stats=False)),
]
# stride 2 on the wide tiles: the loaders walk the strided output grid (ResNet-101 layer2.0 / layer3.0 conv2, resnet.py:66, and their
# weight gradients; the 1x1 form is the shortcut conv, resnet.py:93)
CASES += [
    ("conv3x3 s2 128->128 65x67 relu stats (strided grid, odd sizes)", case_conv(2, 65, 67, 128, 128, 3, 1, stride=2)),
    ("conv3x3 s2 256->256 P=8192 raw addend no stats (tail split)", case_conv(8, 64, 64, 256, 256, 3, 1, lazy=False, addend=True, stats=False, stride=2)),
    ("conv3x3 s2 dil2 160->200 mask ragged", case_conv(3, 41, 38, 160, 200, 3, 2, mask=True, stride=2)),
    ("conv1x1 s2 256->512 33x40 relu bias", case_conv(2, 33, 40, 256, 512, 1, 1, bias=True, stride=2)),
    ("conv1x1 s2 512->1024 raw (x3 wide 1x1)", case_conv(2, 64, 64, 512, 1024, 1, 1, lazy=False, stride=2)),
    ("wgrad3x3 s2 128->128 65x67 relu", case_wgrad(2, 65, 67, 128, 128, 3, 1, stride=2)),
    ("wgrad3x3 s2 256->256 P=16384 raw (x3)", case_wgrad(4, 128, 128, 256, 256, 3, 1, lazy=False, stride=2)),
    ("wgrad3x3 s2 128->128 41 images of 20x20 (rows shorter than a chunk)", case_wgrad(41, 20, 20, 128, 128, 3, 1, stride=2)),
    ("wgrad1x1 s2 256->512 33x40 mask", case_wgrad(2, 33, 40, 256, 512, 1, 1, mask=True, stride=2)),
]


# ---------------------------------------------------------------- input pipeline tail: bit-exact against scipy
def _fundus_like(B, H, W, g):
    """uint8 image + grey-coded mask (255 background, 128 disc rim, 0 cup) with ellipses that touch the border in one sample."""
    import numpy as np
    rs = np.random.RandomState(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g)))
    img = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    lab = np.full((B, H, W), 255, np.uint8)
    for b in range(B):
        cy, cx = (0.15 if b == 0 else rs.uniform(0.35, 0.65)) * H, rs.uniform(0.35, 0.65) * W
        a, c = rs.uniform(0.2, 0.35) * H, rs.uniform(0.2, 0.35) * W
        r = np.sqrt(((yy - cy) / a) ** 2 + ((xx - cx) / c) ** 2)
        lab[b][r <= 1.0] = 128
        lab[b][r <= rs.uniform(0.4, 0.7)] = 0
        lab[b][rs.rand(H, W) < 0.002] = rs.choice([0, 60, 128, 200, 201, 255])      # isolated pixels, threshold values 50/51/200/201
        lab[b][0, :7] = 50
        lab[b][-1, -9:] = 51
    return img, lab


def case_normalize_tf(B, H, W, seed=41):
    """uda_normalize_tf against the reference's own per-sample arithmetic (custom_transforms.py:414-466: numpy + scipy.ndimage)."""
    def run(dev):
        import numpy as np
        from scipy import ndimage
        g = gen(seed)
        img, lab = _fundus_like(B, H, W, g)
        K = hip()
        image, mp, bd = K.normalize_tf(torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev))
        worst = 0.0
        for b in range(B):
            ri = img[b].astype(np.float32)
            ri /= 127.5
            ri -= 1.0
            cup, disc = (lab[b] <= 50).astype(np.float64), (lab[b] <= 200).astype(np.float64)
            ring = np.zeros((H, W), bool)
            for m in (cup, disc):
                d = ndimage.binary_dilation(m, iterations=5).astype(m.dtype)
                e = ndimage.binary_erosion(m, iterations=5).astype(m.dtype)
                s = d + e
                s[s == 2] = 0
                ring |= s > 0
            rb = ndimage.gaussian_filter(ring.astype(np.uint8) * 255, sigma=3) / 255.0
            exact = (torch.equal(image[b].cpu(), torch.from_numpy(ri.transpose(2, 0, 1).copy())) and
                     torch.equal(mp[b].cpu(), torch.from_numpy(np.stack([cup, disc])).float()) and
                     torch.equal(bd[b, 0].cpu(), torch.from_numpy(rb).float()))
            worst = max(worst, 0.0 if exact else 1.0)
        return worst, 0.5          # bit-exact or fail
    return run


def case_elastic(B, H, W, seed=43):
    """uda_field_smooth + uda_elastic_warp against scipy (custom_transforms.py:95-147) on numpy's own uniform draw: the float64
    displacement field and the warped bytes are both bit-identical."""
    def run(dev):
        import numpy as np
        from scipy import ndimage
        g = gen(seed)
        img, lab = _fundus_like(B, H, W, g)
        rs = np.random.RandomState(7)
        alpha, sigma = 2.0 * W, 0.08 * W
        noise = np.stack([[rs.rand(H, W) * 2 - 1 for _ in range(B)] for _ in range(2)])
        K = hip()
        fld = K.field_smooth(torch.from_numpy(noise).to(dev), sigma, alpha)
        ref_f = np.stack([[ndimage.gaussian_filter(noise[q, b], sigma, mode="constant", cval=0) * alpha
                           for b in range(B)] for q in range(2)])
        bad = int((fld.cpu().numpy() != ref_f).sum())
        apply = torch.tensor([1] * (B - 1) + [0], dtype=torch.uint8)
        io, lo = K.elastic_warp(torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev), fld[0], fld[1], apply.to(dev))
        gx, gy = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        for b in range(B):
            if not apply[b]:
                bad += int((io[b].cpu().numpy() != img[b]).sum() + (lo[b].cpu().numpy() != lab[b]).sum())
                continue
            idx = np.reshape(gx + ref_f[0, b], (-1, 1)), np.reshape(gy + ref_f[1, b], (-1, 1))
            ri = np.stack([ndimage.map_coordinates(img[b][:, :, c], idx, order=1).reshape(H, W) for c in range(3)], -1)
            rl = ndimage.map_coordinates(lab[b], idx, order=1, mode="nearest").reshape(H, W)
            bad += int((io[b].cpu().numpy() != ri).sum() + (lo[b].cpu().numpy() != rl).sum())
        return float(bad), 0.5       # bit-exact or fail
    return run


CASES += [
    ("normalize_tf 3 x 96 x 80 vs scipy (bit-exact)", case_normalize_tf(3, 96, 80)),
    ("normalize_tf 2 x 512 x 512 vs scipy (bit-exact)", case_normalize_tf(2, 512, 512)),
    ("elastic field + warp 3 x 96 x 80 vs scipy (bit-exact)", case_elastic(3, 96, 80)),
    ("elastic field + warp 2 x 256 x 256 vs scipy (bit-exact)", case_elastic(2, 256, 256)),
]


# ---------------------------------------------------------------- upsample-then-conv3x3 by low-resolution tap GEMMs + interpolation
def case_upconv(N, h, w, H, W, C, dil=1, addend_rows=None, seed=47):
    def run(dev):
        g = gen(seed)
        K = hip()
        gl = torch.randn(N * h * w, 9 * C, generator=g)
        ad = None if addend_rows is None else padded(addend_rows, C, g)
        o_r = padded(N * H * W, C, g)
        st_r = torch.zeros(16, 2, C, dtype=torch.float64)
        SPEC.upconv_fwd(gl, N, h, w, o_r, H, W, ad, dil, st_r)
        o_h = to_dev(padded(N * H * W, C, g), dev)
        st_h = torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
        K.upconv_fwd(gl.to(dev), N, h, w, o_h, H, W, to_dev(ad, dev), dil, st_h)
        errs = [rel(o_h, o_r), rel(st_h.sum(0), st_r.sum(0))]
        dy = padded(N * H * W, C, g)
        dg_r = torch.empty(N * h * w, 9 * C)
        SPEC.upconv_bwd(dy, N, H, W, dg_r, h, w, dil)
        dg_h = torch.empty(N * h * w, 9 * C, device=dev)
        K.upconv_bwd(to_dev(dy, dev), N, H, W, dg_h, h, w, dil)
        errs.append(rel(dg_h, dg_r))
        return max(errs), 2e-5
    return run


def case_upconv_identity(N, h, w, H, W, Cf, Cl, Cout, seed=48):
    """The identity the engine relies on: conv3x3(cat(up(f), low)) == upconv(f W_taps^T) + conv3x3_low(low), against
    F.interpolate + F.conv2d on the same tensors (HIP kernels for every piece)."""
    def run(dev):
        import torch.nn.functional as F
        g = gen(seed)
        K = hip()
        f = torch.randn(N, Cf, h, w, generator=g)
        low = torch.randn(N, Cl, H, W, generator=g)
        wt = torch.randn(Cout, Cf + Cl, 3, 3, generator=g) / ((Cf + Cl) * 9) ** 0.5
        ref = F.conv2d(torch.cat([F.interpolate(f, size=(H, W), mode="bilinear", align_corners=True), low], 1), wt, None, 1, 1)
        ref = ref.permute(0, 2, 3, 1).reshape(N * H * W, Cout)
        f2 = to_dev(f.permute(0, 2, 3, 1).reshape(N * h * w, Cf).contiguous(), dev)
        l2 = to_dev(low.permute(0, 2, 3, 1).reshape(N * H * W, Cl).contiguous(), dev)
        w_taps = wt[:, :Cf].permute(2, 3, 0, 1).reshape(9 * Cout, Cf, 1, 1).contiguous()
        gl = torch.empty(N * h * w, 9 * Cout, device=dev)
        K.conv(Act(f2, N, h, w), K.relayout_ohwi(w_taps.to(dev)), 1, 1, gl)
        y0 = torch.empty(N * H * W, Cout, device=dev)
        K.conv(Act(l2, N, H, W), K.relayout_ohwi(wt[:, Cf:].contiguous().to(dev)), 3, 1, y0)
        y = torch.empty(N * H * W, Cout, device=dev)
        K.upconv_fwd(gl, N, h, w, y, H, W, y0)
        return rel(y, ref), 2e-5
    return run


CASES += [
    ("upconv fwd/bwd 2 x (8x8 -> 32x32) x 64", case_upconv(2, 8, 8, 32, 32, 64)),
    ("upconv fwd/bwd 1 x (5x7 -> 12x20) x 8, addend", case_upconv(1, 5, 7, 12, 20, 8, addend_rows=240)),
    ("upconv fwd/bwd 4 x (4x4 -> 16x16) x 16, addend shared by 2 reps", case_upconv(4, 4, 4, 16, 16, 16, addend_rows=512)),
    ("upconv fwd/bwd 1 x (16x16 -> 32x32) x 12 dil 2 (pixel kernel + colstats)", case_upconv(1, 16, 16, 32, 32, 12, dil=2)),
    ("upconv fwd/bwd 2 x (16x16 -> 32x32) x 32 (x2: 4-column strips)", case_upconv(2, 16, 16, 32, 32, 32)),
    ("upconv fwd/bwd 1 x (5x7 -> 13x18) x 8 (W % 4 != 0: pixel kernel), addend", case_upconv(1, 5, 7, 13, 18, 8, addend_rows=234)),
    ("upconv fwd/bwd 2 x (1x1 -> 4x4) x 4 (degenerate source)", case_upconv(2, 1, 1, 4, 4, 4)),
    ("upconv fwd/bwd 1 x (43x43 -> 128x128) x 8 (scale exactly 1/3: 4-column strips)", case_upconv(1, 43, 43, 128, 128, 8)),
    ("upconv fwd/bwd 1 x (85x85 -> 128x128) x 8 (scale 0.661: 4-column strips)", case_upconv(1, 85, 85, 128, 128, 8)),
    ("upconv fwd/bwd 1 x (86x86 -> 128x128) x 8 (scale 0.669: pixel kernel)", case_upconv(1, 86, 86, 128, 128, 8)),
    ("upconv fwd/bwd 3 x (32x32 -> 128x128) x 256 (the decoder's shape)", case_upconv(3, 32, 32, 128, 128, 256, addend_rows=16384)),
    ("upconv fwd/bwd 2 x (4x4 -> 16x16) x 256 (tile kernel, one tile; wave-per-pixel bwd)", case_upconv(2, 4, 4, 16, 16, 256, addend_rows=256)),
    ("upconv fwd/bwd 1 x (5x9 -> 17x33) x 256 (exact ratio 1/4, odd sizes: strip / pixel fwd, wave bwd)", case_upconv(1, 5, 9, 17, 33, 256)),
    ("upconv fwd/bwd 1 x (9x5 -> 32x16) x 256 (tile kernel, ratio 0.258 / 0.267), addend", case_upconv(1, 9, 5, 32, 16, 256, addend_rows=512)),
    ("upconv fwd/bwd 2 x (16x16 -> 48x64) x 256 (tile kernel at x3 / x4.2)", case_upconv(2, 16, 16, 48, 64, 256)),
    ("upconv identity vs interpolate+conv2d 2 x (8x8 -> 32x32), 64+16 -> 32", case_upconv_identity(2, 8, 8, 32, 32, 64, 16, 32)),
]


# ---------------------------------------------------------------- evaluation post-processing vs scipy
def _scipy_postprocess(prob, thr_cup, thr_disc):
    """utils/Utils.py:427-463 with skimage's two calls replaced by their scipy.ndimage equivalents: binary_erosion(diamond(7))
    = ndimage.binary_erosion(structure=L1 ball, border_value=1) (skimage erodes with the outside set), measure.label =
    ndimage.label with the full 3x3 structure (both number components in raster order of their first pixel)."""
    import numpy as np
    import scipy.signal
    from scipy import ndimage
    yy, xx = np.mgrid[-7:8, -7:8]
    diamond = (np.abs(yy) + np.abs(xx)) <= 7
    out = np.zeros(prob.shape, np.uint8)
    for c, thr in ((0, thr_cup), (1, thr_disc)):
        m = (prob[c] > thr).astype(np.uint8)
        for _ in range(5):
            m = scipy.signal.medfilt2d(m, 7)
        m = ndimage.binary_erosion(m, structure=diamond, border_value=1).astype(np.uint8)
        lab, n = ndimage.label(m, structure=np.ones((3, 3)))
        if n:
            areas = np.bincount(lab.ravel())[1:]
            m[lab != int(np.argmax(areas)) + 1] = 0
        out[c] = ndimage.binary_fill_holes(m.astype(int)).astype(np.uint8)
    return out


def case_postprocess(B, H, W, seed=61):
    def run(dev):
        import numpy as np
        g = gen(seed)
        K = hip()
        rs = np.random.RandomState(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g)))
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        prob = np.zeros((B, 2, H, W), np.float32)
        for b in range(B):
            for c in range(2):
                p = 0.08 * rs.rand(H, W)
                for _ in range(rs.randint(1, 4)):               # a few blobs (one of them touching the border in sample 0), with holes
                    cy = (0.05 if (b == 0 and c == 1) else rs.uniform(0.25, 0.75)) * H
                    cx, a, d = rs.uniform(0.25, 0.75) * W, rs.uniform(0.12, 0.3) * H, rs.uniform(0.12, 0.3) * W
                    r = np.sqrt(((yy - cy) / a) ** 2 + ((xx - cx) / d) ** 2)
                    p = np.maximum(p, np.clip(1.25 - r, 0, 1))
                    hole = np.sqrt((yy - cy - 0.3 * a) ** 2 + (xx - cx) ** 2) < rs.uniform(2, 9)
                    p[hole] = 0.02
                p += 0.5 * (rs.rand(H, W) < 0.03)                 # salt: isolated pixels the median removes
                p[rs.rand(H, W) < 0.03] = 0.0                     # pepper inside the blobs
                prob[b, c] = np.clip(p, 0, 1)
        out = K.postprocess(torch.from_numpy(prob).to(dev), 0.75, 0.75).cpu().numpy()
        bad = sum(int((out[b] != _scipy_postprocess(prob[b], 0.75, 0.75)).sum()) for b in range(B))
        outd = K.postprocess(torch.from_numpy(prob[:1]).to(dev), 0.1, 0.5).cpu().numpy()
        bad += int((outd[0] != _scipy_postprocess(prob[0], 0.1, 0.5)).sum())
        empty = K.postprocess(torch.zeros(1, 2, H, W, device=dev), 0.75, 0.75)
        bad += int(empty.sum())
        return float(bad), 0.5                  # bit-exact or fail
    return run


CASES += [
    ("postprocess 3 x 2 x 128 x 128 vs scipy (bit-exact)", case_postprocess(3, 128, 128)),
    ("postprocess 2 x 2 x 200 x 136 vs scipy (ragged tiles)", case_postprocess(2, 200, 136, seed=62)),
    ("postprocess 1 x 2 x 512 x 512 vs scipy", case_postprocess(1, 512, 512, seed=63)),
]


# ---------------------------------------------------------------- fused alignment / adversarial glue
def case_proto_align(C, prev, seed=71):
    """uda_proto_align_fwd / bwd (EMA of the eight centroids + intra / inter and the gradient through the current term)."""
    def run(dev):
        g = gen(seed)
        K = hip()
        cs, ct = torch.randn(4, C, generator=g), torch.randn(4, C, generator=g)
        ps, pt = (torch.randn(4, C, generator=g), torch.randn(4, C, generator=g)) if prev else (None, None)
        d = lambda t: None if t is None else t.to(dev)
        ns_r, nt_r, l_r = SPEC.proto_align_fwd(cs, ct, ps, pt, 0.9)
        ns_h, nt_h, l_h = K.proto_align_fwd(cs.to(dev), ct.to(dev), d(ps), d(pt), 0.9)
        gi = torch.tensor([0.37])
        w = 0.9 if prev else 1.0
        a_r, b_r = SPEC.proto_align_bwd(ns_r, nt_r, gi, w, w)
        a_h, b_h = K.proto_align_bwd(ns_h, nt_h, gi.to(dev), w, w)
        exact = torch.equal(ns_h.cpu(), ns_r) and torch.equal(nt_h.cpu(), nt_r)      # the EMA itself: the reference's expression, bit for bit
        return max(rel(l_h, l_r), rel(a_h, a_r), rel(b_h, b_r), 0.0 if exact else 1.0), 2e-6
    return run


def case_adv_loss(n1, n2, label, scale, seed=72):
    def run(dev):
        g = gen(seed)
        K = hip()
        d1, d2 = 3 * torch.randn(n1, generator=g), 3 * torch.randn(n2, generator=g)
        d1[:4] = torch.tensor([40., -40., 100., -100.])
        l_r = SPEC.adv_loss_fwd(d1, d2, label, scale)
        l_h = K.adv_loss_fwd(d1.to(dev), d2.to(dev), label, scale)
        gi = torch.tensor([1.7])
        a_r, b_r = SPEC.adv_loss_bwd(d1, d2, label, scale, gi)
        a_h, b_h = K.adv_loss_bwd(d1.to(dev), d2.to(dev), label, scale, gi.to(dev))
        return max(rel(l_h, l_r), rel(a_h, a_r), rel(b_h, b_r)), 2e-6
    return run


def case_adv_s2d(N, C, H, W, op, seed=73):
    """First discriminator layer fed by logits: z = s2d(sigmoid | uncertainty map) and its adjoint incl. the map's derivative."""
    def run(dev):
        g = gen(seed)
        K = hip()
        x = 3 * torch.randn(N, C, H, W, generator=g)
        x[0, 0, 0, :4] = torch.tensor([30., -30., 90., -90.])          # saturated sigmoids: log(s + 1e-7) at s = 0 and s = 1
        Hz, Wz = (H + 5) // 2, (W + 5) // 2
        z_r, z_h = torch.empty(N * Hz * Wz, 4 * C), torch.empty(N * Hz * Wz, 4 * C, device=dev)
        SPEC.adv_s2d_fwd(x, op, z_r)
        K.adv_s2d_fwd(x.to(dev), op, z_h)
        dz = torch.randn(N * Hz * Wz, 4 * C, generator=g)
        d_r, d_h = torch.empty_like(x), torch.empty(N, C, H, W, device=dev)
        SPEC.adv_s2d_bwd(dz, x, op, d_r)
        K.adv_s2d_bwd(dz.to(dev), x.to(dev), op, d_h)
        return max(rel(z_h, z_r), rel(d_h, d_r)), 5e-6
    return run


CASES += [
    ("proto_align C=305 first use", case_proto_align(305, False)),
    ("proto_align C=305 EMA", case_proto_align(305, True)),
    ("proto_align C=7 EMA", case_proto_align(7, True)),
    ("adv_loss 2x17x17 label 1 scale 0.01", case_adv_loss(578, 578, 1.0, 0.01)),
    ("adv_loss label 0 scale 1 ragged", case_adv_loss(300, 77, 0.0, 1.0)),
    ("adv_s2d sigmoid C=1 64x64", case_adv_s2d(2, 1, 64, 64, 1)),
    ("adv_s2d entropy C=2 50x46", case_adv_s2d(2, 2, 50, 46, 2)),
]
