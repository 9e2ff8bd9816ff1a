"""Where does the HIP path round more than a plain fp32 evaluation?  (TEST INFRASTRUCTURE, run on the GPU box.)

Every kernel call of one training forward + backward of the generator is executed three times on the SAME inputs:
the HIP kernel, the tests' fp32 torch statement (tests/kernel_spec.py, CPU) and that statement in float64.  Per call the
L2 distance of each fp32 result to the float64 one is logged; the table at the end groups calls by kernel and operand
shape and prints  noise(HIP) / noise(torch fp32).  A ratio near 1 = the kernel rounds like any fp32 summation order; a
large ratio on a reduction names the accumulation chain to shorten.

    python tests/noise_report.py [S=256] [B=2]        -> gpurun_out/noise_report_<S>.txt
Then the end-to-end view: per parameter class, gradient distance to the fp64 oracle (HIP | fp32 oracle | ratio)."""
import math
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import model_cases
from dual_kernels import _clone, _tensors
from kernel_spec import SpecKernels
from oracle import deeplab_ref, step_ref
from uda_clr_amd.acts import Act, BNRec
from uda_clr_amd.engine import GeneratorEngine
from uda_clr_amd.kernels import HipKernels


def _to(v, dev=None, dt=None, memo=None):
    memo = {} if memo is None else memo
    if isinstance(v, torch.Tensor):
        key = (v.data_ptr(), tuple(v.shape), tuple(v.stride()), v.dtype)
        if key not in memo:
            if v.dim() == 2 and v.stride(1) == 1 and v.stride(0) != v.shape[1]:       # keep padded row strides (views of wider buffers)
                base = torch.zeros(v.shape[0], v.stride(0), dtype=v.dtype, device=v.device)
                base[:, :v.shape[1]] = v
                t = base.to(device=dev)
                t = t.to(dt) if (dt is not None and t.dtype == torch.float32) else t
                t = t[:, :v.shape[1]]
            else:
                t = v.detach().clone().to(device=dev)
                t = t.to(dt) if (dt is not None and t.dtype == torch.float32) else t
            memo[key] = t
        return memo[key]
    if isinstance(v, Act):
        bn = None if v.bn is None else BNRec(v.bn.key, _to(v.bn.mean, dev, dt, memo), _to(v.bn.invstd, dev, dt, memo), v.bn.count, v.bn.q1_border)
        return Act(_to(v.x, dev, dt, memo), v.N, v.H, v.W, _to(v.scale, dev, dt, memo), _to(v.shift, dev, dt, memo), v.act,
                   _to(v.mask, dev, dt, memo), v.mask_scale, bn)
    return v


class TriKernels:
    """HIP result is what the engine continues with; the two CPU legs only measure."""
    HEAVY = {"conv", "conv_wgrad", "dwconv_fwd", "dwconv_dgrad", "dwconv_wgrad", "stem_fwd", "stem_wgrad", "bnbwd_reduce",
             "bnbwd_apply", "bn_apply", "colstats", "upconv_fwd", "upconv_bwd", "upsample_fwd", "upsample_bwd",
             "head_upsample_fwd", "head_upsample_bwd", "bn_finalize", "bnbwd_finalize", "gap_fwd", "broadcast_rows"}

    def __init__(self, hip, spec, log):
        self.hip, self.spec, self.log = hip, spec, log
        self.rows = []

    def __getattr__(self, meth):
        h, s = getattr(self.hip, meth), getattr(self.spec, meth)
        if meth not in self.HEAVY:
            return h

        def call(*args, **kw):
            m32, m64 = {}, {}
            a32 = [_to(a, "cpu", None, m32) for a in args]
            k32 = {k: _to(v, "cpu", None, m32) for k, v in kw.items()}
            a64 = [_to(a, "cpu", torch.float64, m64) for a in args]
            k64 = {k: _to(v, "cpu", torch.float64, m64) for k, v in kw.items()}
            s(*a32, **k32)
            s(*a64, **k64)
            r = h(*args, **kw)
            named = [("arg%d" % i, a, b, c) for i, (a, b, c) in enumerate(zip(args, a32, a64))] + \
                    [(k, kw[k], k32[k], k64[k]) for k in kw]
            for n, a, b, c in named:
                for (tn, th), (_, t32), (_, t64) in zip(_tensors(n, a), _tensors(n, b), _tensors(n, c)):
                    if th.dtype == torch.uint8:
                        continue
                    ref, vh, vs = t64.double(), th.detach().double().cpu(), t32.double()
                    if th.dtype == torch.float64 and th.dim() == 3 and th.shape[0] == 16:      # slot-replicated accumulators: compare the totals
                        ref, vh, vs = ref.sum(0), vh.sum(0), vs.sum(0)
                    den = ref.norm().item()
                    if den == 0:
                        continue
                    eh = (vh - ref).norm().item() / den
                    es = (vs - ref).norm().item() / den
                    if eh == 0 and es == 0:
                        continue
                    shp = [tuple(x.shape) if isinstance(x, torch.Tensor) else ((tuple(x.x.shape), x.act, x.mask is not None) if isinstance(x, Act) else x) for x in args]
                    self.rows.append((meth, tn, str(shp), eh, es))
            return r
        return call


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    os.makedirs("gpurun_out", exist_ok=True)
    out = open("gpurun_out/noise_report_%d.txt" % S, "w")

    def emit(s):
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    torch.set_num_threads(16)
    m = model_cases.seeded_model(perturb=True).train()
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, S, S, generator=gen)
    tmap = (torch.rand(B, 2, S, S, generator=gen) > 0.5).float()
    tbd = torch.rand(B, 1, S, S, generator=gen)
    masks = deeplab_ref.draw_masks(B, S, S, gen)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}

    def total(outs, dt, dv):
        return step_ref.seg_loss(outs[0], outs[1], tmap.to(dv, dt), tbd.to(dv, dt))

    if "--calls" in sys.argv:
        m.to(dev)
        tri = TriKernels(HipKernels(), SpecKernels(), emit)
        m._engine_override = GeneratorEngine(tri)
        m.set_dropout_masks(masks)
        total(m(x.to(dev)), torch.float32, dev).backward()
        agg = defaultdict(list)
        for meth, tn, shp, eh, es in tri.rows:
            agg[(meth, tn)].append((eh, es, shp))
        emit("per-call rounding noise vs float64 on identical inputs, S=%d B=%d (geometric means over the calls of a kernel)" % (S, B))
        emit("%-22s %-10s %5s %10s %10s %7s   worst call (ratio, shapes)" % ("kernel", "output", "calls", "hip", "torch32", "ratio"))
        for (meth, tn), v in sorted(agg.items()):
            gh = math.exp(sum(math.log(max(a, 1e-12)) for a, _, _ in v) / len(v))
            gs = math.exp(sum(math.log(max(b, 1e-12)) for _, b, _ in v) / len(v))
            w = max(v, key=lambda t: t[0] / max(t[1], 1e-12))
            emit("%-22s %-10s %5d %10.2e %10.2e %7.2f   %.1f %s" % (meth, tn, len(v), gh, gs, gh / gs, w[0] / max(w[1], 1e-12), w[2][:150]))
        m._engine_override = None
        for p in m.parameters():
            p.grad = None
        m.cpu()
        m.load_state_dict(sd0)
    # ---- end to end: gradients vs the fp64 oracle, by parameter class
    o32 = deeplab_ref.canonical_state(sd0, requires_grad=True)
    total(deeplab_ref.deeplab_forward(o32, x, training=True, masks=masks), torch.float32, "cpu").backward()
    o64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone())
           for k, v in deeplab_ref.canonical_state(sd0, requires_grad=True).items()}
    total(deeplab_ref.deeplab_forward(o64, x.double(), training=True, masks=masks), torch.float64, "cpu").backward()
    m.to(dev)
    m.set_dropout_masks(masks)
    total(m(x.to(dev)), torch.float32, dev).backward()
    live = m._flat_state()

    def klass(k):
        t = live[k]
        sec = "backbone" if k.startswith("backbone") else ("aspp" if k.startswith("aspp") else "decoder")
        if t.dim() == 4:
            kind = "dw3x3" if (t.shape[1] == 1 and t.shape[2] == 3) else ("%dx%d" % (t.shape[2], t.shape[3]))
            return "%s conv %s" % (sec, kind)
        if k.endswith(".bias") and (k[:-5] + ".running_mean") not in live and (k[:-5] + ".running_mean_source") not in live:
            return "%s conv bias" % sec
        return "%s bn %s" % (sec, "gamma" if k.endswith("weight") else "beta")
    groups = defaultdict(list)
    allr = []
    for k in deeplab_ref.parameter_keys(o32):
        eh = model_cases.l2rel(live[k].grad, o64[k].grad)
        ef = model_cases.l2rel(o32[k].grad, o64[k].grad)
        groups[klass(k)].append((eh, ef, k))
        allr.append(math.log(max(eh, 1e-7) / max(ef, 1e-7)))
    emit("")
    emit("gradients vs the fp64 oracle, S=%d B=%d (trimmed L2): geometric means per parameter class" % (S, B))
    emit("%-26s %4s %10s %10s %7s   worst tensor" % ("class", "n", "hip", "oracle32", "ratio"))
    for g, v in sorted(groups.items()):
        gh = math.exp(sum(math.log(max(a, 1e-7)) for a, _, _ in v) / len(v))
        gs = math.exp(sum(math.log(max(b, 1e-7)) for _, b, _ in v) / len(v))
        w = max(v, key=lambda t: t[0] / max(t[1], 1e-7))
        emit("%-26s %4d %10.2e %10.2e %7.2f   %s (%.1e vs %.1e)" % (g, len(v), gh, gs, gh / gs, w[2], w[0], w[1]))
    emit("ALL: geometric-mean ratio %.3f over %d tensors" % (math.exp(sum(allr) / len(allr)), len(allr)))
    bad, gmean = model_cases.grads_ok({k: (a, b) for v in groups.values() for a, b, k in v})
    emit("model_cases.grads_ok: %d outside the per-tensor bound, gmean %.3f" % (len(bad), gmean))


if __name__ == "__main__":
    main()
