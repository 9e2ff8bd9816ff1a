"""Debug aid (TEST INFRASTRUCTURE): run every kernel call through BOTH the HIP binding and the
torch statement on cloned arguments, compare every tensor argument afterwards and log the calls
whose results differ.  Lets a whole-model mismatch be traced to the first divergent launch."""
from __future__ import annotations

import copy

import torch

from uda_clr_amd.acts import Act, BNRec


def _clone(v, memo=None):
    """Deep copy that maps aliased tensors (same storage window) onto ONE clone, so in-place
    conventions (out is dU, addend is out) survive in the cloned call."""
    memo = {} if memo is None else memo
    if isinstance(v, torch.Tensor):
        key = (v.data_ptr(), tuple(v.shape), tuple(v.stride()), v.dtype)
        if key not in memo:
            memo[key] = v.clone()
        return memo[key]
    if isinstance(v, Act):
        bn = None if v.bn is None else BNRec(v.bn.key, _clone(v.bn.mean, memo), _clone(v.bn.invstd, memo), v.bn.count,
                                             v.bn.q1_border)
        return Act(_clone(v.x, memo), v.N, v.H, v.W, _clone(v.scale, memo), _clone(v.shift, memo), v.act,
                   _clone(v.mask, memo), v.mask_scale, bn)
    return copy.copy(v)


def _tensors(name, v):
    if isinstance(v, torch.Tensor):
        yield name, v
    elif isinstance(v, Act):
        yield name + ".x", v.x


class DualKernels:
    def __init__(self, hip, spec, tol=1e-4, log=print):
        self.hip, self.spec, self.tol, self.log = hip, spec, tol, log
        self.calls, self.bad = 0, []

    def __getattr__(self, meth):
        h, s = getattr(self.hip, meth), getattr(self.spec, meth)

        def call(*args, **kw):
            self.calls += 1
            if meth.startswith("relayout"):
                a, b = h(*args, **kw), s(*args, **kw)
                self._cmp(meth, "ret", a, b, args)
                return a
            memo = {}
            cargs = [_clone(a, memo) for a in args]
            ckw = {k: _clone(v, memo) for k, v in kw.items()}
            s(*cargs, **ckw)
            r = h(*args, **kw)
            for i, (a, c) in enumerate(zip(args, cargs)):
                for n, t in _tensors("arg%d" % i, a):
                    self._cmp(meth, n, t, dict(_tensors("arg%d" % i, c))[n], args)
            for k in kw:
                for n, t in _tensors(k, kw[k]):
                    self._cmp(meth, n, t, dict(_tensors(k, ckw[k]))[n], args)
            return r
        return call

    def _cmp(self, meth, name, a, b, args):
        if a.dtype == torch.uint8:
            return
        a64, b64 = a.detach().double(), b.detach().double()
        den = max(b64.abs().max().item(), 1e-30)
        err = float("inf") if not torch.isfinite(a64).all() else (a64 - b64).abs().max().item() / den
        if err > self.tol:
            shapes = [tuple(x.shape) if isinstance(x, torch.Tensor) else (tuple(x.x.shape) if isinstance(x, Act) else x) for x in args]
            msg = "call %d %s[%s] rel err %.3e  args %s" % (self.calls, meth, name, err, shapes)
            if a.dim() == 2 and err != float("inf"):
                colerr = (a64 - b64).abs().max(0).values / den
                badc = (colerr > self.tol).nonzero().flatten()
                msg += "  bad cols %d/%d first %s last %s" % (badc.numel(), a.shape[1], badc[:4].tolist(), badc[-4:].tolist())
            self.bad.append(msg)
            self.log("  MISMATCH " + msg)
