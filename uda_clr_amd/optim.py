"""Multi-tensor Adam on the hand-written kernel (``uda_adam_step``; SURVEY.md 2.1 last row).

The entry script builds ``torch.optim.Adam(model_gen.parameters(), lr, betas=(0.9, 0.99))`` (train_use_fix_initial.py:210-214),
may load a checkpointed state into it (:228-256) and hands it to the Trainer, which saves ``optimizer.state_dict()`` into
its checkpoints (Trainer_prototype_full.py:176-190).  ``FlatAdam`` keeps that object as the owner of the hyper-parameters
(``param_groups``: the LR rule writes ``lr`` there) and of the checkpoint layout (``state_dict`` / ``load_state_dict`` are the
torch optimizer's own, per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), but

  * moves the parameters of each group into ONE flat fp32 buffer (every ``p.data`` becomes a view of it, 16-byte aligned),
  * keeps ``exp_avg`` / ``exp_avg_sq`` as views of two more flat buffers, and
  * performs the update of a whole group with one launch of ``uda_adam_step`` on the flat buffers (the gradients are gathered
    into a fourth flat buffer with one multi-tensor copy).

A configuration the kernel does not cover (weight decay, amsgrad, maximize, non-fp32 / non-device parameters) is left to the
torch optimizer.  The first step in which some parameter has no gradient hands the optimizer back to torch's own ``step`` for
good (on the same view tensors): torch skips such a parameter and keeps a step count PER parameter, which one shared count
cannot represent.  Every step checks (host-side pointer compares) that each ``p.data`` still is its view of the flat buffer
and re-flattens otherwise (a ``.to()`` or a manual ``p.data = ...`` after the first step would otherwise leave the optimizer
updating an orphaned buffer).
"""
from __future__ import annotations

import torch


def _eligible(opt):
    if type(opt) is not torch.optim.Adam:
        return False
    for g in opt.param_groups:
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return False
        if isinstance(g["lr"], torch.Tensor):
            return False
        if not all(p.is_cuda and p.dtype == torch.float32 for p in g["params"]):
            return False
    return True


def take_over(optimizer, kernels=None):
    """``FlatAdam`` around a stock Adam over device fp32 parameters; anything else is returned unchanged."""
    if isinstance(optimizer, FlatAdam) or not _eligible(optimizer):
        return optimizer
    return FlatAdam(optimizer, kernels)


class _Group:
    __slots__ = ("params", "offsets", "n", "flat_p", "flat_g", "flat_m", "flat_v", "grad_views")


class FlatAdam:
    def __init__(self, inner, kernels=None):
        self.inner = inner
        self._K = kernels
        self._groups = None
        self._step = 0
        self._torch_steps = False      # True once a step met a parameter without gradient: torch's own rule from then on

    # ---------------------------------------------------------------- torch.optim.Optimizer surface the trainers / scripts use
    @property
    def param_groups(self):
        return self.inner.param_groups

    @property
    def state(self):
        return self.inner.state

    def zero_grad(self, set_to_none=True):
        self.inner.zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        self._sync_steps()
        return self.inner.state_dict()

    def load_state_dict(self, sd):
        self.inner.load_state_dict(sd)
        self._groups = None            # re-flatten from the loaded per-parameter tensors on the next step
        self._step = 0                 # (taken from the loaded state by _build; an empty state starts at 0 again)
        self._torch_steps = False

    # ---------------------------------------------------------------- flat layout
    def _kernels(self):
        if self._K is None:
            from .kernels import HipKernels
            self._K = HipKernels()
        return self._K

    def _build(self):
        self._groups = []
        steps = []
        for g in self.inner.param_groups:
            G = _Group()
            G.params = list(g["params"])
            G.offsets, o = [], 0
            for p in G.params:
                G.offsets.append(o)
                o += (p.numel() + 3) // 4 * 4          # 16-byte aligned slots
            G.n = o
            dev = G.params[0].device
            G.flat_p = torch.zeros(o, dtype=torch.float32, device=dev)
            G.flat_g = torch.zeros(o, dtype=torch.float32, device=dev)
            G.flat_m = torch.zeros(o, dtype=torch.float32, device=dev)
            G.flat_v = torch.zeros(o, dtype=torch.float32, device=dev)
            G.grad_views = []
            with torch.no_grad():
                for p, off in zip(G.params, G.offsets):
                    sl = slice(off, off + p.numel())
                    G.flat_p[sl].copy_(p.detach().reshape(-1))
                    p.data = G.flat_p[sl].view_as(p)
                    st = self.inner.state.get(p)
                    if st:                      # resumed optimizer: adopt its moments
                        G.flat_m[sl].copy_(st["exp_avg"].reshape(-1))
                        G.flat_v[sl].copy_(st["exp_avg_sq"].reshape(-1))
                        steps.append(int(float(st["step"])))
                    self.inner.state[p] = {"step": torch.tensor(0.0), "exp_avg": G.flat_m[sl].view_as(p),
                                           "exp_avg_sq": G.flat_v[sl].view_as(p)}
                    G.grad_views.append(G.flat_g[sl].view_as(p))
            self._groups.append(G)
        if steps:
            if len(set(steps)) != 1:
                raise RuntimeError("FlatAdam: the resumed optimizer state holds different step counts per parameter")
            self._step = steps[0]
        self._sync_steps()

    def _aliased(self):
        """every parameter still is its view of the flat buffer (host-side compares only)"""
        for G in self._groups:
            base = G.flat_p.data_ptr()
            for p, off in zip(G.params, G.offsets):
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32 or p.device != G.flat_p.device:
                    return False
        return True

    def _sync_steps(self):
        if self._groups is None or self._torch_steps:      # (torch's own per-parameter counts are the truth in that mode)
            return
        for G in self._groups:
            for p in G.params:
                st = self.inner.state[p]["step"]
                st.fill_(float(self._step)) if isinstance(st, torch.Tensor) else None

    # ---------------------------------------------------------------- the update
    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdam.step takes no closure")
        if self._torch_steps:
            return self.inner.step()
        if self._groups is None:
            self._build()
        elif not self._aliased():
            self._sync_steps()          # someone rebound a p.data: adopt the current values and moments into fresh flat buffers
            self._build()
        if any(p.grad is None for G in self._groups for p in G.params):
            # some parameter has no gradient: torch's rule skips it and does not advance ITS step count, which the one shared count
            # of the flat update cannot follow - torch's own step (on the same view tensors) from here on
            self._sync_steps()
            self._torch_steps = True
            return self.inner.step()
        self._step += 1
        K = self._kernels()
        for G, g in zip(self._groups, self.inner.param_groups):
            torch._foreach_copy_(G.grad_views, [p.grad for p in G.params])
            b1, b2 = g["betas"]
            K.adam_step(G.flat_p, G.flat_g, G.flat_m, G.flat_v, float(g["lr"]), float(b1), float(b2), float(g["eps"]), self._step)
