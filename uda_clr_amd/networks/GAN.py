"""Patch discriminators of the adversarial branch - drop-in for ``networks/GAN.py:86-148``.

Five 4x4 stride-2 pad-2 bias-free convolutions (1|2 -> 64 -> 128 -> 256 -> 512 -> 1) with LeakyReLU(0.2)
between them, weights ~ N(0, 0.02); same constructor, ``state_dict`` keys (``conv1.weight`` .. ``conv5.weight``)
and seeded initialisation as the reference.  The ``nn.Conv2d`` children only hold the parameters: forward and
backward are ONE autograd node executed by ``uda_clr_amd.gan_engine`` on the HIP kernels.
There is no CPU path: a CPU input (or a missing ``libuda_clr_hip.so``) raises.
"""
import torch
import torch.nn as nn

from ..gan_engine import PatchDiscriminatorEngine


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, need, pre_op, *weights):
        engine = module._engine_for(x)
        out, ectx = engine.forward(x.contiguous().float(), weights, need, pre_op, w_share=module._wshare)
        ctx.engine, ctx.ectx, ctx.module, ctx.weights = engine, ectx, module, weights
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.ectx is None:
            raise RuntimeError("discriminator forward ran without gradient bookkeeping")
        mode = ctx.module.grad_mode
        need_x = ctx.needs_input_grad[1] and mode in ("auto", "input")
        need_w = any(ctx.needs_input_grad[4:]) and mode in ("auto", "weights")
        dx, dws = ctx.engine.backward(ctx.ectx, gout, ctx.weights, need_x, need_w)
        if mode != "input":
            ctx.ectx = None
        if dws is None:
            dws = [None] * len(ctx.weights)
        return (None, dx, None, None) + tuple(dws)


class _PatchDiscriminator(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        widths = [in_channels, 64, 128, 256, 512, 1]
        for i in range(5):
            setattr(self, "conv%d" % (i + 1), nn.Conv2d(widths[i], widths[i + 1], kernel_size=4, stride=2, padding=2, bias=False))
        self.leakyrelu = nn.LeakyReLU(negative_slope=0.2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0.0, 0.02)
        self._engine = None
        self._engine_override = None      # tests only: an engine bound to their torch kernel spec
        self._wshare = None               # inside shared_weight_layouts(): z-space weight operands of the current parameters
        # which gradients a backward pass through this module produces: "auto" = whatever autograd needs;
        # "input" / "weights" let a training loop that back-propagates twice through ONE forward graph (generator
        # step, then discriminator step) skip the half it is not going to use (autograd cannot prune inside a
        # custom node).  The saved activations are released after an "auto" or "weights" pass.
        self.grad_mode = "auto"

    def shared_weight_layouts(self):
        """Context manager: the forwards inside it share one set of kernel-side weight operands (the caller promises that the
        parameters do not change inside the block; see DeepLab.shared_weight_layouts)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            keep, self._wshare = self._wshare, {}
            try:
                yield self
            finally:
                self._wshare = keep
        return scope()

    def _engine_for(self, x):
        if self._engine_override is not None:
            return self._engine_override
        if not x.is_cuda:
            raise RuntimeError("uda_clr_amd discriminators compute only on the MI355X HIP kernels; got a %s tensor "
                               "(there is no CPU fallback)" % x.device)
        if self._engine is None:
            from ..kernels import HipKernels
            self._engine = PatchDiscriminatorEngine(HipKernels())
        return self._engine

    fused_pre = True      # forward(x, pre=...) accepts generator logits (see below); stock modules have no such attribute
    _PRE = {None: 0, "sigmoid": 1, "entropy": 2}

    def forward(self, x, pre=None):
        """``pre`` (extension of the reference's ``forward(x)``): "sigmoid" / "entropy" = x holds generator logits and the
        discriminator reads sigmoid(x) / -sigmoid(x) * log(sigmoid(x) + 1e-7) (the two maps Trainer_prototype_full.py:452-454
        builds with elementwise torch ops); the map is formed inside the first layer's space-to-depth pass and its derivative
        inside the adjoint pass."""
        if x.dim() != 4 or x.shape[1] != self.conv1.weight.shape[1]:
            raise ValueError("expected an [N, %d, H, W] batch" % self.conv1.weight.shape[1])
        weights = [getattr(self, "conv%d" % i).weight for i in range(1, 6)]
        need = torch.is_grad_enabled() and (x.requires_grad or any(w.requires_grad for w in weights))
        return _DiscFn.apply(self, x, need, self._PRE[pre], *weights)


class UncertaintyDiscriminator(_PatchDiscriminator):
    def __init__(self):
        super().__init__(2)


class BoundaryDiscriminator(_PatchDiscriminator):
    def __init__(self):
        super().__init__(1)
