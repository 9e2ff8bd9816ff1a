"""DeepLabV3+ generator (MobileNetV2 or ResNet-101 backbone) - the drop-in for the reference's
``networks/deeplabv3.py``.

Same constructor, same ``state_dict`` keys (675 entries incl. the aliased backbone slices; 899 with
``sync_bn=False``, the TransNorm model of ``--use_TN``), same
seeded initialisation, same 7-tuple from ``forward`` (deeplabv3.py:32-41):

    x1, x2, feature, x_bu_feature, x_feature, x1_before, x2_before

but the whole forward/backward is ONE autograd node executed by ``uda_clr_amd.engine`` on the
hand-written gfx950 kernels.  There is no CPU path: a CPU input (or a missing
``libuda_clr_hip.so``) raises.
"""
import torch
import torch.nn as nn

from ..engine import GeneratorEngine
from ._tree import Holder
from .aspp import build_aspp
from .backbone import build_backbone
from .decoder import build_decoder
from .sync_batchnorm.batchnorm import BatchNorm2d as TransNorm2d

_BN_TYPES = (nn.BatchNorm2d, TransNorm2d)

__all__ = ["DeepLab"]


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, need_grad, keys, *tensors):
        ctx.set_materialize_grads(False)      # unused outputs arrive as None, not as zero tensors to be added
        engine = module._engine_for(x)
        params = module._flat_state()
        masks, module._next_masks = module._next_masks, None
        bn_tr = module._bn_training()
        outs, ectx = engine.forward(params, x, module.training, need_grad, masks, bn_training=bn_tr, w_share=module._wshare)
        ctx.engine, ctx.ectx, ctx.keys, ctx.module = engine, ectx, keys, module
        if ectx is not None and module.training and bn_tr:      # (the MC fast path replays batch statistics: training-mode BN only)
            module._remember(x, ectx)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        if ctx.ectx is None:
            raise RuntimeError("generator forward ran without gradient bookkeeping")
        G = ctx.engine.backward(ctx.ectx, grads)
        ctx.module._forget(ctx.ectx)          # its activations are consumed; parameters are about to change
        ctx.ectx = None
        out = [G.get(k) for k in ctx.keys]
        if ctx.module._fuse_accum:
            # inside fused_grad_accumulation(): autograd would sum the gradients the generator passes of ONE backward run send to the
            # same parameter in its input buffers, one add launch per parameter (~190).  Instead the first node of a run remembers
            # the tensors it hands over and every later node adds its gradients INTO them with one multi-tensor launch and hands
            # over None (an absent contribution).  The references are dropped right away so that AccumulateGrad still holds the last
            # one and keeps the tensor instead of cloning it; an engine callback clears them at the end of the run in any case.
            module = ctx.module
            stash = module._accum_stash
            if stash is None:
                module._accum_stash = {k: g for k, g in zip(ctx.keys, out) if g is not None}
                torch.autograd.Variable._execution_engine.queue_callback(lambda: setattr(module, "_accum_stash", None))
            else:
                idx = [i for i, k in enumerate(ctx.keys) if out[i] is not None and k in stash]
                if idx:
                    torch._foreach_add_([stash[ctx.keys[i]] for i in idx], [out[i] for i in idx])
                    for i in idx:
                        out[i] = None
                module._accum_stash = None
        return (None, None, None, None) + tuple(out)


class DeepLab(Holder):
    def __init__(self, backbone='resnet', output_stride=16, num_classes=21,
                 sync_bn=True, freeze_bn=False, method='prototype'):
        super().__init__()
        if num_classes != 2:
            raise NotImplementedError("the fused heads are built for num_classes=2 (cup, disc)")
        # deeplabv3.py:17-23: sync_bn=True is plain nn.BatchNorm2d, sync_bn=False (--use_TN) is TransNorm
        self.transnorm = not sync_bn
        BatchNorm = TransNorm2d if self.transnorm else nn.BatchNorm2d
        self.output_stride = output_stride
        self.backbone_name = backbone
        self.backbone = build_backbone(backbone, output_stride, BatchNorm)
        self.aspp = build_aspp(backbone, output_stride, BatchNorm)
        self.decoder = build_decoder(num_classes, backbone, method, BatchNorm)
        self._engine = None
        self._recent = []                 # (input data_ptr, shape, input version, generation, engine ctx) of the last training forwards
        self._generation = 0              # bumped whenever parameters / buffers / mode may have changed (note_params_changed)
        self._engine_override = None      # tests only: an engine bound to their torch kernel spec
        self._next_masks = None           # tests only: injected dropout keep-masks for one forward
        self._wshare = None               # inside shared_weight_layouts(): {(key, kind): kernel-side layout} of the current parameters
        self._fuse_accum = False          # inside fused_grad_accumulation(): the passes of one backward run sum their gradients themselves
        self._accum_stash = None          # ... {key: gradient tensor the first pass of the current run handed to autograd}
        if freeze_bn:
            self.freeze_bn()

    # ---------------------------------------------------------------- reference API
    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, _BN_TYPES):
                m.eval()

    def _lr_params(self, roots):
        for root in roots:
            for m in root.modules():
                if isinstance(m, (nn.Conv2d,) + _BN_TYPES):
                    for p in m.parameters(recurse=False):
                        if p.requires_grad:
                            yield p

    def get_1x_lr_params(self):
        return self._lr_params([self.backbone.features if self.backbone_name == 'mobilenet' else self.backbone])

    def get_10x_lr_params(self):
        return self._lr_params([self.aspp, self.decoder])

    # ---------------------------------------------------------------- engine plumbing
    def set_dropout_masks(self, masks):
        """Parity tests: keep-masks ({site: uint8 NCHW}) used by the next training forward."""
        self._next_masks = masks

    def set_dropout_seed(self, seed):
        """Philox key of the dropout masks (data-parallel trainers give every rank its own)."""
        self._dropout_seed = int(seed)
        if self._engine is not None:
            self._engine.seed = self._dropout_seed

    def pop_nonfinite(self):
        """0-dim bool device tensor (or None if no pass ran): a NaN / Inf went through a BatchNorm statistic of a forward or
        backward pass since the last call (see GeneratorEngine.nonfinite).  The trainers fold it into their one host sync."""
        eng = self._engine_override or self._engine
        return None if eng is None else eng.pop_nonfinite()

    def _remember(self, x, ectx):
        self._recent = [(x.data_ptr(), tuple(x.shape), x._version, self._generation, ectx)] + self._recent[:1]

    def _forget(self, ectx):
        self._recent = [r for r in self._recent if r[4] is not ectx]

    def shared_weight_layouts(self):
        """Context manager: inside it every forward / backward of this module builds each kernel-side weight layout (relayouts,
        bf16x3 packed rows) ONCE instead of once per pass.  The caller promises that no parameter changes inside the block
        (torch's fused optimizer steps leave no trace a forward could check); ``note_params_changed`` / ``train`` /
        ``load_state_dict`` drop the shared layouts.  ``Trainer_prototype_full`` holds it around the generator passes of a step
        (target forward, source forward, MC passes and the backward up to ``optim_gen.step()``)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            keep, self._wshare = self._wshare, {}
            try:
                yield self
            finally:
                self._wshare = keep
        return scope()

    def fused_grad_accumulation(self):
        """Context manager around a ``backward()`` that runs SEVERAL generator passes' backward nodes (source + target of one
        step): the later node adds its parameter gradients into the tensors the first one handed to autograd, with one multi-tensor
        launch, and hands over ``None`` (same sums as autograd's per-parameter input-buffer adds, bit for bit).  Only around
        ``backward()`` / ``autograd.grad`` calls that want the SUM over the passes; ``Trainer_prototype_full`` holds it around
        ``loss_all.backward``."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            keep, self._fuse_accum = self._fuse_accum, True
            try:
                yield self
            finally:
                self._fuse_accum = keep
                self._accum_stash = None
        return scope()

    def note_params_changed(self):
        """Invalidate the activations kept for ``mc_dropout_logits``: call after anything that changes parameters or buffers
        outside this module's sight (torch's fused optimizer steps do not bump tensor versions).  The bundled trainers call
        it after every optimizer step; ``train()`` / ``eval()`` / ``load_state_dict`` do it themselves."""
        self._generation += 1
        self._recent = []
        if self._wshare is not None:
            self._wshare = {}              # (a fresh dict: contexts of earlier passes keep theirs for their backward)

    def train(self, mode=True):
        self.note_params_changed()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.note_params_changed()
        return super().load_state_dict(*args, **kwargs)

    def mc_dropout_logits(self, x, passes=4, reps=2, masks=None):
        """Segmentation logits of ``passes`` no-grad training-mode forwards on ``x.repeat(reps,1,1,1)``
        (Trainer_prototype_full.py:358-368: T = passes*reps stochastic predictions per image), as one
        [passes*reps*N, 2, H, W] tensor.  When ``x`` is the unmodified input of a recent grad-mode training
        forward of this module and ``note_params_changed`` has not been called since (the trainers call it after
        every optimizer step), the deterministic pre-dropout activations of that forward are reused and only the dropout-dependent tail is recomputed
        (``GeneratorEngine.mc_forward``); otherwise the passes run as plain forwards."""
        assert self.training, "stochastic passes need training mode (dropout + batch statistics)"
        if self.transnorm and reps == 2 and self._bn_training():
            # TransNorm splits the REPEATED batch into its two copies of x (identical statistics: alpha = 1, gain 2), not into the
            # halves of x the grad-mode forward saw: nothing of that forward can be reused, but the deterministic part of the
            # repeated batch is still one forward of x (GeneratorEngine.forward(repeat_prefix=True)), shared by all passes
            with torch.no_grad():
                engine = self._engine_for(x)
                _, ectx = engine.forward(self._flat_state(), x.contiguous().float(), True, True, None, repeat_prefix=True,
                                         w_share=self._wshare)
                return engine.mc_forward(ectx, reps, passes, masks=masks)
        for ptr, shape, version, generation, ectx in ([] if self.transnorm else self._recent):
            # the SAME tensor (address, shape, not written since) under the SAME parameters (no optimizer step, mode change or
            # state load since): anything else falls through to plain forwards
            if ptr == x.data_ptr() and shape == tuple(x.shape) and version == x._version and generation == self._generation:
                with torch.no_grad():
                    return self._engine_for(x).mc_forward(ectx, reps, passes, masks=masks)
        outs = []
        with torch.no_grad():
            xr = x.repeat(reps, 1, 1, 1)
            for ps in range(passes):
                if masks is not None:
                    self.set_dropout_masks(masks[ps])
                outs.append(self(xr)[0])
        return torch.cat(outs, 0)

    def _flat_state(self):
        sd = {}
        bb = ("backbone.features", self.backbone.features) if self.backbone_name == 'mobilenet' \
            else ("backbone", self.backbone)
        for name, mod in (bb, ("aspp", self.aspp), ("decoder", self.decoder)):
            for k, v in mod.named_parameters(prefix=name):
                sd[k] = v
            for k, v in mod.named_buffers(prefix=name):
                sd[k] = v
        return sd

    def _bn_training(self):
        flags = {m.training for m in self.modules() if isinstance(m, _BN_TYPES)}
        if len(flags) != 1:
            raise NotImplementedError("mixed train/eval BatchNorm layers are not built")
        flag = flags.pop()
        if flag and not self.training:
            raise NotImplementedError("training-mode BatchNorm layers inside an eval-mode model are not built")
        return flag          # False while self.training: freeze_bn() (deeplabv3.py:43-50) - frozen statistics, live dropout

    def _engine_for(self, x):
        if self._engine_override is not None:
            return self._engine_override
        if not x.is_cuda:
            raise RuntimeError("uda_clr_amd.DeepLab computes only on the MI355X HIP kernels; got a "
                               "%s tensor (there is no CPU fallback)" % x.device)
        if self._engine is None:
            from ..kernels import HipKernels
            self._engine = GeneratorEngine(HipKernels(), self.output_stride, backbone=self.backbone_name,
                                           transnorm=self.transnorm, seed=getattr(self, "_dropout_seed", 1337))
        return self._engine

    def forward(self, input):
        if input.dim() != 4 or input.shape[1] != 3:
            raise ValueError("expected an [N, 3, H, W] image batch")
        x = input.contiguous().float()
        state = self._flat_state()
        keys = tuple(k for k, v in state.items() if isinstance(v, nn.Parameter))
        need_grad = torch.is_grad_enabled() and any(state[k].requires_grad for k in keys)
        return _GeneratorFn.apply(self, x, need_grad, keys, *[state[k] for k in keys])
