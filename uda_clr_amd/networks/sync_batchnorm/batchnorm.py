"""TransNorm2d - parameter / buffer holder of the ``--use_TN`` normalisation (compute: uda_clr_amd.engine).

Same class name, constructor defaults and state-dict entries as the reference's
``networks/sync_batchnorm/batchnorm.py:263-324,389-521`` (``BatchNorm2d``, which there is TransNorm, not
torch's batch norm): ``weight`` (ones), ``bias`` (zeros), and the per-domain running statistics
``running_mean_source / running_var_source / running_mean_target / running_var_target`` plus
``num_batches_tracked``, registered in that order (DeepLab then has 899 state-dict keys instead of 675).

What the layer computes (engine.GeneratorEngine with ``transnorm=True``, kernels ``uda_tn_finalize`` /
``uda_tn_eval_coeffs``): in training mode the first N//2 images of a batch are normalised with their own batch
statistics, the remaining images with theirs, and both halves are scaled per channel by 1 + alpha, where alpha
measures how close the two halves' mean/std ratios are (no gradient through alpha); in eval mode the target
running statistics normalise and alpha comes from the running statistics.
"""
import torch
import torch.nn as nn


class BatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        if not (affine and track_running_stats) or eps != 1e-5 or momentum != 0.1:
            raise NotImplementedError("TransNorm2d is built with the defaults the reference networks use")
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean_source", torch.zeros(num_features))
        self.register_buffer("running_var_source", torch.ones(num_features))
        self.register_buffer("running_mean_target", torch.zeros(num_features))
        self.register_buffer("running_var_target", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, *a, **k):  # pragma: no cover - never used
        raise RuntimeError("parameter holder: compute runs in uda_clr_amd.engine, not in submodules")

    def extra_repr(self):
        return "%d, eps=%g, momentum=%g (TransNorm2d)" % (self.num_features, self.eps, self.momentum)
