"""Parameter-tree helpers.

The product network keeps ``torch.nn.Conv2d`` / ``torch.nn.BatchNorm2d`` objects ONLY as named
parameter holders: they are created in the reference's construction order so that a seeded
``torch.manual_seed(s); DeepLab(...)`` yields bit-identical initial weights and identical
``state_dict`` keys; their own ``forward`` is never called - the compute runs
in ``uda_clr_amd.engine`` on the HIP kernels.
"""
import torch.nn as nn


class Holder(nn.Module):
    """A container whose children are addressed by state-dict path components."""

    def forward(self, *a, **k):  # pragma: no cover - never used
        raise RuntimeError("parameter holder: compute runs in uda_clr_amd.engine, not in submodules")

    def __getitem__(self, idx):
        return self._modules[str(idx)]

    def __len__(self):
        return len(self._modules)


def child(root: nn.Module, path: str, leaf: nn.Module = None) -> nn.Module:
    """Walk/create ``Holder`` containers along a dotted path and attach ``leaf`` at its end."""
    parts = path.split(".")
    node = root
    for p in parts[:-1] if leaf is not None else parts:
        if p not in node._modules:
            node.add_module(p, Holder())
        node = node._modules[p]
    if leaf is not None:
        node.add_module(parts[-1], leaf)
        return leaf
    return node


def conv(ci, co, k, stride=1, pad=0, dil=1, groups=1, bias=False):
    return nn.Conv2d(ci, co, k, stride, pad, dil, groups, bias)


def kaiming_bn_init(modules, bn_types):
    """The reference's ``_init_weight`` / ``_initialize_weights`` pass (aspp.py:24-33,
    decoder.py:58-72, mobilenet.py:135-151): He-normal conv weights, BN weight 1 / bias 0."""
    for m in modules:
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, bn_types):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
