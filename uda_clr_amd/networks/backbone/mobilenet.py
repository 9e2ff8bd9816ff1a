"""MobileNetV2 backbone - parameter tree only (compute: uda_clr_amd.engine).

Mirrors the construction order and key names of the reference's
``networks/backbone/mobilenet.py:70-122`` (stem ``features.0``, 17 inverted-residual blocks
``features.1 .. features.17`` with members ``conv.<idx>``, aliases ``low_level_features`` =
``features[0:4]`` and ``high_level_features`` = ``features[4:]``).
"""
import torch.nn as nn

from ...engine import block_plan
from .._tree import Holder, child, conv, kaiming_bn_init


class MobileNetV2(Holder):
    def __init__(self, output_stride=8, BatchNorm=None, width_mult=1., pretrained=True):
        super().__init__()
        if width_mult != 1.:
            raise NotImplementedError("only width_mult=1 is built")
        BatchNorm = BatchNorm or nn.BatchNorm2d
        self.output_stride = output_stride
        feats = Holder()
        self.add_module("features", feats)
        child(feats, "0.0", conv(3, 32, 3, 2, 1))
        child(feats, "0.1", BatchNorm(32))
        for i, (inp, oup, stride, dil, t) in enumerate(block_plan(output_stride), start=1):
            hid, idx = round(inp * t), 0
            if t != 1:
                child(feats, "%d.conv.0" % i, conv(inp, hid, 1))
                child(feats, "%d.conv.1" % i, BatchNorm(hid))
                idx = 3
            child(feats, "%d.conv.%d" % (i, idx), conv(hid, hid, 3, stride, 0, dil, hid))
            child(feats, "%d.conv.%d" % (i, idx + 1), BatchNorm(hid))
            child(feats, "%d.conv.%d" % (i, idx + 3), conv(hid, oup, 1))
            child(feats, "%d.conv.%d" % (i, idx + 4), BatchNorm(oup))
        kaiming_bn_init(self.modules(), (nn.BatchNorm2d, BatchNorm))
        if pretrained:
            self._load_pretrained_model()
        lo, hi = Holder(), Holder()
        for k in range(len(feats)):
            (lo if k < 4 else hi).add_module(str(k), feats[k])   # Sequential slices keep their keys
        self.add_module("low_level_features", lo)
        self.add_module("high_level_features", hi)

    def _load_pretrained_model(self):
        """The reference reads a hard-coded absolute path (mobilenet.py:124-133).  Set
        ``UDA_CLR_MOBILENET_PTH`` to a MobileNetV2 state dict to load it the same key-filtered way;
        unset means seeded random initialisation."""
        import os
        import torch
        path = os.environ.get("UDA_CLR_MOBILENET_PTH")
        if not path:
            return
        pre = torch.load(path, map_location="cpu", weights_only=True)
        own = self.state_dict()
        own.update({k: v for k, v in pre.items() if k in own})
        self.load_state_dict(own)
