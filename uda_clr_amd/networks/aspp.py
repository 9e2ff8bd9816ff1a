"""ASPP head - parameter tree only (compute: uda_clr_amd.engine).  Key names and construction /
initialisation order follow the reference's ``networks/aspp.py:7-95``."""
import torch.nn as nn

from ._tree import Holder, child, conv, kaiming_bn_init


class ASPP(Holder):
    def __init__(self, backbone, output_stride, BatchNorm):
        super().__init__()
        if backbone not in ('mobilenet', 'resnet'):
            raise NotImplementedError("ASPP is built for the mobilenet (320) and resnet (2048) backbones")
        cin = 320 if backbone == 'mobilenet' else 2048           # aspp.py:37-42
        if output_stride not in (16, 8):
            raise NotImplementedError
        dils = (1, 6, 12, 18) if output_stride == 16 else (1, 12, 24, 36)
        bn_t = (nn.BatchNorm2d, BatchNorm)
        for j, d in enumerate(dils, start=1):
            k = 1 if j == 1 else 3
            br = child(self, "aspp%d" % j)
            child(br, "atrous_conv", conv(cin, 256, k, 1, 0 if j == 1 else d, d))
            child(br, "bn", BatchNorm(256))
            kaiming_bn_init(br.modules(), bn_t)          # each branch initialises itself first
        child(self, "global_avg_pool.1", conv(cin, 256, 1))
        child(self, "global_avg_pool.2", BatchNorm(256))
        child(self, "conv1", conv(1280, 256, 1))
        child(self, "bn1", BatchNorm(256))
        kaiming_bn_init(self.modules(), bn_t)            # ... then the head re-initialises all


def build_aspp(backbone, output_stride, BatchNorm):
    return ASPP(backbone, output_stride, BatchNorm)
