"""Decoder (segmentation + boundary heads) - parameter tree only (compute: uda_clr_amd.engine).
Key names / order follow the reference's ``networks/decoder.py:7-74``."""
import torch.nn as nn

from ._tree import Holder, child, conv, kaiming_bn_init


class Decoder(Holder):
    def __init__(self, num_classes, backbone, method, BatchNorm):
        super().__init__()
        if backbone not in ('mobilenet', 'resnet'):
            raise NotImplementedError("decoder is built for the mobilenet (24) and resnet (256) low-level widths")
        low = 24 if backbone == 'mobilenet' else 256             # decoder.py:11-16
        self.method = method
        child(self, "conv1", conv(low, 48, 1))
        child(self, "bn1", BatchNorm(48))
        child(self, "last_conv.0", BatchNorm(305))
        child(self, "last_conv.3", conv(305, num_classes, 1, bias=True))
        child(self, "last_conv_boundary.0", conv(304, 256, 3, 1, 1))
        child(self, "last_conv_boundary.1", BatchNorm(256))
        child(self, "last_conv_boundary.4", conv(256, 256, 3, 1, 1))
        child(self, "last_conv_boundary.5", BatchNorm(256))
        child(self, "last_conv_boundary.8", conv(256, 1, 1, bias=True))
        kaiming_bn_init(self.modules(), (nn.BatchNorm2d, BatchNorm))


def build_decoder(num_classes, backbone, method, BatchNorm):
    return Decoder(num_classes, backbone, method, BatchNorm)
