"""ResNet-101 backbone - parameter tree only (compute: uda_clr_amd.engine).

Key names and construction order of the reference's ``networks/backbone/resnet.py:45-111``
(``conv1, bn1, layer1..layer3`` of Bottlenecks [3, 4, 23], ``layer4`` = multi-grid unit [1, 2, 4]);
``resnet_plan`` is the per-block geometry the engine executes.
"""
import math

import torch.nn as nn

from .._tree import Holder, child, conv


def resnet_plan(output_stride=16, layers=(3, 4, 23)):
    """[(prefix, inplanes, planes, stride, dilation, has_downsample)] (resnet.py:47-70, 72-111)."""
    if output_stride == 16:
        strides, dils = (1, 2, 2, 1), (1, 1, 1, 2)
    elif output_stride == 8:
        strides, dils = (1, 2, 1, 1), (1, 1, 2, 4)
    else:
        raise NotImplementedError
    plan, inp = [], 64
    for li, (planes, n) in enumerate(zip((64, 128, 256), layers), start=1):
        for b in range(n):
            s = strides[li - 1] if b == 0 else 1
            plan.append(("layer%d.%d" % (li, b), inp, planes, s, dils[li - 1], b == 0 and (s != 1 or inp != 4 * planes)))
            inp = 4 * planes
    for b, mg in enumerate((1, 2, 4)):
        s = strides[3] if b == 0 else 1
        plan.append(("layer4.%d" % b, inp, 512, s, mg * dils[3], b == 0 and (s != 1 or inp != 2048)))
        inp = 2048
    return plan


class ResNet(Holder):
    def __init__(self, output_stride, BatchNorm, pretrained=True):
        super().__init__()
        BatchNorm = BatchNorm or nn.BatchNorm2d
        self.output_stride = output_stride
        child(self, "conv1", conv(3, 64, 7, 2, 3))
        child(self, "bn1", BatchNorm(64))
        for pre, inp, planes, stride, dil, has_ds in resnet_plan(output_stride):
            if has_ds:            # the reference builds the shortcut before the block's own convs
                ds0, ds1 = conv(inp, 4 * planes, 1, stride), BatchNorm(4 * planes)
            child(self, pre + ".conv1", conv(inp, planes, 1))
            child(self, pre + ".bn1", BatchNorm(planes))
            child(self, pre + ".conv2", conv(planes, planes, 3, stride, dil, dil))
            child(self, pre + ".bn2", BatchNorm(planes))
            child(self, pre + ".conv3", conv(planes, 4 * planes, 1))
            child(self, pre + ".bn3", BatchNorm(4 * planes))
            if has_ds:
                child(self, pre + ".downsample.0", ds0)
                child(self, pre + ".downsample.1", ds1)
        for m in self.modules():                                   # resnet.py:126-136
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, (nn.BatchNorm2d, BatchNorm)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if pretrained:
            self._load_pretrained_model()

    def _load_pretrained_model(self):
        """The reference downloads torchvision's ImageNet ResNet-101 (resnet.py:138-146); there is no
        network here.  Set ``UDA_CLR_RESNET101_PTH`` to that state dict to load it the same
        key-filtered way; unset means seeded random initialisation."""
        import os
        import torch
        path = os.environ.get("UDA_CLR_RESNET101_PTH")
        if not path:
            return
        pre = torch.load(path, map_location="cpu", weights_only=True)
        own = self.state_dict()
        own.update({k: v for k, v in pre.items() if k in own})
        self.load_state_dict(own)


def ResNet101(output_stride, BatchNorm, pretrained=True):
    return ResNet(output_stride, BatchNorm, pretrained=pretrained)
