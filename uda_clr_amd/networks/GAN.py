"""Patch discriminators of the adversarial branch - drop-in for ``networks/GAN.py:86-148``.

Five 4x4 stride-2 pad-2 bias-free convolutions (1|2 -> 64 -> 128 -> 256 -> 512 -> 1) with
LeakyReLU(0.2) between them, weights ~ N(0, 0.02).  They run on stock PyTorch-ROCm: SURVEY.md 8f-1
ranks their native kernels as the first row AFTER the generator hot path.
"""
import torch.nn as nn


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

    def forward(self, x):
        for i in range(1, 5):
            x = self.leakyrelu(getattr(self, "conv%d" % i)(x))
        return self.conv5(x)


class UncertaintyDiscriminator(_PatchDiscriminator):
    def __init__(self):
        super().__init__(2)


class BoundaryDiscriminator(_PatchDiscriminator):
    def __init__(self):
        super().__init__(1)
