"""Plain-PyTorch restatement of the reference's patch discriminators (networks/GAN.py:86-148).

TEST INFRASTRUCTURE (see oracle/__init__.py): five Conv2d(4, stride 2, pad 2, bias=False) with LeakyReLU(0.2)
between them, weights ~ N(0, 0.02) drawn in construction order.  Pinned through tests/golden/trainer_proto.json:
the rows (seg, adv, D_same, D_diff, intra, inter) the reference's own Trainer_prototype_full loop wrote with the
reference's own discriminators (seeded 1338) are reproduced by tests/test_trainers_cpu.py with these copies.
"""
import torch.nn as nn
import torch.nn.functional as F


class PatchDiscriminator(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        widths = [in_channels, 64, 128, 256, 512, 1]
        for i in range(5):
            setattr(self, "conv%d" % (i + 1), nn.Conv2d(widths[i], widths[i + 1], 4, 2, 2, bias=False))
        for i in range(5):
            getattr(self, "conv%d" % (i + 1)).weight.data.normal_(0.0, 0.02)

    def forward(self, x):
        for i in range(1, 5):
            x = F.leaky_relu(getattr(self, "conv%d" % i)(x), 0.2)
        return self.conv5(x)


def BoundaryDiscriminator():
    return PatchDiscriminator(1)


def UncertaintyDiscriminator():
    return PatchDiscriminator(2)
