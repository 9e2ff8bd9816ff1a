"""-m gpu: the native patch discriminators (one autograd node on the HIP kernels) against the plain-torch
restatement of networks/GAN.py:86-148 evaluated on the CPU (fp32 and fp64)."""
import pytest
import torch

from oracle import gan_ref
from uda_clr_amd.networks import GAN

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rel(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize("kind,B,size", [("BoundaryDiscriminator", 2, 128), ("UncertaintyDiscriminator", 2, 128),
                                         ("UncertaintyDiscriminator", 3, 512)])
def test_discriminator_matches_plain_torch(kind, B, size):
    torch.manual_seed(5)
    mine = getattr(GAN, kind)()
    torch.manual_seed(5)
    ref = getattr(gan_ref, kind)().double()
    mine.to(DEV)
    cin = ref.conv1.weight.shape[1]
    x = torch.rand(B, cin, size, size, generator=torch.Generator().manual_seed(1))
    xa, xb = x.to(DEV).requires_grad_(True), x.double().requires_grad_(True)
    ya, yb = mine(xa), ref(xb)
    assert ya.shape == yb.shape
    assert _rel(ya, yb) < 1e-4
    g = torch.randn(yb.shape, generator=torch.Generator().manual_seed(2))
    ya.backward(g.to(DEV))
    yb.backward(g.double())
    assert _rel(xa.grad, xb.grad) < 1e-3
    for i in range(1, 6):
        assert _rel(getattr(mine, "conv%d" % i).weight.grad, getattr(ref, "conv%d" % i).weight.grad) < 1e-3, i


def test_no_grad_forward_and_detached_input():
    d = GAN.BoundaryDiscriminator().to(DEV)
    x = torch.rand(2, 1, 64, 64, device=DEV)
    with torch.no_grad():
        y = d(x)
    assert not y.requires_grad and y.shape == (2, 1, 5, 5)
    y = d(x)                                  # detached input: weight gradients only
    y.mean().backward()
    assert d.conv1.weight.grad is not None and torch.isfinite(d.conv5.weight.grad).all()
