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
    """fp64 plain torch = truth.  A LeakyReLU whose pre-activation sits within rounding of 0 legitimately takes the
    other slope in another fp32 evaluation order (each flip moves one element by 0.8 x its gradient), so gradients
    are held to a multiple of the distance the fp32 plain-torch run itself has from the fp64 one (trimmed L2)."""
    import model_cases
    torch.manual_seed(5)
    mine = getattr(GAN, kind)()
    torch.manual_seed(5)
    r64 = getattr(gan_ref, kind)().double()
    torch.manual_seed(5)
    r32 = getattr(gan_ref, kind)()
    mine.to(DEV)
    cin = r64.conv1.weight.shape[1]
    x = torch.rand(B, cin, size, size, generator=torch.Generator().manual_seed(1))
    g = torch.randn((B, 1) + tuple(r32(x[:1]).shape[2:]), generator=torch.Generator().manual_seed(2))
    xa, xb, xc = x.to(DEV).requires_grad_(True), x.double().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb, yc = mine(xa), r64(xb), r32(xc)
    assert ya.shape == yb.shape
    assert _rel(ya, yb) < 1e-4
    ya.backward(g.to(DEV))
    yb.backward(g.double())
    yc.backward(g)
    pairs = [(xa.grad, xb.grad, xc.grad)] + [(getattr(mine, "conv%d" % i).weight.grad, getattr(r64, "conv%d" % i).weight.grad,
                                                getattr(r32, "conv%d" % i).weight.grad) for i in range(1, 6)]
    for i, (a, t, f) in enumerate(pairs):
        e, floor = model_cases.l2rel(a, t), model_cases.l2rel(f, t)
        assert e < 10.0 * floor + 1e-4, (i, e, floor)


def test_no_grad_forward_and_detached_input():
    d = GAN.BoundaryDiscriminator().to(DEV)
    x = torch.rand(2, 1, 64, 64, device=DEV)
    with torch.no_grad():
        y = d(x)
    assert not y.requires_grad and y.shape == (2, 1, 3, 3)
    y = d(x)                                  # detached input: weight gradients only
    y.mean().backward()
    assert d.conv1.weight.grad is not None and torch.isfinite(d.conv5.weight.grad).all()


def test_forward_follows_fused_optimizer_steps():
    """torch's fused multi-tensor optimizers update parameters without bumping their autograd version: the kernel-side
    weight layouts must not be cached on it (regression: a discriminator frozen at its initial weights)."""
    torch.manual_seed(7)
    d = GAN.BoundaryDiscriminator().to(DEV)
    torch.manual_seed(7)
    ref = gan_ref.BoundaryDiscriminator()
    opt = torch.optim.SGD(d.parameters(), lr=0.1, momentum=0.9, fused=True)
    opt_ref = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    x = torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(3))
    for _ in range(3):
        for o, m, xx in ((opt, d, x.to(DEV)), (opt_ref, ref, x)):
            o.zero_grad()
            m(xx).square().mean().backward()
            o.step()
    with torch.no_grad():
        ya, yb = d(x.to(DEV)), ref(x)
    assert (yb - gan_ref.BoundaryDiscriminator()(x)).abs().max() > 0        # the weights really moved
    assert _rel(ya, yb) < 2e-3


def test_full_size_batch_independence():
    """16 x 2 x 512 x 512 (the bench shape): a discriminator has no batch coupling, so logits and input gradients of
    image k must not depend on the batch it is evaluated in (large-P tiles vs small-P tiles of the same kernels)."""
    torch.manual_seed(9)
    d = GAN.UncertaintyDiscriminator().to(DEV)
    x = torch.rand(16, 2, 512, 512, generator=torch.Generator().manual_seed(13)).to(DEV)
    xa = x.clone().requires_grad_(True)
    ya = d(xa)
    ya.square().sum().backward()
    for k in (0, 13):
        xb = x[k:k + 2].clone().requires_grad_(True)
        yb = d(xb)
        yb.square().sum().backward()
        import model_cases
        assert _rel(ya[k:k + 2], yb) < 1e-4
        # trimmed L2: a LeakyReLU whose pre-activation sits within rounding of 0 may take the other slope under the
        # other tiling and moves a small patch of the input gradient
        assert model_cases.l2rel(xa.grad[k:k + 2], xb.grad) < 1e-3
