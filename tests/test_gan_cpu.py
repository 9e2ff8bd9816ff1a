"""not-gpu: orchestration of uda_clr_amd.gan_engine (space-to-depth grids, weight layouts, the hand-written
backward) driven by the tests' torch statement of the kernels, against the plain-torch discriminators."""
import pytest
import torch

from kernel_spec import SpecKernels
from oracle import gan_ref
from uda_clr_amd.gan_engine import PatchDiscriminatorEngine
from uda_clr_amd.networks import GAN


def _pair(kind, seed=3):
    torch.manual_seed(seed)
    mine = getattr(GAN, kind)()
    torch.manual_seed(seed)
    ref = getattr(gan_ref, kind)()
    mine._engine_override = PatchDiscriminatorEngine(SpecKernels())
    return mine, ref


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize("kind,size", [("BoundaryDiscriminator", 64), ("UncertaintyDiscriminator", 64),
                                       ("UncertaintyDiscriminator", 50)])
def test_forward_backward_match_plain_torch(kind, size):
    mine, ref = _pair(kind)
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert torch.equal(v, mine.state_dict()[k]), k           # same seeded initialisation
    cin = ref.conv1.weight.shape[1]
    x = torch.rand(2, cin, size, size, generator=torch.Generator().manual_seed(1))
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = mine(xa), ref(xb)
    assert ya.shape == yb.shape
    assert _rel(ya, yb) < 1e-5
    g = torch.randn(yb.shape, generator=torch.Generator().manual_seed(2))
    ya.backward(g)
    yb.backward(g)
    assert _rel(xa.grad, xb.grad) < 1e-4
    for i in range(1, 6):
        a, b = getattr(mine, "conv%d" % i).weight.grad, getattr(ref, "conv%d" % i).weight.grad
        assert _rel(a, b) < 1e-4, i


def test_grad_mode_switches_and_graph_reuse():
    """One forward graph, two backward passes (the training loop's generator step, then its discriminator step)."""
    mine, ref = _pair("BoundaryDiscriminator")
    x = torch.rand(2, 1, 48, 48, generator=torch.Generator().manual_seed(4))
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = mine(xa), ref(xb)
    params = list(mine.parameters())
    mine.grad_mode = "input"
    ya.sum().backward(inputs=[xa], retain_graph=True)
    yb.sum().backward(inputs=[xb], retain_graph=True)
    assert all(p.grad is None for p in params) and _rel(xa.grad, xb.grad) < 1e-4
    mine.grad_mode = "weights"
    (ya * ya).sum().backward(inputs=params)
    (yb * yb).sum().backward(inputs=list(ref.parameters()))
    for p, q in zip(params, ref.parameters()):
        assert _rel(p.grad, q.grad) < 1e-4


def test_cpu_input_fails_loudly():
    d = GAN.BoundaryDiscriminator()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d(torch.zeros(2, 1, 32, 32))


@pytest.mark.parametrize("kind,pre", [("BoundaryDiscriminator", "sigmoid"), ("UncertaintyDiscriminator", "entropy")])
def test_fused_logit_maps_match_the_reference_expressions(kind, pre):
    """forward(logits, pre=...) == the discriminator on sigmoid(logits) / on -sigmoid * log(sigmoid + 1e-7)
    (Trainer_prototype_full.py:452-454), values and gradients into the logits and the weights."""
    mine, ref = _pair(kind)
    cin = ref.conv1.weight.shape[1]
    x = 2.0 * torch.randn(2, cin, 48, 48, generator=torch.Generator().manual_seed(7))     # a seed without a LeakyReLU pre-activation within rounding of 0
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = mine(xa, pre=pre)
    s = torch.sigmoid(xb)
    yb = ref(s if pre == "sigmoid" else -1.0 * s * torch.log(s + 1e-7))
    assert _rel(ya, yb) < 1e-5
    g = torch.randn(yb.shape, generator=torch.Generator().manual_seed(2))
    ya.backward(g)
    yb.backward(g)
    assert _rel(xa.grad, xb.grad) < 1e-4
    for i in range(1, 6):
        assert _rel(getattr(mine, "conv%d" % i).weight.grad, getattr(ref, "conv%d" % i).weight.grad) < 1e-4, i
