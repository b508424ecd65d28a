"""-m gpu: Discriminator.forward (discriminator.py:60-70) as a differentiable nn.Module call -- what user code that
back-propagates through ``loss_f.discriminator(z)`` needs (the reference's own FactorKLoss.call_optimize does:
losses.py:261-306, two forwards of the same batch size, both back-propagated, gradients accumulating)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from disvae_amd.models.discriminator import Discriminator


def _ref_forward(sd, z):
    x = z
    for i in range(1, 7):
        x = F.linear(x, sd["lin%d.weight" % i], sd["lin%d.bias" % i])
        if i < 6:
            x = F.leaky_relu(x, 0.2)
    return x


@pytest.mark.parametrize("M,D", [(37, 10), (128, 10), (9, 6)])
def test_discriminator_forward_is_differentiable(M, D):
    torch.manual_seed(5)
    disc = Discriminator(latent_dim=D).to(DEV)
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in disc.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    z1, zp = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
    z1d, zpd = dev(z1).requires_grad_(True), dev(zp).requires_grad_(True)
    zeros, ones = torch.zeros(M, dtype=torch.long), torch.ones(M, dtype=torch.long)

    def total(d_z, d_zp, zr, on):
        return 0.5 * (F.cross_entropy(d_z, zr) + F.cross_entropy(d_zp, on)) + 0.3 * (d_z[:, 0] - d_z[:, 1]).mean()

    d_z, d_zp = disc(z1d), disc(zpd)                 # two live forwards of the same batch size
    loss = total(d_z, d_zp, zeros.to(DEV), ones.to(DEV))
    loss.backward()
    z1r, zpr = z1.double().requires_grad_(True), zp.double().requires_grad_(True)
    ref = total(_ref_forward(sd, z1r), _ref_forward(sd, zpr), zeros, ones)
    ref.backward()
    check(d_z, _ref_forward(sd, z1r), rtol=1e-5, atol_rel=2e-6, what="D(z)")
    check(loss, ref, rtol=1e-5, what="loss through D")
    # gradients: LeakyReLU' jumps from 0.2 to 1 at 0, so a hidden unit whose pre-activation is within fp32 rounding of zero
    # takes the other slope than in fp64 and moves ITS sample's dL/dz by up to ~1e-3 of the tensor's scale (M x 5000 units:
    # a handful at M = 128, none at M = 9 / 37, which therefore keep the tight bound)
    tol = dict(rtol=1e-4, atol_rel=2e-5) if M < 100 else dict(rtol=1e-3, atol_rel=2e-3)
    check(z1d.grad, z1r.grad, what="dL/dz1", **tol)
    check(zpd.grad, zpr.grad, what="dL/dz_perm", **tol)
    for k, p in disc.named_parameters():
        check(p.grad, sd[k].grad, what="dL/d" + k, **tol)
    # a second backward accumulates (zero_grad(set_to_none=False) semantics of torch.optim)
    before = {k: p.grad.clone() for k, p in disc.named_parameters()}
    total(disc(z1d), disc(zpd), zeros.to(DEV), ones.to(DEV)).backward()
    for k, p in disc.named_parameters():
        check(p.grad, 2 * before[k].cpu().double(), rtol=1e-5, atol_rel=2e-6, what="accumulated dL/d" + k)
    # no_grad inference still works and equals the differentiable forward
    with torch.no_grad():
        assert torch.equal(disc(z1d), d_z.detach())
