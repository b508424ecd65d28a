"""-m gpu: the configuration bench.py TIMES, pinned.

bench.py steps the model with ``torch.optim.Adam(model.flat_parameters(), lr, fused=True)`` (the parameter arena as equal
8192-element chunks) where a drop-in user writes ``optim.Adam(model.parameters(), lr)`` (main.py:208).  Here:
  * both optimizer constructions, from the same state, on the same batches and injected noise, at the bench batch, for
    3 steps: after every step the parameters equal each other and torch's CPU Adam fed with the engine's gradients to <= 1
    ulp (the three implementations differ in operation order only).  The parameters are
    re-synchronised after each comparison: Adam turns an ulp-level difference of a near-zero gradient entry into an
    O(lr) difference of the update, so un-synchronised trajectories measure that amplification, not the optimizers;
  * the 10-step training trajectory of SURVEY.md 8c at the bench batch sizes (btcvae 64x64x3 B = 1024, factor 64x64x1
    tensor 256) with injected noise, stepped with the timed optimizer construction: loss rtol 1e-3 against the oracle
    following its own trajectory (training.py:137-164, losses.py:243-313,356-391); step >= 2 exercises arena reuse, the
    cached gradient views and Adam state at these sizes;
  * the ADVICE r2 hazard: a flat-chunk optimizer combined with the autograd-compatible path raises instead of stepping
    with stale gradients."""
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib  # noqa
from oracle import disvae_oracle as O
from disvae_amd.models.vae import init_specific_model
from disvae_amd.models.losses import get_loss_f

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25, betaB_G=1000,
          factor_G=6.4, latent_dim=10, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)


def _timed_optimizer(model, lr):
    import bench
    return bench.make_optimizer(model, lr)          # the very function the timed loop uses


def _native(loss, img, seed, n_data, lr, lr_disc, flat):
    torch.manual_seed(seed)
    model = init_specific_model("Burgess", img, 10).to(DEV)
    opt = _timed_optimizer(model, lr) if flat else torch.optim.Adam(model.parameters(), lr=lr)
    loss_f = get_loss_f(loss, n_data=n_data, device=torch.device(DEV), lr_disc=lr_disc, **HP)
    loss_f.replay = None
    model.train()
    return model, opt, loss_f


def test_flat_fused_adam_equals_adam_over_the_state_dict_views():
    img, B, seed, n_data, lr = (3, 64, 64), 1024, 1234, 202599, 5e-4
    ma, oa, la = _native("btcvae", img, seed, n_data, lr, 1e-5, True)       # the timed configuration
    mb, ob, lb = _native("btcvae", img, seed, n_data, lr, 1e-5, False)      # the drop-in configuration
    assert torch.equal(ma.arena.flat, mb.arena.flat)
    cpu = [p.detach().cpu().clone().requires_grad_(True) for p in mb.parameters()]
    oc = torch.optim.Adam(cpu, lr=lr)                                       # the oracle's optimizer (torch CPU Adam)
    gen = torch.Generator().manual_seed(seed + 1)
    for step in range(3):
        data = dev(torch.rand((B,) + img, generator=gen))
        eps = dev(torch.randn(B, 10, generator=gen))
        l1 = la.fused_step(data, ma, oa, None, eps=eps)
        l2 = lb.fused_step(data, mb, ob, None, eps=eps)
        # same kernels, same inputs, same parameters, fixed-order reductions: bit-identical gradients
        assert torch.equal(ma.arena.grad, mb.arena.grad), "step %d" % step
        assert l1.item() == l2.item()
        for pc, pb in zip(cpu, mb.parameters()):
            pc.grad = pb.grad.detach().cpu().clone()
        oc.step()
        # <= 1 ulp of the parameter (2^-23 relative; the largest weights, decoder.lin1's, are ~0.8: 6e-8 absolute) or 2e-6
        # of an update (lr = 5e-4 -> 1e-9), whichever is larger
        ulp = lambda a, b: ((a - b).abs() / (b.abs() * 2.0 ** -23 + 1e-9)).max().item()
        da = ulp(ma.arena.flat, mb.arena.flat)
        assert da <= 1.0, "step %d: flat fused Adam vs Adam over the views: %.2f ulp" % (step, da)
        for pc, (k, pb) in zip(cpu, mb.named_parameters()):
            d = ulp(pb.detach().cpu(), pc.detach())
            assert d <= 1.0, "step %d %s: GPU Adam vs torch CPU Adam on the same gradients: %.2f ulp" % (step, k, d)
        # the alignment padding of the arena stays zero under the flat optimizer (zero gradient -> zero update)
        used = torch.zeros_like(ma.arena.flat, dtype=torch.bool)
        for k, (off, n) in ma.arena.offsets.items():
            used[off:off + n] = True
        assert torch.all(ma.arena.flat[~used] == 0)
        # identical parameters again before the next step (see the module docstring); the Adam moments keep their own
        # (ulp-level different) histories
        mb.arena.flat.copy_(ma.arena.flat)
        with torch.no_grad():
            for pc, pa in zip(cpu, ma.parameters()):
                pc.copy_(pa.detach().cpu())


def _update_agreement(p0, p_engine, p_oracle, what, lr, steps):
    """The accumulated parameter update of the engine against the oracle's.  Element-wise bounds cannot be tight here: Adam
    turns an ulp-level difference of a near-zero gradient entry into an O(lr) difference of that entry's update (see the
    module docstring), but a wrong gradient of a whole layer (a missed term, a wrong mask, a stale buffer) rotates or
    rescales the update vector -- so: cosine >= 0.99 between the two update vectors, norms within 3 %, and at most 2 % of
    the entries further apart than lr (one flipped step)."""
    u = (p_engine.double() - p0.double()).flatten()
    v = (p_oracle.double() - p0.double()).flatten()
    nv = v.norm().item()
    if nv < 1e-9:                                    # an untouched tensor stays untouched
        assert u.norm().item() < 1e-9, what
        return
    cos = float(torch.dot(u, v) / (u.norm() * v.norm() + 1e-300))
    assert cos >= 0.99, "%s: update direction differs from the oracle's: cosine %.5f" % (what, cos)
    assert abs(u.norm().item() - nv) <= 0.03 * nv, "%s: update norm %.4e vs the oracle's %.4e" % (what, u.norm().item(), nv)
    far = ((u - v).abs() > lr).double().mean().item()
    assert far <= 0.02, "%s: %.2f %% of the entries differ from the oracle's by more than lr after %d steps" % (what, 100 * far, steps)


@pytest.mark.parametrize("name,loss,img,B,n_data,lr,lr_disc,steps", [
    ("btcvae_celeba", "btcvae", (3, 64, 64), 1024, 202599, 5e-4, 1e-5, 10),
    ("factor_dsprites", "factor", (1, 64, 64), 256, 737280, 1e-4, 1e-4, 10),
    ("btcvae_b128", "btcvae", (3, 64, 64), 128, 202599, 5e-4, 1e-5, 10),   # the per-GPU batch of the 8-GPU headline config
    # BASELINE configs[4]: the two-optimizer step with the 16 MB discriminator arena at tensor 2048 (the CPU oracle takes
    # ~25 s per iteration at this size: 3 steps)
    ("factor_celeba", "factor", (3, 64, 64), 2048, 202599, 1e-4, 1e-5, 3),
])
def test_ten_step_trajectory_at_bench_batch(name, loss, img, B, n_data, lr, lr_disc, steps):
    seed = 1234
    model, opt, loss_f = _native(loss, img, seed, n_data, lr, lr_disc, True)
    if name == "btcvae_b128":
        loss_f.replay = "plan"                       # what `auto` selects at this size: the recorded launch plan
    torch.manual_seed(seed)
    params = O.init_vae_params(img, 10)
    dparams = O.init_disc_params(10) if loss == "factor" else None
    hp = dict(HP, n_data=n_data, lr_disc=lr_disc)
    orc = O.OracleTrainer(loss, hp, img, 10, lr=lr, lr_disc=lr_disc, steps_anneal=HP["reg_anneal"], params=params,
                          dparams=dparams)
    p0 = {k: v.detach().clone() for k, v in orc.params.items()}
    d0 = {k: v.detach().clone() for k, v in orc.dparams.items()} if loss == "factor" else {}
    for k, p in model.named_parameters():            # both start from the same point (bit-identical initialisation)
        assert torch.equal(p.detach().cpu(), p0[k]), k
    gen = torch.Generator().manual_seed(seed + 1)
    data_d = torch.empty((B,) + img, device=DEV)     # the batch keeps its address (plans are keyed on it)
    got, want = [], []
    for step in range(steps):
        data = torch.rand((B,) + img, generator=gen)
        data_d.copy_(data)
        if loss == "factor":
            Bh = B // 2
            eps1, eps2 = torch.randn(Bh, 10, generator=gen), torch.randn(Bh, 10, generator=gen)
            perms = torch.stack([torch.randperm(Bh, generator=gen) for _ in range(10)])
            ref, _ = orc.train_iteration(data, eps=eps1, eps2=eps2, perms=list(perms))
            out = loss_f.call_optimize(data_d, model, opt, None, noise=(dev(eps1), dev(eps2), perms))
        else:
            eps = torch.randn(B, 10, generator=gen)
            ref, _ = orc.train_iteration(data, eps=eps)
            out = loss_f.fused_step(data_d, model, opt, None, eps=dev(eps))
        got.append(out.item())
        want.append(ref)
    np.testing.assert_allclose(got, want, rtol=1e-3, err_msg="%s: loss trajectory" % name)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, err_msg="%s: first loss" % name)
    assert want[-1] < want[0]                        # the workload actually trains
    for k, p in model.named_parameters():            # the accumulated Adam updates agree with the oracle's
        _update_agreement(p0[k], p.detach().cpu(), orc.params[k].detach(), "%s: %s" % (name, k), lr, steps)
    if loss == "factor":
        for k, p in loss_f.discriminator.named_parameters():
            ko = k if k in d0 else k.replace("_", ".")
            _update_agreement(d0[ko], p.detach().cpu(), orc.dparams[ko].detach(), "%s: discriminator %s" % (name, k), lr_disc, steps)


def test_flat_optimizer_with_the_autograd_path_is_refused():
    """autograd delivers gradients to the layer Parameters only; an optimizer built on flat_parameters() would
    step with whatever its chunks' .grad held.  The autograd-compatible backward refuses that combination."""
    img, B = (1, 64, 64), 4
    model, opt, loss_f = _native("btcvae", img, 3, 737280, 5e-4, 1e-4, True)
    data = dev(torch.rand((B,) + img))
    loss_f.fused_step(data, model, opt, None)        # flat chunks now carry arena gradients
    recon, latent_dist, z = model(data)
    loss = loss_f(data, recon, latent_dist, True, None, latent_sample=z)
    with pytest.raises(_lib.DvaeHipError, match="flat_parameters"):
        loss.backward()
