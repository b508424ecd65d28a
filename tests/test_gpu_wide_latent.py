"""-m gpu: latent dimensions ABOVE 16 (main.py:81 takes any --latent-dim; disvae/models/losses.py:452-480, 523-544 and
utils/math.py:8-73 are dimension-agnostic).  The fused kernels cover 1..16; above that the SAME C-ABI entry points run the
run-time-D kernels of csrc/latent_wide.hip on the "wide" layouts of include/dvae_hip.h and the engine runs the FC layers one
launch each.  Everything here goes through the C-ABI and is held to the oracle at the tolerances the narrow path is held to
(tests/test_gpu_kernels.py, tests/test_gpu_step.py)."""
from collections import defaultdict
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib  # noqa
from oracle import disvae_oracle as O
from disvae_amd.models.vae import init_specific_model
from disvae_amd.models.losses import get_loss_f
from disvae_amd.training import Trainer

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25,
          betaB_G=1000, factor_G=6.4, latent_dim=10, lr_disc=1e-4, btcvae_A=1, btcvae_B=6.4,
          btcvae_G=1)


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_layout_helpers_match_the_header_macros():
    """_lib.rowstats_stride / btcvae_tmp_floats / npack / nscal / kl0 == the DVAE_*_D macros of include/dvae_hip.h (evaluated
    by the C preprocessor in tests/test_cabi_symbols.py on the CPU side; here: the values the kernels were compiled with)."""
    for D in (1, 10, 16):
        assert (_lib.rowstats_stride(D), _lib.npack(D), _lib.nscal(D), _lib.kl0(D)) == (32, 32, 32, _lib.S_KL0)
        assert _lib.btcvae_tmp_floats(100, 40, D) == 3 * D * 100
    for D in (17, 28, 29, 64):
        assert _lib.rowstats_stride(D) == (D + 4 + 3) // 4 * 4 and _lib.rowstats_stride(D) >= 4 + D
        assert (_lib.npack(D), _lib.nscal(D), _lib.kl0(D)) == (32 + D, 32 + D, 32)
        assert _lib.btcvae_tmp_floats(100, 40, D) == 3 * D * 100 + 40 * 100


@pytest.mark.parametrize("B,D", [(2, 17), (200, 24), (1500, 40), (64, 100)])
def test_reparam_kl_wide(B, D):
    ml = _rand(B, 2 * D, seed=1, scale=1.5)
    eps = torch.randn(B, D, generator=torch.Generator().manual_seed(2))
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / B
    mld, epsd, coefd = dev(ml), dev(eps), dev(coef)
    mu, lv, z = (torch.empty(B, D, device=DEV) for _ in range(3))
    kl = torch.full((D + 5,), 7.0, device=DEV)
    call("dvae_reparam_kl_fwd", ptr(mld), ptr(epsd), ptr(mu), ptr(lv), ptr(z), ptr(kl), ptr(coefd), B, D, stream())
    m_ref, l_ref = ml.view(B, D, 2).unbind(-1)
    assert torch.equal(mu.cpu(), m_ref) and torch.equal(lv.cpu(), l_ref)
    check(z, O.reparameterize(m_ref.double(), l_ref.double(), eps.double()), what="z")
    check(kl[:D], O.kl_normal_loss(m_ref.double(), l_ref.double())[1], what="kl_dim")
    assert bool((kl[D:] == 7.0).all())                       # D final values, nothing else
    call("dvae_reparam_kl_fwd", ptr(mld), None, ptr(mu), ptr(lv), ptr(z), None, None, B, D, stream())
    assert torch.equal(z.cpu(), m_ref)                       # eval mode: z = mean (vae.py:69-71)
    # the partial-block form does not exist above DVAE_MAX_D: refused, not truncated
    with pytest.raises(_lib.DvaeHipError):
        call("dvae_reparam_kl_fwd", ptr(mld), ptr(epsd), ptr(mu), ptr(lv), ptr(z), ptr(kl), None, B, D, stream())
    # backward (the narrow kernel: elementwise in D)
    dz, dmx, dlx = _rand(B, D, seed=3), _rand(B, D, seed=4), _rand(B, D, seed=5)
    scal = torch.zeros(_lib.NSCAL); scal[_lib.S_KLW] = 2.5
    dml = torch.empty(B, 2 * D, device=DEV)
    call("dvae_reparam_kl_bwd", ptr(dev(dz)), None, None, ptr(dev(dmx)), ptr(dev(dlx)), ptr(dev(m_ref)), ptr(dev(l_ref)), ptr(epsd),
         ptr(dev(scal)), ptr(coefd), ptr(dml), B, D, stream())
    mr, lr = m_ref.double().requires_grad_(True), l_ref.double().requires_grad_(True)
    zz = O.reparameterize(mr, lr, eps.double())
    obj = (zz * dz.double()).sum() + (mr * dmx.double()).sum() + (lr * dlx.double()).sum() + 2.5 * O.kl_normal_loss(mr, lr)[0]
    obj.backward()
    check(dml, torch.stack((mr.grad, lr.grad), -1).reshape(B, 2 * D), what="dml")


@pytest.mark.parametrize("B,n_data,mss,D", [(4, 100, True, 17), (64, 202599, True, 32), (130, 5000, True, 29),
                                            (300, 737280, False, 20), (256, 737280, True, 64), (1024, 202599, True, 24),
                                            (70, 5000, True, 100)])
def test_btcvae_fwd_bwd_wide(B, n_data, mss, D):
    """The estimator's forward and gradient (losses.py:523-544, math.py:8-73) at run-time D vs the fp64 oracle, whole batch
    and row-sharded (the data-parallel use: every call owns its `tmp`, which above 16 also holds the joint log-densities of
    the call's rows)."""
    from disvae_amd.utils.math import log_importance_weights
    g = torch.Generator().manual_seed(B)
    mu = torch.randn(B, D, generator=g)
    lv = torch.randn(B, D, generator=g) * 0.7 - 0.5
    eps = torch.randn(B, D, generator=g)
    z = mu + torch.exp(0.5 * lv) * eps
    lw = torch.zeros(4); lw[:3] = log_importance_weights(B, n_data)
    rstride = _lib.rowstats_stride(D)
    zd, mud, lvd, lwd = dev(z), dev(mu), dev(lv), dev(lw)

    def fwd(row0, rows):
        rs = torch.full((rows, rstride), 7.0, device=DEV)
        tmp = torch.empty(_lib.btcvae_tmp_floats(B, rows, D), device=DEV)
        call("dvae_btcvae_fwd", ptr(zd), ptr(mud), ptr(lvd), B, D, row0, rows, int(mss), ptr(lwd), ptr(tmp), ptr(rs), stream())
        return rs, tmp

    rs, tmp = fwd(0, B)
    ref = O.btcvae_log_densities(z.double(), mu.double(), lv.double(), n_data, mss)
    for k, nm in enumerate(["log_pz", "log_qz", "log_prod_qzi", "log_q_zCx"]):
        check(rs[:, k], ref[k], rtol=2e-6, atol_rel=2e-6, what=nm)
    half = B // 2
    rsa, tmpa = fwd(0, half)
    rsb, tmpb = fwd(half, B - half)
    assert torch.equal(rsa.cpu()[:, :4 + D], rs[:half].cpu()[:, :4 + D])
    assert torch.equal(rsb.cpu()[:, :4 + D], rs[half:].cpu()[:, :4 + D])
    alpha, beta, gamma, anneal = 1.0, 6.4, 1.5, 0.37
    coef = torch.zeros(_lib.NCOEF)
    coef[_lib.C_ALPHA], coef[_lib.C_BETA], coef[_lib.C_GAMMA], coef[_lib.C_ANNEAL] = alpha, beta, gamma, anneal
    coefd = dev(coef)

    def bwd(row0, rows, rs_, tmp_):
        dz, dmu, dlv = torch.empty(rows, D, device=DEV), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
        call("dvae_btcvae_bwd", ptr(zd), ptr(mud), ptr(lvd), ptr(rs_), B, D, row0, rows, int(mss), ptr(lwd), ptr(coefd), ptr(tmp_),
             ptr(dz), ptr(dmu), ptr(dlv), stream())
        return dz, dmu, dlv

    dz, dmu, dlv = bwd(0, B, rs, tmp)
    zr, mr, lr = (t.double().requires_grad_(True) for t in (z, mu, lv))
    mi, tc, dw = O.btcvae_terms(zr, mr, lr, n_data, mss)
    (alpha * mi + beta * tc + anneal * gamma * dw).backward()
    check(dz, zr.grad, rtol=2e-4, atol_rel=1e-5, what="dz")
    check(dmu, mr.grad, rtol=2e-4, atol_rel=1e-5, what="dmu")
    check(dlv, lr.grad, rtol=2e-4, atol_rel=1e-5, what="dlv")
    dza, dma, dla = bwd(0, half, rsa, tmpa)
    dzb, dmb, dlb = bwd(half, B - half, rsb, tmpb)
    check(torch.cat((dza, dzb)), zr.grad, rtol=2e-4, atol_rel=1e-5, what="sharded dz")
    check(dma + dmb, mr.grad, rtol=2e-4, atol_rel=1e-5, what="sharded dmu")
    check(dla + dlb, lr.grad, rtol=2e-4, atol_rel=1e-5, what="sharded dlv")


@pytest.mark.parametrize("kind,B,D", [(_lib.LOSS_BETAH, 8, 17), (_lib.LOSS_BETAB, 700, 40), (_lib.LOSS_BTCVAE, 300, 29),
                                      (_lib.LOSS_FACTOR, 2000, 33)])
def test_loss_epilogue_wide(kind, B, D):
    """dvae_loss_epilogue == dvae_loss_pack -> dvae_loss_finalize bit for bit on the wide layouts, and the scalars are the
    plugins' formulas (losses.py:139-153, 186-202, 268-274, 369-389) of the sums handed in."""
    coef = torch.zeros(_lib.NCOEF)
    coef[_lib.C_INV_B], coef[_lib.C_ANNEAL], coef[_lib.C_BETA] = 1.0 / B, 0.3, 4.0
    coef[_lib.C_ALPHA], coef[_lib.C_GAMMA], coef[_lib.C_CAP] = 1.0, 2.0, 7.0
    coefd = dev(coef)
    partials = _rand(_lib.REC_NPART, seed=3).abs()
    kl = _rand(D, seed=6).abs()
    rstride = _lib.rowstats_stride(D)
    rowstats = _rand(B, rstride, seed=4) if kind == _lib.LOSS_BTCVAE else None
    disc = _rand(4, seed=5) if kind == _lib.LOSS_FACTOR else None
    pd, kd = dev(partials), dev(kl)
    rd = dev(rowstats) if rowstats is not None else None
    dd = dev(disc) if disc is not None else None
    out = []
    for fused in (False, True):
        packed, scal = torch.full((_lib.npack(D),), 7.0, device=DEV), torch.zeros(_lib.nscal(D), device=DEV)
        if fused:
            call("dvae_loss_epilogue", kind, ptr(pd), ptr(kd), 0, D, ptr(rd), B if rd is not None else 0, ptr(dd), B, ptr(coefd),
                 ptr(packed), ptr(scal), stream())
        else:
            call("dvae_loss_pack", ptr(pd), ptr(kd), D, ptr(rd), B if rd is not None else 0, ptr(dd), ptr(packed), stream())
            call("dvae_loss_finalize", kind, ptr(packed), D, B, ptr(coefd), ptr(scal), stream())
        out.append((packed.cpu(), scal.cpu()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    packed, scal = out[0]
    assert torch.equal(packed[32:32 + D], kl) and bool((packed[1:17] == 0).all())
    assert torch.equal(scal[32:32 + D], kl)
    rec = partials.double().sum().item() / B
    klsum = kl.double().sum().item()
    np.testing.assert_allclose(scal[_lib.S_REC].item(), rec, rtol=1e-5)
    np.testing.assert_allclose(scal[_lib.S_KL].item(), klsum, rtol=1e-5)
    if kind == _lib.LOSS_BETAH:
        want = rec + 0.3 * 4.0 * klsum
    elif kind == _lib.LOSS_BETAB:
        want = rec + 4.0 * abs(klsum - 7.0)
    elif kind == _lib.LOSS_BTCVAE:
        s = rowstats.double().sum(0)
        mi, tc, dw = (s[3] - s[1]) / B, (s[1] - s[2]) / B, (s[2] - s[0]) / B
        want = rec + (1.0 * mi + 4.0 * tc + 0.3 * 2.0 * dw).item()
        np.testing.assert_allclose(scal[_lib.S_TC].item(), tc.item(), rtol=1e-4, atol=1e-4)
    else:
        want = rec + klsum + 0.3 * 4.0 * disc[0].item() / B
    np.testing.assert_allclose(scal[_lib.S_LOSS].item(), want, rtol=2e-5, atol=1e-4)
    # a non-zero kl_blocks is the narrow partial-block form: refused above DVAE_MAX_D
    with pytest.raises(_lib.DvaeHipError):
        call("dvae_loss_epilogue", kind, ptr(pd), ptr(kd), 3, D, ptr(rd), B if rd is not None else 0, ptr(dd), B, ptr(coefd),
             ptr(dev(torch.zeros(_lib.npack(D)))), ptr(dev(torch.zeros(_lib.nscal(D)))), stream())


def _model(loss, img, D, seed, n_data, lr, **hp_over):
    torch.manual_seed(seed)
    model = init_specific_model("Burgess", img, D)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    hp = dict(HP, latent_dim=D, n_data=n_data, **hp_over)
    loss_f = get_loss_f(loss, device=torch.device(DEV), **hp)
    model.to(DEV).train()
    return model, opt, loss_f, hp


@pytest.mark.parametrize("loss,img,B,D", [("btcvae", (1, 64, 64), 24, 17), ("btcvae", (3, 64, 64), 40, 32), ("VAE", (1, 32, 32), 9, 20),
                                          ("betaB", (3, 64, 64), 6, 48), ("betaH", (1, 64, 64), 300, 24), ("factor", (1, 64, 64), 24, 20),
                                          ("factor", (3, 64, 64), 10, 33)])
def test_training_step_above_16_latents_vs_oracle(loss, img, B, D):
    """One whole training iteration per loss with a wide latent: loss, every logged scalar (kl_loss_0 .. kl_loss_{D-1}
    included), every gradient, vs the fp64 oracle -- the bounds of test_fused_step_other_latent_dims."""
    seed, n_data, lr = 99, 737280, 5e-4
    model, opt, loss_f, hp = _model(loss, img, D, seed, n_data, lr)
    torch.manual_seed(seed)
    p0 = O.init_vae_params(img, D)
    gen = torch.Generator().manual_seed(seed + 1)
    data = torch.rand((B,) + img, generator=gen)
    st = O.LossState(steps_anneal=HP["reg_anneal"])
    c64 = lambda p: O.clone_params(p, dtype=torch.float64, requires_grad=True)
    storer = defaultdict(list)
    if loss == "factor":
        d0 = O.init_disc_params(D)
        Bh = B // 2
        eps1, eps2 = torch.randn(Bh, D, generator=gen), torch.randn(Bh, D, generator=gen)
        perms = torch.stack([torch.randperm(Bh, generator=gen) for _ in range(D)])
        ref_loss, ref_logs, g64, gd64, _ = O.factor_iteration_grads(hp, st, c64(p0), c64(d0), data.double(), eps1.double(),
                                                                    eps2.double(), list(perms))
        out = loss_f.call_optimize(dev(data), model, opt, storer, noise=(dev(eps1), dev(eps2), perms))
        for k, p in loss_f.discriminator.named_parameters():
            check(p.grad, gd64[k], rtol=1e-3, atol_rel=1e-4, what="D=%d disc grad %s" % (D, k))
    else:
        eps = torch.randn(B, D, generator=gen)
        ref_loss, ref_logs, g64, _ = O.train_iteration_grads(loss, hp, st, c64(p0), data.double(), eps.double())
        out = loss_f.fused_step(dev(data), model, opt, storer, eps=dev(eps))
    np.testing.assert_allclose(out.item(), ref_loss.item(), rtol=2e-5)
    assert list(storer.keys()) == list(ref_logs.keys())
    assert "kl_loss_%d" % (D - 1) in storer
    for k in ref_logs:
        np.testing.assert_allclose(storer[k][0], ref_logs[k].item(), rtol=5e-5, atol=1e-6, err_msg=k)
    for k, p in model.named_parameters():
        check(p.grad, g64[k], rtol=1e-3, atol_rel=1e-4, what="D=%d grad %s" % (D, k))
    assert all(torch.isfinite(p).all() for p in model.parameters())      # Adam stepped


def test_reference_style_loop_above_16_latents_matches_the_native_step():
    """training.py:152-158 verbatim (model(x) -> loss_f(...) -> zero_grad -> backward -> step) through the autograd wrappers ==
    the native iteration, latent_dim 24: same launches for the FC layers here (one per layer both ways), so the bound is
    tighter than at 10 latents."""
    img, B, D = (3, 64, 64), 6, 24
    m1, o1, l1, _ = _model("btcvae", img, D, 7, 202599, 5e-4)
    m2, o2, l2, _ = _model("btcvae", img, D, 7, 202599, 5e-4)
    gen = torch.Generator().manual_seed(3)
    data, eps = dev(torch.rand((B,) + img, generator=gen)), dev(torch.randn(B, D, generator=gen))
    st1, st2 = defaultdict(list), defaultdict(list)
    recon, latent_dist, z = m1(data, eps=eps)
    loss = l1(data, recon, latent_dist, m1.training, st1, latent_sample=z)
    o1.zero_grad()
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in m1.named_parameters()}
    o1.step()
    out = l2.fused_step(data, m2, o2, st2, eps=eps)
    np.testing.assert_allclose(loss.item(), out.item(), rtol=1e-5)
    assert list(st1.keys()) == list(st2.keys())
    for k in st1:
        np.testing.assert_allclose(st1[k][0], st2[k][0], rtol=2e-5, atol=1e-6, err_msg=k)
    for k, p in m2.named_parameters():
        check(g1[k], p.grad, rtol=1e-5, atol_rel=4e-6, what="autograd-vs-native " + k)
    m1.eval()
    with torch.no_grad():
        mu, logvar = m1.encoder(data)
        rec = m1.decoder(mu)
        rec2, (mu2, lv2), z2 = m1(data)
    assert mu.shape == (B, D) and torch.equal(mu, mu2) and torch.equal(rec, rec2) and torch.equal(z2, mu2)


@pytest.mark.parametrize("loss", ["btcvae", "factor"])
def test_replayed_plan_above_16_latents_matches_eager(loss):
    """The recorded launch plan (disvae_amd/graph.py) re-issues the wide iteration bit for bit (fresh batch, noise,
    permutations and annealing coefficient every step)."""
    img, B, D = (1, 64, 64), 16, 20
    runs = []
    for replay in (None, "plan"):
        model, opt, loss_f, _ = _model(loss, img, D, 5, 737280, 5e-4)
        loss_f.replay = replay
        gen = torch.Generator().manual_seed(8)
        data = torch.empty((B,) + img, device=DEV)          # the batch keeps its address: plans are keyed on it
        vals = []
        for step in range(5):
            data.copy_(torch.rand((B,) + img, generator=gen))
            if loss == "factor":
                Bh = B // 2
                noise = (torch.randn(Bh, D, generator=gen).to(DEV), torch.randn(Bh, D, generator=gen).to(DEV),
                         torch.stack([torch.randperm(Bh, generator=gen) for _ in range(D)]))
                vals.append(loss_f.call_optimize(data, model, opt, None, noise=noise).item())
            else:
                vals.append(loss_f.fused_step(data, model, opt, None, eps=torch.randn(B, D, generator=gen).to(DEV)).item())
        if replay:
            assert loss_f._graphs.replays >= 2, "the iteration was never replayed"
        runs.append((vals, model.arena.flat.clone()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])


def test_trainer_evaluator_and_checkpoint_with_32_latents(tmp_path):
    """`python main.py ... --latent-dim 32` end to end on a synthetic loader: Trainer epochs (training.py:64-135), the
    Evaluator's test losses (evaluate.py:97-117) with kl_loss_0..31, save_model / load_model (utils/modelIO.py)."""
    import logging
    from disvae_amd.evaluate import Evaluator
    from disvae_amd.utils.modelIO import save_model, load_model
    img, B, D = (1, 32, 32), 16, 32
    model, opt, loss_f, hp = _model("btcvae", img, D, 11, 60000, 5e-4)
    data = [(torch.rand((B,) + img), torch.zeros(B)) for _ in range(3)] + [(torch.rand((5,) + img), torch.zeros(5))]
    tr = Trainer(model, opt, loss_f, device=torch.device(DEV), logger=logging.getLogger("t"), save_dir=str(tmp_path),
                 is_progress_bar=False)
    tr(data, epochs=2, checkpoint_every=10)
    assert loss_f.n_train_steps == 8 and all(torch.isfinite(p).all() for p in model.parameters())
    log = (tmp_path / "train_losses.log").read_text().splitlines()
    assert any(l.startswith("0,kl_loss_31,") for l in log)
    ev = Evaluator(model, loss_f, device=torch.device(DEV), logger=logging.getLogger("e"), save_dir=str(tmp_path),
                   is_progress_bar=False)
    _, losses = ev(data[:2], is_metrics=False, is_losses=True)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = defaultdict(list)
    for x, _ in data[:2]:
        with torch.no_grad():
            recon, (mu, logvar), z = O.vae_forward(params, x, None)
            _, logs, _ = O.single_optimizer_loss("btcvae", hp, O.LossState(steps_anneal=HP["reg_anneal"]), x, recon, mu, logvar, z, False)
        for k, v in logs.items():
            want[k].append(v.item())
    assert set(want) == set(losses)
    for k, v in want.items():
        np.testing.assert_allclose(losses[k], sum(v) / len(v), rtol=5e-5, atol=1e-5, err_msg=k)
    save_model(model, str(tmp_path))
    again = load_model(str(tmp_path), is_gpu=True)
    assert again.latent_dim == D
    again.eval(); model.eval()
    x = data[0][0].to(DEV)
    with torch.no_grad():
        assert torch.equal(again(x)[0], model(x)[0])


def test_latent_entropy_above_16_latents():
    """dvae_latent_entropy (evaluate.py:233-297) takes any D: vs the fp64 oracle at D = 20."""
    N, D, S = 3000, 20, 500
    g = torch.Generator().manual_seed(4)
    mean, logvar = torch.randn(N, D, generator=g), torch.randn(N, D, generator=g) * 0.5 - 1.0
    draw = torch.randperm(N, generator=g)[:S]
    z_ds = mean.index_select(0, draw).contiguous()             # [S, D], handed over as the [D, S] image (evaluate.py:262)
    ws = torch.empty(_lib.lib().dvae_latent_entropy_ws_floats(N, D, S), device=DEV)
    H = torch.empty(D, device=DEV)
    call("dvae_latent_entropy", ptr(dev(z_ds)), ptr(dev(mean)), ptr(dev(logvar)), N, D, S, ptr(ws), ptr(H), stream())
    zv = z_ds.double().view(D, S)
    m, lv = mean.double(), logvar.double()
    ref = torch.empty(D, dtype=torch.double)
    for d in range(D):
        ld = -0.5 * (math.log(2 * math.pi) + lv[:, d, None]) - 0.5 * (zv[d][None, :] - m[:, d, None]) ** 2 * torch.exp(-lv[:, d, None])
        ref[d] = (math.log(N) - torch.logsumexp(ld, 0)).mean()
    check(H, ref, rtol=1e-4, atol_rel=1e-5, what="H")


@pytest.mark.parametrize("name,loss,img,B,steps", [("btcvae_z32_celeba", "btcvae", (3, 64, 64), 6, 2), ("vae_z24_mnist", "VAE", (1, 32, 32), 8, 2)])
def test_trainer_vs_reference_golden_above_16_latents(name, loss, img, B, steps):
    """The REAL reference's recorded numbers at latent dimensions 32 and 24 (tests/golden/make_golden.py --wide-latent ran
    disvae's own Trainer._train_iteration): same seed -> same initial weights; same data and injected eps -> its losses, storer
    scalars (kl_loss_0 .. kl_loss_{D-1}), gradient and parameter digests -- the bounds of test_trainer_vs_reference_golden."""
    from golden_util import load, tensor_digest, assert_digest_close
    g = load(name)
    D = int(g["latent_dim"])
    model, opt, loss_f, _ = _model(loss, img, D, int(g["seed"]), int(g["n_data"]), float(g["lr"]))
    for k, v in model.state_dict().items():   # same seed -> identical weights (sums: thread-count dependent order)
        d = tensor_digest(v)
        np.testing.assert_array_equal(d[2:], g["init_digest/" + k][2:], err_msg=k)
        np.testing.assert_allclose(d[:2], g["init_digest/" + k][:2], rtol=1e-12, err_msg=k)
    gen = torch.Generator().manual_seed(int(g["seed"]) + 1)
    for s in range(steps):
        data = torch.rand((B,) + tuple(img), generator=gen)
        storer = defaultdict(list)
        out = loss_f.fused_step(dev(data), model, opt, storer, eps=dev(torch.from_numpy(g["step%d/randn0" % s])))
        np.testing.assert_allclose(out.item(), g["step%d/loss" % s], rtol=2e-5 if s == 0 else 1e-3)
        for k, v in storer.items():
            np.testing.assert_allclose(v[0], g["step%d/storer/%s" % (s, k)], rtol=5e-5, atol=1e-6, err_msg=k)
        assert len(storer) == len([k for k in g if k.startswith("step%d/storer/" % s)])
        if s == 0:
            assert "kl_loss_%d" % (D - 1) in storer
        gr, ga = (2e-3, 2e-3) if s == 0 else (3e-2, 3e-2)
        for k, p in model.named_parameters():
            assert_digest_close(tensor_digest(p.grad), g["step%d/grad_digest/%s" % (s, k)], rtol=gr, atol_scale=ga,
                                what="%s step%d grad %s" % (name, s, k))
            assert_digest_close(tensor_digest(p), g["step%d/param_digest/%s" % (s, k)], rtol=1e-2, atol_scale=2e-2,
                                what="%s step%d param %s" % (name, s, k))
    model.eval()
    with torch.no_grad():
        recon, (mu, logvar), z = model(dev(data))
    np.testing.assert_allclose(mu.cpu().numpy(), g["eval/mu"], rtol=2e-3, atol=2e-4)
    assert torch.equal(z, mu)


def test_factor_vs_reference_golden_at_20_latents():
    """FactorVAE with latent_dim 20 against the real reference's recorded iteration (both optimizers, quirks Q1 / Q4): loss,
    storer scalars, VAE and discriminator gradient / parameter digests -- the bounds of test_factor_vs_reference_golden."""
    from golden_util import load, tensor_digest, assert_digest_close
    name, img = "factor_z20_dsprites", (1, 64, 64)
    g = load(name)
    D = int(g["latent_dim"])
    model, opt, loss_f, _ = _model("factor", img, D, int(g["seed"]), int(g["n_data"]), float(g["lr"]))
    for k, v in loss_f.discriminator.state_dict().items():
        d = tensor_digest(v)
        np.testing.assert_array_equal(d[2:], g["dinit_digest/" + k][2:], err_msg=k)
        np.testing.assert_allclose(d[:2], g["dinit_digest/" + k][:2], rtol=1e-12, err_msg=k)
    gen = torch.Generator().manual_seed(int(g["seed"]) + 1)
    for s in range(2):
        data = torch.rand((8,) + tuple(img), generator=gen)
        noise = (dev(torch.from_numpy(g["step%d/randn1" % s])), dev(torch.from_numpy(g["step%d/randn2" % s])),
                 torch.from_numpy(g["step%d/perms" % s]))
        assert noise[0].shape == (4, D) and noise[2].shape == (D, 4)
        storer = defaultdict(list)
        out = loss_f.call_optimize(dev(data), model, opt, storer, noise=noise)
        np.testing.assert_allclose(out.item(), g["step%d/loss" % s], rtol=2e-5 if s == 0 else 1e-3)
        for k, v in storer.items():
            np.testing.assert_allclose(v[0], g["step%d/storer/%s" % (s, k)], rtol=5e-5, atol=1e-6, err_msg=k)
        gr, ga = (2e-3, 2e-3) if s == 0 else (3e-2, 3e-2)
        for k, p in model.named_parameters():
            assert_digest_close(tensor_digest(p.grad), g["step%d/grad_digest/%s" % (s, k)], rtol=gr, atol_scale=ga,
                                what="%s step%d grad %s" % (name, s, k))
        for k, p in loss_f.discriminator.named_parameters():
            assert_digest_close(tensor_digest(p.grad), g["step%d/dgrad_digest/%s" % (s, k)], rtol=gr, atol_scale=ga,
                                what="%s step%d dgrad %s" % (name, s, k))
            assert_digest_close(tensor_digest(p), g["step%d/dparam_digest/%s" % (s, k)], rtol=1e-2, atol_scale=2e-2,
                                what="%s step%d dparam %s" % (name, s, k))
