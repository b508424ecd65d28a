"""CPU: the oracle restatement vs the golden vectors produced by the real reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import disvae_oracle as O
from golden_util import load, tensor_digest, assert_digest_close

pytestmark = pytest.mark.usefixtures("golden_threads")     # the digests were recorded with 8 torch threads (conftest.py)

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25,
          betaB_G=1000, factor_G=6.4, latent_dim=10, lr_disc=1e-4, btcvae_A=1, btcvae_B=6.4,
          btcvae_G=1)

CASES = [
    ("vae_mnist", "VAE", (1, 32, 32), 8, 2),
    ("betaB_mnist", "betaB", (1, 32, 32), 8, 2),
    ("btcvae_dsprites", "btcvae", (1, 64, 64), 8, 3),
    ("btcvae_celeba", "btcvae", (3, 64, 64), 6, 2),
    ("betaH_celeba", "betaH", (3, 64, 64), 4, 2),
    ("betaH_mnist_gaussian", "betaH", (1, 32, 32), 4, 1),
    ("betaH_mnist_laplace", "betaH", (1, 32, 32), 4, 1),
]


def test_kats():
    g = load("kats")
    i = torch.arange(12, dtype=torch.float32).view(4, 3)
    mu, logvar, eps = torch.sin(i), 0.5 * torch.cos(i), torch.cos(2 * i + 1)
    z = mu + torch.exp(0.5 * logvar) * eps
    for mss in (True, False):
        r = O.btcvae_log_densities(z, mu, logvar, 100, is_mss=mss)
        for nm, v in zip(["log_pz", "log_qz", "log_prod_qzi", "log_q_zCx"], r):
            np.testing.assert_allclose(v.numpy(), g["kat_%s_mss%d" % (nm, mss)], rtol=1e-6, atol=1e-6)
    # the values quoted in SURVEY.md 8c
    r = O.btcvae_log_densities(z, mu, logvar, 100, True)
    np.testing.assert_allclose(r[1].numpy(), [-6.53220654, -5.76705217, -5.91961098, -5.92125273], rtol=1e-6)
    np.testing.assert_allclose(O.kl_normal_loss(mu, logvar)[0].item(), 0.84495693, rtol=1e-6)
    np.testing.assert_allclose(O.kl_normal_loss(mu, logvar)[0].numpy(), g["kat_kl"], rtol=1e-6)
    x = (torch.arange(32).view(2, 1, 4, 4) % 5).float() / 4
    rr = torch.sigmoid(torch.sin(torch.arange(32).float())).view(2, 1, 4, 4)
    for d, want in [("bernoulli", 12.00147820), ("gaussian", 624.41442871), ("laplace", 16.03404999)]:
        got = O.reconstruction_loss(x, rr, d).item()
        np.testing.assert_allclose(got, g["kat_rec_" + d], rtol=1e-6)
        np.testing.assert_allclose(got, want, rtol=1e-6)
    # saturated sigmoids: ATen's -100 clamp on the fp32 p (recorded from the real reference's _reconstruction_loss)
    ar = torch.arange(2 * 3 * 64 * 64)
    xs = ((ar % 5).float() / 4).view(2, 3, 64, 64)
    x8 = (torch.tensor([0, 128, 255], dtype=torch.uint8)[ar % 3]).view(2, 3, 64, 64)
    for i, v in enumerate(torch.from_numpy(g["kat_rec_sat_logits"])):
        p = torch.sigmoid(v.view(1, 3, 1, 1).expand(2, 3, 64, 64).contiguous())
        assert O.reconstruction_loss(xs, p, "bernoulli").item() == g["kat_rec_sat_loss"][i]
        assert O.reconstruction_loss(x8.float() / 255.0, p, "bernoulli").item() == g["kat_rec_sat_loss_u8"][i]
    st = O.LossState(rec_dist="bernoulli", steps_anneal=10000)
    hp = dict(n_data=100, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)
    loss, logs, keep = O.single_optimizer_loss("btcvae", hp, st, x, rr, mu[:2], logvar[:2], z[:2], True)
    assert keep
    np.testing.assert_allclose(loss.item(), g["kat_btcvae_loss"], rtol=1e-6)
    np.testing.assert_allclose(loss.item(), 6.12298536, rtol=1e-6)
    for k in ["mi_loss", "tc_loss", "dw_kl_loss", "kl_loss"]:
        np.testing.assert_allclose(logs[k].item(), g["kat_btcvae_" + k], rtol=2e-6, atol=1e-6)
    assert O.linear_annealing(0, 1, 1, 10000) == g["kat_anneal_a"] == 1e-4
    assert O.linear_annealing(0, 25, 5000, 100000) == g["kat_anneal_b"] == 1.25
    assert O.linear_annealing(0, 1, 5, 0) == 1
    for (b, n) in [(4, 100), (8, 737280), (64, 202599)]:
        np.testing.assert_array_equal(O.log_importance_weight_matrix(b, n).numpy(), g["kat_logiw_%d_%d" % (b, n)])
    # quirk Q2: column-wise, not diagonal
    W = O.log_importance_weight_matrix(4, 100).exp()
    np.testing.assert_allclose(W[0].numpy(), [.01, .3233333, .3333333, .3333333], rtol=1e-5)
    np.testing.assert_allclose(W[2].numpy(), [.3233333, .3233333, .3333333, .3333333], rtol=1e-5)


@pytest.mark.parametrize("name,loss,img,batch,steps", CASES)
def test_single_optimizer_cases(name, loss, img, batch, steps):
    g = load(name)
    seed = int(g["seed"])
    torch.manual_seed(seed)
    params = O.init_vae_params(img, 10)
    for k, v in params.items():   # same seeds -> bit-identical initial weights
        np.testing.assert_array_equal(tensor_digest(v), g["init_digest/" + k], err_msg=k)
    rec_dist = name.split("_")[-1] if name.endswith(("gaussian", "laplace")) else "bernoulli"
    hp = dict(HP, n_data=int(g["n_data"]))
    tr = O.OracleTrainer(loss, hp, img, 10, lr=float(g["lr"]), rec_dist=rec_dist,
                         steps_anneal=HP["reg_anneal"], params=params)
    gen = torch.Generator().manual_seed(seed + 1)
    for s in range(steps):
        data = torch.rand((batch,) + tuple(img), generator=gen)
        eps = torch.from_numpy(g["step%d/randn0" % s])
        loss_val, logs = tr.train_iteration(data, eps=eps)
        np.testing.assert_allclose(loss_val, g["step%d/loss" % s], rtol=2e-6)
        if s == 0:  # storer is kept on step 1 only (n_train_steps % 50 == 1)
            for k in logs:
                np.testing.assert_allclose(logs[k].item(), g["step0/storer/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
            assert set(logs) == {k.split("/")[-1] for k in g if k.startswith("step0/storer/")}
        else:
            assert not any(k.startswith("step%d/storer/" % s) for k in g)
        for k, p in tr.params.items():
            assert_digest_close(tensor_digest(p.grad), g["step%d/grad_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d grad %s" % (name, s, k))
            assert_digest_close(tensor_digest(p), g["step%d/param_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d param %s" % (name, s, k))


@pytest.mark.parametrize("name,loss,img,batch,steps,zdim", [
    ("btcvae_z16_dsprites", "btcvae", (1, 64, 64), 8, 2, 16),
    ("btcvae_z3_mnist", "btcvae", (1, 32, 32), 6, 2, 3),
    ("betaB_z16_celeba", "betaB", (3, 64, 64), 4, 2, 16),
    ("btcvae_z32_celeba", "btcvae", (3, 64, 64), 6, 2, 32),       # above 16: `make_golden.py --wide-latent`
    ("vae_z24_mnist", "VAE", (1, 32, 32), 8, 2, 24),
])
def test_other_latent_dimensions(name, loss, img, batch, steps, zdim):
    """main.py -z: latent dimensions other than 10 (the largest the FUSED kernels of the native engine take, a small one, and
    two above 16 for its run-time-D kernels), recorded from the real reference with `make_golden.py --latent` /
    `--wide-latent`: initial weights bit for bit, losses, storer scalars, gradients, parameters"""
    g = load(name)
    assert int(g["latent_dim"]) == zdim
    seed = int(g["seed"])
    torch.manual_seed(seed)
    params = O.init_vae_params(img, zdim)
    for k, v in params.items():
        np.testing.assert_array_equal(tensor_digest(v), g["init_digest/" + k], err_msg=k)
    hp = dict(HP, n_data=int(g["n_data"]), latent_dim=zdim)
    tr = O.OracleTrainer(loss, hp, img, zdim, lr=float(g["lr"]), rec_dist="bernoulli", steps_anneal=HP["reg_anneal"], params=params)
    gen = torch.Generator().manual_seed(seed + 1)
    for s in range(steps):
        data = torch.rand((batch,) + tuple(img), generator=gen)
        eps = torch.from_numpy(g["step%d/randn0" % s])
        assert eps.shape == (batch, zdim)
        loss_val, logs = tr.train_iteration(data, eps=eps)
        np.testing.assert_allclose(loss_val, g["step%d/loss" % s], rtol=2e-6)
        if s == 0:
            for k in logs:
                np.testing.assert_allclose(logs[k].item(), g["step0/storer/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
            assert set(logs) == {k.split("/")[-1] for k in g if k.startswith("step0/storer/")}
            assert sum(k.startswith("kl_loss_") for k in logs) == zdim
        for k, p in tr.params.items():
            assert_digest_close(tensor_digest(p.grad), g["step%d/grad_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d grad %s" % (name, s, k))
            assert_digest_close(tensor_digest(p), g["step%d/param_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d param %s" % (name, s, k))


@pytest.mark.parametrize("name,img", [("factor_dsprites", (1, 64, 64)), ("factor_celeba", (3, 64, 64)),
                                      ("factor_z20_dsprites", (1, 64, 64))])
def test_factor_cases(name, img):
    g = load(name)
    seed = int(g["seed"])
    Z = int(g["latent_dim"]) if "latent_dim" in g else 10      # 10, or 20 (`make_golden.py --wide-latent`)
    torch.manual_seed(seed)
    params = O.init_vae_params(img, Z)
    dparams = O.init_disc_params(Z)
    for k, v in params.items():
        np.testing.assert_array_equal(tensor_digest(v), g["init_digest/" + k], err_msg=k)
    for k, v in dparams.items():
        np.testing.assert_array_equal(tensor_digest(v), g["dinit_digest/" + k], err_msg=k)
    hp = dict(HP, n_data=int(g["n_data"]), latent_dim=Z)
    tr = O.OracleTrainer("factor", hp, img, Z, lr=float(g["lr"]), lr_disc=HP["lr_disc"],
                         steps_anneal=HP["reg_anneal"], params=params, dparams=dparams)
    gen = torch.Generator().manual_seed(seed + 1)
    B = 8
    for s in range(2):
        data = torch.rand((B,) + tuple(img), generator=gen)
        assert g["step%d/randn0" % s].shape == (B, Z)         # wasted full-batch draw (Q4)
        eps1 = torch.from_numpy(g["step%d/randn1" % s])
        eps2 = torch.from_numpy(g["step%d/randn2" % s])
        perms = [torch.from_numpy(p) for p in g["step%d/perms" % s]]
        assert len(perms) == Z and perms[0].numel() == B // 2
        loss_val, logs = tr.train_iteration(data, eps=eps1, eps2=eps2, perms=perms)
        np.testing.assert_allclose(loss_val, g["step%d/loss" % s], rtol=2e-6)
        if s == 0:
            for k in logs:
                np.testing.assert_allclose(logs[k].item(), g["step0/storer/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
        for k, p in tr.params.items():
            assert_digest_close(tensor_digest(p.grad), g["step%d/grad_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d grad %s" % (name, s, k))
            assert_digest_close(tensor_digest(p), g["step%d/param_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d param %s" % (name, s, k))
        for k, p in tr.dparams.items():
            assert_digest_close(tensor_digest(p.grad), g["step%d/dgrad_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d dgrad %s" % (name, s, k))
            assert_digest_close(tensor_digest(p), g["step%d/dparam_digest/%s" % (s, k)], rtol=2e-5,
                                what="%s step%d dparam %s" % (name, s, k))


@pytest.mark.parametrize("name,loss", [("btcvae_dsprites_b256", "btcvae"), ("factor_dsprites_b256", "factor")])
def test_bench_size_cases(name, loss):
    """Step 0 of the two dsprites BASELINE workloads at their OWN batch (256), recorded from the real reference with
    `make_golden.py --bench-size`: the oracle reproduces the reference at a bench size too (same bounds as the small cases)."""
    g = load(name)
    img, B = (1, 64, 64), 256
    seed = int(g["seed"])
    torch.manual_seed(seed)
    params = O.init_vae_params(img, 10)
    dparams = O.init_disc_params(10) if loss == "factor" else None
    hp = dict(HP, n_data=int(g["n_data"]))
    tr = O.OracleTrainer(loss, hp, img, 10, lr=float(g["lr"]), lr_disc=HP["lr_disc"], steps_anneal=HP["reg_anneal"],
                         params=params, dparams=dparams)
    data = torch.rand((B,) + img, generator=torch.Generator().manual_seed(seed + 1))
    if loss == "factor":
        perms = [torch.from_numpy(p) for p in g["step0/perms"]]
        assert len(perms) == 10 and perms[0].numel() == B // 2
        loss_val, logs = tr.train_iteration(data, eps=torch.from_numpy(g["step0/randn1"]), eps2=torch.from_numpy(g["step0/randn2"]),
                                            perms=perms)
    else:
        loss_val, logs = tr.train_iteration(data, eps=torch.from_numpy(g["step0/randn0"]))
    np.testing.assert_allclose(loss_val, g["step0/loss"], rtol=2e-6)
    for k in logs:
        np.testing.assert_allclose(logs[k].item(), g["step0/storer/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    for k, p in tr.params.items():
        assert_digest_close(tensor_digest(p.grad), g["step0/grad_digest/" + k], rtol=2e-5, what="%s grad %s" % (name, k))
        assert_digest_close(tensor_digest(p), g["step0/param_digest/" + k], rtol=2e-5, what="%s param %s" % (name, k))
    if loss == "factor":
        for k, p in tr.dparams.items():
            assert_digest_close(tensor_digest(p.grad), g["step0/dgrad_digest/" + k], rtol=2e-5, what="%s dgrad %s" % (name, k))


def test_eval_forward_matches_golden():
    """eval-mode forward (z = mu) after the golden training steps."""
    g = load("btcvae_dsprites")
    torch.manual_seed(int(g["seed"]))
    params = O.init_vae_params((1, 64, 64), 10)
    hp = dict(HP, n_data=int(g["n_data"]))
    tr = O.OracleTrainer("btcvae", hp, (1, 64, 64), 10, lr=float(g["lr"]),
                         steps_anneal=HP["reg_anneal"], params=params)
    gen = torch.Generator().manual_seed(int(g["seed"]) + 1)
    for s in range(3):
        data = torch.rand((8, 1, 64, 64), generator=gen)
        tr.train_iteration(data, eps=torch.from_numpy(g["step%d/randn0" % s]))
    with torch.no_grad():
        recon, (mu, logvar), z = O.vae_forward(tr.params, data, None)
    np.testing.assert_allclose(mu.numpy(), g["eval/mu"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(logvar.numpy(), g["eval/logvar"], rtol=1e-4, atol=1e-6)
    assert torch.equal(z, mu)
    assert_digest_close(tensor_digest(recon, 64), g["eval/recon_digest"], rtol=1e-5, what="recon")


def test_metrics_oracle_vs_reference_golden():
    """MIG / AAM estimator restatement (oracle) vs the fixture recorded from the real reference Evaluator
    (tests/golden/make_golden.py --metrics; disvae/evaluate.py:119-317), same injected randperm draws."""
    g = load("metrics")
    lat_sizes = [int(k) for k in g["lat_sizes"]]
    mean, logvar = torch.from_numpy(g["mean"]), torch.from_numpy(g["logvar"])
    samples = mean.clone()
    S, D = int(g["n_samples"]), mean.shape[1]
    H_z = O.estimate_latent_entropies(samples, mean, logvar, torch.from_numpy(g["H_z/perm"])[:S], S)
    np.testing.assert_allclose(H_z.numpy(), g["H_z"], rtol=2e-6)
    H_z40 = O.estimate_latent_entropies(samples, mean, logvar, torch.from_numpy(g["H_z40/perm"])[:40], 40)
    np.testing.assert_allclose(H_z40.numpy(), g["H_z40"], rtol=2e-6)
    perms = [torch.from_numpy(g["H_zCv/perm%d" % j])[:S] for j in range(sum(lat_sizes))]
    H_zCv = O.estimate_H_zCv(samples.view(*lat_sizes, D), mean.view(*lat_sizes, D), logvar.view(*lat_sizes, D), lat_sizes, perms, S)
    np.testing.assert_allclose(H_zCv.numpy(), g["H_zCv"], rtol=2e-6)
    mig, aam, sorted_mi = O.metrics_from_entropies(H_z, H_zCv, lat_sizes)
    np.testing.assert_allclose(sorted_mi.numpy(), g["sorted_mut_info"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mig.item(), g["MIG"], rtol=1e-5)
    np.testing.assert_allclose(aam.item(), g["AAM"], rtol=1e-5)
