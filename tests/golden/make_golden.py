"""Generate golden vectors by running the REAL reference (read-only at /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py [--latent | --wide-latent | --metrics | --bench-size]

The reference is imported unmodified with the two import shims of SURVEY.md section 8c
(stub ``imageio``; ``numpy.product = numpy.prod``).  Noise is recorded by wrapping
``torch.randn_like`` / ``torch.randperm`` so that the oracle and the HIP engine can
replay the same eps / permutations.  Outputs are small ``.npz`` files committed next to
this script; nothing from the reference's sources is copied.
"""
import os
import sys
import types
import logging
from collections import defaultdict, OrderedDict

import numpy as np
import torch
torch.set_num_threads(8)      # the digests compare fp32 CPU gradients at rtol 2e-5: the thread count fixes the reduction order (tests/conftest.py)

REF = os.environ.get("DVAE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25,
          betaB_G=1000, factor_G=6.4, latent_dim=10, lr_disc=1e-4, btcvae_A=1, btcvae_B=6.4,
          btcvae_G=1)


def import_reference():
    sys.modules.setdefault("imageio", types.ModuleType("imageio"))
    if not hasattr(np, "product"):
        np.product = np.prod
    sys.path.insert(0, REF)
    import disvae  # noqa
    from disvae.models import losses, vae, discriminator
    from disvae.utils import math as dmath
    from disvae import training
    return disvae, losses, vae, discriminator, dmath, training


class NoiseRecorder:
    """Wrap torch.randn_like / torch.randperm, keep what the reference drew."""

    def __enter__(self):
        self.randn, self.perm = [], []
        self._rl, self._rp = torch.randn_like, torch.randperm

        def randn_like(t, *a, **k):
            out = self._rl(t, *a, **k)
            self.randn.append(out.detach().clone())
            return out

        def randperm(n, *a, **k):
            out = self._rp(n, *a, **k)
            self.perm.append(out.detach().clone())
            return out

        torch.randn_like, torch.randperm = randn_like, randperm
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.randperm = self._rl, self._rp


def tensor_digest(t, n_samples=24):
    """sum, abs-sum, and a few deterministic sample entries of a tensor."""
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n_samples, f.numel())).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def kats(losses, dmath):
    """RNG-free known-answer vectors."""
    out = {}
    i = torch.arange(12, dtype=torch.float32).view(4, 3)
    mu, logvar, eps = torch.sin(i), 0.5 * torch.cos(i), torch.cos(2 * i + 1)
    z = mu + torch.exp(0.5 * logvar) * eps
    for mss in (True, False):
        r = losses._get_log_pz_qz_prodzi_qzCx(z, (mu, logvar), 100, is_mss=mss)
        for nm, v in zip(["log_pz", "log_qz", "log_prod_qzi", "log_q_zCx"], r):
            out["kat_%s_mss%d" % (nm, mss)] = v.numpy()
    out["kat_kl"] = losses._kl_normal_loss(mu, logvar).numpy()
    x = (torch.arange(32).view(2, 1, 4, 4) % 5).float() / 4
    r = torch.sigmoid(torch.sin(torch.arange(32).float())).view(2, 1, 4, 4)
    for d in ["bernoulli", "gaussian", "laplace"]:
        out["kat_rec_" + d] = losses._reconstruction_loss(x, r, distribution=d).numpy()
    # saturated Bernoulli likelihood (losses.py:430): logits whose fp32 sigmoid rounds to 1 (v >= ~16.64) or is 0 (exp(-v)
    # overflows, v < -88.72) -- F.binary_cross_entropy's -100 clamp decides those terms; three logits = one per channel of a
    # [2, 3, 64, 64] reconstruction, targets (i % 5) / 4 (fp32) and 255-level uint8 pixels (i % 3) -> {0, 128, 255}
    sat = torch.tensor([[16.6, 17., 20.], [-16.6, -17., -20.], [40., 88., 90.], [-40., -88., -90.], [104., -104., 0.5],
                        [8.5, -12., 3.]])
    ar = torch.arange(2 * 3 * 64 * 64)
    xs = ((ar % 5).float() / 4).view(2, 3, 64, 64)
    x8 = (torch.tensor([0, 128, 255], dtype=torch.uint8)[ar % 3]).view(2, 3, 64, 64)
    out["kat_rec_sat_logits"] = sat.numpy()
    for nm, tgt in (("", xs), ("_u8", x8.float() / 255.0)):
        out["kat_rec_sat_loss" + nm] = np.stack([
            losses._reconstruction_loss(tgt, torch.sigmoid(v.view(1, 3, 1, 1).expand(2, 3, 64, 64).contiguous()),
                                        distribution="bernoulli").numpy() for v in sat])
    lf = losses.BtcvaeLoss(100, alpha=1, beta=6.4, gamma=1, steps_anneal=10000)
    st = defaultdict(list)
    loss = lf(x, r, (mu[:2], logvar[:2]), True, st, latent_sample=z[:2])
    out["kat_btcvae_loss"] = loss.numpy()
    for k in ["mi_loss", "tc_loss", "dw_kl_loss", "kl_loss"]:
        out["kat_btcvae_" + k] = np.float32(st[k][0])
    out["kat_anneal_a"] = np.float64(losses.linear_annealing(0, 1, 1, 10000))
    out["kat_anneal_b"] = np.float64(losses.linear_annealing(0, 25, 5000, 100000))
    for (b, n) in [(4, 100), (8, 737280), (64, 202599)]:
        out["kat_logiw_%d_%d" % (b, n)] = dmath.log_importance_weight_matrix(b, n).numpy()
    return out


def run_case(ref, loss_name, img_size, batch, n_steps, seed, n_data, lr, rec_dist="bernoulli", latent_dim=10, absmax=False):
    """Run the reference Trainer._train_iteration n_steps times; record everything."""
    disvae, losses, vae, discriminator, dmath, training = ref
    torch.manual_seed(seed)
    model = vae.init_specific_model("Burgess", img_size, latent_dim)
    init_state = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    kw = dict(HP)
    kw["rec_dist"] = rec_dist
    kw["latent_dim"] = latent_dim
    loss_f = losses.get_loss_f(loss_name, n_data=n_data, device=torch.device("cpu"), **kw)
    init_dstate = None
    if loss_name == "factor":
        init_dstate = OrderedDict((k, v.detach().clone()) for k, v in loss_f.discriminator.state_dict().items())
    logging.disable(logging.CRITICAL)
    tr = training.Trainer(model, opt, loss_f, device=torch.device("cpu"), save_dir="/tmp",
                          is_progress_bar=False)
    model.train()
    gen = torch.Generator().manual_seed(seed + 1)
    out = {}
    out["seed"] = np.int64(seed)
    out["n_data"] = np.int64(n_data)
    out["lr"] = np.float64(lr)
    out["latent_dim"] = np.int64(latent_dim)
    for k, v in init_state.items():
        out["init_digest/" + k] = tensor_digest(v)
    if init_dstate is not None:
        for k, v in init_dstate.items():
            out["dinit_digest/" + k] = tensor_digest(v)
    for step in range(n_steps):
        data = torch.rand((batch,) + tuple(img_size), generator=gen)
        storer = defaultdict(list)
        with NoiseRecorder() as rec:
            loss_val = tr._train_iteration(data, storer)
        out["step%d/loss" % step] = np.float64(loss_val)
        for k, v in storer.items():
            out["step%d/storer/%s" % (step, k)] = np.float64(v[0])
        for j, e in enumerate(rec.randn):
            out["step%d/randn%d" % (step, j)] = e.numpy()
        if rec.perm:
            out["step%d/perms" % step] = torch.stack(rec.perm).numpy()
        for k, p in model.named_parameters():
            out["step%d/grad_digest/%s" % (step, k)] = tensor_digest(p.grad)
            out["step%d/param_digest/%s" % (step, k)] = tensor_digest(p)
            if absmax:         # scale of the element-wise bound two fp32 arithmetics are held to (ReLU gates at rounding level)
                out["step%d/grad_absmax/%s" % (step, k)] = np.float64(p.grad.abs().max().item())
        if loss_name == "factor":
            for k, p in loss_f.discriminator.named_parameters():
                if absmax:
                    out["step%d/dgrad_absmax/%s" % (step, k)] = np.float64(p.grad.abs().max().item())
                out["step%d/dgrad_digest/%s" % (step, k)] = tensor_digest(p.grad)
                out["step%d/dparam_digest/%s" % (step, k)] = tensor_digest(p)
    # full small tensors for the last forward (eval-mode forward on the last batch)
    model.eval()
    with torch.no_grad():
        recon, (mu, logvar), z = model(data)
    out["eval/mu"] = mu.numpy()
    out["eval/logvar"] = logvar.numpy()
    out["eval/recon_digest"] = tensor_digest(recon, 64)
    return out


def metrics_case():
    """MIG / AAM pieces of the REAL reference Evaluator (disvae/evaluate.py:119-317) on a small synthetic latent table:
    marginal entropies (two sample counts), conditional entropies, the whole metric.  The Evaluator object is built
    without a model (its estimator methods only use device / logger / progress-bar flag); `_estimate_H_zCv` calls
    `_estimate_latent_entropies` with its default n_samples = 10000, which requires every conditional slice to hold
    >= 10000 points -- the slices here are tiny, so the default is overridden (the arithmetic is unchanged)."""
    import_reference()
    from disvae import evaluate as ev_mod
    import functools
    lat_sizes = np.array([3, 4, 5])
    N, D, S = 60, 4, 10
    gen = torch.Generator().manual_seed(2024)
    mean = torch.randn(N, D, generator=gen) * 1.5
    logvar = torch.randn(N, D, generator=gen) * 0.8 - 1.0
    # make two latent dims informative about two factors so that MIG / AAM are not ~0
    grid = np.stack(np.meshgrid(*[np.arange(k) for k in lat_sizes], indexing="ij"), -1).reshape(N, 3)
    mean[:, 0] += torch.from_numpy(grid[:, 0]).float() * 2.0
    mean[:, 2] += torch.from_numpy(grid[:, 2]).float() * 1.2
    samples = mean.clone()                                   # Evaluator.__call__ puts the model in eval mode: z = mean
    ev = object.__new__(ev_mod.Evaluator)
    ev.device, ev.is_progress_bar, ev.logger, ev.save_dir = torch.device("cpu"), True, logging.getLogger("golden"), "/tmp"
    out = dict(lat_sizes=lat_sizes, mean=mean.numpy(), logvar=logvar.numpy(), n_samples=np.int64(S))
    logging.disable(logging.CRITICAL)
    with NoiseRecorder() as rec:
        torch.manual_seed(5)
        H_z = ev._estimate_latent_entropies(samples, (mean, logvar), n_samples=S)
    out["H_z"], out["H_z/perm"] = H_z.numpy(), rec.perm[0].numpy()
    with NoiseRecorder() as rec:
        H_z40 = ev._estimate_latent_entropies(samples, (mean, logvar), n_samples=40)
    out["H_z40"], out["H_z40/perm"] = H_z40.numpy(), rec.perm[0].numpy()
    ev._estimate_latent_entropies = functools.partial(ev_mod.Evaluator._estimate_latent_entropies, ev, n_samples=S)
    with NoiseRecorder() as rec:
        H_zCv = ev._estimate_H_zCv(samples.view(*lat_sizes, D), tuple(p.view(*lat_sizes, D) for p in (mean, logvar)),
                                   lat_sizes, ["a", "b", "c"])
    out["H_zCv"] = H_zCv.numpy()
    for j, p in enumerate(rec.perm):
        out["H_zCv/perm%d" % j] = p.numpy()
    mut_info = -H_zCv + H_z
    sorted_mi = torch.sort(mut_info, dim=1, descending=True)[0].clamp(min=0)
    out["sorted_mut_info"] = sorted_mi.numpy()
    out["MIG"] = ev._mutual_information_gap(sorted_mi, lat_sizes).numpy()
    out["AAM"] = ev._axis_aligned_metric(sorted_mi).numpy()
    return out


def main():
    if "--metrics" in sys.argv:          # only the MIG / AAM fixture (the training-step fixtures stay as committed)
        out = metrics_case()
        np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
        print("metrics: MIG", out["MIG"], "AAM", out["AAM"], "H_z", out["H_z"])
        return
    ref = import_reference()
    _, losses, vae, discriminator, dmath, training = ref
    if "--kats" in sys.argv:             # only the RNG-free known-answer vectors
        out = kats(losses, dmath)
        np.savez_compressed(os.path.join(HERE, "kats.npz"), **out)
        print("kats:", {k: np.asarray(v).tolist() for k, v in out.items() if k.startswith("kat_rec")})
        return
    if "--bench-size" in sys.argv:    # step 0 of the two dsprites BASELINE workloads at their own batch (digests + noise only)
        for name, loss, img, b, steps, seed, n_data, lr in [
                ("btcvae_dsprites_b256", "btcvae", (1, 64, 64), 256, 1, 1234, 737280, 5e-4),
                ("factor_dsprites_b256", "factor", (1, 64, 64), 256, 1, 1234, 737280, 1e-4)]:
            out = run_case(ref, loss, img, b, steps, seed, n_data, lr, absmax=True)
            drop = ("eval/",) + (("step0/randn0",) if loss == "factor" else ())     # factor: the wasted full-batch draw (Q4)
            for k in [k_ for k_ in out if k_.startswith(drop)]:          # [256, 10] tensors nobody replays: keep the file small
                del out[k]
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            print(name, "loss", [out["step%d/loss" % s] for s in range(steps)], os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
        return
    if "--wide-latent" in sys.argv:   # latent dimensions above 16 (the engine's run-time-D kernels): only these files are written
        for name, loss, img, b, steps, seed, n_data, lr, zdim in [
                ("btcvae_z32_celeba", "btcvae", (3, 64, 64), 6, 2, 4321, 202599, 5e-4, 32),
                ("factor_z20_dsprites", "factor", (1, 64, 64), 8, 2, 4321, 737280, 1e-4, 20),
                ("vae_z24_mnist", "VAE", (1, 32, 32), 8, 2, 4321, 60000, 5e-4, 24)]:
            out = run_case(ref, loss, img, b, steps, seed, n_data, lr, latent_dim=zdim)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            print(name, "loss", [out["step%d/loss" % s] for s in range(steps)])
        return
    if "--latent" in sys.argv:        # latent dimensions other than the default 10 (main.py -z): only these files are written
        for name, loss, img, b, steps, seed, n_data, lr, zdim in [
                ("btcvae_z16_dsprites", "btcvae", (1, 64, 64), 8, 2, 4321, 737280, 5e-4, 16),
                ("btcvae_z3_mnist", "btcvae", (1, 32, 32), 6, 2, 4321, 60000, 5e-4, 3),
                ("betaB_z16_celeba", "betaB", (3, 64, 64), 4, 2, 4321, 202599, 5e-4, 16)]:
            out = run_case(ref, loss, img, b, steps, seed, n_data, lr, latent_dim=zdim)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            print(name, "loss", [out["step%d/loss" % s] for s in range(steps)])
        return
    np.savez_compressed(os.path.join(HERE, "kats.npz"), **kats(losses, dmath))
    cases = [
        # name,           loss,     img_size,    B, steps, seed, n_data,  lr
        ("vae_mnist",      "VAE",    (1, 32, 32), 8, 2, 1234, 60000, 5e-4),
        ("betaB_mnist",    "betaB",  (1, 32, 32), 8, 2, 1234, 60000, 5e-4),
        ("btcvae_dsprites", "btcvae", (1, 64, 64), 8, 3, 1234, 737280, 5e-4),
        ("btcvae_celeba",  "btcvae", (3, 64, 64), 6, 2, 1234, 202599, 5e-4),
        ("betaH_celeba",   "betaH",  (3, 64, 64), 4, 2, 1234, 202599, 5e-4),
        ("factor_dsprites", "factor", (1, 64, 64), 8, 2, 1234, 737280, 1e-4),
        ("factor_celeba",  "factor", (3, 64, 64), 8, 2, 1234, 202599, 1e-4),
    ]
    for name, loss, img, b, steps, seed, n_data, lr in cases:
        out = run_case(ref, loss, img, b, steps, seed, n_data, lr)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "loss", [out["step%d/loss" % s] for s in range(steps)])
    # reconstruction-distribution variants (one step each)
    for dist in ["gaussian", "laplace"]:
        out = run_case(ref, "betaH", (1, 32, 32), 4, 1, 7, 60000, 5e-4, rec_dist=dist)
        np.savez_compressed(os.path.join(HERE, "betaH_mnist_%s.npz" % dist), **out)
        print(dist, out["step0/loss"])


if __name__ == "__main__":
    main()
