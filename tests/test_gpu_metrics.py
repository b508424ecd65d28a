"""-m gpu: the MIG / AAM entropy estimator (disvae/evaluate.py:119-317) on the HIP kernel vs the values
recorded from the real reference (tests/golden/metrics.npz) and vs the oracle."""
import logging
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib  # noqa
from golden_util import load
from oracle import disvae_oracle as O
from disvae_amd.evaluate import Evaluator
from disvae_amd.models.vae import init_specific_model
from disvae_amd.models.losses import get_loss_f


def _entropy(z_sd, mean, logvar, S):
    """z_sd: the gathered [S, D] samples (memory = the reference's [D, S] view)."""
    N, D = mean.shape
    ws = torch.empty(_lib.lib().dvae_latent_entropy_ws_floats(N, D, S), device=DEV)
    H = torch.empty(D, device=DEV)
    call("dvae_latent_entropy", ptr(dev(z_sd)), ptr(dev(mean)), ptr(dev(logvar)), N, D, S, ptr(ws), ptr(H), stream())
    return H, ws


def test_entropy_kernel_vs_reference_golden():
    g = load("metrics")
    mean, logvar = torch.from_numpy(g["mean"]), torch.from_numpy(g["logvar"])
    for key, S in (("H_z", int(g["n_samples"])), ("H_z40", 40)):
        idx = torch.from_numpy(g[key + "/perm"])[:S]
        H, _ = _entropy(mean.index_select(0, idx), mean, logvar, S)
        np.testing.assert_allclose(H.cpu().numpy(), g[key], rtol=1e-5, err_msg=key)


def test_entropy_of_a_column_without_any_finite_density_is_inf_not_nan():
    """exp(-logvar) overflows for logvar < -88.7: every log-density of such a latent is -inf, torch.logsumexp returns -inf
    and the entropy estimate is +inf (evaluate.py:273-289); the kernel's running-max rescale must not turn that into NaN."""
    N, D, S = 40, 2, 5
    gen = torch.Generator().manual_seed(7)
    mean = torch.randn(N, D, generator=gen)
    logvar = torch.randn(N, D, generator=gen) * 0.3
    logvar[:, 0] = -200.0
    z = torch.randn(S, D, generator=gen) + 3.0           # S x D block = the [D, S] image the kernel reads
    H, _ = _entropy(z, mean, logvar, S)
    zv = z.reshape(D, S)
    ld = -0.5 * (math.log(2 * math.pi) + logvar[:, 1:2]) - 0.5 * (zv[1].view(1, S) - mean[:, 1:2]) ** 2 * torch.exp(-logvar[:, 1:2])
    ref1 = (math.log(N) - torch.logsumexp(ld.double(), 0)).mean()
    got = H.cpu()
    assert torch.isinf(got[0]) and got[0] > 0, got
    np.testing.assert_allclose(got[1].item(), ref1.item(), rtol=1e-5)


@pytest.mark.parametrize("N,D,S", [(5003, 10, 777), (17, 3, 5), (40000, 16, 300)])
def test_entropy_kernel_vs_oracle(N, D, S):
    gen = torch.Generator().manual_seed(N)
    mean = torch.randn(N, D, generator=gen) * 2
    logvar = torch.randn(N, D, generator=gen) * 0.7 - 1.5
    S = min(S, N)
    idx = torch.randperm(N, generator=gen)[:S]
    H, _ = _entropy(mean.index_select(0, idx), mean, logvar, S)
    ref = O.estimate_latent_entropies(mean.double(), mean.double(), logvar.double(), idx, S, mini_batch_size=50)
    check(H, ref, rtol=1e-5, atol_rel=1e-6, what="H_z N=%d" % N)


def test_entropy_kernel_at_dsprites_scale():
    """N = 737 280 (dSprites), D = 10, S = 10 000 -- 7.4e13 log-densities; the oracle cannot run that, so 12 of the
    10 000 per-sample logsumexps are checked against fp64 over the full data set and the mean against their definition."""
    import time
    N, D, S = 737280, 10, 10000
    gen = torch.Generator().manual_seed(1)
    mean = torch.randn(N, D, generator=gen) * 2
    logvar = torch.randn(N, D, generator=gen) * 0.5 - 2.0
    idx = torch.randperm(N, generator=gen)[:S]
    z_sd = mean.index_select(0, idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    H, ws = _entropy(z_sd, mean, logvar, S)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("dvae_latent_entropy N=737280 D=10 S=10000: %.2f s (incl. upload)" % dt)
    chunks = min(64, (N + 16383) // 16384)
    lse = ws[3 * D * N + chunks * D * S * 2:][:D * S].view(D, S).cpu().double()
    np.testing.assert_allclose(H.cpu().double().numpy(), (math.log(N) - lse).mean(1).numpy(), rtol=1e-5)
    z_ds = z_sd.reshape(D, S).double()                     # the reference's view (evaluate.py:262)
    m64, l64 = mean.double(), logvar.double()
    for s in range(0, S, 900):
        for d in (0, 7):
            ld = O.log_density_gaussian(z_ds[d, s], m64[:, d], l64[:, d])
            np.testing.assert_allclose(lse[d, s].item(), torch.logsumexp(ld, 0).item(), rtol=2e-6, atol=2e-6)


class _FactorData:
    """tiny data set with known factors, iterated in factor order (what compute_metrics requires)."""
    lat_sizes = np.array([3, 4, 5])
    lat_names = ("a", "b", "c")

    def __init__(self, images):
        self.images = images

    def __len__(self):
        return self.images.shape[0]


class _Loader:
    def __init__(self, images, bs):
        self.dataset, self.bs = _FactorData(images), bs

    def __len__(self):
        return (len(self.dataset) + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(0, len(self.dataset), self.bs):
            yield self.dataset.images[i:i + self.bs], 0


def test_compute_metrics_end_to_end(tmp_path):
    """Evaluator(...)(loader, is_metrics=True) (evaluate.py:60-95,119-158): native encoder -> entropies -> MIG / AAM,
    vs the oracle evaluated on the same q(z|x) table with the same injected randperm draws."""
    img, N, S = (1, 64, 64), 60, 10
    torch.manual_seed(3)
    model = init_specific_model("Burgess", img, 10)
    loss_f = get_loss_f("btcvae", device=torch.device(DEV), n_data=N, rec_dist="bernoulli", reg_anneal=0, btcvae_A=1,
                        btcvae_B=6, btcvae_G=1)
    gen = torch.Generator().manual_seed(4)
    images = torch.rand((N,) + img, generator=gen)
    loader = _Loader(images, 16)
    ev = Evaluator(model, loss_f, device=torch.device(DEV), logger=logging.getLogger("m"), save_dir=str(tmp_path),
                   is_progress_bar=False)
    lat_sizes = [3, 4, 5]
    draws = [torch.randperm(N, generator=gen)[:S]] + [torch.randperm(N // k, generator=gen)[:S] for k in lat_sizes for _ in range(k)]
    model.train()
    ev.model.eval()
    metrics = ev.compute_metrics(loader, sample_idx=draws, n_samples=S)
    with torch.no_grad():
        mean, logvar = model.encoder(images.to(DEV))
    mean, logvar = mean.cpu().double(), logvar.cpu().double()
    D = 10
    H_z = O.estimate_latent_entropies(mean, mean, logvar, draws[0], S)
    H_zCv = O.estimate_H_zCv(mean.view(*lat_sizes, D), mean.view(*lat_sizes, D), logvar.view(*lat_sizes, D), lat_sizes, draws[1:], S)
    mig, aam, _ = O.metrics_from_entropies(H_z.float(), H_zCv.float(), lat_sizes)
    helpers = torch.load(tmp_path / "metric_helpers.pth")
    np.testing.assert_allclose(helpers["marginal_entropies"].numpy(), H_z.numpy(), rtol=1e-5)
    np.testing.assert_allclose(helpers["cond_entropies"].numpy(), H_zCv.numpy(), rtol=1e-5)
    np.testing.assert_allclose(metrics["MIG"], mig.item(), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(metrics["AAM"], aam.item(), rtol=1e-3, atol=1e-6)
    # the public entry point writes metrics.log and restores train mode; data without factors is refused like the reference
    model.train()
    ev(loader, is_metrics=False, is_losses=True)   # (metrics through __call__ need >= 10000 points per slice: see compute_metrics above)
    assert model.training
    class _NoFactors:
        def __init__(self):
            self.dataset = [0, 1, 2, 3]

    with pytest.raises(ValueError, match="known true factors"):
        ev.compute_metrics(_NoFactors())
