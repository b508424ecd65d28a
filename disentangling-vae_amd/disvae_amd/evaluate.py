"""Evaluation with the API of disvae/evaluate.py on the native model.

* Test losses (``Evaluator.compute_losses``, evaluate.py:60-117): eval-mode forward (z = mean, vae.py:69-71) + the loss
  plugins with ``is_train=False`` (storer always kept, annealing = 1, losses.py:109,146-147), one stream of HIP launches
  per batch and ONE device->host copy for all logged scalars.
  Reference quirk: ``compute_losses`` returns from inside its batch loop (evaluate.py:116-117), i.e. it
  evaluates only the FIRST batch and divides by the number of batches.  Here every batch is evaluated and the mean is
  returned; ``reference_early_return=True`` reproduces the reference's numbers.
* MIG / AAM disentanglement metrics (``compute_metrics``, evaluate.py:119-317): q(z|x) of the whole data
  set through the native encoder, then the marginal and conditional entropy estimators -- N x D x 10 000 Gaussian
  log-densities + logsumexp per entropy, the kernel family of the beta-TCVAE estimator -- as ONE HIP launch sequence per
  entropy (``dvae_latent_entropy``) instead of 1 000 Python-driven [N, D, 10] torch chunks.  The few-element table
  arithmetic (mutual information, sort, MIG, AAM) stays on the host as in the reference.
"""
import logging
import math
import os
from collections import defaultdict
from functools import reduce
from timeit import default_timer

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr
from .engine import _stream
from .models.losses import FactorKLoss
from .utils.modelIO import save_metadata

TEST_LOSSES_FILE = "test_losses.log"
METRICS_FILENAME = "metrics.log"
METRIC_HELPERS_FILE = "metric_helpers.pth"


class Evaluator:
    def __init__(self, model, loss_f, device=torch.device("cpu"), logger=logging.getLogger(__name__),
                 save_dir="results", is_progress_bar=True, reference_early_return=False):
        self.device = device
        self.loss_f = loss_f
        self.model = model.to(self.device)
        self.logger = logger
        self.save_dir = save_dir
        self.is_progress_bar = is_progress_bar
        self.reference_early_return = reference_early_return
        self.logger.info("Testing Device: {}".format(self.device))

    def __call__(self, data_loader, is_metrics=False, is_losses=True):
        """evaluate.py:60-95."""
        start = default_timer()
        is_still_training = self.model.training
        self.model.eval()
        metric, losses = None, None
        if is_metrics:
            self.logger.info('Computing metrics...')
            metrics = self.compute_metrics(data_loader)
            self.logger.info('Losses: {}'.format(metrics))
            os.makedirs(self.save_dir, exist_ok=True)
            save_metadata(metrics, self.save_dir, filename=METRICS_FILENAME)
        if is_losses:
            self.logger.info('Computing losses...')
            losses = self.compute_losses(data_loader)
            self.logger.info('Losses: {}'.format(losses))
            os.makedirs(self.save_dir, exist_ok=True)
            save_metadata(losses, self.save_dir, filename=TEST_LOSSES_FILE)
        if is_still_training:
            self.model.train()
        self.logger.info('Finished evaluating after {:.1f} min.'.format((default_timer() - start) / 60))
        return metric, losses

    def compute_losses(self, dataloader):
        """evaluate.py:97-117."""
        storer = defaultdict(list)
        n = len(dataloader)
        for data, _ in dataloader:
            data = data.to(self.device)
            if isinstance(self.loss_f, FactorKLoss):
                self.loss_f.call_optimize(data, self.model, None, storer)     # evaluate.py:112-114
            else:
                self.loss_f.fused_step(data, self.model, None, storer)
            if self.reference_early_return:
                return {k: sum(v) / n for k, v in storer.items()}
        return {k: sum(v) / len(v) for k, v in storer.items()}

    # ------------------------------------------------------------------ MIG / AAM (evaluate.py:119-317)
    def compute_metrics(self, dataloader, sample_idx=None, n_samples=10000):
        """evaluate.py:119-158.  ``dataloader.dataset`` must expose ``lat_sizes`` / ``lat_names`` (data with known,
        balanced factors of variation, e.g. dSprites) and the loader must iterate the data set in factor order
        (``shuffle=False``), as the reference requires.  sample_idx: optional iterator of injected ``randperm`` draws
        (parity tests): first the marginal one, then one per (factor, value) in loop order."""
        try:
            lat_sizes = dataloader.dataset.lat_sizes
            lat_names = dataloader.dataset.lat_names
        except AttributeError:
            raise ValueError("Dataset needs to have known true factors of variations to compute the metric. This does not "
                             "seem to be the case for {}".format(type(dataloader.__dict__["dataset"]).__name__))
        draws = iter(sample_idx) if sample_idx is not None else None
        self.logger.info("Computing the empirical distribution q(z|x).")
        samples_zCx, params_zCx = self._compute_q_zCx(dataloader)
        len_dataset, latent_dim = samples_zCx.shape
        self.logger.info("Estimating the marginal entropy.")
        H_z = self._estimate_latent_entropies(samples_zCx, params_zCx, n_samples=n_samples,
                                              sample_idx=None if draws is None else next(draws))
        samples_zCx = samples_zCx.view(*lat_sizes, latent_dim)
        params_zCx = tuple(p.view(*lat_sizes, latent_dim) for p in params_zCx)
        H_zCv = self._estimate_H_zCv(samples_zCx, params_zCx, lat_sizes, lat_names, n_samples=n_samples, draws=draws)
        H_z, H_zCv = H_z.cpu(), H_zCv.cpu()
        # I[z_j;v_k] = -H[z_j|v_k] + H[z_j]   (evaluate.py:147-149)
        mut_info = -H_zCv + H_z
        sorted_mut_info = torch.sort(mut_info, dim=1, descending=True)[0].clamp(min=0)
        metric_helpers = {'marginal_entropies': H_z, 'cond_entropies': H_zCv}
        mig = self._mutual_information_gap(sorted_mut_info, lat_sizes, storer=metric_helpers)
        aam = self._axis_aligned_metric(sorted_mut_info, storer=metric_helpers)
        metrics = {'MIG': mig.item(), 'AAM': aam.item()}
        os.makedirs(self.save_dir, exist_ok=True)
        torch.save(metric_helpers, os.path.join(self.save_dir, METRIC_HELPERS_FILE))
        return metrics

    def _mutual_information_gap(self, sorted_mut_info, lat_sizes, storer=None):
        """evaluate.py:160-180 (balanced factors: H(v_k) = log |V_k|)."""
        delta_mut_info = sorted_mut_info[:, 0] - sorted_mut_info[:, 1]
        H_v = torch.from_numpy(np.asarray(lat_sizes)).float().log()
        mig_k = delta_mut_info / H_v
        mig = mig_k.mean()
        if storer is not None:
            storer["mig_k"] = mig_k
            storer["mig"] = mig
        return mig

    def _axis_aligned_metric(self, sorted_mut_info, storer=None):
        """evaluate.py:182-194."""
        numerator = (sorted_mut_info[:, 0] - sorted_mut_info[:, 1:].sum(dim=1)).clamp(min=0)
        aam_k = numerator / sorted_mut_info[:, 0]
        aam_k[torch.isnan(aam_k)] = 0
        aam = aam_k.mean()
        if storer is not None:
            storer["aam_k"] = aam_k
            storer["aam"] = aam
        return aam

    def _compute_q_zCx(self, dataloader):
        """evaluate.py:196-231: (mean, logvar) of q(z|x) for every x, through the native encoder; the model is in eval
        mode here (Evaluator.__call__), so the "sample" of q(z|x) is its mean (vae.py:69-71)."""
        len_dataset = len(dataloader.dataset)
        latent_dim = self.model.latent_dim
        mean = torch.zeros(len_dataset, latent_dim, device=self.device)
        logvar = torch.zeros(len_dataset, latent_dim, device=self.device)
        n = 0
        with torch.no_grad():
            for x, _label in dataloader:
                batch_size = x.size(0)
                mean[n:n + batch_size], logvar[n:n + batch_size] = self.model.encoder(x.to(self.device))
                n += batch_size
        samples_zCx = self.model.reparameterize(mean, logvar)
        return samples_zCx, (mean, logvar)

    def _estimate_latent_entropies(self, samples_zCx, params_zCX, n_samples=10000, sample_idx=None):
        """evaluate.py:233-297 on the device: H(z_j) = E_z[-log q(z_j)], q(z_j) = 1/N sum_n q(z_j|x_n).
        The reference re-views the gathered [n_samples, D] samples as [D, n_samples] (:262, a reshape): the gathered
        buffer is handed to the kernel as that [D, S] image, which reproduces it exactly."""
        len_dataset, latent_dim = samples_zCx.shape
        device = samples_zCx.device
        if sample_idx is None:
            sample_idx = torch.randperm(len_dataset, device=device)[:n_samples]                     # :259
        sample_idx = sample_idx.to(device)[:n_samples]
        if sample_idx.numel() != n_samples:
            raise RuntimeError("shape '[{}, {}]' is invalid for input of size {}".format(      # what .view raises at :262
                latent_dim, n_samples, sample_idx.numel() * latent_dim))
        z_ds = samples_zCx.index_select(0, sample_idx).contiguous()                                 # memory = the [D, S] view
        mean, logvar = params_zCX[0].contiguous(), params_zCX[1].contiguous()
        n_ws = _lib.lib().dvae_latent_entropy_ws_floats(len_dataset, latent_dim, n_samples)
        ws = getattr(self, "_metric_ws", None)
        if ws is None or ws.numel() < n_ws or ws.device != device:
            ws = self._metric_ws = torch.empty(n_ws, dtype=torch.float32, device=device)
        H_z = torch.empty(latent_dim, dtype=torch.float32, device=device)
        call("dvae_latent_entropy", ptr(z_ds), ptr(mean), ptr(logvar), len_dataset, latent_dim, n_samples, ptr(ws), ptr(H_z),
             _stream())
        return H_z

    def _estimate_H_zCv(self, samples_zCx, params_zCx, lat_sizes, lat_names, n_samples=10000, draws=None):
        """evaluate.py:299-317: conditional entropies H[z|v], averaged over the values of every factor."""
        latent_dim = samples_zCx.size(-1)
        len_dataset = reduce((lambda x, y: x * y), [int(k) for k in lat_sizes])
        H_zCv = torch.zeros(len(lat_sizes), latent_dim, device=self.device)
        for i_fac_var, (lat_size, lat_name) in enumerate(zip(lat_sizes, lat_names)):
            lat_size = int(lat_size)
            idcs = [slice(None)] * len(lat_sizes)
            for i in range(lat_size):
                self.logger.info("Estimating conditional entropies for the {}th value of {}.".format(i, lat_name))
                idcs[i_fac_var] = i
                sl = tuple(idcs)
                samples_zxCv = samples_zCx[sl].contiguous().view(len_dataset // lat_size, latent_dim)
                params_zxCv = tuple(p[sl].contiguous().view(len_dataset // lat_size, latent_dim) for p in params_zCx)
                H_zCv[i_fac_var] += self._estimate_latent_entropies(
                    samples_zxCv, params_zxCv, n_samples=n_samples,
                    sample_idx=None if draws is None else next(draws)) / lat_size
        return H_zCv
