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
        """Mutual Information Gap and Axis Alignment Metric of the model on a data set with known, balanced factors of
        variation (evaluate.py:119-158): ``dataloader.dataset`` must expose ``lat_sizes`` / ``lat_names`` and the loader
        must iterate the data set in factor order (``shuffle=False``), as the reference requires.  Returns
        ``{'MIG': ..., 'AAM': ...}`` and writes the reference's ``metric_helpers.pth``.
        sample_idx: optional sequence of injected ``randperm`` draws (parity tests): the marginal one first, then one per
        (factor, value) in the reference's loop order."""
        ds = dataloader.dataset
        if not (hasattr(ds, "lat_sizes") and hasattr(ds, "lat_names")):
            raise ValueError("Dataset needs to have known true factors of variations to compute the metric. This does not "
                             "seem to be the case for {}".format(type(ds).__name__))
        lat_sizes = [int(k) for k in ds.lat_sizes]
        draws = iter(sample_idx) if sample_idx is not None else None
        take = (lambda: None) if draws is None else (lambda: next(draws))
        self.logger.info("Computing the empirical distribution q(z|x).")
        table = _LatentTable(*self._encode_dataset(dataloader))
        if table.n != int(np.prod(lat_sizes)):
            raise ValueError("data set of %d images does not enumerate lat_sizes=%s" % (table.n, lat_sizes))
        self.logger.info("Estimating the marginal entropy.")
        H_z = self._entropies(table, None, n_samples, take())                      # H[z_j]                 [D]
        H_zCv = torch.zeros(len(lat_sizes), table.dim, device=self.device)         # H[z_j | v_k]           [K, D]
        for k, (size, name) in enumerate(zip(lat_sizes, ds.lat_names)):
            for value in range(size):
                self.logger.info("Estimating conditional entropies for the {}th value of {}.".format(value, name))
                rows = table.rows_where(lat_sizes, k, value)
                H_zCv[k] += self._entropies(table, rows, n_samples, take()) / size
        scores = disentanglement_scores(H_z.cpu(), H_zCv.cpu(), lat_sizes)
        os.makedirs(self.save_dir, exist_ok=True)
        torch.save(scores, os.path.join(self.save_dir, METRIC_HELPERS_FILE))        # same keys as evaluate.py:151-156
        return {'MIG': scores["mig"].item(), 'AAM': scores["aam"].item()}

    def _encode_dataset(self, dataloader):
        """(mean, logvar) of q(z|x) for every image of the data set, [N, D] each on the device, through the native encoder
        (evaluate.py:196-231; in eval mode -- Evaluator.__call__ -- the reference's "sample" of q(z|x) is its mean,
        vae.py:69-71, so the table of means doubles as the table of samples)."""
        chunks = []
        with torch.no_grad():
            for x, _label in dataloader:
                chunks.append(self.model.encoder(x.to(self.device)))
        return torch.cat([m for m, _ in chunks]), torch.cat([lv for _, lv in chunks])

    def _entropies(self, table, rows, n_samples, draw):
        """H[z_j] = E_z[-log q(z_j)] with q(z_j) = 1/N sum_n q(z_j | x_n) over the `rows` of the table (None: all of it),
        estimated on n_samples of its own samples (evaluate.py:233-297), as one dvae_latent_entropy launch sequence.
        draw: the ``randperm(N)[:n_samples]`` of evaluate.py:259, or None to draw it here.  The reference re-views the
        gathered [n_samples, D] block as [D, n_samples] (:262, a reshape, not a transpose): the gathered block is handed to
        the kernel as that [D, S] image, which reproduces it exactly."""
        mean, logvar = table.select(rows)
        n, dim = mean.shape
        if draw is None:
            draw = torch.randperm(n, device=mean.device)
        draw = draw.to(mean.device)[:n_samples]
        if draw.numel() != n_samples:              # the reference's .view(latent_dim, n_samples) fails the same way
            raise RuntimeError("shape '[{}, {}]' is invalid for input of size {}".format(dim, n_samples, draw.numel() * dim))
        z_ds = mean.index_select(0, draw).contiguous()
        need = _lib.lib().dvae_latent_entropy_ws_floats(n, dim, n_samples)
        ws = getattr(self, "_metric_ws", None)
        if ws is None or ws.numel() < need or ws.device != mean.device:
            ws = self._metric_ws = torch.empty(need, dtype=torch.float32, device=mean.device)
        H = torch.empty(dim, dtype=torch.float32, device=mean.device)
        call("dvae_latent_entropy", ptr(z_ds), ptr(mean), ptr(logvar), n, dim, n_samples, ptr(ws), ptr(H), _stream())
        return H


class _LatentTable:
    """q(z|x) of a whole data set: mean / logvar [N, D] on the device, rows in the data set's factor order."""

    def __init__(self, mean, logvar):
        self.mean, self.logvar = mean.contiguous(), logvar.contiguous()
        self.n, self.dim = self.mean.shape
        self._index = None

    def rows_where(self, lat_sizes, factor, value):
        """Row numbers of the images whose `factor`-th factor of variation takes its `value`-th value, in data-set order
        (= the order of the reference's ``samples_zCx[..., value, ...]`` slice flattened, evaluate.py:311-314)."""
        if self._index is None:
            self._index = torch.arange(self.n, device=self.mean.device).view(*lat_sizes)
        return self._index.select(factor, value).reshape(-1)

    def select(self, rows):
        if rows is None:
            return self.mean, self.logvar
        return self.mean.index_select(0, rows), self.logvar.index_select(0, rows)


def disentanglement_scores(H_z, H_zCv, lat_sizes):
    """MIG (evaluate.py:160-180) and AAM (:182-194) from the marginal entropies H_z [D] and the conditional entropies
    H_zCv [K, D] of K balanced factors of variation (H[v_k] = log |V_k|).  I[z_j; v_k] = H[z_j] - H[z_j | v_k], negative
    estimates count as 0; per factor: MIG_k = (largest - second largest information) / H[v_k], AAM_k = max(0, largest -
    all the others) / largest (0 where no latent carries information).  Returns the reference's metric_helpers dict."""
    info = (H_z.unsqueeze(0) - H_zCv).clamp(min=0)                 # [K, D]
    ranked = torch.sort(info, dim=1, descending=True)[0]
    best, runner_up = ranked[:, 0], ranked[:, 1] if ranked.shape[1] > 1 else torch.zeros_like(ranked[:, 0])
    mig_k = (best - runner_up) / torch.tensor([float(k) for k in lat_sizes]).log()
    others = ranked[:, 1:].sum(dim=1)
    aam_k = torch.where(best > 0, (best - others).clamp(min=0) / best, torch.zeros_like(best))
    return {"marginal_entropies": H_z, "cond_entropies": H_zCv, "mig_k": mig_k, "mig": mig_k.mean(),
            "aam_k": aam_k, "aam": aam_k.mean()}
