"""Test-loss evaluation with the API of disvae/evaluate.py:60-117 (``Evaluator.compute_losses``)
on the native model: eval-mode forward (z = mean, vae.py:69-71) + the loss plugins with
``is_train=False`` (storer always kept, annealing = 1, losses.py:109,146-147), one stream of HIP
launches per batch and ONE device->host copy for all logged scalars.

Reference quirk: ``compute_losses`` returns from inside its batch loop
(evaluate.py:116-117), i.e. it evaluates only the FIRST batch and divides by the number of batches.
Here every batch is evaluated and the mean is returned; ``reference_early_return=True`` reproduces
the reference's numbers.  The MIG / AAM disentanglement metrics (evaluate.py:119-317) are offline
analysis outside the training-step hot path and are not provided.
"""
import logging
import os
from collections import defaultdict
from timeit import default_timer

import torch

from .models.losses import FactorKLoss
from .utils.modelIO import save_metadata

TEST_LOSSES_FILE = "test_losses.log"


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
            raise NotImplementedError("MIG / AAM metrics are outside the accelerated training-step path")
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
