"""Trainer with the API of disvae/training.py:17-164 driving the native (HIP) model + losses.

``_train_iteration`` keeps the reference's contract (returns a python float).  When the
model is a native ``disvae_amd`` VAE on an MI355X and the loss is a native plugin, the
iteration is ONE stream of kernel launches (forward + loss + backward in libdvae_hip.so,
then ``optimizer.step()``) instead of ~600 dispatched ATen ops; otherwise the generic
reference control flow (model -> loss -> zero_grad -> backward -> step, ValueError ->
call_optimize) is used, which also works with the native model through its autograd
wrappers.  ``_train_epoch`` defers the per-iteration ``loss.item()`` host sync to the end of
the epoch unless a progress bar needs the value.
"""
import logging
import os
from collections import defaultdict
from timeit import default_timer

import torch
from tqdm import trange

from .models.vae import VAE
from .models.losses import BaseLoss, FactorKLoss
from .utils.modelIO import save_model

TRAIN_LOSSES_LOGFILE = "train_losses.log"


class Trainer():
    def __init__(self, model, optimizer, loss_f, device=torch.device("cpu"), logger=logging.getLogger(__name__),
                 save_dir="results", gif_visualizer=None, is_progress_bar=True, replay=None):
        """``replay="plan"`` re-issues the device side of the native iteration from a recorded
        launch list, ``replay="graph"`` from a hipGraph (disvae_amd/graph.py): worthwhile below
        ~512 images per GPU, where issuing the ~75 launches from Python takes longer than the GPU
        needs to run them.  ``False`` forces the eager path; ``None`` keeps the loss's setting."""
        self.device = device
        self.model = model.to(self.device)
        self.loss_f = loss_f
        self.optimizer = optimizer
        self.save_dir = save_dir
        self.is_progress_bar = is_progress_bar
        self.logger = logger
        self.losses_logger = LossesLogger(os.path.join(self.save_dir, TRAIN_LOSSES_LOGFILE))
        self.gif_visualizer = gif_visualizer
        if replay is not None and isinstance(loss_f, BaseLoss):
            loss_f.replay = replay or None
        self.logger.info("Training Device: {}".format(self.device))

    def __call__(self, data_loader, epochs=10, checkpoint_every=10):
        """training.py:64-102."""
        start = default_timer()
        self.model.train()
        for epoch in range(epochs):
            storer = defaultdict(list)
            mean_epoch_loss = self._train_epoch(data_loader, storer, epoch)
            self.logger.info('Epoch: {} Average loss per image: {:.2f}'.format(epoch + 1, mean_epoch_loss))
            self.losses_logger.log(epoch, storer)
            if self.gif_visualizer is not None:
                self.gif_visualizer()
            if epoch % checkpoint_every == 0:
                save_model(self.model, self.save_dir, filename="model-{}.pt".format(epoch))
        if self.gif_visualizer is not None:
            self.gif_visualizer.save_reset()
        self.model.eval()
        delta_time = (default_timer() - start) / 60
        self.logger.info('Finished training after {:.1f} min.'.format(delta_time))

    def _train_epoch(self, data_loader, storer, epoch):
        """training.py:104-135; the epoch loss is accumulated on the device when no progress
        bar needs per-iteration values (one host sync per epoch instead of one per step)."""
        kwargs = dict(desc="Epoch {}".format(epoch + 1), leave=False, disable=not self.is_progress_bar)
        defer = (not self.is_progress_bar) and self._is_native()
        epoch_loss = 0.
        dev_losses = []
        with trange(len(data_loader), **kwargs) as t:
            for _, (data, _) in enumerate(data_loader):
                if defer:
                    dev_losses.append(self._train_iteration_async(data, storer).clone())
                else:
                    iter_loss = self._train_iteration(data, storer)
                    epoch_loss += iter_loss
                    t.set_postfix(loss=iter_loss)
                t.update()
        if defer and dev_losses:
            epoch_loss = float(torch.stack(dev_losses).sum().item())
        return epoch_loss / len(data_loader)

    def _is_native(self):
        return (isinstance(self.model, VAE) and isinstance(self.loss_f, BaseLoss)
                and next(self.model.parameters()).device.type == "cuda")

    def _train_iteration_async(self, data, storer):
        """One native training iteration; returns the loss as a 0-d DEVICE tensor (no sync)."""
        data = data.to(self.device, non_blocking=True)
        if isinstance(self.loss_f, FactorKLoss):
            if self.model.training:
                # the reference runs (and discards) a full-batch forward before the ValueError
                # (training.py:153): keep its N(0,1) draw so the device RNG stream matches (Q4)
                torch.randn(data.shape[0], self.model.latent_dim, dtype=torch.float32, device=data.device)
            return self.loss_f.call_optimize(data, self.model, self.optimizer, storer)
        return self.loss_f.fused_step(data, self.model, self.optimizer, storer)

    def _train_iteration(self, data, storer):
        """training.py:137-164."""
        if self._is_native():
            return self._train_iteration_async(data, storer).item()
        batch_size, channel, height, width = data.size()
        data = data.to(self.device)
        try:
            recon_batch, latent_dist, latent_sample = self.model(data)
            loss = self.loss_f(data, recon_batch, latent_dist, self.model.training, storer,
                               latent_sample=latent_sample)
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
        except ValueError:
            loss = self.loss_f.call_optimize(data, self.model, self.optimizer, storer)
        return loss.item()


class LossesLogger(object):
    """training.py:167-190: CSV 'Epoch,Loss,Value' through the process-global logger."""

    def __init__(self, file_path_name):
        if os.path.isfile(file_path_name):
            os.remove(file_path_name)
        os.makedirs(os.path.dirname(file_path_name) or ".", exist_ok=True)
        self.logger = logging.getLogger("losses_logger")
        self.logger.setLevel(1)
        file_handler = logging.FileHandler(file_path_name)
        file_handler.setLevel(1)
        self.logger.addHandler(file_handler)
        header = ",".join(["Epoch", "Loss", "Value"])
        self.logger.debug(header)

    def log(self, epoch, losses_storer):
        for k, v in losses_storer.items():
            log_string = ",".join(str(item) for item in [epoch, k, mean(v)])
            self.logger.debug(log_string)


def mean(l):
    return sum(l) / len(l)
