"""Trainer with the API of disvae/training.py:17-164 driving the native (HIP) model + losses.

``_train_iteration`` keeps the reference's contract (returns a python float).  When the
model is a native ``disvae_amd`` VAE on an MI355X and the loss is a native plugin, the
iteration is a fixed sequence of launches from libdvae_hip.so on two HIP streams (forward + loss +
backward, then ``optimizer.step()``) instead of ~600 dispatched ATen ops; otherwise the generic
reference control flow (model -> loss -> zero_grad -> backward -> step, ValueError ->
call_optimize) is used, which also works with the native model through its autograd
wrappers.  ``_train_epoch`` defers the per-iteration ``loss.item()`` host sync to the end of
the epoch unless a progress bar needs the value.
"""
import logging
import os
import time
from collections import defaultdict

import torch
from tqdm import tqdm

from .models.vae import VAE
from .models.losses import BaseLoss, FactorKLoss
from .utils.modelIO import save_model

TRAIN_LOSSES_LOGFILE = "train_losses.log"


class Trainer():
    def __init__(self, model, optimizer, loss_f, device=torch.device("cpu"), logger=logging.getLogger(__name__),
                 save_dir="results", gif_visualizer=None, is_progress_bar=True, replay=None):
        """Same arguments as training.py:46-51, plus ``replay``: how the launches of a native iteration are
        issued -- "plan" re-issues them from a recorded launch list, "graph" from a hipGraph
        (disvae_amd/graph.py: worthwhile below ~512 images per GPU, where issuing the launches from Python
        takes longer than the GPU needs to run them), ``False`` forces the eager path, ``None`` keeps the
        loss's setting ("auto")."""
        self.device = device
        self.model = model.to(self.device)
        self.optimizer = optimizer
        self.loss_f = loss_f
        self.logger = logger
        self.save_dir = save_dir
        self.is_progress_bar = is_progress_bar
        self.gif_visualizer = gif_visualizer
        self.losses_logger = LossesLogger(os.path.join(self.save_dir, TRAIN_LOSSES_LOGFILE))
        if replay is not None and isinstance(loss_f, BaseLoss):
            loss_f.replay = replay or None
        if self._is_native() and fuse_plain_adam(self.optimizer):
            self.logger.info("optimizer: torch.optim.Adam switched to its fused multi-tensor kernel (same update)")
        self.logger.info("Training Device: {}".format(self.device))

    # ------------------------------------------------------------------ epochs (training.py:64-102)
    def __call__(self, data_loader, epochs=10, checkpoint_every=10):
        t_begin = time.perf_counter()
        self.model.train()
        for epoch in range(epochs):
            scalars = defaultdict(list)                       # filled by the loss on its logging steps
            per_image = self._train_epoch(data_loader, scalars, epoch)
            self._end_of_epoch(epoch, per_image, scalars, checkpoint_every)
        if self.gif_visualizer is not None:
            self.gif_visualizer.save_reset()
        self.model.eval()
        self.logger.info('Finished training after {:.1f} min.'.format((time.perf_counter() - t_begin) / 60))

    def _end_of_epoch(self, epoch, per_image, scalars, checkpoint_every):
        self.logger.info('Epoch: {} Average loss per image: {:.2f}'.format(epoch + 1, per_image))
        self.losses_logger.log(epoch, scalars)
        if self.gif_visualizer is not None:
            self.gif_visualizer()
        if epoch % checkpoint_every == 0:                     # epoch 0 is always saved, as in the reference
            save_model(self.model, self.save_dir, filename="model-{}.pt".format(epoch))

    def _train_epoch(self, data_loader, storer, epoch):
        """Mean loss of the epoch's iterations (training.py:104-135).  Without a progress bar the native
        path keeps the iteration losses on the device and syncs with the host ONCE per epoch."""
        n_iter = len(data_loader)
        on_device = (not self.is_progress_bar) and self._is_native()
        pending, total = [], 0.0
        bar = tqdm(total=n_iter, desc="Epoch {}".format(epoch + 1), leave=False, disable=not self.is_progress_bar)
        try:
            for batch, _labels in data_loader:
                if on_device:
                    pending.append(self._train_iteration_async(batch, storer).clone())
                else:
                    value = self._train_iteration(batch, storer)
                    total += value
                    bar.set_postfix(loss=value)
                bar.update()
        finally:
            bar.close()
        if pending:
            total = float(torch.stack(pending).sum().item())
        return total / n_iter

    # ------------------------------------------------------------------ one iteration (training.py:137-164)
    def _is_native(self):
        return (isinstance(self.model, VAE) and isinstance(self.loss_f, BaseLoss)
                and next(self.model.parameters()).device.type == "cuda")

    def _train_iteration_async(self, data, storer):
        """One native training iteration; returns the loss as a 0-d DEVICE tensor (no sync)."""
        data = data.to(self.device, non_blocking=True)
        if not isinstance(self.loss_f, FactorKLoss):
            return self.loss_f.fused_step(data, self.model, self.optimizer, storer)
        if self.model.training:
            # the reference runs (and discards) a full-batch forward before the ValueError
            # (training.py:153): keep its N(0,1) draw so the device RNG stream matches (Q4)
            torch.randn(data.shape[0], self.model.latent_dim, dtype=torch.float32, device=data.device)
        return self.loss_f.call_optimize(data, self.model, self.optimizer, storer)

    def _train_iteration(self, data, storer):
        if self._is_native():
            return self._train_iteration_async(data, storer).item()
        # any other model / loss combination: the reference's control flow, in which a loss that needs its
        # own optimisation schedule announces itself with ValueError
        data = data.to(self.device)
        try:
            recon, latent_dist, latent_sample = self.model(data)
            loss = self.loss_f(data, recon, latent_dist, self.model.training, storer, latent_sample=latent_sample)
        except ValueError:
            return self.loss_f.call_optimize(data, self.model, self.optimizer, storer).item()
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss.item()


def fuse_plain_adam(optimizer):
    """main.py:208 builds ``optim.Adam(model.parameters(), lr)``: torch then steps the 28 parameter tensors through its
    "foreach" implementation -- a dozen multi-tensor launches per step plus their host work, 0.1 ms on a 1.1 ms iteration.
    A stock Adam (exactly ``torch.optim.Adam``, no amsgrad / capturable / differentiable / explicit foreach or fused choice)
    over fp32 device tensors is switched to torch's own fused multi-tensor kernel: the same element-wise update (<= 1 ulp per
    step, tests/test_gpu_timed_config.py), ONE launch.  Param groups, hyper-parameters and ``state_dict()`` keep their stock
    layout (the state of a fused Adam loads into a foreach one and back).  Returns True when the switch was made."""
    if type(optimizer) is not torch.optim.Adam:
        return False
    params = [p_ for g in optimizer.param_groups for p_ in g["params"]]
    if not params or any(p_.device.type != "cuda" or p_.dtype != torch.float32 for p_ in params):
        return False
    for g in optimizer.param_groups:
        if g.get("fused") is not None or g.get("foreach") is not None:      # the user chose an implementation
            return False
        if g.get("amsgrad") or g.get("capturable") or g.get("differentiable") or isinstance(g.get("lr"), torch.Tensor):
            return False
    for g in optimizer.param_groups:
        g["fused"] = True
    for p_, st in optimizer.state.items():            # steps taken before the switch: the fused kernel keeps `step` on the device
        if "step" in st and (not torch.is_tensor(st["step"]) or st["step"].device != p_.device):
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p_.device)
    optimizer._step_supports_amp_scaling = True
    return True


class LossesLogger(object):
    """'Epoch,Loss,Value' CSV of the per-epoch means of the logged scalars (the file training.py:167-190
    produces).  The reference routes the lines through the process-global logger "losses_logger", which
    accumulates one FileHandler per Trainer ever constructed; here the file is simply written."""

    def __init__(self, file_path_name):
        self.path = file_path_name
        os.makedirs(os.path.dirname(file_path_name) or ".", exist_ok=True)
        with open(self.path, "w") as f:                       # truncates a log left by an earlier run
            f.write("Epoch,Loss,Value\n")

    def log(self, epoch, losses_storer):
        with open(self.path, "a") as f:
            for name, values in losses_storer.items():
                f.write("{},{},{}\n".format(epoch, name, mean(values)))


def mean(values):
    return sum(values) / len(values)
