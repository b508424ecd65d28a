"""CPU oracle for the Burgess-VAE training step (TEST INFRASTRUCTURE ONLY).

This module is a functional CPU restatement (torch CPU tensors, fp32 by default, fp64 on
request) of the reference's hot path.  It is *the checker*, never the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The shipped engine (``disvae_amd``) never imports
anything from ``oracle/`` and has no CPU fallback.

Parity status: **pinned**.  ``tests/golden/make_golden.py`` imports the real reference
from ``/root/reference`` (read-only) in the build container, runs its own
``Trainer._train_iteration`` / loss functions with recorded noise and commits the
resulting vectors under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
this restatement against every one of them (plus the RNG-free known-answer vectors of
SURVEY.md section 8c).  The arithmetic primitives themselves (conv, linear, logsumexp,
BCE, Adam) live in PyTorch (un-vendored third-party dependency of the reference,
``requirements.txt:1`` unpinned; this image has torch 2.10.0) and are called, not
re-derived.

Every function cites the reference file:line (relative to /root/reference) it follows.
"""
import math

import numpy as np
from collections import OrderedDict

import torch
import torch.nn.functional as F

LOSSES = ["VAE", "betaH", "betaB", "factor", "btcvae"]  # disvae/models/losses.py:17
RECON_DIST = ["bernoulli", "laplace", "gaussian"]        # disvae/models/losses.py:18

HID_CHANNELS = 32   # disvae/models/encoders.py:45
KERNEL_SIZE = 4     # disvae/models/encoders.py:46
HIDDEN_DIM = 256    # disvae/models/encoders.py:47
DISC_HIDDEN = 1000  # disvae/models/discriminator.py:13
DISC_SLOPE = 0.2    # disvae/models/discriminator.py:11


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def vae_param_shapes(img_size, latent_dim=10):
    """Ordered name -> shape of the Burgess VAE state_dict.

    Follows the registration order of disvae/models/encoders.py:54-67 and
    disvae/models/decoders.py:53-65 (note convT_64 is registered before convT1).
    """
    c, h, w = img_size
    if [h, w] not in ([32, 32], [64, 64]):
        raise RuntimeError("{} sized images not supported".format(img_size))  # vae.py:41-42
    is64 = (h == 64)
    hc, k, hd = HID_CHANNELS, KERNEL_SIZE, HIDDEN_DIM
    shapes = OrderedDict()

    def add(name, wshape, nb):
        shapes[name + ".weight"] = tuple(wshape)
        shapes[name + ".bias"] = (nb,)

    add("encoder.conv1", (hc, c, k, k), hc)
    add("encoder.conv2", (hc, hc, k, k), hc)
    add("encoder.conv3", (hc, hc, k, k), hc)
    if is64:
        add("encoder.conv_64", (hc, hc, k, k), hc)
    add("encoder.lin1", (hd, hc * k * k), hd)
    add("encoder.lin2", (hd, hd), hd)
    add("encoder.mu_logvar_gen", (2 * latent_dim, hd), 2 * latent_dim)
    add("decoder.lin1", (hd, latent_dim), hd)
    add("decoder.lin2", (hd, hd), hd)
    add("decoder.lin3", (hc * k * k, hd), hc * k * k)
    if is64:
        add("decoder.convT_64", (hc, hc, k, k), hc)
    add("decoder.convT1", (hc, hc, k, k), hc)
    add("decoder.convT2", (hc, hc, k, k), hc)
    add("decoder.convT3", (hc, c, k, k), c)
    return shapes


def disc_param_shapes(latent_dim=10, hidden=DISC_HIDDEN):
    """disvae/models/discriminator.py:51-56."""
    shapes = OrderedDict()
    dims = [latent_dim] + [hidden] * 5 + [2]
    for i in range(6):
        shapes["lin%d.weight" % (i + 1)] = (dims[i + 1], dims[i])
        shapes["lin%d.bias" % (i + 1)] = (dims[i + 1],)
    return shapes


def _default_then_kaiming(shapes, transposed_names=()):
    """Draw parameters exactly like the reference does with the global torch RNG.

    1. construction: every nn.Conv2d / nn.ConvTranspose2d / nn.Linear draws its default
       init in registration order (weight: kaiming_uniform_(a=sqrt(5)), bias:
       U(+-1/sqrt(fan_in)) -- torch.nn.modules.{conv,linear}.reset_parameters);
    2. ``self.apply(weights_init)`` (vae.py:87-88, discriminator.py:72-73) then re-draws
       every weight with kaiming_uniform_(nonlinearity='relu') in the same order
       (initialization.py:33-61); biases keep the draw of step 1.
    fan_in is weight.size(1) * receptive field, also for ConvTranspose2d
    (initialization.py uses nn.init, which does not special-case transposed convs).
    """
    params = OrderedDict()
    names = [n[:-len(".weight")] for n in shapes if n.endswith(".weight")]
    for n in names:
        w = torch.empty(shapes[n + ".weight"])
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        fan_in = w.size(1) * (w[0][0].numel() if w.dim() > 2 else 1)
        if n in transposed_names:
            # nn.ConvTranspose2d: _calculate_fan_in_and_fan_out also uses size(1)*rf
            pass
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        b = torch.empty(shapes[n + ".bias"])
        torch.nn.init.uniform_(b, -bound, bound)
        params[n + ".weight"] = w
        params[n + ".bias"] = b
    for n in names:
        torch.nn.init.kaiming_uniform_(params[n + ".weight"], nonlinearity="relu")
    return params


def init_vae_params(img_size, latent_dim=10):
    """Parameters of ``init_specific_model('Burgess', img_size, latent_dim)`` (vae.py:15-26)
    drawn from the *current* global torch RNG state (call torch.manual_seed first)."""
    return _default_then_kaiming(vae_param_shapes(img_size, latent_dim))


def init_disc_params(latent_dim=10):
    """Parameters of ``Discriminator(latent_dim=...)`` (discriminator.py:9-58)."""
    return _default_then_kaiming(disc_param_shapes(latent_dim))


def clone_params(params, dtype=None, requires_grad=False):
    out = OrderedDict()
    for k, v in params.items():
        t = v.detach().clone()
        if dtype is not None:
            t = t.to(dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


# --------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------
# Activation gates (test infrastructure for comparing fp32 and fp64 evaluations of the SAME network):
# torch.relu / F.leaky_relu are discontinuous in their derivative at 0, so a unit whose pre-activation is
# within fp32 rounding of zero can be "on" in one arithmetic and "off" in the other, which changes
# gradients by a finite amount.  ``with gates(masks):`` makes the forward functions below use the
# given on/off pattern (name -> list of boolean tensors in reference layout, consumed in call order)
# instead of the sign of their own pre-activation; ``with gates(None, record=log):`` leaves the
# arithmetic untouched and appends (name, pre-activation) to ``log``.  Default: plain relu / leaky_relu.
_GATES = None
_GATE_LOG = None


class gates:
    def __init__(self, masks, record=None):
        self.masks = None if masks is None else {k: list(v) for k, v in masks.items()}
        self.record = record

    def __enter__(self):
        global _GATES, _GATE_LOG
        self._old = (_GATES, _GATE_LOG)
        _GATES, _GATE_LOG = self.masks, self.record
        return self

    def __exit__(self, *exc):
        global _GATES, _GATE_LOG
        _GATES, _GATE_LOG = self._old
        return False


def _act(pre, name, slope=0.0):
    """relu (slope 0) / leaky_relu of layer `name`, honouring an active ``gates`` context."""
    if _GATE_LOG is not None:
        _GATE_LOG.append((name, pre.detach()))
    if _GATES is not None and _GATES.get(name):
        g = _GATES[name].pop(0).to(pre.device)
        assert g.shape == pre.shape, (name, g.shape, pre.shape)
        return pre * torch.where(g, torch.ones((), dtype=pre.dtype), torch.full((), slope, dtype=pre.dtype))
    return torch.relu(pre) if slope == 0.0 else F.leaky_relu(pre, slope)


def encoder_forward(p, x, want_acts=False):
    """EncoderBurgess.forward, disvae/models/encoders.py:69-89."""
    acts = OrderedDict()
    h = x
    names = ["conv1", "conv2", "conv3"] + (["conv_64"] if "encoder.conv_64.weight" in p else [])
    for n in names:
        h = _act(F.conv2d(h, p["encoder.%s.weight" % n], p["encoder.%s.bias" % n],
                          stride=2, padding=1), "encoder." + n)   # encoders.py:73-77
        acts["encoder." + n] = h
    h = h.reshape(x.size(0), -1)                                 # encoders.py:80 (c,h,w order)
    h = _act(F.linear(h, p["encoder.lin1.weight"], p["encoder.lin1.bias"]), "encoder.lin1")  # :81
    acts["encoder.lin1"] = h
    h = _act(F.linear(h, p["encoder.lin2.weight"], p["encoder.lin2.bias"]), "encoder.lin2")  # :82
    acts["encoder.lin2"] = h
    ml = F.linear(h, p["encoder.mu_logvar_gen.weight"], p["encoder.mu_logvar_gen.bias"])  # :86
    acts["encoder.mu_logvar_gen"] = ml
    latent_dim = ml.size(1) // 2
    mu, logvar = ml.view(-1, latent_dim, 2).unbind(-1)           # :87 (interleaved, quirk Q5)
    if want_acts:
        return mu, logvar, acts
    return mu, logvar


def reparameterize(mu, logvar, eps):
    """VAE.reparameterize, disvae/models/vae.py:52-71.  ``eps`` is the injected N(0,1)
    draw (the reference calls torch.randn_like(std), vae.py:67); ``eps=None`` is eval
    mode (z = mean, vae.py:69-71)."""
    if eps is None:
        return mu
    std = torch.exp(0.5 * logvar)
    return mu + std * eps


def decoder_forward(p, z, want_acts=False):
    """DecoderBurgess.forward, disvae/models/decoders.py:67-84."""
    acts = OrderedDict()
    h = z
    for n in ["lin1", "lin2", "lin3"]:
        h = _act(F.linear(h, p["decoder.%s.weight" % n], p["decoder.%s.bias" % n]), "decoder." + n)  # :71-73
        acts["decoder." + n] = h
    h = h.view(z.size(0), HID_CHANNELS, KERNEL_SIZE, KERNEL_SIZE)                            # :74
    names = (["convT_64"] if "decoder.convT_64.weight" in p else []) + ["convT1", "convT2"]
    for n in names:
        h = _act(F.conv_transpose2d(h, p["decoder.%s.weight" % n], p["decoder.%s.bias" % n],
                                    stride=2, padding=1), "decoder." + n)                   # :77-80
        acts["decoder." + n] = h
    h = torch.sigmoid(F.conv_transpose2d(h, p["decoder.convT3.weight"], p["decoder.convT3.bias"],
                                         stride=2, padding=1))                               # :82
    acts["decoder.convT3"] = h
    if want_acts:
        return h, acts
    return h


def vae_forward(p, x, eps):
    """VAE.forward, disvae/models/vae.py:73-85 -> (recon, (mu, logvar), z)."""
    mu, logvar = encoder_forward(p, x)
    z = reparameterize(mu, logvar, eps)
    recon = decoder_forward(p, z)
    return recon, (mu, logvar), z


def discriminator_forward(dp, z):
    """Discriminator.forward, disvae/models/discriminator.py:60-70."""
    h = z
    for i in range(1, 6):
        h = _act(F.linear(h, dp["lin%d.weight" % i], dp["lin%d.bias" % i]), "disc.lin%d" % i, DISC_SLOPE)
    return F.linear(h, dp["lin6.weight"], dp["lin6.bias"])


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def linear_annealing(init, fin, step, annealing_steps):
    """disvae/models/losses.py:511-518."""
    if annealing_steps == 0:
        return fin
    assert fin > init
    delta = fin - init
    return min(init + delta * step / annealing_steps, fin)


def reconstruction_loss(data, recon, distribution="bernoulli"):
    """_reconstruction_loss, disvae/models/losses.py:394-449 (sum over pixels / batch)."""
    batch_size = recon.size(0)
    if distribution == "bernoulli":
        loss = F.binary_cross_entropy(recon, data, reduction="sum")               # :430
    elif distribution == "gaussian":
        loss = F.mse_loss(recon * 255, data * 255, reduction="sum") / 255          # :433
    elif distribution == "laplace":
        loss = F.l1_loss(recon, data, reduction="sum")                             # :437
        loss = loss * 3                                                             # :438
        loss = loss * (loss != 0)                                                   # :439
    else:
        raise ValueError("Unkown distribution: {}".format(distribution))           # :442
    return loss / batch_size                                                        # :444


def kl_normal_loss(mu, logvar):
    """_kl_normal_loss, disvae/models/losses.py:452-480 -> (total, per-dim)."""
    latent_kl = 0.5 * (-1 - logvar + mu.pow(2) + logvar.exp()).mean(dim=0)         # :472
    return latent_kl.sum(), latent_kl                                              # :473


def log_density_gaussian(x, mu, logvar):
    """disvae/utils/math.py:34-51."""
    normalization = -0.5 * (math.log(2 * math.pi) + logvar)
    inv_var = torch.exp(-logvar)
    return normalization - 0.5 * ((x - mu) ** 2 * inv_var)


def log_importance_weight_matrix(batch_size, dataset_size, dtype=torch.float32):
    """disvae/utils/math.py:54-73 -- including the flat-stride quirk: with
    M = B-1 the strided writes use step M+1 == B, i.e. they hit COLUMN 0 and COLUMN 1 of
    every row, not the diagonal."""
    N = dataset_size
    M = batch_size - 1
    strat_weight = (N - M) / (N * M)
    W = torch.empty(batch_size, batch_size, dtype=torch.float32).fill_(1 / M)
    W.view(-1)[::M + 1] = 1 / N
    W.view(-1)[1::M + 1] = strat_weight
    W[M - 1, 0] = strat_weight
    return W.log().to(dtype)


def btcvae_log_densities(z, mu, logvar, n_data, is_mss=True):
    """_get_log_pz_qz_prodzi_qzCx, disvae/models/losses.py:523-544."""
    B, D = z.shape
    log_q_zCx = log_density_gaussian(z, mu, logvar).sum(dim=1)                     # :527
    zeros = torch.zeros_like(z)
    log_pz = log_density_gaussian(z, zeros, zeros).sum(1)                          # :531-532
    mat = log_density_gaussian(z.view(B, 1, D), mu.view(1, B, D), logvar.view(1, B, D))  # :534
    if is_mss:
        log_iw = log_importance_weight_matrix(B, n_data, dtype=z.dtype)            # :538
        mat = mat + log_iw.view(B, B, 1)                                           # :539 (Q3)
    log_qz = torch.logsumexp(mat.sum(2), dim=1)                                    # :541
    log_prod_qzi = torch.logsumexp(mat, dim=1).sum(1)                              # :542
    return log_pz, log_qz, log_prod_qzi, log_q_zCx


def btcvae_terms(z, mu, logvar, n_data, is_mss=True):
    """mi / tc / dw_kl batch means, disvae/models/losses.py:368-373."""
    log_pz, log_qz, log_prod_qzi, log_q_zCx = btcvae_log_densities(z, mu, logvar, n_data, is_mss)
    mi = (log_q_zCx - log_qz).mean()
    tc = (log_qz - log_prod_qzi).mean()
    dw_kl = (log_prod_qzi - log_pz).mean()
    return mi, tc, dw_kl


def permute_dims(z, perms):
    """_permute_dims, disvae/models/losses.py:483-508; ``perms[d]`` is the injected
    torch.randperm(B) of latent dimension d (the reference draws it on the CPU, :505)."""
    out = torch.zeros_like(z)
    for d in range(z.size(1)):
        out[:, d] = z[perms[d], d]
    return out


class LossState:
    """Host-side state of BaseLoss (disvae/models/losses.py:71-75,105-114)."""

    def __init__(self, record_loss_every=50, rec_dist="bernoulli", steps_anneal=0):
        self.n_train_steps = 0
        self.record_loss_every = record_loss_every
        self.rec_dist = rec_dist
        self.steps_anneal = steps_anneal

    def pre_call(self, is_train):
        if is_train:
            self.n_train_steps += 1
        return (not is_train) or (self.n_train_steps % self.record_loss_every == 1)


def single_optimizer_loss(name, hp, state, data, recon, mu, logvar, z, is_train=True):
    """Loss value + logged scalars of BetaHLoss/BetaBLoss/BtcvaeLoss.__call__
    (losses.py:139-153, 186-202, 356-391).  ``hp`` holds the hyper-parameters consumed by
    get_loss_f (losses.py:22-46).  Returns (loss, dict_of_scalars, keep_storer)."""
    keep = state.pre_call(is_train)
    logs = OrderedDict()
    rec = reconstruction_loss(data, recon, state.rec_dist)
    logs["recon_loss"] = rec
    if name in ("VAE", "betaH"):
        beta = 1 if name == "VAE" else hp["betaH_B"]
        kl, kl_i = kl_normal_loss(mu, logvar)
        anneal = linear_annealing(0, 1, state.n_train_steps, state.steps_anneal) if is_train else 1
        loss = rec + anneal * (beta * kl)                                          # :149
        logs["kl_loss"] = kl
        for i in range(kl_i.numel()):
            logs["kl_loss_%d" % i] = kl_i[i]
        logs["loss"] = loss
    elif name == "betaB":
        kl, kl_i = kl_normal_loss(mu, logvar)
        C = (linear_annealing(hp["betaB_initC"], hp["betaB_finC"], state.n_train_steps,
                              state.steps_anneal) if is_train else hp["betaB_finC"])  # :194-195
        loss = rec + hp["betaB_G"] * (kl - C).abs()                                # :197
        logs["kl_loss"] = kl
        for i in range(kl_i.numel()):
            logs["kl_loss_%d" % i] = kl_i[i]
        logs["loss"] = loss
    elif name == "btcvae":
        mi, tc, dw_kl = btcvae_terms(z, mu, logvar, hp["n_data"], hp.get("is_mss", True))
        anneal = linear_annealing(0, 1, state.n_train_steps, state.steps_anneal) if is_train else 1
        loss = rec + (hp["btcvae_A"] * mi + hp["btcvae_B"] * tc + anneal * hp["btcvae_G"] * dw_kl)  # :379-381
        logs["loss"] = loss
        logs["mi_loss"] = mi
        logs["tc_loss"] = tc
        logs["dw_kl_loss"] = dw_kl
        kl, kl_i = kl_normal_loss(mu, logvar)                                      # :389 (logging only)
        logs["kl_loss"] = kl
        for i in range(kl_i.numel()):
            logs["kl_loss_%d" % i] = kl_i[i]
    else:
        raise ValueError("Uknown loss : {}".format(name))
    return loss, logs, keep


# --------------------------------------------------------------------------------------
# whole training iterations (loss + gradients), noise injected
# --------------------------------------------------------------------------------------
def train_iteration_grads(name, hp, state, params, data, eps):
    """Forward + loss + backward of Trainer._train_iteration for the single-optimizer
    losses (disvae/training.py:152-158): returns (loss, logs, grads, outs).  ``params``
    must be leaf tensors with requires_grad=True; the optimizer step is the caller's."""
    recon, (mu, logvar), z = vae_forward(params, data, eps)
    loss, logs, keep = single_optimizer_loss(name, hp, state, data, recon, mu, logvar, z, True)
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    grads = OrderedDict((k, g) for k, g in zip(params.keys(), grads))
    outs = dict(recon=recon.detach(), mu=mu.detach(), logvar=logvar.detach(), z=z.detach())
    return loss.detach(), OrderedDict((k, v.detach()) for k, v in logs.items()), grads, outs


def factor_iteration_grads(hp, state, params, dparams, data, eps1, eps2, perms):
    """FactorKLoss.call_optimize in train mode (disvae/models/losses.py:243-313) up to (not
    including) the two optimizer steps.  Reproduces quirk Q1: d_z is not detached, so
    d_tc_loss.backward() adds d[0.5*CE(D(z1),0)]/d(theta_enc) on top of the vae_loss
    gradients; discriminator gradients come only from d_tc_loss (optimizer_d.zero_grad at
    :303 discards those of vae_loss.backward).

    eps1 / eps2: injected N(0,1) draws for model(data1) (:254) and sample_latent(data2)
    (:286); perms: list of D index tensors for _permute_dims.  The wasted full-batch
    forward of training.py:153 (quirk Q4) only consumes RNG and is not restated.
    Returns (vae_loss, logs, vae_grads, disc_grads, outs)."""
    state.pre_call(True)                                                           # :244
    half = data.size(0) // 2
    data1, data2 = data.split(half)[:2]                                            # :247-251
    recon, (mu, logvar), z1 = vae_forward(params, data1, eps1)                     # :254
    rec = reconstruction_loss(data1, recon, state.rec_dist)                        # :255-257
    kl, kl_i = kl_normal_loss(mu, logvar)                                          # :259
    d_z = discriminator_forward(dparams, z1)                                       # :261
    tc = (d_z[:, 0] - d_z[:, 1]).mean()                                            # :265
    anneal = linear_annealing(0, 1, state.n_train_steps, state.steps_anneal)      # :268
    vae_loss = rec + kl + anneal * hp["factor_G"] * tc                             # :270
    plist = list(params.values())
    dlist = list(dparams.values())
    g_vae = torch.autograd.grad(vae_loss, plist, retain_graph=True, allow_unused=True)  # :282
    mu2, logvar2 = encoder_forward(params, data2)                                  # :286
    z2 = reparameterize(mu2, logvar2, eps2)
    z_perm = permute_dims(z2, perms).detach()                                      # :287
    d_z_perm = discriminator_forward(dparams, z_perm)                              # :288
    ones = torch.ones(half, dtype=torch.long)
    zeros = torch.zeros_like(ones)
    d_tc = 0.5 * (F.cross_entropy(d_z, zeros) + F.cross_entropy(d_z_perm, ones))  # :295
    g_d = torch.autograd.grad(d_tc, plist + dlist, allow_unused=True)              # :304 (leaks into encoder)
    vae_grads = OrderedDict()
    for i, k in enumerate(params.keys()):
        g = g_vae[i]
        extra = g_d[i]
        if g is None:
            g = torch.zeros_like(plist[i])
        if extra is not None:
            g = g + extra
        vae_grads[k] = g
    disc_grads = OrderedDict((k, g) for k, g in zip(dparams.keys(), g_d[len(plist):]))
    logs = OrderedDict(recon_loss=rec.detach(), kl_loss=kl.detach())
    for i in range(kl_i.numel()):
        logs["kl_loss_%d" % i] = kl_i[i].detach()
    logs["loss"] = vae_loss.detach()
    logs["tc_loss"] = tc.detach()
    logs["discrim_loss"] = d_tc.detach()
    outs = dict(recon=recon.detach(), mu=mu.detach(), logvar=logvar.detach(), z1=z1.detach(),
                z2=z2.detach(), z_perm=z_perm.detach(), d_z=d_z.detach(), d_z_perm=d_z_perm.detach())
    return vae_loss.detach(), logs, vae_grads, disc_grads, outs


class OracleTrainer:
    """Minimal CPU Trainer used for the ``cpu_baseline`` leg of bench.py and for
    trajectory tests: Trainer._train_iteration (disvae/training.py:137-164) with
    torch.optim.Adam (main.py:208, losses.py:238), noise drawn from the torch CPU
    generator in the reference's order (quirk Q4) unless injected."""

    def __init__(self, loss_name, hp, img_size, latent_dim=10, lr=5e-4, lr_disc=5e-5,
                 rec_dist="bernoulli", steps_anneal=0, params=None, dparams=None):
        self.loss_name = loss_name
        self.hp = dict(hp)
        self.latent_dim = latent_dim
        self.state = LossState(rec_dist=rec_dist, steps_anneal=steps_anneal)
        self.params = clone_params(params if params is not None else init_vae_params(img_size, latent_dim),
                                   requires_grad=True)
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr)
        if loss_name == "factor":
            self.dparams = clone_params(dparams if dparams is not None else init_disc_params(latent_dim),
                                        requires_grad=True)
            self.opt_d = torch.optim.Adam(list(self.dparams.values()), lr=lr_disc, betas=(0.5, 0.9))

    def train_iteration(self, data, eps=None, eps2=None, perms=None):
        B = data.size(0)
        D = self.latent_dim
        if self.loss_name == "factor":
            half = B // 2
            if eps is None:
                # training.py:153 runs a full-batch model(data) (autograd graph included) before
                # FactorKLoss.__call__ raises ValueError (quirk Q4): its work and its N(0,1) draw are
                # part of what the reference's CPU iteration costs, so the timed oracle does them too
                _wasted = vae_forward(self.params, data, torch.randn(B, D))
                del _wasted
                eps = torch.randn(half, D)
                eps2 = torch.randn(half, D)
                perms = [torch.randperm(half) for _ in range(D)]
            loss, logs, g, gd, _ = factor_iteration_grads(self.hp, self.state, self.params, self.dparams,
                                                          data, eps, eps2, perms)
            for p_, k in zip(self.params.values(), g):
                p_.grad = g[k]
            for p_, k in zip(self.dparams.values(), gd):
                p_.grad = gd[k]
            self.opt.step()
            self.opt_d.step()
        else:
            if eps is None:
                eps = torch.randn(B, D)
            loss, logs, g, _ = train_iteration_grads(self.loss_name, self.hp, self.state, self.params, data, eps)
            for p_, k in zip(self.params.values(), g):
                p_.grad = g[k]
            self.opt.step()
        return float(loss), logs


# --------------------------------------------------------------------------------------
# MIG / AAM disentanglement metrics (disvae/evaluate.py:119-317) -- SURVEY 8 f-4
# --------------------------------------------------------------------------------------
def estimate_latent_entropies(samples_zCx, mean, logvar, sample_idx, n_samples, mini_batch_size=10):
    """Evaluator._estimate_latent_entropies (evaluate.py:233-297).  ``sample_idx`` = the
    ``torch.randperm(len_dataset)[:n_samples]`` draw (:259), injected.  Quirk reproduced: the [n_samples, D] gather is
    re-VIEWED as [D, n_samples] (:262) -- a reshape, not a transpose."""
    N, D = samples_zCx.shape
    z = samples_zCx.index_select(0, sample_idx).view(D, n_samples)                     # :259-262
    H = torch.zeros(D, dtype=samples_zCx.dtype)
    log_N = math.log(N)
    m3, l3 = mean.unsqueeze(-1), logvar.unsqueeze(-1)
    for k in range(0, n_samples, mini_batch_size):                                     # :270
        zc = z[:, k:k + mini_batch_size].unsqueeze(0)                                  # [1, D, mb] broadcast over N
        log_q_zCx = log_density_gaussian(zc, m3, l3)                                   # :273-275 -> [N, D, mb]
        log_q_z = -log_N + torch.logsumexp(log_q_zCx, dim=0)                           # :282
        H += (-log_q_z).sum(1)                                                         # :285
    return H / n_samples                                                               # :289


def estimate_H_zCv(samples_zCx, mean, logvar, lat_sizes, sample_idx_list, n_samples):
    """Evaluator._estimate_H_zCv (evaluate.py:299-317); inputs shaped [*lat_sizes, D]; one injected randperm per
    (factor, value) in the reference's loop order."""
    D = samples_zCx.size(-1)
    N = 1
    for k in lat_sizes:
        N *= int(k)
    H = torch.zeros(len(lat_sizes), D, dtype=samples_zCx.dtype)
    it = iter(sample_idx_list)
    for f, lat_size in enumerate(lat_sizes):
        lat_size = int(lat_size)
        for i in range(lat_size):
            idcs = [slice(None)] * len(lat_sizes)
            idcs[f] = i
            sl = tuple(idcs)
            s_ = samples_zCx[sl].contiguous().view(N // lat_size, D)                    # :310
            m_ = mean[sl].contiguous().view(N // lat_size, D)
            l_ = logvar[sl].contiguous().view(N // lat_size, D)
            H[f] += estimate_latent_entropies(s_, m_, l_, next(it), n_samples) / lat_size   # :315
    return H


def mutual_information_gap(sorted_mut_info, lat_sizes):
    """Evaluator._mutual_information_gap (evaluate.py:160-180)."""
    delta = sorted_mut_info[:, 0] - sorted_mut_info[:, 1]
    H_v = torch.as_tensor(np.asarray(lat_sizes)).float().log()
    return (delta / H_v).mean()


def axis_aligned_metric(sorted_mut_info):
    """Evaluator._axis_aligned_metric (evaluate.py:182-194)."""
    num = (sorted_mut_info[:, 0] - sorted_mut_info[:, 1:].sum(dim=1)).clamp(min=0)
    aam_k = num / sorted_mut_info[:, 0]
    aam_k[torch.isnan(aam_k)] = 0
    return aam_k.mean()


def metrics_from_entropies(H_z, H_zCv, lat_sizes):
    """evaluate.py:148-157: mutual information table -> (MIG, AAM, sorted table)."""
    mut_info = -H_zCv + H_z
    sorted_mi = torch.sort(mut_info, dim=1, descending=True)[0].clamp(min=0)
    return mutual_information_gap(sorted_mi, lat_sizes), axis_aligned_metric(sorted_mi), sorted_mi
