"""FactorVAE discriminator (disvae/models/discriminator.py): 6-layer MLP
latent_dim -> 1000 x5 -> 2 with LeakyReLU(0.2); forward / backward are chains of the fp32
MFMA GEMM kernel of libdvae_hip.so.  Parameters live in one flat arena like the VAE's."""
from collections import OrderedDict

import torch
from torch import nn

from ..engine import _stream
from ..engine import ParamArena
from .. import _lib
from .._lib import call, ptr, ACT_NONE, ACT_LEAKY02
from ..utils.initialization import reference_init_


class _Layer(nn.Module):
    def __init__(self, weight, bias):
        super().__init__()
        self.weight = nn.Parameter(weight)
        self.bias = nn.Parameter(bias)


class Discriminator(nn.Module):
    def __init__(self, neg_slope=0.2, latent_dim=10, hidden_units=1000):
        super().__init__()
        if neg_slope != 0.2:
            raise ValueError("the HIP epilogue implements LeakyReLU(0.2) only (discriminator.py:11)")
        self.neg_slope = neg_slope
        self.z_dim = latent_dim
        self.hidden_units = hidden_units
        dims = [latent_dim] + [hidden_units] * 5 + [2]
        self.dims = dims
        shapes = OrderedDict()
        for i in range(6):
            shapes["lin%d.weight" % (i + 1)] = (dims[i + 1], dims[i])
            shapes["lin%d.bias" % (i + 1)] = (dims[i + 1],)
        self._arena = ParamArena(shapes, "cpu")
        self._layer_names = ["lin%d" % (i + 1) for i in range(6)]
        for n in self._layer_names:
            self.add_module(n, _Layer(self._arena.view(n + ".weight"), self._arena.view(n + ".bias")))
        self._acts = {}
        self.reset_parameters()

    def reset_parameters(self):
        """discriminator.py:72-73 (weights_init => kaiming_uniform 'relu' on every Linear)."""
        dev = self._arena.flat.device
        if dev.type != "cpu":
            self._move(torch.device("cpu"))
        reference_init_(self._arena, self._layer_names)
        if dev.type != "cpu":
            self._move(dev)

    def _move(self, device):
        self._arena.to(device)
        for n in self._layer_names:
            layer = getattr(self, n)
            layer.weight.data = self._arena.view(n + ".weight")
            layer.bias.data = self._arena.view(n + ".bias")
        self._acts = {}

    def _apply(self, fn, *args, **kwargs):
        new_flat = fn(self._arena.flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("the HIP engine computes in fp32 only")
        if new_flat.device != self._arena.flat.device:
            self._move(new_flat.device)
        return self

    @property
    def arena(self):
        return self._arena

    def assign_grads(self):
        """Point every Parameter.grad at its slice of the gradient arena (views built once)."""
        cache = getattr(self, "_grad_views", None)
        if cache is None or cache[0] is not self._arena.grad:
            pairs = []
            for n in self._layer_names:
                layer = getattr(self, n)
                pairs.append((layer.weight, self._arena.view(n + ".weight", grad=True)))
                pairs.append((layer.bias, self._arena.view(n + ".bias", grad=True)))
            cache = self._grad_views = (self._arena.grad, pairs)
        pairs = cache[1]
        if pairs[0][0].grad is pairs[0][1] and pairs[-1][0].grad is pairs[-1][1]:
            return
        for p_, g_ in pairs:
            p_.grad = g_

    # ---- raw (non-autograd) engine used by FactorKLoss ------------------------------------
    def _act_buffers(self, M):
        b = self._acts.get(M)
        if b is None:
            _lib.note_alloc()
            dev = self._arena.flat.device
            f = lambda n: torch.empty(M, n, dtype=torch.float32, device=dev)
            b = dict(h=[f(self.dims[i + 1]) for i in range(6)],           # outputs of lin1..lin6
                     g=[f(self.dims[i]) for i in range(6)],               # grads w.r.t. inputs of lin1..lin6
                     g2=[f(self.dims[i]) for i in range(6)])              # second (dgrad-only) chain
            self._acts[M] = b
        return b

    def _ws(self, side=False):
        """Split-K workspace of the GEMM launches (one per stream: launches of different streams run concurrently).
        side: False = the caller's stream, True = the engine's side stream, "aux" = its third stream."""
        name = "_wsbuf_aux" if side == "aux" else ("_wsbuf_side" if side else "_wsbuf")
        if getattr(self, name, None) is None or getattr(self, name).device != self._arena.flat.device:
            _lib.note_alloc()
            setattr(self, name, torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), dtype=torch.float32,
                                            device=self._arena.flat.device))
        return getattr(self, name)

    def _fresh_buffers(self, M):
        """Activation / gradient buffers owned by ONE forward call (the autograd path: several forwards of the same batch
        size may be alive at once, losses.py:261,292)."""
        dev = self._arena.flat.device
        f = lambda n: torch.empty(M, n, dtype=torch.float32, device=dev)
        return dict(h=[f(self.dims[i + 1]) for i in range(6)], g=[f(self.dims[i]) for i in range(6)])

    def forward_raw(self, z, M, acts=None):
        """z[M,latent] -> logits[M,2] (discriminator.py:60-70); activations kept for backward (in the module's workspace
        for this batch size, or in `acts`)."""
        if self._arena.flat.device.type != "cuda":
            raise _lib.DvaeHipError("the native Discriminator computes only on an MI355X (no CPU fallback)")
        s = _stream()
        b = self._act_buffers(M) if acts is None else acts
        x = z
        for i, n in enumerate(self._layer_names):
            act = ACT_LEAKY02 if i < 5 else ACT_NONE
            call("dvae_linear_fwd", ptr(x), ptr(self._arena.view(n + ".weight")), ptr(self._arena.view(n + ".bias")),
                 ptr(b["h"][i]), M, self.dims[i], self.dims[i + 1], act, ptr(self._ws()), s)
            x = b["h"][i]
        return x

    def backward_raw(self, z, g_logits, M, rows=None, wgrad=True, chain="g", acts=None, side=None, stream=None, ws=False):
        """Back-propagate g_logits[rows,2] through the MLP evaluated by forward_raw(z, M).
        rows < M restricts to the first `rows` samples (dgrad-only chain of quirk Q1).
        Returns the gradient w.r.t. z ([rows, latent]).
        stream / ws: raw HIP stream to enqueue on instead of the current one, with the split-K workspace that belongs to it
        (_ws's argument) -- FactorKLoss runs its second, input-gradient-only chain beside the first one.
        side: a VAEEngine -- the six weight gradients are then launched on ITS side stream, after ONE fork behind the chain of
        input gradients (every operand -- the saved activations, this chain's gradients -- is final by then): the weight
        gradients are only due at the end of the iteration, the chain's result is the next thing the critical path needs
        (163 us of weight gradients per step at 2048 rows: profiles/r04_final2_factor_celeba_timeline.md)."""
        s = _stream() if stream is None else stream
        b = self._act_buffers(M) if acts is None else acts
        R = M if rows is None else rows
        on_side = wgrad and side is not None and not side.single_stream
        wsp = ptr(self._ws(ws))
        dy = g_logits
        dys = []
        for i in range(5, -1, -1):
            n = self._layer_names[i]
            x_in = z if i == 0 else b["h"][i - 1]
            if wgrad and not on_side:
                call("dvae_linear_wgrad", ptr(x_in), ptr(dy), ptr(self._arena.view(n + ".weight", grad=True)),
                     ptr(self._arena.view(n + ".bias", grad=True)), R, self.dims[i], self.dims[i + 1], wsp, s)
            dys.append((i, x_in, dy))
            gx = b[chain][i]
            call("dvae_linear_dgrad", ptr(dy), ptr(self._arena.view(n + ".weight")), None if i == 0 else ptr(x_in),
                 ACT_LEAKY02 if i > 0 else ACT_NONE, ptr(gx), R, self.dims[i], self.dims[i + 1], wsp, s)
            dy = gx
        if on_side:
            side.fork_side()
            ss = side._side_raw()
            for i, x_in, dy_i in dys:
                n = self._layer_names[i]
                call("dvae_linear_wgrad", ptr(x_in), ptr(dy_i), ptr(self._arena.view(n + ".weight", grad=True)),
                     ptr(self._arena.view(n + ".bias", grad=True)), R, self.dims[i], self.dims[i + 1], ptr(self._ws(side=True)), ss)
        return dy

    def param_list(self):
        out = []
        for n in self._layer_names:
            layer = getattr(self, n)
            out += [layer.weight, layer.bias]
        return out

    def unalias_grads(self):
        """After FactorKLoss.call_optimize ``Parameter.grad`` IS a view of the gradient arena (assign_grads); the
        autograd-compatible backward writes its results there and hands autograd clones, which autograd would then
        accumulate into the very buffer just written.  Give every aliased ``.grad`` its own storage first."""
        lo = self._arena.grad.data_ptr()
        hi = lo + self._arena.grad.numel() * 4
        for p_ in self.parameters():
            g = p_.grad
            if g is not None and lo <= g.data_ptr() < hi:
                p_.grad = g.clone()

    def forward(self, z):
        """discriminator.py:60-70 as an nn.Module call: logits[M,2], differentiable w.r.t. ``z`` and the parameters
        (user code that back-propagates through ``loss_f.discriminator(z)``, e.g. the reference's own
        FactorKLoss.call_optimize, losses.py:261-306: two forwards of the same batch size, both back-propagated).  Every
        call owns its activations."""
        return _DiscFn.apply(self, z, *self.param_list())


class _DiscFn(torch.autograd.Function):
    """Discriminator.forward with autograd: forward_raw / backward_raw (the GEMM chains of libdvae_hip.so)."""

    @staticmethod
    def forward(ctx, disc, z, *params):
        if z.dtype != torch.float32 or z.device != disc.arena.flat.device:
            raise _lib.DvaeHipError("Discriminator input must be fp32 on %s" % disc.arena.flat.device)
        z = z.contiguous()
        M = z.shape[0]
        if not any(ctx.needs_input_grad):
            # inference (torch.no_grad(), evaluation): nothing is kept for a backward pass -- the module's shared workspace,
            # no per-call activation / gradient buffers
            return disc.forward_raw(z, M).clone()
        acts = disc._fresh_buffers(M)
        logits = disc.forward_raw(z, M, acts=acts).clone()
        ctx.disc, ctx.M, ctx.acts = disc, M, acts
        ctx.version = disc.arena.flat._version      # backward reads the CURRENT weights: they must not change in between
        ctx.save_for_backward(z)
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        disc, M = ctx.disc, ctx.M
        (z,) = ctx.saved_tensors
        if disc.arena.flat._version != ctx.version:
            raise RuntimeError("the discriminator's parameters were modified in place (optimizer step, load_state_dict) between "
                               "this forward and its backward: the gradients would be computed with the new weights")
        disc.unalias_grads()
        dz = disc.backward_raw(z, g_logits.contiguous(), M, wgrad=True, chain="g", acts=ctx.acts)
        grads = []
        for n in disc._layer_names:
            grads.append(disc.arena.view(n + ".weight", grad=True).clone())
            grads.append(disc.arena.view(n + ".bias", grad=True).clone())
        return (None, dz.clone()) + tuple(grads)
