"""Native (HIP) VAE with the API of disvae/models/vae.py: ``init_specific_model``, ``VAE`` with
``forward -> (recon, (mu, logvar), z)``, ``reparameterize``, ``sample_latent``,
``reset_parameters`` and the reference's state_dict names/shapes.

Parameters live in one flat fp32 arena (``engine.ParamArena``); the ``nn.Parameter`` objects
handed to ``torch.optim.Adam`` are views into it, their ``.grad`` are views into one flat
gradient arena that the HIP backward fills (and a single RCCL all-reduce covers).
All network arithmetic runs in libdvae_hip.so; there is no PyTorch/CPU fallback for it.
"""
import torch
from torch import nn

from ..engine import _stream
from ..engine import VAEEngine, ParamArena, vae_param_shapes
from .. import _lib
from .._lib import call, ptr
from ..utils.initialization import reference_init_
from .._debug import knob

MODELS = ["Burgess"]  # disvae/models/vae.py:12


def init_specific_model(model_type, img_size, latent_dim):
    """disvae/models/vae.py:15-26."""
    model_type = model_type.lower().capitalize()
    if model_type not in MODELS:
        err = "Unkown model_type={}. Possible values: {}"
        raise ValueError(err.format(model_type, MODELS))
    model = VAE(img_size, None, None, latent_dim)
    model.model_type = model_type
    return model


class _Layer(nn.Module):
    """Holder of one layer's weight/bias views (keeps the reference's state_dict keys)."""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight = nn.Parameter(weight)
        self.bias = nn.Parameter(bias)


class _Half(nn.Module):
    """``model.encoder`` / ``model.decoder``: callable like the reference's sub-modules."""

    def __init__(self, vae, which):
        super().__init__()
        object.__setattr__(self, "_vae", vae)   # not registered: avoid a module cycle
        self._which = which

    def forward(self, t):
        if self._which == "encoder":
            mu, logvar = _EncodeFn.apply(self._vae, t, *self._vae.param_list())
            return mu, logvar
        return _DecodeFn.apply(self._vae, t, *self._vae.param_list())


class VAE(nn.Module):
    def __init__(self, img_size, encoder=None, decoder=None, latent_dim=10):
        """disvae/models/vae.py:30-50.  ``encoder`` / ``decoder`` class arguments of the
        reference are accepted and ignored (the only model family is Burgess)."""
        super().__init__()
        if list(img_size[1:]) not in [[32, 32], [64, 64]]:
            raise RuntimeError("{} sized images not supported. Only (None, 32, 32) and (None, 64, 64) supported. "
                               "Build your own architecture or reshape images!".format(img_size))
        # any dimension, like main.py:81: the fused kernels cover 1..16 (_lib.MAX_LATENT_DIM), above that the engine runs the FC
        # layers one launch each and the latent / loss kernels their run-time-D forms (csrc/latent_wide.hip)
        if not (isinstance(latent_dim, int) and not isinstance(latent_dim, bool) and latent_dim >= 1):
            raise ValueError("latent_dim={!r}: expected a positive integer".format(latent_dim))
        self.latent_dim = latent_dim
        self.img_size = tuple(img_size)
        self.num_pixels = self.img_size[1] * self.img_size[2]
        self.model_type = "Burgess"
        shapes = vae_param_shapes(self.img_size, latent_dim)
        self._arena = ParamArena(shapes, "cpu")
        self._layer_names = [k[:-len(".weight")] for k in shapes if k.endswith(".weight")]
        self.encoder = _Half(self, "encoder")
        self.decoder = _Half(self, "decoder")
        for name in self._layer_names:
            half, lname = name.split(".")
            getattr(self, half).add_module(lname, _Layer(self._arena.view(name + ".weight"),
                                                         self._arena.view(name + ".bias")))
        self._engine = None
        self._fwd_version = 0
        # the arena as a few (unregistered) 1-D chunk Parameters: see flat_parameters()
        object.__setattr__(self, "_flat_params", [nn.Parameter(c) for c in self._arena_chunks()])
        self.reset_parameters()

    # ---- parameter plumbing -------------------------------------------------------------
    def reset_parameters(self):
        """vae.py:87-88 -- same RNG stream as the reference (CPU generator)."""
        dev = self._arena.flat.device
        if dev.type != "cpu":
            self._move(torch.device("cpu"))
        reference_init_(self._arena, self._layer_names)
        if dev.type != "cpu":
            self._move(dev)

    def _rebind(self):
        for name in self._layer_names:
            half, lname = name.split(".")
            layer = getattr(getattr(self, half), lname)
            layer.weight.data = self._arena.view(name + ".weight")
            layer.bias.data = self._arena.view(name + ".bias")
        for p_, c in zip(self._flat_params, self._arena_chunks()):
            p_.data = c
        self._engine = None

    def _move(self, device):
        self._arena.to(device)
        self._rebind()

    def _apply(self, fn, *args, **kwargs):
        # nn.Module.to()/cuda()/cpu(): move the ARENA and re-bind the parameter views so that
        # Parameter objects (already handed to an optimizer) keep their identity.
        new_flat = fn(self._arena.flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("the HIP engine computes in fp32 only")
        if new_flat.device != self._arena.flat.device:
            self._move(new_flat.device)
        return self

    def param_list(self):
        out = []
        for name in self._layer_names:
            half, lname = name.split(".")
            layer = getattr(getattr(self, half), lname)
            out += [layer.weight, layer.bias]
        return out

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        # load_state_dict copies into param.data (the arena views): nothing else to do
        return r

    @property
    def arena(self):
        return self._arena

    @property
    def engine(self):
        if self._arena.flat.device.type != "cuda":
            raise _lib.DvaeHipError("the native VAE computes only on an MI355X (model is on %s); move it with "
                                    ".to('cuda') -- there is no CPU fallback" % self._arena.flat.device)
        if self._engine is None:
            self._engine = VAEEngine(self.img_size, self.latent_dim, self._arena)
        return self._engine

    FLAT_CHUNK = int(knob("DVAE_FLAT_CHUNK", 16384))

    def _arena_chunks(self, grad=False):
        buf = self._arena.grad if grad else self._arena.flat
        return list(torch.split(buf, self.FLAT_CHUNK))

    def flat_parameters(self):
        """The parameter arena as equal 1-D chunk Parameters (same storage as ``parameters()``).
        ``torch.optim.Adam(model.flat_parameters(), ...)`` performs exactly the same element-wise
        update as ``Adam(model.parameters(), ...)`` (the 16-byte alignment padding has zero
        gradient and stays zero); the uniform 8192-element chunks give torch's fused multi-tensor
        Adam kernel ~60 equally sized workgroups instead of 8 large ones.
        Only the native training step (``loss_f.fused_step`` / ``call_optimize``) fills the chunks' ``.grad``: the
        autograd-compatible path (``model(x)`` ... ``loss.backward()``) delivers gradients to the layer Parameters and
        refuses to run once the chunks have been handed out (an optimizer over them would step with stale gradients)."""
        self._flat_handed_out = True
        return list(self._flat_params)

    def assign_grads(self):
        """Point every Parameter.grad at its slice of the flat gradient arena.  The views are
        built once per arena placement; afterwards this is two identity checks per step."""
        cache = getattr(self, "_grad_views", None)
        if cache is None or cache[0] is not self._arena.grad:
            pairs = list(zip(self._flat_params, self._arena_chunks(grad=True)))
            for name in self._layer_names:
                half, lname = name.split(".")
                layer = getattr(getattr(self, half), lname)
                pairs.append((layer.weight, self._arena.view(name + ".weight", grad=True)))
                pairs.append((layer.bias, self._arena.view(name + ".bias", grad=True)))
            cache = self._grad_views = (self._arena.grad, pairs)
        pairs = cache[1]
        if pairs[0][0].grad is pairs[0][1] and pairs[-1][0].grad is pairs[-1][1]:
            return
        for p_, g_ in pairs:
            p_.grad = g_

    def unalias_grads(self):
        """The autograd-compatible backward writes its results into the flat gradient arena and hands autograd CLONES
        of it.  After a fused step ``Parameter.grad`` IS a view of that arena (assign_grads): autograd would then
        accumulate the clone into the very buffer the kernels just wrote (doubling the gradient under
        ``zero_grad(set_to_none=False)``).  Give every aliased ``.grad`` its own storage first (values kept)."""
        if getattr(self, "_flat_handed_out", False):
            raise _lib.DvaeHipError("this model's flat_parameters() were handed to an optimizer: autograd delivers gradients "
                                    "to model.parameters() only, so optimizer.step() would use stale gradients -- build "
                                    "the optimizer on model.parameters() to use loss.backward(), or train through the "
                                    "native step (Trainer / loss_f.fused_step / call_optimize)")
        lo = self._arena.grad.data_ptr()
        hi = lo + self._arena.grad.numel() * 4
        for p_ in list(self.parameters()) + list(self._flat_params):
            g = p_.grad
            if g is not None and lo <= g.data_ptr() < hi:
                p_.grad = g.clone()

    # ---- reference API --------------------------------------------------------------------
    def reparameterize(self, mean, logvar):
        """vae.py:52-71 (stand-alone use by callers; forward() fuses it into a HIP kernel)."""
        if self.training:
            std = torch.exp(0.5 * logvar)
            eps = torch.randn_like(std)
            return mean + std * eps
        return mean

    def forward(self, x, eps=None):
        """vae.py:73-85.  ``eps`` optionally injects the N(0,1) draw (parity tests); by
        default it is drawn on the device like the reference's torch.randn_like."""
        if self.training and eps is None:
            eps = torch.randn(x.shape[0], self.latent_dim, dtype=torch.float32, device=x.device)
        if not self.training:
            eps = None
        recon, mu, logvar, z = _VAEFn.apply(self, x, eps, True, *self.param_list())
        return recon, (mu, logvar), z

    def sample_latent(self, x, eps=None):
        """vae.py:90-101."""
        if self.training and eps is None:
            eps = torch.randn(x.shape[0], self.latent_dim, dtype=torch.float32, device=x.device)
        if not self.training:
            eps = None
        _, _, _, z = _VAEFn.apply(self, x, eps, False, *self.param_list())
        return z


def _check_input(model, x):
    if x.device != model.arena.flat.device:
        raise _lib.DvaeHipError("input must be on %s" % model.arena.flat.device)
    if x.dtype == torch.uint8:                 # pixel batch: ToTensor on the device (the fused step skips even this pass)
        x = x.contiguous()
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        call("dvae_u8_to_f32", ptr(x), ptr(out), x.numel(), _stream())
        return out
    if x.dtype != torch.float32:
        raise _lib.DvaeHipError("input must be fp32 (or uint8 pixels) on %s" % model.arena.flat.device)
    return x.contiguous()


def _zeros_scal(dev):
    return torch.zeros(_lib.NSCAL, dtype=torch.float32, device=dev)    # (dvae_reparam_kl_bwd reads the fixed slots only)


def _param_grads(model):
    model_grads = []
    for name in model._layer_names:
        model_grads.append(model.arena.view(name + ".weight", grad=True).clone())
        model_grads.append(model.arena.view(name + ".bias", grad=True).clone())
    return model_grads


class _VAEFn(torch.autograd.Function):
    """model(x): encoder -> reparameterise -> (decoder), backward through the HIP kernels.
    One forward per batch size may be in flight (activations live in the engine workspace)."""

    @staticmethod
    def forward(ctx, model, x, eps, decode, *params):
        eng = model.engine
        x = _check_input(model, x)
        B = x.shape[0]
        buf = eng.buffers(B)
        eng.encode(x, buf)
        eng.reparam(buf, eps)
        if decode:
            eng.decode(buf.z, buf, staged=True)      # encode() above staged this pass's weight images
        model._fwd_version += 1
        ctx.model, ctx.x, ctx.eps, ctx.decode, ctx.version = model, x, eps, decode, model._fwd_version
        recon = buf.recon.clone() if decode else buf.recon.new_zeros(())
        return recon, buf.mu.clone(), buf.logvar.clone(), buf.z.clone()

    @staticmethod
    def backward(ctx, g_recon, g_mu, g_lv, g_z):
        model, x = ctx.model, ctx.x
        if model._fwd_version != ctx.version:
            raise _lib.DvaeHipError("backward through a stale forward: the engine workspace was overwritten by a "
                                    "later forward of the same model")
        eng = model.engine
        model.unalias_grads()
        B = x.shape[0]
        buf = eng.buffers(B)
        s = _stream()
        dz = g_z.contiguous() if g_z is not None else None
        if ctx.decode and g_recon is not None:
            call("dvae_sigmoid_bwd", ptr(g_recon.contiguous()), ptr(buf.recon), ptr(buf.g_logit), buf.recon.numel(), s)
            eng.decode_backward(buf.z, buf, defer_fc_wgrad=True)     # encode_backward below launches all six FC wgrads
            if dz is not None:
                call("dvae_add", ptr(buf.dz), ptr(dz), ptr(buf.dz), buf.dz.numel(), s)
            dz = buf.dz
        elif ctx.decode:
            for n in model._layer_names:
                if n.startswith("decoder."):
                    model.arena.view(n + ".weight", grad=True).zero_()
                    model.arena.view(n + ".bias", grad=True).zero_()
        else:
            for n in model._layer_names:
                if n.startswith("decoder."):
                    model.arena.view(n + ".weight", grad=True).zero_()
                    model.arena.view(n + ".bias", grad=True).zero_()
        scal = _zeros_scal(x.device)                    # KL weight 0: KL is a separate loss node here
        coef = torch.ones(_lib.NCOEF, dtype=torch.float32, device=x.device)
        gm = g_mu.contiguous() if g_mu is not None else None
        gl = g_lv.contiguous() if g_lv is not None else None
        call("dvae_reparam_kl_bwd", ptr(dz), None, None, ptr(gm), ptr(gl), ptr(buf.mu), ptr(buf.logvar), ptr(ctx.eps), ptr(scal),
             ptr(coef), ptr(buf.dml), B, model.latent_dim, s)
        eng.encode_backward(x, buf)
        return (None, None, None, None) + tuple(_param_grads(model))


class _EncodeFn(torch.autograd.Function):
    """model.encoder(x) -> (mu, logvar)  (encoders.py:69-89)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model.engine
        x = _check_input(model, x)
        buf = eng.buffers(x.shape[0])
        eng.encode(x, buf)
        eng.reparam(buf, None)
        model._fwd_version += 1
        ctx.model, ctx.x, ctx.version = model, x, model._fwd_version
        return buf.mu.clone(), buf.logvar.clone()

    @staticmethod
    def backward(ctx, g_mu, g_lv):
        model, x = ctx.model, ctx.x
        if model._fwd_version != ctx.version:
            raise _lib.DvaeHipError("backward through a stale forward")
        eng = model.engine
        model.unalias_grads()
        buf = eng.buffers(x.shape[0])
        s = _stream()
        scal = _zeros_scal(x.device)
        coef = torch.ones(_lib.NCOEF, dtype=torch.float32, device=x.device)
        gm = g_mu.contiguous() if g_mu is not None else None
        gl = g_lv.contiguous() if g_lv is not None else None
        call("dvae_reparam_kl_bwd", None, None, None, ptr(gm), ptr(gl), ptr(buf.mu), ptr(buf.logvar), None, ptr(scal), ptr(coef),
             ptr(buf.dml), x.shape[0], model.latent_dim, s)
        eng.encode_backward(x, buf)
        grads = []
        for n in model._layer_names:
            for suffix in (".weight", ".bias"):
                gv = model.arena.view(n + suffix, grad=True)
                grads.append(gv.clone() if n.startswith("encoder.") else None)
        return (None, None) + tuple(grads)


class _DecodeFn(torch.autograd.Function):
    """model.decoder(z) -> recon  (decoders.py:67-84)."""

    @staticmethod
    def forward(ctx, model, z, *params):
        eng = model.engine
        z = _check_input(model, z)
        buf = eng.buffers(z.shape[0])
        eng.decode(z, buf)
        model._fwd_version += 1
        ctx.model, ctx.z, ctx.version = model, z, model._fwd_version
        return buf.recon.clone()

    @staticmethod
    def backward(ctx, g_recon):
        model, z = ctx.model, ctx.z
        if model._fwd_version != ctx.version:
            raise _lib.DvaeHipError("backward through a stale forward")
        eng = model.engine
        model.unalias_grads()
        buf = eng.buffers(z.shape[0])
        s = _stream()
        call("dvae_sigmoid_bwd", ptr(g_recon.contiguous()), ptr(buf.recon), ptr(buf.g_logit), buf.recon.numel(), s)
        eng.decode_backward(z, buf)
        grads = []
        for n in model._layer_names:
            for suffix in (".weight", ".bias"):
                gv = model.arena.view(n + suffix, grad=True)
                grads.append(gv.clone() if n.startswith("decoder.") else None)
        return (None, buf.dz.clone()) + tuple(grads)
