"""Host-side orchestration of the HIP kernels for the Burgess VAE (forward + backward).

This is plumbing only: torch tensors are used as device-memory containers, every FLOP of
the network runs in libdvae_hip.so through the C-ABI (``_lib.call``).  Layout contract:
  * API boundary (input batch, reconstruction): NCHW fp32, like the reference;
  * internal conv activations: NHWC (one pixel = one 128-byte line of 32 channels);
  * the 4x4x32 tensors between the conv stack and the FC stack are kept in the reference's
    (c,h,w) flatten order (encoders.py:80, decoders.py:74) = NCHW, so that lin1 / lin3 weights
    keep their state_dict layout; the conv kernels at that end read / write NCHW directly.
Two streams: the caller's (critical path) and a side stream for everything that is only due at the
end of the backward pass (weight gradients; the loss plugins also put the B x B estimator there).
Reference being replaced: EncoderBurgess.forward (encoders.py:69-89), VAE.reparameterize
(vae.py:52-71), DecoderBurgess.forward (decoders.py:67-84) and their autograd backward
(training.py:157).
"""
import os
from collections import OrderedDict

import torch

from . import _lib
from ._lib import call, ptr, record_py, NCHW, NHWC, ACT_NONE, ACT_RELU, ACT_SIGMOID

HID = 32
HIDDEN_DIM = 256


def vae_param_shapes(img_size, latent_dim=10):
    """name -> shape in the reference's registration order (encoders.py:54-67,
    decoders.py:53-65; convT_64 precedes convT1)."""
    c, h, w = img_size
    if [h, w] not in ([32, 32], [64, 64]):
        raise RuntimeError("{} sized images not supported. Only (None, 32, 32) and (None, 64, 64) supported. "
                           "Build your own architecture or reshape images!".format(img_size))
    is64 = h == 64
    shapes = OrderedDict()

    def add(name, wshape, nb):
        shapes[name + ".weight"] = tuple(wshape)
        shapes[name + ".bias"] = (nb,)

    add("encoder.conv1", (HID, c, 4, 4), HID)
    add("encoder.conv2", (HID, HID, 4, 4), HID)
    add("encoder.conv3", (HID, HID, 4, 4), HID)
    if is64:
        add("encoder.conv_64", (HID, HID, 4, 4), HID)
    add("encoder.lin1", (HIDDEN_DIM, HID * 16), HIDDEN_DIM)
    add("encoder.lin2", (HIDDEN_DIM, HIDDEN_DIM), HIDDEN_DIM)
    add("encoder.mu_logvar_gen", (2 * latent_dim, HIDDEN_DIM), 2 * latent_dim)
    add("decoder.lin1", (HIDDEN_DIM, latent_dim), HIDDEN_DIM)
    add("decoder.lin2", (HIDDEN_DIM, HIDDEN_DIM), HIDDEN_DIM)
    add("decoder.lin3", (HID * 16, HIDDEN_DIM), HID * 16)
    if is64:
        add("decoder.convT_64", (HID, HID, 4, 4), HID)
    add("decoder.convT1", (HID, HID, 4, 4), HID)
    add("decoder.convT2", (HID, HID, 4, 4), HID)
    add("decoder.convT3", (HID, c, 4, 4), c)
    return shapes


class ParamArena:
    """All parameters of a module in ONE flat fp32 buffer (+ one flat gradient buffer):
    a single RCCL all-reduce covers every gradient, and nn.Parameter views keep the
    reference's state_dict names/shapes so torch.optim.Adam works unchanged."""

    def __init__(self, shapes, device="cpu"):
        self.shapes = OrderedDict(shapes)
        self.offsets = OrderedDict()
        off = 0
        for k, s in self.shapes.items():
            n = 1
            for d in s:
                n *= d
            self.offsets[k] = (off, n)
            off += (n + 3) // 4 * 4   # keep every tensor 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)

    def span(self, prefix, grad=True):
        """Contiguous slice of the (gradient) arena covering every tensor whose name starts with
        `prefix` (the reference's registration order keeps encoder.* and decoder.* contiguous)."""
        offs = [(o, n) for k, (o, n) in self.offsets.items() if k.startswith(prefix)]
        lo = min(o for o, _ in offs)
        hi = max(o + (n + 3) // 4 * 4 for o, n in offs)
        return (self.grad if grad else self.flat)[lo:min(hi, self.numel)]

    def view(self, name, grad=False):
        """Shaped view of one tensor of the (gradient) arena; views are cached per placement."""
        cache = self.__dict__.get("_views")
        if cache is None or cache[0] is not self.flat or cache[1] is not self.grad:
            cache = self._views = (self.flat, self.grad, {})
        v = cache[2].get((name, grad))
        if v is None:
            off, n = self.offsets[name]
            buf = self.grad if grad else self.flat
            v = cache[2][(name, grad)] = buf[off:off + n].view(self.shapes[name])
        return v

    def to(self, device):
        _lib.note_alloc()
        self.flat = self.flat.to(device)
        self.grad = self.grad.to(device)
        return self


_CONV_WGRAD_MAIN = os.environ.get("DVAE_CONV_WGRAD_MAIN", "0") == "1"
# which encoder conv weight gradients the MAIN stream computes itself at the very end of the backward pass (after conv1's),
# instead of leaving them in the side stream's queue: the side stream is the tail of the iteration (timeline:
# profiles/r02_run9_timeline.md), the main stream is idle from the end of conv1's weight gradient to the join
_TAIL_MAIN = [n_ for n_ in os.environ.get("DVAE_TAIL_MAIN", "conv3,conv_64").split(",") if n_]
# 1 = conv weight gradients leave their partial sums and ONE grouped launch reduces all layers at the end of the backward
# pass.  Measured (profiles/r02_run6_ab.txt): 8 reduce launches fewer but +1.5 % step time at B=1024 -- the per-layer
# reductions hide in the side stream, the grouped one (137 MB of partials, ~37 us) sits on the critical path -> default 0
_DEFER_REDUCE = os.environ.get("DVAE_DEFER_REDUCE", "0") == "1"


def _stream():
    """hipStream_t of torch's current stream (the raw accessor is ~20x cheaper than building a
    torch.cuda.Stream object per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


class _Buffers:
    """Activation / gradient workspace for one batch size."""

    def __init__(self, eng, B):
        _lib.note_alloc()
        dev = eng.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.B = B
        D = eng.latent_dim
        self.enc_act = [f(B, h, h, HID) for h in eng.enc_sizes]       # NHWC outputs of the conv layers
        self.enc_gact = [f(B, h, h, HID) for h in eng.enc_sizes]
        self.a_flat = f(B, HID * 16)
        self.ga_flat = f(B, HID * 16)
        self.h1, self.h2 = f(B, HIDDEN_DIM), f(B, HIDDEN_DIM)
        self.gh1, self.gh2 = f(B, HIDDEN_DIM), f(B, HIDDEN_DIM)
        self.ml, self.dml = f(B, 2 * D), f(B, 2 * D)
        self.mu, self.logvar, self.z = f(B, D), f(B, D), f(B, D)
        self.d1, self.d2, self.d3 = f(B, HIDDEN_DIM), f(B, HIDDEN_DIM), f(B, HID * 16)
        self.gd1, self.gd2, self.gd3 = f(B, HIDDEN_DIM), f(B, HIDDEN_DIM), f(B, HID * 16)
        self.dec_act = [f(B, h, h, HID) for h in eng.dec_sizes]       # NHWC outputs of the hidden convT layers
        self.dec_gact = [f(B, h, h, HID) for h in eng.dec_sizes]
        c, hh, ww = eng.img_size
        self.recon = f(B, c, hh, ww)
        self.g_logit = f(B, c, hh, ww)
        self.dz = f(B, D)


class VAEEngine:
    """Forward / backward of the Burgess VAE on one MI355X through libdvae_hip.so."""

    def __init__(self, img_size, latent_dim, arena):
        _lib.lib()  # fail loudly if the HIP library is missing
        self.img_size = tuple(img_size)
        self.latent_dim = latent_dim
        self.arena = arena
        c, h, w = self.img_size
        self.is64 = h == 64
        self.enc_names = ["conv1", "conv2", "conv3"] + (["conv_64"] if self.is64 else [])
        self.enc_sizes = [h >> (i + 1) for i in range(len(self.enc_names))]   # output H of each conv
        self.dec_names = (["convT_64"] if self.is64 else []) + ["convT1", "convT2"]
        self.dec_sizes = [8 << i for i in range(len(self.dec_names))]         # output H of each hidden convT
        self._bufs = {}
        self._ws = None
        self._ws_side = None
        self._side = None      # side HIP stream: the FC weight-gradient GEMMs run beside the dgrad chain
        # small batches: everything on the caller's stream.  Below ~256 images the iteration is bound by the latency of
        # dependent launches, a fork / join between hardware queues costs ~6 us each (5 forks + 1 join per iteration) and
        # the weight-gradient kernels that the side stream would overlap are a few microseconds long.  Set per step by the
        # loss plugins (BaseLoss._streams).
        self.single_stream = False
        self._fc_pending = []  # FC weight-gradient problems waiting for the grouped launch (decoder's, deferred)
        self._fc_descs = {}    # host descriptor arrays of the grouped launches, kept alive for recorded plans
        self._defer_reduce = False   # conv weight gradients: partial sums now, ONE grouped reduction at the end of the backward pass
        self._reduce_pending = []
        self._layer_ws = {}    # one partial-sum workspace per conv layer (deferred reductions need them all alive)

    @property
    def device(self):
        return self.arena.flat.device

    def p(self, name):
        return self.arena.view(name)

    def g(self, name):
        return self.arena.view(name, grad=True)

    def buffers(self, B):
        b = self._bufs.get(B)
        if b is None or b.recon.device != self.device:
            b = _Buffers(self, B)
            self._bufs[B] = b
        if self._ws is None or self._ws.device != self.device:
            n = _lib.lib().dvae_conv_wgrad_ws_floats()
            _lib.note_alloc()
            self._ws = torch.empty(n, dtype=torch.float32, device=self.device)
            self._ws_side = torch.empty(n, dtype=torch.float32, device=self.device)
            self._side = torch.cuda.Stream(device=self.device)
        return b

    # ---- fork / join of the side stream (weight-gradient GEMMs of the FC layers) ----------------
    def fork_side(self):
        """Order the side stream after everything enqueued so far on the current stream.  A fork
        costs the current stream ~6 us (event signal between hardware queues, profiles/r01_run19
        timeline), so the FC weight gradients fork once per chain, not once per layer."""
        if self.single_stream:
            return
        record_py(self._side.wait_stream, torch.cuda.current_stream())

    @property
    def side_stream(self):
        return torch.cuda.current_stream() if self.single_stream else self._side

    def _side_raw(self):
        return _stream() if self.single_stream else self._side.cuda_stream

    def _side_wgrad(self, x, dy, dw, db, M, K, N):
        """dw, db <- wgrad(x, dy) on the side stream (after a fork_side): overlaps with whatever the
        current stream does next."""
        call("dvae_linear_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(db), M, K, N, ptr(self._ws_side),
             self._side_raw())

    def _side_wgrad_grouped(self, problems):
        """All FC weight gradients of `problems` = [(x, dy, dw, db, M, K, N)] (tensors) in ONE launch on the side stream
        (dvae_linear_wgrad_grouped): ~400 short-lived workgroups instead of six launches that each leave most of
        the chip idle and delay the conv weight gradients queued behind them."""
        key = tuple((ptr(x), ptr(dy), ptr(dw), ptr(db), M, K, N) for x, dy, dw, db, M, K, N in problems)
        ent = self._fc_descs.get(key)
        if ent is None:
            if len(self._fc_descs) >= 64:        # recorded plans hold the addresses of these arrays: invalidate them
                self._fc_descs.clear()
                _lib.note_alloc()
            ent = self._fc_descs[key] = _lib.wgrad_descs(key)
        call("dvae_linear_wgrad_grouped", ent[1], len(problems), self._side_raw())

    def _ws_of(self, key):
        w = self._layer_ws.get(key)
        if w is None or w.device != self.device:
            _lib.note_alloc()
            w = self._layer_ws[key] = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), dtype=torch.float32, device=self.device)
        return w

    def _conv_wgrad(self, fn, *args, fork=True, main=False):
        """Conv / convT weight gradient `fn(*args, ws, stream)`: off the dgrad critical path, so it
        goes to the side stream (after a fork) and co-runs with the dgrad chain (main=True: the current stream);
        DVAE_CONV_WGRAD_MAIN=1 keeps it on the current stream.  While reductions are deferred (tuned 64x64 geometry)
        only the accumulation kernel runs here, into the layer's own workspace; `_reduce_convs` finishes every layer of
        the backward pass in ONE launch (8 latency-bound ~10 us reduce kernels less on the weight-gradient stream)."""
        on_main = main or _CONV_WGRAD_MAIN
        if fork and not on_main:
            self.fork_side()
        stream = _stream() if on_main else self._side_raw()
        if self._defer_reduce:
            x, xl, dy, dyl, dw, db, N, Cin, H, W, Cout = args
            ws = self._ws_of((fn, dw))
            call(fn + "_partial", x, xl, dy, dyl, N, Cin, H, W, Cout, ptr(ws), stream)
            self._reduce_pending.append((ptr(ws), dw, db, N, Cin, H, W, Cout, 1 if fn.startswith("dvae_convT") else 0))
            return
        call(fn, *args, ptr(self._ws if on_main else self._ws_side), stream)

    def _reduce_convs(self):
        """One launch: the fixed-order reductions of every deferred conv / convT weight gradient (current stream; the
        side stream must have been joined)."""
        pend, self._reduce_pending = self._reduce_pending, []
        if not pend:
            return
        key = ("wgr",) + tuple(pend)
        ent = self._fc_descs.get(key)
        if ent is None:
            if len(self._fc_descs) >= 64:
                self._fc_descs.clear()
                _lib.note_alloc()
            ent = self._fc_descs[key] = _lib.conv_wgrad_descs(pend)
        call("dvae_conv_wgrad_reduce_grouped", ent[1], len(pend), _stream())

    def _join_side(self):
        if self.single_stream:
            return
        record_py(torch.cuda.current_stream().wait_stream, self._side)

    # ------------------------------------------------------------------ input
    @property
    def u8_fused(self):
        """uint8 batches are consumed as they are (ToTensor's /255 fused into conv1 forward, conv1 weight gradient
        and the likelihood target: dvae_*_u8) for the tuned geometry: 64x64 images with 1 or 3 channels."""
        c, h, w = self.img_size
        return h == 64 and w == 64 and c in (1, 3)

    def input(self, x, buf):
        """Batch as the kernels will read it.  fp32 [B,C,H,W]: itself.  uint8 [B,C,H,W] (pixels 0..255 as the datasets
        store them, utils/datasets.py:204-213,282-291): itself when the fused uint8 kernels cover the geometry, else its
        ToTensor image (float(v)/255, dvae_u8_to_f32) in the engine workspace."""
        if x.dtype == torch.float32:
            return x
        if x.dtype != torch.uint8:
            raise _lib.DvaeHipError("input batches must be float32 in [0,1] or uint8 pixels, got %s" % x.dtype)
        if self.u8_fused:
            return x
        if getattr(buf, "x_f32", None) is None or buf.x_f32.shape != x.shape:
            _lib.note_alloc()
            buf.x_f32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        call("dvae_u8_to_f32", ptr(x), ptr(buf.x_f32), x.numel(), _stream())
        return buf.x_f32

    # ------------------------------------------------------------------ forward
    def encode(self, x, buf, n=None):
        """x[B,C,H,W] (NCHW; fp32, or uint8 for the fused geometry: see input()) -> buf.ml[B,2D] (interleaved mu/logvar)."""
        s = _stream()
        ws = ptr(self._ws)
        B = x.shape[0] if n is None else n
        c, H, _ = self.img_size
        src, src_layout, cin, h = x, NCHW, c, H
        last = len(self.enc_names) - 1
        for k, (name, act) in enumerate(zip(self.enc_names, buf.enc_act)):
            # the last conv writes its 4x4x32 output NCHW = the (c,h,w) flatten order lin1 consumes
            # (encoders.py:80), straight into a_flat: no relayout pass; no conv kernel reads that tensor
            dst, dst_layout = (buf.a_flat, NCHW) if k == last else (act, NHWC)
            if k == 0 and x.dtype == torch.uint8:
                call("dvae_conv4s2_fwd_u8", ptr(src), ptr(self.p("encoder.%s.weight" % name)),
                     ptr(self.p("encoder.%s.bias" % name)), ptr(dst), B, cin, h, h, HID, ACT_RELU, s)
            else:
                call("dvae_conv4s2_fwd", ptr(src), src_layout, ptr(self.p("encoder.%s.weight" % name)),
                     ptr(self.p("encoder.%s.bias" % name)), ptr(dst), dst_layout, B, cin, h, h, HID, ACT_RELU, s)
            src, src_layout, cin, h = act, NHWC, HID, h // 2
        call("dvae_linear_fwd", ptr(buf.a_flat), ptr(self.p("encoder.lin1.weight")), ptr(self.p("encoder.lin1.bias")),
             ptr(buf.h1), B, HID * 16, HIDDEN_DIM, ACT_RELU, ws, s)
        call("dvae_linear_fwd", ptr(buf.h1), ptr(self.p("encoder.lin2.weight")), ptr(self.p("encoder.lin2.bias")),
             ptr(buf.h2), B, HIDDEN_DIM, HIDDEN_DIM, ACT_RELU, ws, s)
        call("dvae_linear_fwd", ptr(buf.h2), ptr(self.p("encoder.mu_logvar_gen.weight")),
             ptr(self.p("encoder.mu_logvar_gen.bias")), ptr(buf.ml), B, HIDDEN_DIM, 2 * self.latent_dim, ACT_NONE, ws, s)

    def reparam(self, buf, eps, kl_dim=None, coef=None, n=None):
        B = buf.B if n is None else n
        call("dvae_reparam_kl_fwd", ptr(buf.ml), ptr(eps), ptr(buf.mu), ptr(buf.logvar), ptr(buf.z), ptr(kl_dim),
             ptr(coef), B, self.latent_dim, _stream())

    def decode(self, z, buf, n=None, fuse_loss=None):
        """z[B,D] -> buf.recon[B,C,H,W] (NCHW, post-sigmoid).  fuse_loss = (target, dist_code, coef,
        partials): the last layer also evaluates the reconstruction likelihood against `target`
        (partial sums -> partials) and writes dLoss/dlogit into buf.g_logit in the same pass."""
        s = _stream()
        ws = ptr(self._ws)
        B = z.shape[0] if n is None else n
        D = self.latent_dim
        call("dvae_linear_fwd", ptr(z), ptr(self.p("decoder.lin1.weight")), ptr(self.p("decoder.lin1.bias")),
             ptr(buf.d1), B, D, HIDDEN_DIM, ACT_RELU, ws, s)
        call("dvae_linear_fwd", ptr(buf.d1), ptr(self.p("decoder.lin2.weight")), ptr(self.p("decoder.lin2.bias")),
             ptr(buf.d2), B, HIDDEN_DIM, HIDDEN_DIM, ACT_RELU, ws, s)
        call("dvae_linear_fwd", ptr(buf.d2), ptr(self.p("decoder.lin3.weight")), ptr(self.p("decoder.lin3.bias")),
             ptr(buf.d3), B, HIDDEN_DIM, HID * 16, ACT_RELU, ws, s)
        # lin3's output [B, 32*4*4] in (c,h,w) order IS the NCHW 4x4x32 input of the first convT
        # (decoders.py:74): read as such, no relayout pass
        src, src_layout, h = buf.d3, NCHW, 4
        for name, act in zip(self.dec_names, buf.dec_act):
            call("dvae_convT4s2_fwd", ptr(src), src_layout, ptr(self.p("decoder.%s.weight" % name)),
                 ptr(self.p("decoder.%s.bias" % name)), ptr(act), NHWC, B, HID, h, h, HID, ACT_RELU, s)
            src, src_layout, h = act, NHWC, h * 2
        c = self.img_size[0]
        if fuse_loss is None:
            call("dvae_convT4s2_fwd", ptr(src), NHWC, ptr(self.p("decoder.convT3.weight")),
                 ptr(self.p("decoder.convT3.bias")), ptr(buf.recon), NCHW, B, HID, h, h, c, ACT_SIGMOID, s)
        else:
            target, dist_code, coef, partials = fuse_loss
            if target.dtype == torch.uint8:
                call("dvae_convT4s2_sigmoid_recon_fwd_u8", ptr(src), ptr(self.p("decoder.convT3.weight")),
                     ptr(self.p("decoder.convT3.bias")), ptr(target), ptr(buf.recon), ptr(buf.g_logit), dist_code,
                     ptr(coef), ptr(partials), B, HID, h, h, c, s)
            else:
                call("dvae_convT4s2_sigmoid_recon_fwd", ptr(src), NHWC, ptr(self.p("decoder.convT3.weight")),
                     ptr(self.p("decoder.convT3.bias")), ptr(target), ptr(buf.recon), ptr(buf.g_logit), dist_code,
                     ptr(coef), ptr(partials), B, HID, h, h, c, s)

    # ------------------------------------------------------------------ backward
    def decode_backward(self, z, buf, n=None, join=True, defer_fc_wgrad=False):
        """buf.g_logit (grad w.r.t. the pre-sigmoid output) -> decoder weight grads, buf.dz.
        defer_fc_wgrad: the three FC weight gradients are not launched here but handed to the next
        encode_backward, which computes all six FC weight gradients of the step in one grouped launch."""
        s = _stream()
        B = z.shape[0] if n is None else n
        D = self.latent_dim
        c = self.img_size[0]
        ws = ptr(self._ws)
        # an encode_backward follows (defer_fc_wgrad): leave the conv reductions to its grouped launch as well
        self._defer_reduce = bool(defer_fc_wgrad) and self.is64 and _DEFER_REDUCE
        if not self._defer_reduce:
            self._reduce_pending = []
        acts = [buf.d3] + buf.dec_act           # inputs of convT_64/convT1/convT2/convT3 (the first one NCHW = lin3's output)
        gacts = [buf.gd3] + buf.dec_gact
        names = self.dec_names + ["convT3"]
        couts = [HID] * len(self.dec_names) + [c]
        hs = [4 << i for i in range(len(names))]  # input H of each convT
        dy, dy_layout = buf.g_logit, NCHW
        # Weight gradients are off the critical path and only due at the end of the backward pass.  The
        # dgrads of the two big layers (convT3, convT2) fill the chip by themselves; everything after them
        # on this stream is small (8x8 / 4x4 layers, the FC chain, the latent glue, the encoder's FC chain)
        # and leaves most CUs idle -- so the big weight gradients are forked THERE (fork 1, after the last
        # big dgrad), and the rest after the FC dgrads (fork 2).  Every fork costs this stream ~6 us.
        pending, deferred, queued = [], [], []
        for k in range(len(names) - 1, -1, -1):
            name, x_in, gx, h = names[k], acts[k], gacts[k], hs[k]
            wargs = ("dvae_convT4s2_wgrad", ptr(x_in), NCHW if k == 0 else NHWC, ptr(dy), dy_layout,
                     ptr(self.g("decoder.%s.weight" % name)), ptr(self.g("decoder.%s.bias" % name)),
                     B, HID, h, h, couts[k])
            (pending if h >= 16 else deferred).append(wargs)
            if k == 0:
                # the first decoder layer's input gradient leaves NCHW = (c,h,w) order, straight into gd3 (the
                # gradient of lin3's output; ReLU mask = lin3's output d3 in the same order): no relayout pass
                call("dvae_convT4s2_dgrad", ptr(dy), dy_layout, ptr(self.p("decoder.%s.weight" % name)), ptr(buf.d3),
                     ptr(buf.gd3), NCHW, B, HID, h, h, couts[k], s)
            else:
                call("dvae_convT4s2_dgrad", ptr(dy), dy_layout, ptr(self.p("decoder.%s.weight" % name)), ptr(x_in), ptr(gx),
                     NHWC, B, HID, h, h, couts[k], s)
            dy, dy_layout = gx, NHWC
            for w_ in queued:                    # side launches of the previous fork, issued AFTER this stream's next kernel
                self._conv_wgrad(*w_, fork=False)
            queued = []
            if h == 16 or (k == 0 and pending):  # last big dgrad is enqueued: its inputs and those of `pending` are final
                self.fork_side()
                queued, pending = pending, []
        for w_ in queued:
            self._conv_wgrad(*w_, fork=False)
        call("dvae_linear_dgrad", ptr(buf.gd3), ptr(self.p("decoder.lin3.weight")), ptr(buf.d2), ACT_RELU, ptr(buf.gd2),
             B, HIDDEN_DIM, HID * 16, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gd2), ptr(self.p("decoder.lin2.weight")), ptr(buf.d1), ACT_RELU, ptr(buf.gd1),
             B, HIDDEN_DIM, HIDDEN_DIM, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gd1), ptr(self.p("decoder.lin1.weight")), None, ACT_NONE, ptr(buf.dz),
             B, D, HIDDEN_DIM, ws, s)
        # small conv layers + the three FC weight gradients: one fork, then they co-run with whatever follows
        self.fork_side()
        for wargs in deferred:
            self._conv_wgrad(*wargs, fork=False)
        fc = [(buf.d2, buf.gd3, self.g("decoder.lin3.weight"), self.g("decoder.lin3.bias"), B, HIDDEN_DIM, HID * 16),
              (buf.d1, buf.gd2, self.g("decoder.lin2.weight"), self.g("decoder.lin2.bias"), B, HIDDEN_DIM, HIDDEN_DIM),
              (z, buf.gd1, self.g("decoder.lin1.weight"), self.g("decoder.lin1.bias"), B, D, HIDDEN_DIM)]
        if defer_fc_wgrad:
            self._fc_pending = fc
        else:
            self._side_wgrad_grouped(fc)
        self._defer_reduce = False
        if join:
            self._join_side()

    def encode_backward(self, x, buf, n=None):
        """buf.dml (grad w.r.t. the interleaved mu/logvar output) -> encoder weight grads."""
        s = _stream()
        B = x.shape[0] if n is None else n
        c, H, _ = self.img_size
        ws = ptr(self._ws)
        self._reduce_pending = [p_ for p_ in self._reduce_pending if p_[3] == B]     # the decoder's, if it deferred them
        self._defer_reduce = self.is64 and _DEFER_REDUCE
        call("dvae_linear_dgrad", ptr(buf.dml), ptr(self.p("encoder.mu_logvar_gen.weight")), ptr(buf.h2), ACT_RELU,
             ptr(buf.gh2), B, HIDDEN_DIM, 2 * self.latent_dim, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gh2), ptr(self.p("encoder.lin2.weight")), ptr(buf.h1), ACT_RELU, ptr(buf.gh1),
             B, HIDDEN_DIM, HIDDEN_DIM, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gh1), ptr(self.p("encoder.lin1.weight")), ptr(buf.a_flat), ACT_RELU,
             ptr(buf.ga_flat), B, HID * 16, HIDDEN_DIM, ws, s)
        # weight gradients wait for the next fork (they only have to be done by the end of the backward pass): the
        # encoder's three FC layers + the decoder's three when decode_backward deferred them = one grouped launch
        pend, self._fc_pending = [p_ for p_ in self._fc_pending if p_[4] == B], []
        # largest problems first (128, 128, 64, 64, 8, 8 tiles): the long-running workgroups start first
        fc = ([(buf.a_flat, buf.gh1, self.g("encoder.lin1.weight"), self.g("encoder.lin1.bias"), B, HID * 16, HIDDEN_DIM)]
              + pend[:1]
              + [(buf.h1, buf.gh2, self.g("encoder.lin2.weight"), self.g("encoder.lin2.bias"), B, HIDDEN_DIM, HIDDEN_DIM)]
              + pend[1:]
              + [(buf.h2, buf.dml, self.g("encoder.mu_logvar_gen.weight"), self.g("encoder.mu_logvar_gen.bias"),
                  B, HIDDEN_DIM, 2 * self.latent_dim)])
        deferred = [lambda fc=fc: self._side_wgrad_grouped(fc)]
        tail_main = []                      # weight gradients the main stream computes after conv1's (load balance of the tail)
        last = len(self.enc_names) - 1
        for k in range(last, -1, -1):
            name = self.enc_names[k]
            h_in = self.enc_sizes[k] * 2
            if k > 0:
                x_in, x_layout, cin = buf.enc_act[k - 1], NHWC, HID
            else:
                x_in, x_layout, cin = x, NCHW, c
            # the last conv's output gradient is lin1's input gradient, (c,h,w) order = NCHW 4x4x32: read as such
            dy, dy_layout = (buf.ga_flat, NCHW) if k == last else (buf.enc_gact[k], NHWC)
            # forks: one before the first big layer (h_in >= 32; the small layers' weight gradients ride
            # along with it), one per big layer after that
            big = h_in >= 32
            wargs = ("dvae_conv4s2_wgrad", ptr(x_in), x_layout, ptr(dy), dy_layout,
                     ptr(self.g("encoder.%s.weight" % name)), ptr(self.g("encoder.%s.bias" % name)),
                     B, cin, h_in, h_in, HID)
            side = []
            if k == 0:
                # the first layer has no dgrad: this stream has nothing else left, so it computes the last
                # weight gradient itself (no fork) while the side stream drains its queue
                if deferred:
                    self.fork_side()
                    side, deferred = deferred, []
                if x.dtype == torch.uint8:
                    call("dvae_conv4s2_wgrad_u8", ptr(x), ptr(dy), ptr(self.g("encoder.%s.weight" % name)),
                         ptr(self.g("encoder.%s.bias" % name)), B, cin, h_in, h_in, HID, ptr(self._ws), s)
                else:
                    self._conv_wgrad(*wargs, fork=False, main=True)
                for w_ in tail_main:
                    self._conv_wgrad(*w_, fork=False, main=True)
            elif name in _TAIL_MAIN and self.is64 and not self.single_stream:
                tail_main.append(wargs)
            elif big:
                self.fork_side()
                side, deferred = deferred + [lambda wargs=wargs: self._conv_wgrad(*wargs, fork=False)], []
            else:
                deferred.append(lambda wargs=wargs: self._conv_wgrad(*wargs, fork=False))
            if k > 0:                            # this stream's next kernel first, then the side launches
                call("dvae_conv4s2_dgrad", ptr(dy), dy_layout, ptr(self.p("encoder.%s.weight" % name)), ptr(x_in),
                     ptr(buf.enc_gact[k - 1]), NHWC, B, cin, h_in, h_in, HID, s)
            for launch in side:
                launch()
        if deferred:
            self.fork_side()
            for launch in deferred:
                launch()
        self._join_side()
        self._defer_reduce = False
        self._reduce_convs()
