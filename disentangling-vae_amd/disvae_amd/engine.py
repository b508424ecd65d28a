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
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from ._debug import knob
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


# which encoder conv weight gradients the MAIN stream computes itself at the very end of the backward pass (after conv1's),
# instead of leaving them in the side stream's queue: the side stream is the tail of the iteration (timeline:
# profiles/r02_final_timeline.md), the main stream is idle from the end of conv1's weight gradient to the join
# (measured -1.5 %: profiles/r02_run12_tail_ab.txt)
_TAIL_MAIN = ("conv3", "conv_64")


_DEVICE_STREAMS = {}
_WG2_STREAMS = {}


def wg2_stream(device):
    """Second weight-gradient stream of `device` (the three-queue schedule of small steps), one per process like device_streams."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _WG2_STREAMS.get(key)
    if st is None:
        st = _WG2_STREAMS[key] = _lib.new_stream(device)
    return st


def device_streams(device):
    """The (side, exchange) HIP streams of `device`, ONE pair per process: every engine on the device uses the same two.  HIP
    multiplexes streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES); a process that builds many models -- bench.py's
    legs, a trainer next to an evaluator -- would otherwise collect streams until two streams of one iteration share a queue
    and serialise (the twelfth engine of a bench run: 1.32 ms per 128-image iteration instead of 0.35,
    profiles/r05_final1_bench.json).  Iterations of different engines in one process run one after the other anyway.
    Created through the C-ABI (dvae_stream_create), not taken from torch's pool: include/dvae_hip.h says why."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _DEVICE_STREAMS.get(key)
    if st is None:
        st = _DEVICE_STREAMS[key] = (_lib.new_stream(device), _lib.new_stream(device))
    return st


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
        # z, mu, logvar: consecutive slabs of ONE buffer -- the sharded beta-TCVAE step all-gathers them with a single
        # collective, no packing pass (parallel.Comm.all_gather_latents)
        self.lat3 = f(3, B, D)
        self.z, self.mu, self.logvar = self.lat3[0], self.lat3[1], self.lat3[2]
        self.d1, self.d2, self.d3 = f(B, HIDDEN_DIM), f(B, HIDDEN_DIM), f(B, HID * 16)
        self.gd1, self.gd2, self.gd3 = f(B, HIDDEN_DIM), f(B, HIDDEN_DIM), f(B, HID * 16)
        self.dec_act = [f(B, h, h, HID) for h in eng.dec_sizes]       # NHWC outputs of the hidden convT layers
        self.dec_gact = [f(B, h, h, HID) for h in eng.dec_sizes]
        # ReLU masks as bit planes (one uint32 per pixel of a 32-channel NHWC activation) for the two 32x32x32 activations of
        # the 64x64 geometry: conv1's output (mask of conv2's input gradient) and convT2's output (mask of convT3's)
        self.bits_conv1 = self.bits_convT2 = None
        if eng.mask_bits:
            self.bits_conv1 = torch.empty(B * 32 * 32, dtype=torch.int32, device=dev)
            self.bits_convT2 = torch.empty(B * 32 * 32, dtype=torch.int32, device=dev)
        c, hh, ww = eng.img_size
        self.recon = f(B, c, hh, ww)
        self.g_logit = f(B, c, hh, ww)
        self.dz = f(B, D)


class _Images:
    """Pre-staged weight images of one parameter placement (dvae_stage_weights): the two 64 KB LDS images of every
    32 <-> 32 channel conv / convT layer and the k-chunked forward / input-gradient operand streams of the six FC layers, in
    ONE device buffer, plus the host descriptor tables of the staging launch."""

    FC = ["encoder.lin1", "encoder.lin2", "encoder.mu_logvar_gen", "decoder.lin1", "decoder.lin2", "decoder.lin3"]

    def __init__(self, eng):
        _lib.note_alloc()
        arena = eng.arena
        self.flat_ptr = arena.flat.data_ptr()
        conv = ["encoder." + n for n in eng.enc_names[1:]] + ["decoder." + n for n in eng.dec_names]
        sizes, off = {}, 0
        for name in conv:
            for kind in ("down", "up"):
                sizes[(name, kind)] = off
                off += 16384
        # latent dimensions above _lib.MAX_LATENT_DIM: the FC layers run one launch each on the raw weights (fc_chain_fwd below),
        # no operand streams are staged
        self.FC = [] if _lib.wide(eng.latent_dim) else list(_Images.FC)
        for name in self.FC:
            N, K = arena.shapes[name + ".weight"]
            sizes[(name, "fwd")] = off
            off += (K + 3) // 4 * N * 4
            sizes[(name, "bwd")] = off
            off += (N + 3) // 4 * K * 4
        # the last decoder layer (C = 1 / 3 output channels at 64x64): operand-pair records of the packed-FMA forward kernel
        c = eng.img_size[0]
        self.thin_C = c if (eng.is64 and c in (1, 3)) else 0
        if self.thin_C:
            sizes[("decoder.convT3", "pairs")] = off
            off += 32 * _lib.thin_pair_floats(c)
        self.buf = torch.empty(off, dtype=torch.float32, device=arena.flat.device)
        base = self.buf.data_ptr()
        self.ptrs = {k: base + 4 * o for k, o in sizes.items()}
        self.conv_descs = (_lib.ConvImageDesc * len(conv))()
        for d, name in zip(self.conv_descs, conv):
            d.w, d.img_down, d.img_up = ptr(arena.view(name + ".weight")), self.ptrs[(name, "down")], self.ptrs[(name, "up")]
        self.fc_descs = (_lib.FcImageDesc * max(len(self.FC), 1))()
        self.n_fc = len(self.FC)
        for d, name in zip(self.fc_descs, self.FC):
            N, K = arena.shapes[name + ".weight"]
            d.w, d.img_fwd, d.img_bwd, d.N, d.K = (ptr(arena.view(name + ".weight")), self.ptrs[(name, "fwd")],
                                                   self.ptrs[(name, "bwd")], N, K)
        self.thin_desc = None
        if self.thin_C:
            self.thin_desc = _lib.ThinImageDesc()
            self.thin_desc.w, self.thin_desc.img_pairs, self.thin_desc.C = (ptr(arena.view("decoder.convT3.weight")),
                                                                            self.ptrs[("decoder.convT3", "pairs")], self.thin_C)
        self.coef_vals = (ctypes.c_float * 8)()


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
        self._ws_wg2 = None    # partial-sum workspace of the second weight-gradient stream (three_streams)
        self._wg2 = None
        self._side = None      # side HIP stream: the weight-gradient kernels run beside the dgrad chain
        self._aux = None       # exchange stream of sharded steps (see buffers())
        # small batches: everything on the caller's stream.  Below ~256 images the iteration is bound by the latency of
        # dependent launches, a fork / join between hardware queues costs ~6 us each (5 forks + 1 join per iteration) and
        # the weight-gradient kernels that the side stream would overlap are a few microseconds long.  Set per step by the
        # loss plugins (BaseLoss._streams).
        self.single_stream = False
        # how the weight gradients are scheduled against the chain of input gradients (two streams):
        #   False: batch-sized schedule -- the big layers' weight gradients are forked behind the big input gradients (two
        #          chip-filling persistent kernels do not co-run: what matters is that the small kernels of the critical path
        #          find idle CUs), the tail is balanced between the streams (measured at B = 1024: DESIGN.md section 5);
        #   True : dependency-driven -- every weight gradient is launched on the side stream as soon as its two operands
        #          exist (a fork per layer), beside the input gradient of the same layer.  Below a few hundred images per
        #          step no kernel fills the chip, the iteration is a latency chain, and the side stream should start as
        #          early as the data allows (profiles/r03_v2_timeline_b128.md: backward pass 287 us against ~150 us of
        #          dependent work).  Set per step by the loss plugins (BaseLoss._streams).
        self.eager_wgrad = False
        # encoder weight gradients the MAIN stream computes after conv1's, at the very end of the backward pass (the balance of
        # the two streams' tails).  Set per step by the loss plugins (BaseLoss._streams).
        self.tail_main = _TAIL_MAIN
        self._fork_hook = None
        # 64x64 images with 1 / 3 channels: the forward kernels of conv1 and convT2 also emit the sign bits of their outputs and
        # the input-gradient kernels of conv2 and convT3 read those instead of the 32x32x32 fp32 activations (dvae_*_bits)
        self.mask_bits = self.is64 and c in (1, 3) and knob("DVAE_MASK_BITS", "1") != "0"   # (A/B knob: DVAE_DEBUG=1 only)
        # fused FC chain: the 8x8 <-> 4x4 layers (conv_64 / convT_64 at 64x64, conv3 / convT1 at 32x32) and their input
        # gradients run INSIDE the chain launches (dvae_fc_chain_fwd / _bwd, conv_in / convT_gout fields: csrc/conv4_end.h) --
        # four launches fewer on the critical path
        # Up to fuse_ends_max_rows rows per launch, where the step is a chain of dependent launches and each one saved counts
        # (same box, three alternations, profiles/r06_s2_chain3.txt: factor 64x64x1 tensor 256 0.590 -> 0.569 ms, btcvae 64x64x3 at
        # 64 / 128 / 256 images 0.311 -> 0.302, 0.347 -> 0.344, 0.450 -> 0.447 ms); from 512 rows up the fused launches -- 150 KB
        # of LDS, a whole CU per workgroup -- can no longer slip in beside the other stream's persistent kernels the way the
        # small conv launches do: 0.643 -> 0.652 ms at 512 images, 1.060 -> 1.082 ms at 1024 (profiles/r06_s2_chain2.txt) -- in
        # the BACKWARD pass, that is (fuse_ends_max_rows); the forward chain has its own limit below
        self.fuse_ends = not _lib.wide(latent_dim) and knob("DVAE_FUSE_ENDS", "1") != "0"   # (A/B knob: DVAE_DEBUG=1 only)
        self.fuse_ends_max_rows = int(knob("DVAE_FUSE_ENDS_MAX_ROWS", "256"))
        # the FORWARD chain's own limit: beside it the other stream carries only the estimator's small kernels, nothing a 150 KB
        # workgroup could block -- 384 / 512 / 1024 images 0.545 -> 0.543, 0.631 -> 0.629, 1.039 -> 1.030 ms; level at 2048 rows,
        # where the 8-row variant runs (profiles/r06_s2_fwd_ends.txt)
        self.fuse_ends_max_rows_fwd = int(knob("DVAE_FUSE_ENDS_MAX_ROWS_FWD", "1024"))
        # Round 6: convT3's weight gradient (bandwidth-bound) is forked one kernel earlier -- behind convT3's input gradient, beside
        # the matrix-bound input gradient of convT2 -- instead of behind both.  The side stream's serial chain of weight gradients
        # is what small steps end on, and it now starts ~15 us sooner: 128 / 256 images 0.341 -> 0.330, 0.443 -> 0.431 ms,
        # btcvae 64x64x1 B = 256 0.412 -> 0.398, 1024 images 1.054 -> 1.048 ms (profiles/r06_s2_sched2.txt).  Mode 2 (in FRONT of
        # convT3's input gradient) wins another 1-2 % at 128 images and loses 1.6 % at 256, 0.7 % at 1024 (r06_s2_sched3.txt): used at
        # 112-128 images only (decode_backward).
        # Moving the main stream's tail (tail_main) to the side stream loses 2-8 % at every small batch (same file).
        _et = knob("DVAE_EARLY_THIN", "auto")                          # (A/B knob: DVAE_DEBUG=1 only: 0 / 1 / 2 force a mode)
        self.early_thin_wgrad = 1 if _et == "auto" else int(_et)
        self.early_thin_auto = _et == "auto"
        self.sharded = False   # this step runs under data parallelism (set per step by the loss plugins, BaseLoss._streams)
        # the weight gradients on TWO side streams, each launched at the first fork behind the kernel that produces its last operand
        # (decode_backward's `three` branch / _encode_backward_3s).  Set per step by the loss plugins (BaseLoss._streams: FactorVAE
        # from 2048 rows; slower for every other step measured).
        self.three_streams = False
        # steps of 129-320 images end on the side stream (its weight-gradient grid is the smaller one there, conv_wgrad_ws.hip): the
        # grouped FC weight gradients become the LAST launch of the main stream's tail instead -- 256 images 0.425 -> 0.415 ms,
        # btcvae 64x64x1 B = 256 0.386 -> 0.375; outside that band the main stream is the tail already: 64 / 128 / 512 / 1024
        # images +1.3 / +1.4 / +2.1 / +0.6 % (profiles/r06_s2_fcw_main.txt)
        self.fcw_main = knob("DVAE_FCW_MAIN", "1") == "1"
        self.fcw_main_rows = tuple(int(v) for v in knob("DVAE_FCW_MAIN_ROWS", "129,320").split(","))
        self._ends_on = False  # this forward pass: set by encode_convs(chain=True), read by fc_chain_fwd / decode_convs
        self._fc_pending = []  # FC weight-gradient problems waiting for the grouped launch (decoder's, deferred)
        self._fc_descs = {}    # host descriptor arrays / argument structs of launches, kept alive for recorded plans
        self._images = None

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
            if b is not None:
                self._fc_descs.clear()     # cached argument structs hold pointers into the workspace being replaced
            b = _Buffers(self, B)
            self._bufs[B] = b
        if self._ws is None or self._ws.device != self.device:
            n = _lib.lib().dvae_conv_wgrad_ws_floats()
            _lib.note_alloc()
            self._ws = torch.empty(n, dtype=torch.float32, device=self.device)
            self._ws_side = torch.empty(n, dtype=torch.float32, device=self.device)
            self._ws_wg2 = torch.empty(n, dtype=torch.float32, device=self.device)
            self._wg2 = wg2_stream(self.device)
            # (a high-priority side stream measured the same step time: profiles/r04_v45_side_priority.txt)
            # third stream (sharded batches): the exchange-bound part of a step -- latent all-gather, the estimator over the
            # global batch, column-gradient reduce-scatter, the all-reduce of the loss sums -- must not sit in front of the
            # weight gradients on the side stream, which is the tail of the iteration
            self._side, self._aux = device_streams(self.device)
        return b

    # ---- per-step weight staging -------------------------------------------------------------------
    @property
    def images(self):
        im = self._images
        if im is None or im.flat_ptr != self.arena.flat.data_ptr():
            self._fc_descs.clear()         # cached argument structs hold pointers into the old images / parameter arena
            im = self._images = _Images(self)
        return im

    def stage(self, coef=None, coef_host=None):
        """ONE launch at the head of a forward pass: the LDS weight images of the 32-channel conv layers and the operand
        streams of the FC chain are rebuilt from the current parameters (they change in optimizer.step(), training.py:158, or
        under the caller's hands: load_state_dict, reset_parameters); `coef` (device) <- `coef_host` (8 floats) rides along
        (= dvae_set_coef).  Everything downstream in the pass -- forward and backward -- reads the images."""
        im = self.images
        cv = None
        if coef is not None:
            for i, v in enumerate(coef_host):
                im.coef_vals[i] = v
            cv = ctypes.addressof(im.coef_vals)
        call("dvae_stage_weights", ctypes.addressof(im.conv_descs), len(im.conv_descs), ctypes.addressof(im.fc_descs),
             im.n_fc, None if im.thin_desc is None else ctypes.addressof(im.thin_desc), ptr(coef), cv, _stream())

    def _img(self, layer, kind):
        return self._images.ptrs[(layer, kind)]

    def _args(self, key, cls, **fields):
        """Host argument struct of a C-ABI call (cached: recorded launch plans hold its address)."""
        ent = self._fc_descs.get(key)
        if ent is None:
            if len(self._fc_descs) >= 64:        # recorded plans hold the addresses of these structs: invalidate them
                self._fc_descs.clear()
                _lib.note_alloc()
            ent = self._fc_descs[key] = _lib.struct_of(cls, **fields)
        return ent[1]

    # ---- fork / join of the side stream (weight-gradient kernels) ----------------------------------
    def fork_side(self):
        """Order the side stream after everything enqueued so far on the current stream.  A fork
        costs the current stream ~6 us (event signal between hardware queues, profiles/r01_run19
        timeline), so the FC weight gradients fork once per chain, not once per layer."""
        if self.single_stream:
            return
        call("dvae_stream_order", _stream(), self._side.cuda_stream)
        hook, self._fork_hook = self._fork_hook, None
        if hook is not None:
            hook()

    def at_next_fork(self, fn):
        """Side-stream work that needs everything enqueued on the current stream SO FAR but is not urgent: `fn()` is called
        right after the next fork_side() instead of paying for a fork of its own (each costs the current stream ~6 us)."""
        if knob("DVAE_FORK_HOOK", "1") == "0":      # A/B (DVAE_DEBUG=1): a fork of its own, right here
            self._fork_hook = None
            self.fork_side()
            fn()
            return
        self._fork_hook = fn                   # (the loss plugins clear a hook left behind by a step that raised)

    def flush_fork_hook(self):
        """A consumer of the deferred side-stream work is about to be enqueued: if no fork has happened yet, fork now."""
        if self._fork_hook is not None:
            self.fork_side()

    @property
    def side_stream(self):
        return torch.cuda.current_stream() if self.single_stream else self._side

    def _side_raw(self):
        return _stream() if self.single_stream else self._side.cuda_stream

    @property
    def aux_stream(self):
        return torch.cuda.current_stream() if self.single_stream else self._aux

    def _aux_raw(self):
        return _stream() if self.single_stream else self._aux.cuda_stream

    def _side_wgrad_grouped(self, problems, stream=None):
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
        call("dvae_linear_wgrad_grouped", ent[1], len(problems), self._side_raw() if stream is None else stream)

    def _conv_wgrad(self, fn, *args, fork=True, main=False):
        """Conv / convT weight gradient `fn(*args, ws, stream)`: off the dgrad critical path, so it
        goes to the side stream (after a fork) and co-runs with the dgrad chain (main=True: the current stream).
        (Capping the side stream's chip-filling launches at 96-224 workgroups so that the other stream's short kernels find
        free CUs measured 0.3-4.7 % SLOWER at 1024 images: profiles/r05_v23_side_cap_ab.txt.)"""
        if fork and not main:
            self.fork_side()
        call(fn, *args, ptr(self._ws if main else self._ws_side), _stream() if main else self._side_raw())

    def _join_side(self):
        if self.single_stream:
            return
        call("dvae_stream_order", self._side.cuda_stream, _stream())

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
    def encode_convs(self, x, buf, n=None, chain=False):
        """x[B,C,H,W] (NCHW; fp32, or uint8 for the fused geometry: see input()) -> buf.a_flat[B,512]: the conv stack of
        encoders.py:73-80.  The 32-channel layers read their pre-staged weight images (stage() must precede).
        chain: fc_chain_fwd follows -- with fuse_ends it computes conv_64 itself, the stack stops at conv3's output."""
        s = _stream()
        B = x.shape[0] if n is None else n
        c, H, _ = self.img_size
        src, h = x, H
        last = len(self.enc_names) - 1
        self._ends_on = chain and self.fuse_ends and B <= self.fuse_ends_max_rows_fwd
        for k, (name, act) in enumerate(zip(self.enc_names, buf.enc_act)):
            if self._ends_on and k == last:
                break
            # the last conv writes its 4x4x32 output NCHW = the (c,h,w) flatten order lin1 consumes
            # (encoders.py:80), straight into a_flat: no relayout pass; no conv kernel reads that tensor
            dst, dst_layout = (buf.a_flat, NCHW) if k == last else (act, NHWC)
            lname = "encoder.%s" % name
            if k > 0:
                call("dvae_conv32_down", ptr(src), self._img(lname, "down"), ptr(self.p(lname + ".bias")), None, ptr(dst),
                     dst_layout, B, h // 2, ACT_RELU, s)
            elif self.mask_bits:
                call("dvae_conv1_fwd_bits", ptr(src), int(x.dtype == torch.uint8), ptr(self.p(lname + ".weight")),
                     ptr(self.p(lname + ".bias")), ptr(dst), ptr(buf.bits_conv1), B, c, s)
            elif x.dtype == torch.uint8:
                call("dvae_conv4s2_fwd_u8", ptr(src), ptr(self.p(lname + ".weight")), ptr(self.p(lname + ".bias")),
                     ptr(dst), B, c, h, h, HID, ACT_RELU, s)
            else:
                call("dvae_conv4s2_fwd", ptr(src), NCHW, ptr(self.p(lname + ".weight")), ptr(self.p(lname + ".bias")),
                     ptr(dst), dst_layout, B, c, h, h, HID, ACT_RELU, s)
            src, h = act, h // 2

    def encode(self, x, buf, n=None):
        """x -> buf.ml[B,2D] (interleaved mu/logvar), layer by layer (the autograd-compatible entry points; the native
        training step runs the FC layers as one launch: fc_chain_fwd)."""
        s = _stream()
        ws = ptr(self._ws)
        B = x.shape[0] if n is None else n
        self.stage()
        self.encode_convs(x, buf, n)
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

    def kl_blocks(self, n_enc):
        """Number of KL partial blocks fc_chain_fwd leaves at kl_dim + 16 (dvae_loss_epilogue / dvae_kl_finish argument).
        0 above _lib.MAX_LATENT_DIM: kl_dim then holds the D final values and there is nothing to finish."""
        if _lib.wide(self.latent_dim):
            return 0
        r = _lib.fc_chain_rows(n_enc)
        return (n_enc + r - 1) // r

    def fc_chain_fwd(self, buf, eps, kl_dim, n_enc, n_kl=None, n_dec=None, coef=None):
        """buf.a_flat -> h1, h2, ml, mu, logvar, z (rows < n_enc; KL partial blocks from rows < n_kl at kl_dim + 16) and
        d1, d2, d3 (rows < n_dec) in ONE launch (dvae_fc_chain_fwd): encoders.py:81-87, vae.py:52-71, losses.py:470,
        decoders.py:71-73.  eps [n_enc, D] or None (z = mu).
        Latent dimensions above _lib.MAX_LATENT_DIM: the same tensors from one launch per layer (_fc_layers_fwd); kl_dim then
        receives the D FINAL per-dimension values, normalised by coef[INV_B] (`coef` is required with kl_dim)."""
        n_kl = n_enc if n_kl is None else n_kl
        n_dec = n_enc if n_dec is None else n_dec
        if _lib.wide(self.latent_dim):
            return self._fc_layers_fwd(buf, eps, kl_dim, n_enc, n_kl, n_dec, coef)
        if n_enc > _lib.FC_CHAIN_MAX_ROWS:
            raise _lib.DvaeHipError("fc_chain_fwd: at most %d rows per launch" % _lib.FC_CHAIN_MAX_ROWS)
        P, I = self.p, self._img
        addr = self._args(("fcf", id(buf), ptr(eps), ptr(kl_dim), n_enc, n_kl, n_dec, self._images.buf.data_ptr(), self._ends_on),
                          _lib.FcChainFwdArgs, a_flat=ptr(buf.a_flat),
                          w_e1=I("encoder.lin1", "fwd"), w_e2=I("encoder.lin2", "fwd"), w_ml=I("encoder.mu_logvar_gen", "fwd"),
                          w_d1=I("decoder.lin1", "fwd"), w_d2=I("decoder.lin2", "fwd"), w_d3=I("decoder.lin3", "fwd"),
                          b_e1=ptr(P("encoder.lin1.bias")), b_e2=ptr(P("encoder.lin2.bias")),
                          b_ml=ptr(P("encoder.mu_logvar_gen.bias")), b_d1=ptr(P("decoder.lin1.bias")),
                          b_d2=ptr(P("decoder.lin2.bias")), b_d3=ptr(P("decoder.lin3.bias")), eps=ptr(eps),
                          h1=ptr(buf.h1), h2=ptr(buf.h2), ml=ptr(buf.ml), mu=ptr(buf.mu), logvar=ptr(buf.logvar), z=ptr(buf.z),
                          kl_part=None if kl_dim is None else ptr(kl_dim) + 64, d1=ptr(buf.d1), d2=ptr(buf.d2), d3=ptr(buf.d3),
                          n_enc=n_enc, n_kl=n_kl, n_dec=n_dec, D=self.latent_dim, **self._ends_fwd(buf, n_dec))
        call("dvae_fc_chain_fwd", addr, _stream())

    def _ends(self, rows):
        """Do the chain launches over `rows` rows carry the 4x4 conv ends?"""
        return self.fuse_ends and rows <= self.fuse_ends_max_rows

    def _ends_fwd(self, buf, n_dec):
        """dvae_fc_chain_fwd_args' conv_in .. convT_out: conv_64 in front of the chain, convT_64 behind it (fuse_ends)."""
        if not self._ends_on:
            return {}
        P, I = self.p, self._img
        enc, dec = "encoder." + self.enc_names[-1], "decoder." + self.dec_names[0]
        d = dict(conv_in=ptr(buf.enc_act[-2]), conv_w=I(enc, "down"), conv_b=ptr(P(enc + ".bias")))
        if n_dec > 0:
            d.update(convT_w=I(dec, "up"), convT_b=ptr(P(dec + ".bias")), convT_out=ptr(buf.dec_act[0]))
        return d

    def _fc_layers_fwd(self, buf, eps, kl_dim, n_enc, n_kl, n_dec, coef):
        """fc_chain_fwd for any latent dimension: dvae_linear_fwd x 3, dvae_reparam_kl_fwd (its run-time-D form), x 3."""
        s = _stream()
        ws = ptr(self._ws)
        D = self.latent_dim
        P = self.p
        if kl_dim is not None and n_kl > 0 and coef is None:
            raise _lib.DvaeHipError("fc_chain_fwd: latent_dim > %d needs `coef` with kl_dim" % _lib.MAX_LATENT_DIM)
        for x, name, y, K, N in ((buf.a_flat, "encoder.lin1", buf.h1, HID * 16, HIDDEN_DIM),
                                 (buf.h1, "encoder.lin2", buf.h2, HIDDEN_DIM, HIDDEN_DIM),
                                 (buf.h2, "encoder.mu_logvar_gen", buf.ml, HIDDEN_DIM, 2 * D)):
            call("dvae_linear_fwd", ptr(x), ptr(P(name + ".weight")), ptr(P(name + ".bias")), ptr(y), n_enc, K, N,
                 ACT_NONE if name.endswith("gen") else ACT_RELU, ws, s)
        # KL over the rows < n_kl only (FactorVAE: the first half batch, losses.py:255-259): two row ranges
        with_kl = kl_dim is not None and n_kl > 0
        n0 = n_kl if with_kl else n_enc
        call("dvae_reparam_kl_fwd", ptr(buf.ml), ptr(eps), ptr(buf.mu), ptr(buf.logvar), ptr(buf.z),
             ptr(kl_dim) if with_kl else None, ptr(coef) if with_kl else None, n0, D, s)
        if n0 < n_enc:
            o = n0 * D * 4
            call("dvae_reparam_kl_fwd", ptr(buf.ml) + 2 * o, None if eps is None else ptr(eps) + o, ptr(buf.mu) + o,
                 ptr(buf.logvar) + o, ptr(buf.z) + o, None, None, n_enc - n0, D, s)
        if n_dec > 0:
            for x, name, y, K, N in ((buf.z, "decoder.lin1", buf.d1, D, HIDDEN_DIM),
                                     (buf.d1, "decoder.lin2", buf.d2, HIDDEN_DIM, HIDDEN_DIM),
                                     (buf.d2, "decoder.lin3", buf.d3, HIDDEN_DIM, HID * 16)):
                call("dvae_linear_fwd", ptr(x), ptr(P(name + ".weight")), ptr(P(name + ".bias")), ptr(y), n_dec, K, N,
                     ACT_RELU, ws, s)

    def _fc_layers_bwd(self, buf, eps, dz2, dz3, dmu_x, dlv_x, scal, coef, n):
        """fc_chain_bwd for any latent dimension: dvae_linear_dgrad x 3, dvae_reparam_kl_bwd, x 3 (the launches of the
        autograd-compatible path: decode_backward / encode_backward without fc_chain)."""
        s = _stream()
        ws = ptr(self._ws)
        D = self.latent_dim
        P = self.p
        call("dvae_linear_dgrad", ptr(buf.gd3), ptr(P("decoder.lin3.weight")), ptr(buf.d2), ACT_RELU, ptr(buf.gd2),
             n, HIDDEN_DIM, HID * 16, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gd2), ptr(P("decoder.lin2.weight")), ptr(buf.d1), ACT_RELU, ptr(buf.gd1),
             n, HIDDEN_DIM, HIDDEN_DIM, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gd1), ptr(P("decoder.lin1.weight")), None, ACT_NONE, ptr(buf.dz),
             n, D, HIDDEN_DIM, ws, s)
        call("dvae_reparam_kl_bwd", ptr(buf.dz), ptr(dz2), ptr(dz3), ptr(dmu_x), ptr(dlv_x), ptr(buf.mu), ptr(buf.logvar),
             ptr(eps), ptr(scal), ptr(coef), ptr(buf.dml), n, D, s)
        call("dvae_linear_dgrad", ptr(buf.dml), ptr(P("encoder.mu_logvar_gen.weight")), ptr(buf.h2), ACT_RELU, ptr(buf.gh2),
             n, HIDDEN_DIM, 2 * D, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gh2), ptr(P("encoder.lin2.weight")), ptr(buf.h1), ACT_RELU, ptr(buf.gh1),
             n, HIDDEN_DIM, HIDDEN_DIM, ws, s)
        call("dvae_linear_dgrad", ptr(buf.gh1), ptr(P("encoder.lin1.weight")), ptr(buf.a_flat), ACT_RELU, ptr(buf.ga_flat),
             n, HID * 16, HIDDEN_DIM, ws, s)

    def fc_chain_bwd(self, buf, eps, dz2, dz3, dmu_x, dlv_x, scal, coef, n):
        """buf.gd3 -> gd2, gd1, dz, dml, gh2, gh1, ga_flat (rows < n) in ONE launch (dvae_fc_chain_bwd): the input gradients
        of the six FC layers with dvae_reparam_kl_bwd's arithmetic in the middle (training.py:157).  Latent dimensions above
        _lib.MAX_LATENT_DIM: one launch per layer (_fc_layers_bwd)."""
        if _lib.wide(self.latent_dim):
            return self._fc_layers_bwd(buf, eps, dz2, dz3, dmu_x, dlv_x, scal, coef, n)
        I = self._img
        ends = {}
        if self._ends(n):      # convT_64's input gradient in front of the chain, conv_64's behind it (dvae_fc_chain_bwd_args)
            ends = dict(convT_gout=ptr(buf.dec_gact[0]), convT_w=I("decoder." + self.dec_names[0], "down"), d3=ptr(buf.d3),
                        conv_w=I("encoder." + self.enc_names[-1], "up"), conv_act=ptr(buf.enc_act[-2]),
                        conv_gin=ptr(buf.enc_gact[-2]))
        addr = self._args(("fcb", id(buf), ptr(eps), ptr(dz2), ptr(dz3), ptr(dmu_x), ptr(dlv_x), ptr(scal), ptr(coef), n,
                           self._images.buf.data_ptr()),
                          _lib.FcChainBwdArgs, gd3=ptr(buf.gd3),
                          w_d3=I("decoder.lin3", "bwd"), w_d2=I("decoder.lin2", "bwd"), w_d1=I("decoder.lin1", "bwd"),
                          w_ml=I("encoder.mu_logvar_gen", "bwd"), w_e2=I("encoder.lin2", "bwd"), w_e1=I("encoder.lin1", "bwd"),
                          d2=ptr(buf.d2), d1=ptr(buf.d1), h2=ptr(buf.h2), h1=ptr(buf.h1), a_flat=ptr(buf.a_flat),
                          mu=ptr(buf.mu), logvar=ptr(buf.logvar), eps=ptr(eps), dz2=ptr(dz2), dz3=ptr(dz3),
                          dmu_x=ptr(dmu_x), dlv_x=ptr(dlv_x), scal=ptr(scal), coef=ptr(coef),
                          gd2=ptr(buf.gd2), gd1=ptr(buf.gd1), dz=ptr(buf.dz), dml=ptr(buf.dml), gh2=ptr(buf.gh2),
                          gh1=ptr(buf.gh1), ga_flat=ptr(buf.ga_flat), n=n, D=self.latent_dim, **ends)
        call("dvae_fc_chain_bwd", addr, _stream())

    def decode_convs(self, buf, n, fuse_loss=None, chain=False):
        """buf.d3[B,512] -> buf.recon[B,C,H,W] (NCHW, post-sigmoid): the convT stack of decoders.py:74-82.
        fuse_loss = (target, dist_code, coef, partials): the last layer also evaluates the reconstruction likelihood
        against `target` (partial sums -> partials) and writes dLoss/dlogit into buf.g_logit in the same pass.
        chain: fc_chain_fwd preceded -- with fuse_ends it has computed convT_64 already (buf.dec_act[0])."""
        s = _stream()
        B = n
        # lin3's output [B, 32*4*4] in (c,h,w) order IS the NCHW 4x4x32 input of the first convT
        # (decoders.py:74): read as such, no relayout pass
        src, src_layout, h = buf.d3, NCHW, 4
        for name, act in zip(self.dec_names, buf.dec_act):
            lname = "decoder.%s" % name
            if chain and self._ends_on and h == 4:
                src, src_layout, h = act, NHWC, 8
                continue
            if self.mask_bits and h == 16:          # convT2: also emits the sign bits of its output (convT3's backward mask)
                call("dvae_conv32_up_bits", ptr(src), self._img(lname, "up"), ptr(self.p(lname + ".bias")), None, ptr(act),
                     ptr(buf.bits_convT2), B, ACT_RELU, s)
            else:
                call("dvae_conv32_up", ptr(src), src_layout, self._img(lname, "up"), ptr(self.p(lname + ".bias")), None,
                     ptr(act), B, h, ACT_RELU, s)
            src, src_layout, h = act, NHWC, h * 2
        c = self.img_size[0]
        if self._images.thin_C:
            # tuned geometry: the packed-FMA kernel on the staged pair records, with or without the fused likelihood
            target, dist_code, coef, partials = fuse_loss if fuse_loss is not None else (None, 0, None, None)
            call("dvae_convT3_fwd_staged", ptr(src), self._img("decoder.convT3", "pairs"), ptr(self.p("decoder.convT3.bias")),
                 ptr(target), int(target is not None and target.dtype == torch.uint8), ptr(buf.recon),
                 None if target is None else ptr(buf.g_logit), dist_code, ptr(coef), ptr(partials), B, c, s)
        elif fuse_loss is None:
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

    def decode(self, z, buf, n=None, fuse_loss=None, staged=False):
        """z[B,D] -> buf.recon, layer by layer (decoders.py:67-84; the autograd-compatible entry points)."""
        s = _stream()
        ws = ptr(self._ws)
        B = z.shape[0] if n is None else n
        D = self.latent_dim
        if not staged:
            self.stage()
        call("dvae_linear_fwd", ptr(z), ptr(self.p("decoder.lin1.weight")), ptr(self.p("decoder.lin1.bias")),
             ptr(buf.d1), B, D, HIDDEN_DIM, ACT_RELU, ws, s)
        call("dvae_linear_fwd", ptr(buf.d1), ptr(self.p("decoder.lin2.weight")), ptr(self.p("decoder.lin2.bias")),
             ptr(buf.d2), B, HIDDEN_DIM, HIDDEN_DIM, ACT_RELU, ws, s)
        call("dvae_linear_fwd", ptr(buf.d2), ptr(self.p("decoder.lin3.weight")), ptr(self.p("decoder.lin3.bias")),
             ptr(buf.d3), B, HIDDEN_DIM, HID * 16, ACT_RELU, ws, s)
        self.decode_convs(buf, B, fuse_loss)

    # ---- three-queue schedule of small steps ----------------------------------------------------------------------------
    def _three(self, chain):
        """The backward pass of this step puts its weight gradients on two side streams (64x64 geometry, native step)."""
        return bool(self.three_streams and chain and self.is64 and not self.single_stream and not self.eager_wgrad)

    def _fork_q(self, *qs):
        """Order side stream 1 (side) and / or 2 (wg2) behind everything enqueued so far on the current stream."""
        for q in qs:
            if q == 1:
                self.fork_side()
            else:
                call("dvae_stream_order", _stream(), self._wg2.cuda_stream)

    def _wgrad_q(self, q, fn, *args):
        """Conv / convT weight gradient (+ its fixed-order reduction) on side stream q, with that stream's partial-sum workspace."""
        call(fn, *args, ptr(self._ws_side if q == 1 else self._ws_wg2), (self._side if q == 1 else self._wg2).cuda_stream)

    # ------------------------------------------------------------------ backward
    def decode_backward(self, z, buf, n=None, join=True, defer_fc_wgrad=False, fc_chain=None):
        """buf.g_logit (grad w.r.t. the pre-sigmoid output) -> decoder weight grads, buf.dz.
        defer_fc_wgrad: the three FC weight gradients are not launched here but handed to the next
        encode_backward, which computes all six FC weight gradients of the step in one grouped launch.
        fc_chain: callable that enqueues fc_chain_bwd (the native training step): it replaces the three FC input-gradient
        launches here AND the latent glue + the encoder's three of the following encode_backward(fc_chain=True)."""
        s = _stream()
        B = z.shape[0] if n is None else n
        D = self.latent_dim
        c = self.img_size[0]
        ws = ptr(self._ws)
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
        eager = self.eager_wgrad and not self.single_stream
        # where convT3's weight gradient is forked (early_thin_wgrad): behind its input gradient -- or, at the 128 images of one
        # rank of the 8-GPU headline configuration, in FRONT of it (beside it): 0.332 -> 0.325 ms there, level at 32 / 64, +1.3 %
        # at 96, +1.6 % at 256 images (profiles/r06_s2_sched3.txt, r06_s2_sched4.txt).  Not under data parallelism: the fork carries the
        # late epilogue with its collectives, whose host-side issue would then stand in front of convT3's input gradient
        early_mode = 2 if (self.early_thin_auto and 112 <= B <= 128 and not self.sharded) else self.early_thin_wgrad
        three = self._three(fc_chain is not None) and defer_fc_wgrad and not join
        pending, deferred, queued = [], [], []
        W3 = {}                                  # three-queue schedule: every layer's weight-gradient launch, by layer
        for k in range(len(names) - 1, -1, -1):
            name, x_in, gx, h = names[k], acts[k], gacts[k], hs[k]
            lname = "decoder.%s" % name
            wargs = ("dvae_convT4s2_wgrad", ptr(x_in), NCHW if k == 0 else NHWC, ptr(dy), dy_layout,
                     ptr(self.g(lname + ".weight")), ptr(self.g(lname + ".bias")), B, HID, h, h, couts[k])
            if three:
                W3[k] = wargs
            elif eager:
                # both operands of this layer's weight gradient exist (dy: the previous input gradient or g_logit): side
                # stream, now, beside this layer's input gradient
                self._conv_wgrad(*wargs, fork=True)
            elif early_mode == 2 and h == 32 and not self.single_stream:
                self.fork_side()                 # (its launch follows this stream's next kernel, like every side launch)
                queued.append(wargs)
            else:
                (pending if h >= 16 else deferred).append(wargs)
            # the first decoder layer's input gradient leaves NCHW = (c,h,w) order, straight into gd3 (the
            # gradient of lin3's output; ReLU mask = lin3's output d3 in the same order): no relayout pass
            out_layout = NCHW if k == 0 else NHWC
            if k == 0 and fc_chain is not None and self._ends(B):
                pass                                 # convT_64's input gradient: the prologue of fc_chain_bwd
            elif couts[k] == HID:
                call("dvae_conv32_down", ptr(dy), self._img(lname, "down"), None, ptr(x_in), ptr(gx), out_layout, B, h,
                     ACT_NONE, s)
            elif self.mask_bits:
                call("dvae_convT3_dgrad_bits", ptr(dy), ptr(self.p(lname + ".weight")), ptr(buf.bits_convT2), ptr(gx), B,
                     couts[k], s)
            else:
                call("dvae_convT4s2_dgrad", ptr(dy), dy_layout, ptr(self.p(lname + ".weight")), ptr(x_in), ptr(gx),
                     out_layout, B, HID, h, h, couts[k], s)
            dy, dy_layout = gx, NHWC
            if three:
                # a weight gradient needs its layer's OUTPUT gradient, i.e. the input gradient of the layer above: behind convT3's
                # input gradient both convT3's and convT2's are due (one per side stream), behind convT2's convT1's
                last = len(names) - 1
                for q, w_ in queued:
                    self._wgrad_q(q, *w_)
                queued = []
                if k == last:
                    self._fork_q(1, 2)
                    queued = [(1, wargs)]        # (W3[last - 1] does not exist yet: built at the head of the next iteration)
                elif k == last - 1:
                    self._wgrad_q(2, *wargs)     # convT2's: forked behind convT3's input gradient, its operands were final there
                    self._fork_q(1)
                elif k == last - 2:
                    self._wgrad_q(1, *wargs)     # convT1's: forked behind convT2's input gradient
                continue
            if eager:
                continue
            for w_ in queued:                    # side launches of the previous fork, issued AFTER this stream's next kernel
                self._conv_wgrad(*w_, fork=False)
            queued = []
            early = early_mode == 1 and h == 32 and pending and not self.single_stream
            if h == 16 or (k == 0 and pending) or early:  # last big dgrad is enqueued: its inputs and those of `pending` are final
                self.fork_side()
                queued, pending = pending, []
        if three:
            for q, w_ in queued:
                self._wgrad_q(q, *w_)
            fc_chain()
            # behind the chain of FC input gradients: convT_64's weight gradient here, the FC layers' and the encoder's 4x4 end
            # in _encode_backward_3s (same fork)
            self._fork_q(1, 2)
            if len(names) == 4:
                self._wgrad_q(1, *W3[0])
            self._fc_pending = [(buf.d2, buf.gd3, self.g("decoder.lin3.weight"), self.g("decoder.lin3.bias"), B, HIDDEN_DIM, HID * 16),
                                (buf.d1, buf.gd2, self.g("decoder.lin2.weight"), self.g("decoder.lin2.bias"), B, HIDDEN_DIM, HIDDEN_DIM),
                                (z, buf.gd1, self.g("decoder.lin1.weight"), self.g("decoder.lin1.bias"), B, D, HIDDEN_DIM)]
            return
        for w_ in queued:
            self._conv_wgrad(*w_, fork=False)
        if fc_chain is not None:
            fc_chain()
        else:
            call("dvae_linear_dgrad", ptr(buf.gd3), ptr(self.p("decoder.lin3.weight")), ptr(buf.d2), ACT_RELU, ptr(buf.gd2),
                 B, HIDDEN_DIM, HID * 16, ws, s)
            call("dvae_linear_dgrad", ptr(buf.gd2), ptr(self.p("decoder.lin2.weight")), ptr(buf.d1), ACT_RELU, ptr(buf.gd1),
                 B, HIDDEN_DIM, HIDDEN_DIM, ws, s)
            call("dvae_linear_dgrad", ptr(buf.gd1), ptr(self.p("decoder.lin1.weight")), None, ACT_NONE, ptr(buf.dz),
                 B, D, HIDDEN_DIM, ws, s)
        # small conv layers + the three FC weight gradients: one fork, then they co-run with whatever follows
        if deferred or not defer_fc_wgrad:
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
        if join:
            self._join_side()

    def _encode_backward_3s(self, x, buf, B, fc, fused_end):
        """encode_backward of the three-queue schedule; _decode_backward's fork behind the FC chain covers the first launches."""
        s = _stream()
        c, H, _ = self.img_size
        last = len(self.enc_names) - 1
        W = {}
        for k in range(last, -1, -1):
            lname = "encoder.%s" % self.enc_names[k]
            h_in = self.enc_sizes[k] * 2
            x_in, x_layout, cin = (buf.enc_act[k - 1], NHWC, HID) if k > 0 else (x, NCHW, c)
            dy, dy_layout = (buf.ga_flat, NCHW) if k == last else (buf.enc_gact[k], NHWC)
            W[k] = (("dvae_conv4s2_wgrad", ptr(x_in), x_layout, ptr(dy), dy_layout, ptr(self.g(lname + ".weight")),
                     ptr(self.g(lname + ".bias")), B, cin, h_in, h_in, HID), dy, dy_layout, lname, x_in, h_in)
        # all six FC weight gradients + conv_64's (ga_flat is final): second side stream
        self._side_wgrad_grouped(fc, stream=self._wg2.cuda_stream)
        self._wgrad_q(2, *W[last][0])
        if not fused_end:                        # conv_64's input gradient as a launch of its own: conv3's weight gradient waits for it
            _, dy, dyl, lname, x_in, h_in = W[last]
            self._conv_dgrad(dy, dyl, lname, x_in, buf, last, B, h_in, s)
            self._fork_q(1)
        for k in range(last - 1, 0, -1):
            _, dy, dyl, lname, x_in, h_in = W[k]
            self._conv_dgrad(dy, dyl, lname, x_in, buf, k, B, h_in, s)     # this stream's next kernel first, then the side launch
            if k == last - 1:
                self._wgrad_q(1, *W[k][0])       # conv3's (operand: fc_chain_bwd's epilogue / the launch above)
                self._fork_q(2)                  # behind conv3's input gradient: conv2's weight gradient
            else:
                self._wgrad_q(2, *W[k][0])
        if x.dtype == torch.uint8:
            lname = "encoder.%s" % self.enc_names[0]
            call("dvae_conv4s2_wgrad_u8", ptr(x), ptr(buf.enc_gact[0]), ptr(self.g(lname + ".weight")),
                 ptr(self.g(lname + ".bias")), B, c, H, H, HID, ptr(self._ws), s)
        else:
            self._conv_wgrad(*W[0][0], fork=False, main=True)
        self._join_side()
        call("dvae_stream_order", self._wg2.cuda_stream, s)

    def _conv_dgrad(self, dy, dy_layout, lname, x_in, buf, k, B, h_in, s):
        """Input gradient of encoder conv layer k (> 0) into buf.enc_gact[k - 1], masked by the ReLU of layer k - 1: by the
        bit plane conv1's forward emitted where there is one (conv2 at the 64x64 geometry), else by the fp32 activation."""
        if self.mask_bits and k == 1:
            call("dvae_conv32_up_bits", ptr(dy), self._img(lname, "up"), None, ptr(buf.bits_conv1), ptr(buf.enc_gact[0]),
                 None, B, ACT_NONE, s)
        else:
            call("dvae_conv32_up", ptr(dy), dy_layout, self._img(lname, "up"), None, ptr(x_in), ptr(buf.enc_gact[k - 1]),
                 B, h_in // 2, ACT_NONE, s)

    def encode_backward(self, x, buf, n=None, fc_chain=False):
        """buf.dml (grad w.r.t. the interleaved mu/logvar output) -> encoder weight grads.  fc_chain: the three FC input
        gradients were already computed by fc_chain_bwd (buf.gh2, gh1, ga_flat are final)."""
        s = _stream()
        B = x.shape[0] if n is None else n
        c, H, _ = self.img_size
        ws = ptr(self._ws)
        if not fc_chain:
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
        eager = self.eager_wgrad and not self.single_stream
        fused_end = bool(fc_chain) and self._ends(B)        # conv_64's input gradient: fc_chain_bwd's epilogue wrote enc_gact[2]
        if self._three(bool(fc_chain)) and len(pend) == 3:
            return self._encode_backward_3s(x, buf, B, fc, fused_end)
        # the grouped FC weight gradients: side stream (in front of conv2's weight gradient) -- or, fcw_main, the LAST launch of the
        # main stream's tail (a step of a few hundred images ends on the side stream: profiles/r06_final4_dsprites_timeline.md)
        fcw_main = (self.fcw_main and self.is64 and not self.single_stream
                    and self.fcw_main_rows[0] <= B <= self.fcw_main_rows[1])
        deferred = [] if fcw_main else [lambda fc=fc: self._side_wgrad_grouped(fc)]
        tail_main = []                      # weight gradients the main stream computes after conv1's (load balance of the tail)
        last = len(self.enc_names) - 1
        if eager:
            # dependency-driven schedule: the six FC weight gradients now (every operand exists: the chain of input gradients
            # is behind us), then each conv layer's weight gradient beside its input gradient; conv1 has no input gradient:
            # its weight gradient is the main stream's last kernel
            self.fork_side()
            self._side_wgrad_grouped(fc)
            for k in range(last, -1, -1):
                name = self.enc_names[k]
                lname = "encoder.%s" % name
                h_in = self.enc_sizes[k] * 2
                x_in, x_layout, cin = (buf.enc_act[k - 1], NHWC, HID) if k > 0 else (x, NCHW, c)
                dy, dy_layout = (buf.ga_flat, NCHW) if k == last else (buf.enc_gact[k], NHWC)
                wargs = ("dvae_conv4s2_wgrad", ptr(x_in), x_layout, ptr(dy), dy_layout,
                         ptr(self.g(lname + ".weight")), ptr(self.g(lname + ".bias")), B, cin, h_in, h_in, HID)
                if k == 0:
                    if x.dtype == torch.uint8:
                        call("dvae_conv4s2_wgrad_u8", ptr(x), ptr(dy), ptr(self.g(lname + ".weight")),
                             ptr(self.g(lname + ".bias")), B, cin, h_in, h_in, HID, ptr(self._ws), s)
                    else:
                        self._conv_wgrad(*wargs, fork=False, main=True)
                    break
                self._conv_wgrad(*wargs, fork=k < last)      # (k == last: the fork above covers it)
                if not (k == last and fused_end):
                    self._conv_dgrad(dy, dy_layout, lname, x_in, buf, k, B, h_in, s)
            self._join_side()
            return
        for k in range(last, -1, -1):
            name = self.enc_names[k]
            lname = "encoder.%s" % name
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
                     ptr(self.g(lname + ".weight")), ptr(self.g(lname + ".bias")), B, cin, h_in, h_in, HID)
            side = []
            if k == 0:
                # the first layer has no dgrad: this stream has nothing else left, so it computes the last
                # weight gradient itself (no fork) while the side stream drains its queue
                if deferred:
                    self.fork_side()
                    side, deferred = deferred, []
                if x.dtype == torch.uint8:
                    call("dvae_conv4s2_wgrad_u8", ptr(x), ptr(dy), ptr(self.g(lname + ".weight")),
                         ptr(self.g(lname + ".bias")), B, cin, h_in, h_in, HID, ptr(self._ws), s)
                else:
                    self._conv_wgrad(*wargs, fork=False, main=True)
                for w_ in tail_main:
                    self._conv_wgrad(*w_, fork=False, main=True)
                if fcw_main:
                    self._side_wgrad_grouped(fc, stream=s)
            elif name in self.tail_main and self.is64 and not self.single_stream:
                tail_main.append(wargs)
            elif big:
                self.fork_side()
                side, deferred = deferred + [lambda wargs=wargs: self._conv_wgrad(*wargs, fork=False)], []
            else:
                deferred.append(lambda wargs=wargs: self._conv_wgrad(*wargs, fork=False))
            if k > 0 and not (k == last and fused_end):   # this stream's next kernel first, then the side launches
                self._conv_dgrad(dy, dy_layout, lname, x_in, buf, k, B, h_in, s)
            for launch in side:
                launch()
        if deferred:
            self.fork_side()
            for launch in deferred:
                launch()
        self._join_side()
