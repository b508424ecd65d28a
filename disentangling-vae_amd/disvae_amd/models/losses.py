"""Loss plugins with the API of disvae/models/losses.py (get_loss_f, BaseLoss, BetaHLoss,
BetaBLoss, FactorKLoss, BtcvaeLoss, LOSSES, RECON_DIST) on the HIP kernels.

Two ways in:
  * ``loss(data, recon, latent_dist, is_train, storer, latent_sample=...)`` -- the reference
    signature (losses.py:78); returns a 0-d tensor wired into autograd through thin
    torch.autograd.Function wrappers around the kernels, so the reference's own
    Trainer / Evaluator code keeps working with the native model;
  * ``loss.fused_step(data, model, optimizer, storer)`` -- what the native Trainer uses: one
    stream of kernel launches for forward + loss + backward (no autograd graph, gradients
    land in the flat arena, all scalars in one small device buffer), then optimizer.step().
Host-side state (n_train_steps, annealing, storer cadence) follows losses.py:71-75,105-114.
"""
import abc

import torch

from .. import _lib
from .. import optim
from .._lib import call, ptr, record_on_stream, record_py
from ..utils.math import log_importance_weights
from .discriminator import Discriminator
from ..graph import StepGraphs
from ..parallel import scale_, copy_flat_
from .._debug import knob

LOSSES = ["VAE", "betaH", "betaB", "factor", "btcvae"]  # losses.py:17
RECON_DIST = ["bernoulli", "laplace", "gaussian"]        # losses.py:18


def get_loss_f(loss_name, **kwargs_parse):
    """losses.py:22-49 -- same keys consumed."""
    kwargs_all = dict(rec_dist=kwargs_parse["rec_dist"], steps_anneal=kwargs_parse["reg_anneal"])
    if loss_name == "betaH":
        return BetaHLoss(beta=kwargs_parse["betaH_B"], **kwargs_all)
    elif loss_name == "VAE":
        return BetaHLoss(beta=1, **kwargs_all)
    elif loss_name == "betaB":
        return BetaBLoss(C_init=kwargs_parse["betaB_initC"], C_fin=kwargs_parse["betaB_finC"],
                         gamma=kwargs_parse["betaB_G"], **kwargs_all)
    elif loss_name == "factor":
        return FactorKLoss(kwargs_parse["device"], gamma=kwargs_parse["factor_G"],
                           disc_kwargs=dict(latent_dim=kwargs_parse["latent_dim"]),
                           optim_kwargs=dict(lr=kwargs_parse["lr_disc"], betas=(0.5, 0.9)), **kwargs_all)
    elif loss_name == "btcvae":
        return BtcvaeLoss(kwargs_parse["n_data"], alpha=kwargs_parse["btcvae_A"], beta=kwargs_parse["btcvae_B"],
                          gamma=kwargs_parse["btcvae_G"], **kwargs_all)
    else:
        assert loss_name not in LOSSES
        raise ValueError("Uknown loss : {}".format(loss_name))


def linear_annealing(init, fin, step, annealing_steps):
    """losses.py:511-518."""
    if annealing_steps == 0:
        return fin
    assert fin > init
    delta = fin - init
    return min(init + delta * step / annealing_steps, fin)


from ..engine import _stream  # noqa: E402


class _Scratch:
    """Small device buffers shared by the loss kernels of one loss object."""

    def __init__(self, device):
        _lib.note_alloc()
        f = lambda n: torch.zeros(n, dtype=torch.float32, device=device)
        self.device = device
        self.coef = f(_lib.NCOEF)
        self.coef_host = [0.0] * _lib.NCOEF
        self.scal = f(_lib.NSCAL)
        self.packed = f(_lib.NPACK)
        self.partials = f(_lib.REC_NPART)
        self.kl_dim = f(_lib.KL_FLOATS)   # DVAE_KL_FLOATS: per-dim KL + per-workgroup partial blocks
        self.disc_sums = f(4)
        self.log_w = f(4)
        self._log_w_key = None
        self.rowstats = None
        self.lat = {}

    def for_latent_dim(self, D):
        """`scal` / `packed` sized for latent dimension D (include/dvae_hip.h: above DVAE_MAX_D the per-dimension KL values follow
        the 32 fixed slots) -- grown once; recorded launch plans are invalidated with the allocation."""
        if self.scal.numel() < _lib.nscal(D):
            _lib.note_alloc()
            self.scal = torch.zeros(_lib.nscal(D), dtype=torch.float32, device=self.device)
            self.packed = torch.zeros(_lib.npack(D), dtype=torch.float32, device=self.device)
        if self.kl_dim.numel() < D:
            _lib.note_alloc()
            self.kl_dim = torch.zeros(D, dtype=torch.float32, device=self.device)
        return self

    def set_coef(self, **kw):
        h = self.coef_host
        for k, v in kw.items():
            h[getattr(_lib, "C_" + k)] = float(v)
        # values travel as kernel arguments: ordered with the stream, no host sync
        call("dvae_set_coef", ptr(self.coef), *h, _stream())

    def set_coef_host(self, **kw):
        """Host copy only: the values reach the device with the step's weight-staging launch (engine.stage)."""
        h = self.coef_host
        for k, v in kw.items():
            h[getattr(_lib, "C_" + k)] = float(v)

    def set_log_w(self, batch, n_data):
        key = (batch, n_data)
        if key != self._log_w_key:
            self.log_w[:3].copy_(log_importance_weights(batch, n_data))
            self._log_w_key = key

    def latent(self, name, rows, cols):
        t = self.lat.get((name, rows, cols))
        if t is None:
            _lib.note_alloc()
            t = torch.empty(rows, cols, dtype=torch.float32, device=self.device)
            self.lat[(name, rows, cols)] = t
        return t


class BaseLoss(abc.ABC):
    """losses.py:53-114."""

    def __init__(self, record_loss_every=50, rec_dist="bernoulli", steps_anneal=0):
        self.n_train_steps = 0
        self.record_loss_every = record_loss_every
        self.rec_dist = rec_dist
        self.steps_anneal = steps_anneal
        self._scratch = None
        self.comm = None   # set by disvae_amd.parallel.data_parallel for sharded batches
        # sharded batches: "global" = the B x B estimator / permute_dims couple the GLOBAL batch (equal to the
        # single-process step on the concatenated batch); "local" = every rank's shard is its own minibatch
        # (what running the reference under DistributedDataParallel would compute: a different estimator)
        self.estimator = "global"
        # how the device side of the native training iteration is issued (graph.py): None = eager
        # Python; "plan" = recorded launch list (the same launches on the same streams, bit-identical
        # results); "graph" = hipGraph; "auto" (default) = plan while the iteration is launch-bound
        # (batch tensor <= AUTO_PLAN_ELEMS elements: measured cross-over), eager
        # above.  Sharded steps replay too: collectives are recorded plan entries (parallel.py)
        mode = knob("DVAE_REPLAY", "auto")
        if mode not in ("plan", "graph", "eager", "auto"):
            raise ValueError("DVAE_REPLAY={!r}: expected one of auto, eager, plan, graph".format(mode))
        self.replay = {"plan": "plan", "graph": "graph", "eager": None, "auto": "auto"}[mode]
        self._graphs = StepGraphs()
        self._static = {}
        # dvae_event_record / dvae_event_wait slot of this loss object: "the estimator and the scalar loss of this step are final"
        self._ev_slot = _lib.next_event_slot()

    def _static_buf(self, name, like):
        """Persistent device buffer with the shape/dtype of `like`, refreshed with its contents."""
        key = (name, tuple(like.shape), like.dtype)
        t = self._static.get(key)
        if t is None:
            _lib.note_alloc()
            t = self._static[key] = torch.empty(like.shape, dtype=like.dtype, device=self._scratch.device)
        if t.data_ptr() != like.data_ptr():
            t.copy_(like, non_blocking=True)
        return t

    # (round 6: up to 1024 images -- the same step time at 512 / 1024 images single process, 0.23-0.27 instead of 0.36 ms of host
    # time per step; one rank of two of configs[3] (512 images through the sharded path) spends 0.55 ms of host per 0.63 ms step
    # when it is issued eagerly: profiles/r06_s2_shard_world.txt)
    AUTO_PLAN_ELEMS = 1024 * 3 * 64 * 64
    # one HIP stream instead of two below this many input elements per step (engine.single_stream); DVAE_STREAMS=1|2 forces
    # (round 2 measured the cross-over at 64 images, profiles/r02_run10_streams.txt; with the round-5 schedule two streams win at
    # 32 and 64 images as well: 0.291 / 0.301 against 0.338 / 0.346 ms, profiles/r05_v26_sweep.txt; round 6: at 4 / 8 / 16 images
    # too -- 0.306 -> 0.267, 0.312 -> 0.271, 0.315 -> 0.277 ms -- and at the 32x32 geometry level at 16 / 64 images, -3 % at 128:
    # profiles/r06_s2_streams_small.txt.  Two streams at every size.)
    SINGLE_STREAM_ELEMS = int(knob("DVAE_SINGLE_STREAM_ELEMS", 0))

    # dependency-driven weight-gradient schedule (engine.eager_wgrad: a fork per layer) up to this many input elements per
    # step; above, the batch-sized schedule (two forks per half of the backward pass).  Round 3 measured the two within noise
    # of each other up to 384 images; with the round-5 kernels the batch-sized schedule wins at every batch measured (128
    # images: 0.349 vs 0.359 ms, btcvae_dsprites 0.421 vs 0.433 ms: profiles/r05_v25_schedule_ab.txt) -- each fork costs the
    # critical path an event and the small weight gradients it frees early are not what the iteration waits for
    EAGER_WGRAD_ELEMS = int(knob("DVAE_EAGER_WGRAD_ELEMS", 0))
    # weight gradients on TWO side streams (engine.three_streams) from this many batch rows per step (single process).  Measured
    # (profiles/r06_s2_three1.txt, same box, three alternations): SLOWER for every beta-TCVAE step -- 64 / 128 / 256 / 512 / 1024
    # images 0.288 -> 0.300, 0.328 -> 0.347, 0.435 -> 0.441, 0.634 -> 0.648, 1.050 -> 1.060 ms: whatever runs beside the main
    # stream's chain of small kernels slows that chain by more than the side streams gain -- and faster only where the side
    # stream also carries the discriminator's chain: factor 64x64x3 tensor 2048 1.853 -> 1.832 ms (FactorKLoss below)
    THREE_STREAM_MIN_ROWS = 1 << 30
    # sharded batches up to this many input elements per rank: ONE all-reduce of the whole gradient arena at the end instead of
    # two overlapped spans (the step is a latency chain; every collective costs the host and both streams more than the
    # overlap of 1 MB buys).  Round 6: at EVERY size -- as one rank of two (512 images) the two-span path takes 1.06-1.16 ms
    # against 0.63 with one all-reduce (single process: 0.63), FactorVAE tensor 1024 / 512 per rank 1.31 / 0.93 against 1.18 / 0.83
    # (profiles/r06_s2_shard_world.txt: mirrored world, C-ABI transport); the 2 MB arena is ~20 us of xGMI time, there is
    # nothing worth overlapping.  (The spans stay reachable for A/B: DVAE_DEBUG=1 DVAE_SMALL_SHARD_ELEMS=<elements>.)
    SMALL_SHARD_ELEMS = int(knob("DVAE_SMALL_SHARD_ELEMS", 1 << 40))

    def _streams(self, model, data):
        mode = knob("DVAE_STREAMS", "auto")
        single = mode == "1" or (mode == "auto" and data.numel() <= self.SINGLE_STREAM_ELEMS)
        model.engine.single_stream = bool(single) and self._world()[0] == 1
        model.engine.eager_wgrad = data.numel() <= int(knob("DVAE_EAGER_WGRAD_ELEMS", self.EAGER_WGRAD_ELEMS))
        model.engine.sharded = self._world()[0] > 1
        model.engine.three_streams = (not single and self._world()[0] == 1
                                      and data.shape[0] >= int(knob("DVAE_THREE_STREAM_MIN_ROWS", self.THREE_STREAM_MIN_ROWS)))
        tm = knob("DVAE_TAIL_MAIN", "default")      # A/B (DVAE_DEBUG=1): which encoder weight gradients end the main stream
        if tm != "default":
            model.engine.tail_main = tuple(v for v in tm.split(",") if v)
        return model.engine.single_stream

    def _replay_mode(self, is_train, data):
        # (sharded batches replay too: the collectives and the torch ops around them are recorded plan entries -- C-ABI calls
        # with the RCCL transport, host callables re-entering their stream with torch.distributed: disvae_amd/parallel.py)
        if not is_train:
            return None
        if self.replay == "auto":
            return "plan" if data.numel() <= self.AUTO_PLAN_ELEMS else None
        return self.replay

    def _replay_key(self, model, data, injected):
        """Everything a recorded launch freezes: buffers (allocation generation), arenas, the
        batch pointer, the stream, and the few Python-side switches passed as scalars."""
        return (id(model), data.shape, data.data_ptr(), injected, _stream(), model.arena.flat.data_ptr(),
                model.arena.grad.data_ptr(), _lib.ALLOC_GEN[0], self.rec_dist, getattr(self, "is_mss", None),
                model.engine.single_stream, model.engine.eager_wgrad, model.engine.tail_main, id(self.comm), self.estimator,
                model.engine.three_streams, model.engine.sharded)

    @abc.abstractmethod
    def __call__(self, data, recon_data, latent_dist, is_train, storer, **kwargs):
        pass

    def _pre_call(self, is_train, storer):
        if is_train:
            self.n_train_steps += 1
        if not is_train or self.n_train_steps % self.record_loss_every == 1:
            storer = storer
        else:
            storer = None
        return storer

    def scratch(self, device):
        if self._scratch is None or self._scratch.device != device:
            self._scratch = _Scratch(device)
        return self._scratch

    def _pack_sums(self, sc, klb, D, rowstats, Bl, disc_sums, packed, stream):
        """This rank's partial sums -> `packed` (sharded batches: sum-all-reduced over the ranks before dvae_loss_finalize) in
        ONE launch: dvae_loss_epilogue without `scal` is dvae_kl_finish (the klb KL partial blocks the fused FC chain left) +
        dvae_loss_pack, bit for bit (tests/test_gpu_kernels.py::test_loss_epilogue_equals_pack_then_finalize) -- one dependent
        launch less on the exchange stream, which the FC chain's input gradients wait for."""
        call("dvae_loss_epilogue", _lib.LOSS_BETAH, ptr(sc.partials), ptr(sc.kl_dim), klb, D, ptr(rowstats), Bl, ptr(disc_sums), 1,
             ptr(sc.coef), ptr(packed), None, stream)

    def _rec_code(self):
        if self.rec_dist not in _lib.REC:
            assert self.rec_dist not in RECON_DIST
            raise ValueError("Unkown distribution: {}".format(self.rec_dist))  # losses.py:442
        return _lib.REC[self.rec_dist]

    # world size / rank of the data-parallel group (1 / 0 without a communicator)
    def _world(self):
        return (1, 0) if self.comm is None else (self.comm.world_size, self.comm.rank)

    def _est_world(self):
        """(world, rank) as seen by the batch-coupled estimators."""
        return (1, 0) if (self.comm is None or self.estimator == "local") else (self.comm.world_size, self.comm.rank)

    @staticmethod
    def _store_common(storer, vals, D):
        storer['recon_loss'].append(vals[_lib.S_REC])

    @staticmethod
    def _store_kl(storer, vals, D):
        storer['kl_loss'].append(vals[_lib.S_KL])
        for i in range(D):
            storer['kl_loss_' + str(i)].append(vals[_lib.kl0(D) + i])


# ------------------------------------------------------------------------------------------
# autograd-compatible pieces (reference call signature)
# ------------------------------------------------------------------------------------------
class _ReconLossFn(torch.autograd.Function):
    """_reconstruction_loss (losses.py:394-449): sum over pixels / batch."""

    @staticmethod
    def forward(ctx, recon, data, dist_code, scratch):
        recon, data = recon.contiguous(), data.contiguous()
        B = recon.shape[0]
        scratch.set_coef(INV_B=1.0 / B)
        g = torch.empty_like(recon)
        call("dvae_recon_loss", ptr(recon), ptr(data), recon.numel(), dist_code, ptr(scratch.coef),
             ptr(scratch.partials), ptr(g), 0, _stream())
        ctx.save_for_backward(g)
        out = torch.empty((), dtype=torch.float32, device=recon.device)
        call("dvae_reduce_sum", ptr(scratch.partials), _lib.REC_NPART, 1.0 / B, ptr(out), _stream())
        return out

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * gout, None, None, None


class _KLFn(torch.autograd.Function):
    """_kl_normal_loss (losses.py:452-480) -> per-dim KL [D] (mean over batch)."""

    @staticmethod
    def forward(ctx, mu, logvar, scratch):
        B, D = mu.shape
        ml = torch.stack((mu, logvar), dim=-1).reshape(B, 2 * D).contiguous()
        scratch.set_coef(INV_B=1.0 / B)
        tmp = torch.empty(3, B, D, dtype=torch.float32, device=mu.device)
        kl_dim = torch.empty(max(16 + 64 * 16, D), dtype=torch.float32, device=mu.device)
        call("dvae_reparam_kl_fwd", ptr(ml), None, ptr(tmp[0]), ptr(tmp[1]), ptr(tmp[2]), ptr(kl_dim),
             ptr(scratch.coef), B, D, _stream())
        ctx.save_for_backward(mu, logvar)
        return kl_dim[:D].clone()

    @staticmethod
    def backward(ctx, gout):
        mu, logvar = ctx.saved_tensors
        B, D = mu.shape
        mu, logvar, gout = mu.contiguous(), logvar.contiguous(), gout.contiguous()
        gmu, glv = torch.empty_like(mu), torch.empty_like(logvar)
        call("dvae_kl_normal_bwd", ptr(gout), ptr(mu), ptr(logvar), ptr(gmu), ptr(glv), B, D, _stream())
        return gmu, glv, None


class _BtcvaeFn(torch.autograd.Function):
    """_get_log_pz_qz_prodzi_qzCx + the three batch means (losses.py:364-373, 523-544) ->
    tensor [mi, tc, dw_kl]; backward through dvae_btcvae_bwd for each requested term."""

    @staticmethod
    def forward(ctx, z, mu, logvar, n_data, is_mss, scratch):
        z, mu, logvar = z.contiguous(), mu.contiguous(), logvar.contiguous()
        B, D = z.shape
        scratch.set_log_w(B, n_data)
        rowstats = torch.empty(B, _lib.rowstats_stride(D), dtype=torch.float32, device=z.device)
        tmp = torch.empty(_lib.btcvae_tmp_floats(B, B, D), dtype=torch.float32, device=z.device)
        call("dvae_btcvae_fwd", ptr(z), ptr(mu), ptr(logvar), B, D, 0, B, int(is_mss), ptr(scratch.log_w),
             ptr(tmp), ptr(rowstats), _stream())
        # batch means of the four log-densities -> (mi, tc, dw_kl) by the scalar epilogue kernels (losses.py:369-373); no KL
        # values are passed (kl_dim = NULL): D only tells the kernels the row stride of `rowstats`
        packed = torch.empty(_lib.npack(D), dtype=torch.float32, device=z.device)
        scal = torch.empty(_lib.nscal(D), dtype=torch.float32, device=z.device)
        call("dvae_loss_pack", ptr(scratch.partials), None, D, ptr(rowstats), B, None, ptr(packed), _stream())
        call("dvae_loss_finalize", _lib.LOSS_BTCVAE, ptr(packed), D, B, ptr(scratch.coef), ptr(scal), _stream())
        ctx.save_for_backward(z, mu, logvar, rowstats, tmp)
        ctx.is_mss, ctx.scratch = is_mss, scratch
        return scal[_lib.S_MI:_lib.S_DWKL + 1].clone()       # [mi, tc, dw_kl]

    @staticmethod
    def backward(ctx, gout):
        z, mu, logvar, rowstats, tmp = ctx.saved_tensors
        B, D = z.shape
        # d(a*mi + b*tc + c*dw)/d(.) with (alpha, beta, gamma*anneal) = (a, b, c)
        a, b, c = [float(v) for v in gout.tolist()]
        coef = torch.zeros(_lib.NCOEF, dtype=torch.float32)
        coef[_lib.C_ALPHA], coef[_lib.C_BETA], coef[_lib.C_GAMMA], coef[_lib.C_ANNEAL] = a, b, c, 1.0
        coef = coef.to(z.device)
        dz, dmu, dlv = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        call("dvae_btcvae_bwd", ptr(z), ptr(mu), ptr(logvar), ptr(rowstats), B, D, 0, B, int(ctx.is_mss),
             ptr(ctx.scratch.log_w), ptr(coef), ptr(tmp), ptr(dz), ptr(dmu), ptr(dlv), _stream())
        return dz, dmu, dlv, None, None, None


def u8_to_f32(x):
    """ToTensor's arithmetic on a uint8 device tensor: float(v) / 255 (dvae_u8_to_f32)."""
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("dvae_u8_to_f32", ptr(x), ptr(out), x.numel(), _stream())
    return out


def _reconstruction_loss(data, recon_data, distribution="bernoulli", storer=None, scratch=None):
    """losses.py:394-449."""
    if distribution not in _lib.REC:
        assert distribution not in RECON_DIST
        raise ValueError("Unkown distribution: {}".format(distribution))
    scratch = scratch or _Scratch(recon_data.device)
    if data.dtype == torch.uint8:              # pixel batches: ToTensor (utils/datasets.py:207-209) on the device
        data = u8_to_f32(data)
    loss = _ReconLossFn.apply(recon_data, data, _lib.REC[distribution], scratch)
    if distribution == "laplace":
        loss = loss * (loss != 0)
    if storer is not None:
        storer['recon_loss'].append(loss.item())
    return loss


def _kl_normal_loss(mean, logvar, storer=None, scratch=None):
    """losses.py:452-480."""
    scratch = scratch or _Scratch(mean.device)
    latent_kl = _KLFn.apply(mean, logvar, scratch)
    total_kl = latent_kl.sum()
    if storer is not None:
        storer['kl_loss'].append(total_kl.item())
        for i in range(mean.size(1)):
            storer['kl_loss_' + str(i)].append(latent_kl[i].item())
    return total_kl


def _permute_dims(latent_sample, perms=None):
    """losses.py:483-508; the permutations are drawn with torch.randperm on the CPU generator
    like the reference (:505) unless injected (perms: int64 [D,B])."""
    B, D = latent_sample.shape
    if perms is None:
        perms = torch.stack([torch.randperm(B) for _ in range(D)])
    perms = perms.to(device=latent_sample.device, dtype=torch.int64).contiguous()
    out = torch.empty_like(latent_sample)
    call("dvae_permute_dims", ptr(latent_sample.contiguous()), ptr(perms), ptr(out), B, D, _stream())
    return out


# ------------------------------------------------------------------------------------------
# loss classes
# ------------------------------------------------------------------------------------------
class _SingleOptimizerLoss(BaseLoss):
    """Shared fused step of BetaH / BetaB / Btcvae (training.py:152-158 + the loss __call__)."""

    KIND = None

    def _coefs(self, is_train):
        raise NotImplementedError

    def _store(self, storer, vals, D):
        raise NotImplementedError

    def fused_step(self, data, model, optimizer, storer, eps=None):
        is_train = model.training
        storer = self._pre_call(is_train, storer)
        B, D = data.shape[0], model.latent_dim
        world, rank = self._world()
        sc = self.scratch(data.device).for_latent_dim(D)
        # ONE launch: this step's weight images (32-channel conv layers, FC chain) + its loss coefficients
        sc.set_coef_host(INV_B=1.0 / (B * world), **self._coefs(is_train))
        model.engine.stage(sc.coef, sc.coef_host)
        data = data.contiguous()
        self._streams(model, data)
        if self.KIND == _lib.LOSS_BTCVAE:
            sc.set_log_w(B * self._est_world()[0], self.n_data)
        mode = self._replay_mode(is_train, data)
        if mode:
            # a replay re-issues launches with frozen pointers: injected noise goes through a static
            # buffer; the batch pointer is part of the plan key (a hipGraph needs it static as well)
            if mode == "graph":
                data = self._static_buf("data", data)
            if eps is not None:
                eps = self._static_buf("eps", eps)
            self._graphs.run(self._replay_key(model, data, eps is not None),
                             lambda: self._device_step(data, model, sc, eps, True), mode)
        else:
            self._device_step(data, model, sc, eps, is_train)
        if is_train:
            model.assign_grads()          # optimizer.zero_grad(); loss.backward()  (training.py:156-157)
            optim.step(optimizer)         # optimizer.step(), training.py:158 (one launch for a stock Adam: disvae_amd/optim.py)
        if storer is not None:
            vals = sc.scal.tolist()       # ONE device->host copy for every logged scalar
            self._store(storer, vals, D)
        return sc.scal[_lib.S_LOSS]

    def _device_step(self, data, model, sc, eps, is_train):
        """Forward + loss + backward as one stream of launches; no host-dependent values."""
        eng = model.engine
        eng._fork_hook = None                  # a hook left behind by a step that raised is dropped, not run (nor recorded)
        B, D = data.shape[0], model.latent_dim
        world, rank = self._world()
        Bg = B * world
        buf = eng.buffers(B)
        s = _stream()
        data = eng.input(data, buf)            # uint8 pixel batches: fused /255 (or one ToTensor pass), see engine.input
        if is_train and eps is None:
            eps = sc.latent("eps", B, D)
            record_py(eps.normal_)             # = torch.randn_like (vae.py:67): same Philox consumption
        if not is_train:
            eps = None
        eng.encode_convs(data, buf, chain=True)
        # the FC core in one launch: lin1 -> lin2 -> mu_logvar -> reparameterise (+ KL partial blocks) -> lin1 -> lin2 -> lin3
        # (latent dimensions above 16: one launch per layer, kl_dim final at once and klb = 0 -- engine.fc_chain_fwd)
        eng.fc_chain_fwd(buf, eps, sc.kl_dim, B, coef=sc.coef)
        klb = eng.kl_blocks(B)            # single process: the one-launch loss epilogue finishes the KL partials
        npk = _lib.npack(D)
        lat = {"rowstats": None, "dz": None, "dmu": None, "dlv": None, "xbuf": None}
        btc = self.KIND == _lib.LOSS_BTCVAE

        def estimator():
            # the B x B estimator (forward AND backward: it needs z, mu, logvar and the coefficients only) on the side stream
            # while the decoder forward occupies the current one
            ew, er = self._est_world()            # the estimator's view of the sharding (local mode: one shard = one batch)
            Be = B * ew
            with torch.cuda.stream(eng.aux_stream if world > 1 else eng.side_stream):
                ss = _stream()
                zg, mug, lvg = buf.z, buf.mu, buf.logvar
                if ew > 1:
                    zg, mug, lvg = self.comm.all_gather_latents(buf.z, buf.mu, buf.logvar)
                rowstats = lat["rowstats"] = sc.latent("rowstats", B, _lib.rowstats_stride(D))
                tc_tmp = sc.latent("tc_tmp", 1, _lib.btcvae_tmp_floats(Be, B, D))
                call("dvae_btcvae_fwd", ptr(zg), ptr(mug), ptr(lvg), Be, D, er * B, B, int(self.is_mss), ptr(sc.log_w),
                     ptr(tc_tmp), ptr(rowstats), ss)
                if is_train:
                    dz_x = sc.latent("dz_tc", B, D)
                    # (dmu, dlogvar) of ALL columns: two slabs of one buffer, followed by the packed loss sums -- sharded, the
                    # lot is summed over the ranks by ONE all-reduce in the step's late epilogue (Comm.all_reduce_cols_sums)
                    xbuf = sc.latent("xbuf", 1, 2 * Be * D + npk).view(-1)
                    dmu_all, dlv_all = xbuf[:Be * D].view(Be, D), xbuf[Be * D:2 * Be * D].view(Be, D)
                    call("dvae_btcvae_bwd", ptr(zg), ptr(mug), ptr(lvg), ptr(rowstats), Be, D, er * B, B,
                         int(self.is_mss), ptr(sc.log_w), ptr(sc.coef), ptr(tc_tmp), ptr(dz_x), ptr(dmu_all), ptr(dlv_all), ss)
                    if ew > 1:
                        lat["xbuf"] = xbuf
                        dmu_x, dlv_x = dmu_all[er * B:(er + 1) * B], dlv_all[er * B:(er + 1) * B]
                    else:
                        dmu_x, dlv_x = dmu_all, dlv_all
                    if world > ew:                # local estimator: its mean runs over B, the loss over B * world
                        for t_ in (dz_x, dmu_x, dlv_x):
                            scale_(t_, 1.0 / world)
                    lat["dz"], lat["dmu"], lat["dlv"] = dz_x, dmu_x, dlv_x

        fuse = (data, self._rec_code(), sc.coef, sc.partials)
        if btc and world == 1:
            eng.fork_side()
            estimator()
        elif btc:
            # sharded: the estimator with its exchanges on a stream of its own (engine.buffers: the side stream is the tail of
            # the iteration, nothing may queue in front of its weight gradients)
            call("dvae_stream_order", s, eng._aux_raw())
        # decoder convT stack; its last layer also evaluates the reconstruction likelihood and dL/dlogit
        eng.decode_convs(buf, B, fuse_loss=fuse, chain=True)
        if btc and world > 1:
            # sharded: this stream's launches are issued FIRST -- the exchanges make the side stream's part long to issue, and
            # at a hundred images per GPU the host is what the critical path would wait for
            estimator()
        rowstats = lat["rowstats"]
        # Nothing on this stream needs the estimator (or the scalar loss) before the FC chain's input gradients, a whole convT
        # backward later -- the estimator's backward kernels run past the end of the decoder forward, and joining here left
        # this stream idle for ~20 us plus the epilogue (profiles/r04_v35_btcvae_celeba_timeline.md).  The epilogue goes to the
        # side stream behind them, an event slot marks the lot, fc_chain() waits for the slot.  Sharded batches (any loss):
        # the all-reduce of the packed loss sums sits between the two halves of the epilogue, on the side stream as well.
        late_join = (is_train and not eng.single_stream and (btc or world > 1) and knob("DVAE_LATE_JOIN", "1") != "0")
        if btc and not late_join:
            if world > 1:
                call("dvae_stream_order", eng._aux_raw(), s)
            else:
                eng._join_side()
        if late_join:
            def epilogue():                       # after the next fork (the backward pass's first): no fork of its own
                ss = eng._side_raw()
                if world == 1:
                    call("dvae_loss_epilogue", self.KIND, ptr(sc.partials), ptr(sc.kl_dim), klb, D, ptr(rowstats), B, None, Bg,
                         ptr(sc.coef), ptr(sc.packed), ptr(sc.scal), ss)
                else:
                    # on the exchange stream, behind the estimator; ordered after this stream's fork point through the side
                    # stream (whose only queued work at this moment is the wait for that fork): no second event on this stream
                    ss = eng._aux_raw()
                    call("dvae_stream_order", eng._side_raw(), ss)
                    xbuf = lat["xbuf"]
                    packed = sc.packed if xbuf is None else xbuf[xbuf.numel() - npk:]
                    self._pack_sums(sc, klb, D, rowstats, B, None, packed, ss)
                    with torch.cuda.stream(eng.aux_stream):
                        if xbuf is None:
                            self.comm.all_reduce(packed)
                        else:                     # + the estimator's column gradients: one collective
                            self.comm.all_reduce_cols_sums(xbuf, B, D, npk)
                    call("dvae_loss_finalize", self.KIND, ptr(packed), D, Bg, ptr(sc.coef), ptr(sc.scal), ss)
                call("dvae_event_record", self._ev_slot, ss)
            eng.at_next_fork(epilogue)
        elif world > 1:
            self._pack_sums(sc, klb, D, rowstats, B, None, sc.packed, s)
            self.comm.all_reduce(sc.packed)
            call("dvae_loss_finalize", self.KIND, ptr(sc.packed), D, Bg, ptr(sc.coef), ptr(sc.scal), s)
        else:
            call("dvae_loss_epilogue", self.KIND, ptr(sc.partials), ptr(sc.kl_dim), klb, D, ptr(rowstats), B, None, Bg,
                 ptr(sc.coef), ptr(sc.packed), ptr(sc.scal), s)
        if not is_train:
            return

        def fc_chain():        # the six FC input gradients + the reparameterisation / KL backward in ONE launch
            if late_join:
                eng.flush_fork_hook()
                call("dvae_event_wait", self._ev_slot, s)
            eng.fc_chain_bwd(buf, eps, lat["dz"], None, lat["dmu"], lat["dlv"], sc.scal, sc.coef, B)

        # one join, at the end of the backward pass.  Single process: ONE grouped launch for all six FC weight gradients
        # (issued by encode_backward).  Sharded: the decoder's three are launched with the decoder's conv weight gradients --
        # every kernel that writes a decoder gradient goes to the side stream, so the all-reduce of the decoder span is ordered
        # behind the SIDE stream and overlaps the encoder backward; this stream never waits for it before the end.  Small shards
        # (SMALL_SHARD_ELEMS: the step is a latency chain, and every collective costs the host and both streams more than the
        # overlap of 1 MB buys): ONE all-reduce of the whole arena after the final join.
        spans = world > 1 and data.numel() > self.SMALL_SHARD_ELEMS
        eng.decode_backward(buf.z, buf, join=False, defer_fc_wgrad=not spans, fc_chain=fc_chain)
        pending = []
        if spans:
            with torch.cuda.stream(eng.side_stream):
                pending.append(self.comm.all_reduce_async(model.arena.span("decoder.")))
        eng.encode_backward(data, buf, fc_chain=True)
        if spans:
            pending.append(self.comm.all_reduce_async(model.arena.span("encoder.")))
            for h_ in pending:
                h_.wait()
        elif world > 1:
            self.comm.all_reduce(model.arena.grad)


class BetaHLoss(_SingleOptimizerLoss):
    """losses.py:117-153."""
    KIND = _lib.LOSS_BETAH

    def __init__(self, beta=4, **kwargs):
        super().__init__(**kwargs)
        self.beta = beta

    def _coefs(self, is_train):
        anneal = linear_annealing(0, 1, self.n_train_steps, self.steps_anneal) if is_train else 1
        return dict(ANNEAL=anneal, BETA=self.beta)

    def _store(self, storer, vals, D):
        storer['recon_loss'].append(vals[_lib.S_REC])
        self._store_kl(storer, vals, D)
        storer['loss'].append(vals[_lib.S_LOSS])

    def __call__(self, data, recon_data, latent_dist, is_train, storer, **kwargs):
        storer = self._pre_call(is_train, storer)
        sc = self.scratch(recon_data.device)
        rec_loss = _reconstruction_loss(data, recon_data, storer=storer, distribution=self.rec_dist, scratch=sc)
        kl_loss = _kl_normal_loss(*latent_dist, storer, scratch=sc)
        anneal_reg = linear_annealing(0, 1, self.n_train_steps, self.steps_anneal) if is_train else 1
        loss = rec_loss + anneal_reg * (self.beta * kl_loss)
        if storer is not None:
            storer['loss'].append(loss.item())
        return loss


class BetaBLoss(_SingleOptimizerLoss):
    """losses.py:156-202."""
    KIND = _lib.LOSS_BETAB

    def __init__(self, C_init=0., C_fin=20., gamma=100., **kwargs):
        super().__init__(**kwargs)
        self.gamma = gamma
        self.C_init = C_init
        self.C_fin = C_fin

    def _capacity(self, is_train):
        return (linear_annealing(self.C_init, self.C_fin, self.n_train_steps, self.steps_anneal)
                if is_train else self.C_fin)

    def _coefs(self, is_train):
        return dict(ANNEAL=1.0, BETA=self.gamma, CAP=self._capacity(is_train))

    def _store(self, storer, vals, D):
        storer['recon_loss'].append(vals[_lib.S_REC])
        self._store_kl(storer, vals, D)
        storer['loss'].append(vals[_lib.S_LOSS])

    def __call__(self, data, recon_data, latent_dist, is_train, storer, **kwargs):
        storer = self._pre_call(is_train, storer)
        sc = self.scratch(recon_data.device)
        rec_loss = _reconstruction_loss(data, recon_data, storer=storer, distribution=self.rec_dist, scratch=sc)
        kl_loss = _kl_normal_loss(*latent_dist, storer, scratch=sc)
        C = self._capacity(is_train)
        loss = rec_loss + self.gamma * (kl_loss - C).abs()
        if storer is not None:
            storer['loss'].append(loss.item())
        return loss


class BtcvaeLoss(_SingleOptimizerLoss):
    """losses.py:316-391 (is_mss=True default, never overridden by get_loss_f)."""
    KIND = _lib.LOSS_BTCVAE

    def __init__(self, n_data, alpha=1., beta=6., gamma=1., is_mss=True, **kwargs):
        super().__init__(**kwargs)
        self.n_data = n_data
        self.beta = beta
        self.alpha = alpha
        self.gamma = gamma
        self.is_mss = is_mss

    def _coefs(self, is_train):
        anneal = linear_annealing(0, 1, self.n_train_steps, self.steps_anneal) if is_train else 1
        return dict(ANNEAL=anneal, ALPHA=self.alpha, BETA=self.beta, GAMMA=self.gamma)

    def _store(self, storer, vals, D):
        storer['recon_loss'].append(vals[_lib.S_REC])
        storer['loss'].append(vals[_lib.S_LOSS])
        storer['mi_loss'].append(vals[_lib.S_MI])
        storer['tc_loss'].append(vals[_lib.S_TC])
        storer['dw_kl_loss'].append(vals[_lib.S_DWKL])
        self._store_kl(storer, vals, D)

    def __call__(self, data, recon_batch, latent_dist, is_train, storer, latent_sample=None):
        storer = self._pre_call(is_train, storer)
        sc = self.scratch(recon_batch.device)
        rec_loss = _reconstruction_loss(data, recon_batch, storer=storer, distribution=self.rec_dist, scratch=sc)
        terms = _BtcvaeFn.apply(latent_sample, latent_dist[0], latent_dist[1], self.n_data, self.is_mss, sc)
        mi_loss, tc_loss, dw_kl_loss = terms[0], terms[1], terms[2]
        anneal_reg = linear_annealing(0, 1, self.n_train_steps, self.steps_anneal) if is_train else 1
        loss = rec_loss + (self.alpha * mi_loss + self.beta * tc_loss + anneal_reg * self.gamma * dw_kl_loss)
        if storer is not None:
            storer['loss'].append(loss.item())
            storer['mi_loss'].append(mi_loss.item())
            storer['tc_loss'].append(tc_loss.item())
            storer['dw_kl_loss'].append(dw_kl_loss.item())
            with torch.no_grad():
                _ = _kl_normal_loss(latent_dist[0].detach(), latent_dist[1].detach(), storer, scratch=sc)
        return loss


class FactorKLoss(BaseLoss):
    """losses.py:205-313.  ``call_optimize`` runs the whole two-optimizer iteration on the HIP
    kernels, including quirk Q1 (the encoder also receives d[0.5 CE(D(z1),0)]/dz1 because the
    reference does not detach d_z and steps the VAE optimizer after d_tc_loss.backward())."""

    # the side stream carries the discriminator's chain as well: from 2048 rows per step the VAE's weight gradients go to two
    # side streams (BaseLoss.THREE_STREAM_MIN_ROWS: tensor 2048 1.853 -> 1.832 ms, tensor 256 0.561 -> 0.592)
    THREE_STREAM_MIN_ROWS = 2048

    def __init__(self, device, gamma=10., disc_kwargs={}, optim_kwargs=dict(lr=5e-5, betas=(0.5, 0.9)), **kwargs):
        super().__init__(**kwargs)
        self.gamma = gamma
        self.device = device
        self.discriminator = Discriminator(**disc_kwargs).to(self.device)
        optim_kwargs = dict(optim_kwargs)
        if torch.device(self.device).type == "cuda":
            optim_kwargs.setdefault("fused", True)   # same Adam arithmetic, one multi-tensor kernel
        self.optimizer_d = torch.optim.Adam(self.discriminator.parameters(), **optim_kwargs)

    def __call__(self, *args, **kwargs):
        raise ValueError("Use `call_optimize` to also train the discriminator")  # losses.py:240-241

    _PERM_RING = 4

    def _draw_perms(self, D, n):
        """D independent torch.randperm(n) draws (losses.py:505, CPU generator, the reference's order) into the next slot of
        a small ring of pinned [D, n] int64 buffers -> (buffer, slot); slot[1] is the event that marks the slot's last
        host-to-device copy (waited for before the slot is overwritten: normally long past)."""
        ring = self.__dict__.setdefault("_perm_ring", {})
        ent = ring.get((D, n))
        if ent is None:
            ent = ring[(D, n)] = {"slots": [[torch.empty(D, n, dtype=torch.int64).pin_memory(), None]
                                            for _ in range(self._PERM_RING)], "next": 0}
        slot = ent["slots"][ent["next"]]
        ent["next"] = (ent["next"] + 1) % self._PERM_RING
        if slot[1] is not None:
            slot[1].synchronize()
        for d in range(D):
            torch.randperm(n, out=slot[0][d])
        return slot[0], slot

    def _device_step(self, data, model, sc, eps1, eps2, perms):
        """Training iteration of FactorVAE as one stream of launches (no host-dependent values):
        VAE forward on both halves, discriminator on (z1, z_perm), both backward passes."""
        eng = model.engine
        eng._fork_hook = None                  # a hook left behind by a step that raised is dropped, not run (nor recorded)
        disc = self.discriminator
        D = model.latent_dim
        B = data.size(0)
        Bh = B // 2
        world, rank = self._world()
        Bhg = Bh * world
        dev = data.device
        s = _stream()
        # the N(0,1) draws of both halves in ONE [2*Bh, D] buffer (rows < Bh: data1, losses.py:254; the rest:
        # sample_latent(data2), losses.py:286) -- two separate draws, like the reference, into its two halves
        eps12 = sc.latent("eps12", 2 * Bh, D)
        if eps1 is None:
            record_py(eps12[:Bh].normal_)
            record_py(eps12[Bh:].normal_)
        else:
            record_py(eps12[:Bh].copy_, eps1)
            record_py(eps12[Bh:].copy_, eps2)
        eps1 = eps12[:Bh]
        buf = eng.buffers(B)
        data = eng.input(data, buf)
        eng.encode_convs(data, buf, n=2 * Bh, chain=True)                         # data1 and data2 in one pass
        # FC core of both halves in one launch; KL only over data1 with the half batch as denominator (losses.py:255-259),
        # decoder only for data1
        eng.fc_chain_fwd(buf, eps12, sc.kl_dim, 2 * Bh, n_kl=Bh, n_dec=Bh, coef=sc.coef)
        klb = eng.kl_blocks(2 * Bh)
        eng.decode_convs(buf, Bh, fuse_loss=(data, self._rec_code(), sc.coef, sc.partials), chain=True)
        off = Bh
        # z_perm: permute across the (global) half batch, losses.py:287
        zin = sc.latent("disc_in", 2 * Bh, D)
        copy_flat_(zin[:Bh], buf.z[:Bh])
        z2 = buf.z[off:off + Bh]
        ew, er = self._est_world()                # scope of permute_dims: global half batch, or this shard ("local")
        if ew > 1:
            z2g = self.comm.all_gather_rows(z2)
        else:
            z2g = z2
        zperm_g = sc.latent("zperm_g", Bh * ew, D)
        call("dvae_permute_dims", ptr(z2g.contiguous()), ptr(perms), ptr(zperm_g), Bh * ew, D, s)
        copy_flat_(zin[Bh:], zperm_g[er * Bh:(er + 1) * Bh])
        logits = disc.forward_raw(zin, 2 * Bh)                        # D(z1) and D(z_perm) in one pass
        g_dtc = sc.latent("g_dtc", 2 * Bh, 2)
        g_tc = sc.latent("g_tc", Bh, 2)
        call("dvae_disc_losses", ptr(logits), Bh, ptr(sc.coef), ptr(sc.disc_sums), ptr(g_dtc), ptr(g_tc), s)
        if world > 1:
            # the CE / tc means run over the global half batch
            scale_(g_dtc, 1.0 / world)
            scale_(g_tc, 1.0 / world)

        # the scalar epilogue (13 us; sharded: KL finish + pack + all-reduce of the packed sums + finalize) is first needed by
        # the FC chain's input gradients, after the discriminator's and the decoder's backward passes: it runs on the side
        # stream, an event slot marks it (as in the btcvae step)
        def epilogue(on_side):
            stream = eng._side_raw() if on_side else s
            if world > 1 and on_side:             # sharded: the exchange stream (see the btcvae step), ordered through the side stream
                stream = eng._aux_raw()
                call("dvae_stream_order", eng._side_raw(), stream)
            if world == 1:
                call("dvae_loss_epilogue", _lib.LOSS_FACTOR, ptr(sc.partials), ptr(sc.kl_dim), klb, D, None, 0,
                     ptr(sc.disc_sums), Bhg, ptr(sc.coef), ptr(sc.packed), ptr(sc.scal), stream)
                return
            self._pack_sums(sc, klb, D, None, 0, sc.disc_sums, sc.packed, stream)
            with torch.cuda.stream(eng.aux_stream if on_side else torch.cuda.current_stream()):
                self.comm.all_reduce(sc.packed)
            call("dvae_loss_finalize", _lib.LOSS_FACTOR, ptr(sc.packed), D, Bhg, ptr(sc.coef), ptr(sc.scal), stream)
            if on_side:
                call("dvae_event_record", self._ev_slot, stream)
        late_epi = not eng.single_stream and knob("DVAE_LATE_JOIN", "1") != "0"
        if late_epi:                          # after the next fork (the backward pass's first): no fork of its own
            def deferred():
                epilogue(True)
                if world == 1:
                    call("dvae_event_record", self._ev_slot, eng._side_raw())
            eng.at_next_fork(deferred)
        else:
            epilogue(False)
        # discriminator backward of d_tc_loss (weight grads + dz), losses.py:303-304
        # (its six weight gradients: on the side stream, behind one fork after the input-gradient chain)
        side_wg = not eng.single_stream and knob("DVAE_DISC_WGRAD_SIDE", "1") != "0"
        # the second chain of input gradients (the tc term of vae_loss through D: first half, no weight gradients) depends on
        # nothing the first one computes and could run beside it on the engine's third stream.  Measured, NOT shipped (A/B under
        # DVAE_DEBUG=1, same box: factor_dsprites 0.583 vs 0.579-0.585 ms, factor_celeba 1.905 vs 1.885-1.892, tensor 512
        # 0.831 vs 0.788-0.819: profiles/r05_v35_disc_chain2_ab.txt): the fork and the join cost the critical path what the
        # overlap of two launch-bound chains buys, and at 2048 rows both chains fill the chip anyway.
        par2 = (not eng.single_stream and world == 1 and knob("DVAE_DISC_CHAIN2_AUX", "0") == "1")
        dz_b = None
        if par2:
            call("dvae_stream_order", s, eng._aux_raw())
            dz_b = disc.backward_raw(zin, g_tc, 2 * Bh, rows=Bh, wgrad=False, chain="g2", stream=eng._aux_raw(), ws="aux")
        dz_a = disc.backward_raw(zin, g_dtc, 2 * Bh, wgrad=True, chain="g", side=eng if side_wg else None)
        pending = []
        if world > 1:      # the 16 MB discriminator gradients are final: their all-reduce runs under the whole VAE backward
            with torch.cuda.stream(eng.side_stream if side_wg else torch.cuda.current_stream()):
                pending.append(self.comm.all_reduce_async(disc.arena.grad))
        # tc term of vae_loss through D: dgrad only, first half (its disc weight grads are zeroed at :303)
        if dz_b is None:
            dz_b = disc.backward_raw(zin, g_tc, 2 * Bh, rows=Bh, wgrad=False, chain="g2")

        def fc_chain():
            # dz_a: quirk Q1 (the encoder also receives d[0.5 CE(D(z1),0)]/dz1); dz_b: the tc term through D
            if par2:
                call("dvae_stream_order", eng._aux_raw(), s)
            if late_epi:
                eng.flush_fork_hook()
                call("dvae_event_wait", self._ev_slot, s)
            eng.fc_chain_bwd(buf, eps1, dz_a, dz_b, None, None, sc.scal, sc.coef, Bh)

        # one join, at the end of encode_backward; sharded: the decoder span's all-reduce is ordered behind the side stream
        # (every decoder gradient is written there) and overlaps the encoder backward
        spans = world > 1 and data.numel() > self.SMALL_SHARD_ELEMS      # small shards: ONE all-reduce of the whole VAE arena (see the btcvae step)
        eng.decode_backward(buf.z, buf, n=Bh, join=False, defer_fc_wgrad=not spans, fc_chain=fc_chain)
        if spans:
            with torch.cuda.stream(eng.side_stream):
                pending.append(self.comm.all_reduce_async(model.arena.span("decoder.")))
        eng.encode_backward(data, buf, n=Bh, fc_chain=True)
        if spans:
            pending.append(self.comm.all_reduce_async(model.arena.span("encoder.")))
        elif world > 1:
            self.comm.all_reduce(model.arena.grad)
        for h_ in pending:
            h_.wait()

    def call_optimize(self, data, model, optimizer, storer, noise=None):
        """noise: optional (eps1[Bh,D], eps2[Bh,D], perms int64[D,Bh]) injected for parity;
        by default eps are drawn on the device and the permutations with torch.randperm on the
        CPU generator, in the reference's order (losses.py:254,286,505)."""
        is_train = model.training
        storer = self._pre_call(is_train, storer)
        eng = model.engine
        disc = self.discriminator
        D = model.latent_dim
        B = data.size(0)
        Bh = B // 2
        world, rank = self._world()
        Bhg = Bh * world
        dev = data.device
        s = _stream()
        sc = self.scratch(dev).for_latent_dim(D)
        anneal = linear_annealing(0, 1, self.n_train_steps, self.steps_anneal) if is_train else 1
        sc.set_coef_host(INV_B=1.0 / Bhg, ANNEAL=anneal, BETA=self.gamma)
        eng.stage(sc.coef, sc.coef_host)       # ONE launch: this step's weight images + its loss coefficients
        data = data.contiguous()
        self._streams(model, data)
        if noise is not None:
            eps1, eps2, perms = noise
        else:
            eps1 = eps2 = perms = None       # training draws them on the device in _device_step
        if is_train:
            slot = None
            if perms is None:
                # CPU generator (shared seed across ranks), reference order losses.py:505 -- drawn straight into a pinned
                # staging buffer: the copy to the device is then truly asynchronous (from pageable memory it blocks the host
                # until every launch enqueued before it has run, i.e. the host could never run ahead of the GPU)
                perms, slot = self._draw_perms(D, Bh * self._est_world()[0])
            perms = perms.to(dtype=torch.int64)
            mode = self._replay_mode(True, data)
            if mode == "graph":
                data = self._static_buf("data", data)
            perms = self._static_buf("perms", perms)      # device copy (non-blocking from the pinned ring)
            if slot is not None:
                slot[1] = torch.cuda.Event()
                slot[1].record()                          # the staging buffer is free again once this has passed
            if mode:
                if noise is not None:
                    eps1, eps2 = self._static_buf("eps1", eps1), self._static_buf("eps2", eps2)
                self._graphs.run(self._replay_key(model, data, noise is not None) + (disc.arena.flat.data_ptr(),),
                                 lambda: self._device_step(data, model, sc, eps1, eps2, perms), mode)
            else:
                self._device_step(data, model, sc, eps1, eps2, perms)
        else:
            buf = eng.buffers(B)
            data = eng.input(data, buf)
            eng.encode_convs(data, buf, n=Bh, chain=True)
            # z = mean; KL over data1 with the half batch as denominator (losses.py:255-259)
            eng.fc_chain_fwd(buf, None, sc.kl_dim, Bh, coef=sc.coef)
            if eng.kl_blocks(Bh):
                call("dvae_kl_finish", ptr(sc.kl_dim), eng.kl_blocks(Bh), ptr(sc.coef), D, s)
            eng.decode_convs(buf, Bh, fuse_loss=(data, self._rec_code(), sc.coef, sc.partials), chain=True)
            # evaluation: vae_loss only (losses.py:276-278); discriminator on z1
            logits = disc.forward_raw(buf.z, Bh)
            g_dtc = sc.latent("g_dtc", 2 * Bh, 2)
            lg2 = sc.latent("lg2", 2 * Bh, 2)
            lg2[:Bh].copy_(logits[:Bh]); lg2[Bh:].copy_(logits[:Bh])
            call("dvae_disc_losses", ptr(lg2), Bh, ptr(sc.coef), ptr(sc.disc_sums), ptr(g_dtc), None, s)
            call("dvae_loss_pack", ptr(sc.partials), ptr(sc.kl_dim), D, None, 0, ptr(sc.disc_sums), ptr(sc.packed), s)
            if world > 1:
                self.comm.all_reduce(sc.packed)
            call("dvae_loss_finalize", _lib.LOSS_FACTOR, ptr(sc.packed), D, Bhg, ptr(sc.coef), ptr(sc.scal), s)
            if storer is not None:
                vals = sc.scal.tolist()
                storer['recon_loss'].append(vals[_lib.S_REC])
                self._store_kl(storer, vals, D)
                storer['loss'].append(vals[_lib.S_LOSS])
                storer['tc_loss'].append(vals[_lib.S_TC])
            return sc.scal[_lib.S_LOSS]
        model.assign_grads()
        disc.assign_grads()
        optim.step(optimizer)         # optimizer.step(), losses.py:307
        optim.step(self.optimizer_d)  # losses.py:308
        if storer is not None:
            vals = sc.scal.tolist()
            storer['recon_loss'].append(vals[_lib.S_REC])
            self._store_kl(storer, vals, D)
            storer['loss'].append(vals[_lib.S_LOSS])
            storer['tc_loss'].append(vals[_lib.S_TC])
            storer['discrim_loss'].append(vals[_lib.S_DTC])
        return sc.scal[_lib.S_LOSS]
