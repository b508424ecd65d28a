"""Data parallelism over the GPUs of one node: one process per GPU.

The reference is single-process; this is new.  What is exchanged per step:
  * every loss      : sum-all-reduce of the flat gradient arena in two spans (decoder half as soon as the decoder's
                      gradients are final, encoder half at the end; 2.0 MB together) and one of the 32-float packed
                      loss sums;
  * btcvae          : ONE all-gather of the packed (z, mu, logvar) so every rank evaluates its ROW block of the global
                      B x B estimator (reference parity at the global batch), and ONE reduce-scatter of the packed
                      [B_global, D] column gradients (dmu, dlogvar);
  * factor          : all-gather of the second half-batch latents (permute_dims permutes across the GLOBAL half batch;
                      every rank applies the same, shared-seed, CPU-generated permutations and keeps its slice); the
                      16 MB discriminator gradient arena is all-reduced as soon as the discriminator's backward pass has
                      produced it, under the whole VAE backward.
Encoder / decoder / reconstruction / KL are independent per image: no exchange.

Two transports with the same interface:
  * ``Comm``      -- torch.distributed collectives (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests);
  * ``RcclComm``  -- the collectives of libdvae_hip.so's C-ABI (``dvae_comm_*``: RCCL enqueued on the caller's HIP
                     stream, no torch types in the data path); torch.distributed is then only the rendezvous that ships
                     the 128-byte RCCL unique id.  Select with ``data_parallel(..., transport="rccl")`` / DVAE_COMM=rccl.
Gather / scatter buffers are allocated once per shape and reused.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import call, ptr, record_on_stream


class _Done:
    def wait(self):
        return None


class Comm:
    """Collectives over a torch.distributed process group (plumbing, not compute)."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._bufs = {}

    # ---- primitives --------------------------------------------------------------------------
    # Every collective (and every torch op around one) is issued through _lib.record_on_stream / record_py: a recorded
    # launch plan (disvae_amd/graph.py) then contains them and replays them in place, so the plan stays usable when the
    # batch is sharded.
    def all_reduce(self, t):
        record_on_stream(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
        return t

    def all_reduce_async(self, t):
        """Start a sum-all-reduce of `t` (ordered after the work already enqueued on the current stream) and return a
        handle; ``handle.wait()`` orders the current stream after it.  Used to overlap the decoder half of the gradient
        arena (and the discriminator arena) with the rest of the backward pass."""
        group = self.group

        class _Pending:
            work = None

            def start(self_inner):
                self_inner.work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)

            def finish(self_inner):
                self_inner.work.wait()

            def wait(self_inner):
                record_on_stream(self_inner.finish)

        h = _Pending()
        record_on_stream(h.start)
        return h

    def all_gather_into(self, out, t):
        """out[world * n] <- concatenation of every rank's t[n] in rank order."""
        record_on_stream(lambda: dist.all_gather_into_tensor(out.view(-1), t.reshape(-1), group=self.group))

    def reduce_scatter_into(self, out, t):
        """out[n] <- this rank's chunk of the element-wise sum over ranks of t[world * n]."""
        record_on_stream(lambda: dist.reduce_scatter_tensor(out.view(-1), t.reshape(-1), op=dist.ReduceOp.SUM, group=self.group))

    def broadcast(self, t, src=0):
        dist.broadcast(t, src=src, group=self.group)
        return t

    def group_start(self):
        pass

    def group_end(self):
        pass

    # ---- buffers ------------------------------------------------------------------------------
    def _buf(self, name, shape, like):
        key = (name, tuple(shape), like.dtype, like.device)
        b = self._bufs.get(key)
        if b is None:
            _lib.note_alloc()
            b = self._bufs[key] = torch.empty(shape, dtype=like.dtype, device=like.device)
        return b

    # ---- what the loss plugins call -----------------------------------------------------------------
    def all_gather_rows(self, t, name="rows"):
        """t[B, D] on every rank -> [world*B, D] in rank order (persistent buffer)."""
        t = t.contiguous()
        out = self._buf(name, (self.world_size * t.shape[0],) + tuple(t.shape[1:]), t)
        self.all_gather_into(out, t)
        return out

    def all_gather_latents(self, z, mu, logvar):
        """(z, mu, logvar) local [B, D] -> global [world*B, D] each, in rank order: ONE collective.  The engine keeps the
        three tensors as consecutive slabs of one [3, B, D] buffer, which is sent as it is (anything else is packed first);
        one copy kernel re-orders the received [world, 3, B, D] into three contiguous [world*B, D] tensors."""
        B, D = z.shape
        n = B * D * z.element_size()
        if (z.is_contiguous() and mu.is_contiguous() and logvar.is_contiguous()
                and mu.data_ptr() == z.data_ptr() + n and logvar.data_ptr() == z.data_ptr() + 2 * n
                and z.untyped_storage().data_ptr() == logvar.untyped_storage().data_ptr()):
            send = z.as_strided((3, B, D), (B * D, D, 1))
        else:
            send = self._buf("lat_send", (3, B, D), z)
            record_on_stream(lambda: torch.stack((z, mu, logvar), out=send))
        recv = self._buf("lat_recv", (self.world_size, 3, B, D), z)
        self.all_gather_into(recv, send)
        glob = self._buf("lat_glob", (3, self.world_size * B, D), z)
        record_on_stream(glob.view(3, self.world_size, B, D).copy_, recv.permute(1, 0, 2, 3))
        return glob[0], glob[1], glob[2]

    def reduce_scatter_cols(self, dmu_all, dlv_all):
        """Column gradients [B_global, D] summed over ranks -> this rank's rows [B, D] (rows of rank r are contiguous):
        ONE collective on the pair packed as [world, 2, B, D]."""
        W = self.world_size
        B = dmu_all.shape[0] // W
        D = dmu_all.shape[1]
        send = self._buf("cols_send", (W, 2, B, D), dmu_all)
        record_on_stream(lambda: torch.stack((dmu_all.reshape(W, B, D), dlv_all.reshape(W, B, D)), dim=1, out=send))
        out = self._buf("cols_loc", (2, B, D), dmu_all)
        self.reduce_scatter_into(out, send)
        return out[0], out[1]

    def close(self):
        pass


class RcclComm(Comm):
    """Same interface, data path through the C-ABI (``dvae_comm_*`` of libdvae_hip.so): RCCL collectives enqueued on the
    current HIP stream.  The torch.distributed group is used once, to broadcast rank 0's RCCL unique id."""

    def __init__(self, group=None):
        super().__init__(group)
        h = _lib.lib()
        path = os.environ.get("DVAE_RCCL_LIB")
        if path is None:                      # prefer the RCCL torch itself is linked with (one RCCL per process)
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            path = cand if os.path.exists(cand) else None
        call("dvae_comm_load", path.encode() if path else None)
        dev = torch.device("cuda", torch.cuda.current_device())
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_char * 128)()
            call("dvae_comm_unique_id", ctypes.addressof(buf))
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world_size > 1:
            backend = dist.get_backend(group)
            t = uid.to(dev) if backend == "nccl" else uid
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = t.cpu()
        self._uid = bytes(uid.numpy().tobytes())
        handle = ctypes.c_void_p()
        call("dvae_comm_init", ctypes.addressof(handle), self._uid, self.world_size, self.rank)
        self._h = handle.value
        self._side = torch.cuda.Stream(device=dev)
        assert h.dvae_comm_world(self._h) == self.world_size

    @staticmethod
    def _s():
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())

    def all_reduce(self, t):
        call("dvae_comm_allreduce", self._h, ptr(t), t.numel(), self._s())
        return t

    def all_reduce_async(self, t):
        """The collective goes to a communication stream forked from the current one; wait() joins it back."""
        cur = torch.cuda.current_stream()
        _lib.record_py(self._side.wait_stream, cur)
        call("dvae_comm_allreduce", self._h, ptr(t), t.numel(), self._side.cuda_stream)
        side = self._side

        class _Handle:
            def wait(self_inner):
                _lib.record_py(torch.cuda.current_stream().wait_stream, side)

        return _Handle()

    def all_gather_into(self, out, t):
        call("dvae_comm_allgather", self._h, ptr(t), ptr(out), t.numel(), self._s())

    def reduce_scatter_into(self, out, t):
        call("dvae_comm_reducescatter", self._h, ptr(t), ptr(out), out.numel(), self._s())

    def broadcast(self, t, src=0):
        call("dvae_comm_broadcast", self._h, ptr(t), t.numel(), src, self._s())
        return t

    def group_start(self):
        call("dvae_comm_group_start")

    def group_end(self):
        call("dvae_comm_group_end")

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize()
            call("dvae_comm_destroy", self._h)
            self._h = None


class MirroredWorldComm(Comm):
    """Measurement and test aid, NOT a transport: one process stands in for rank `rank` of `world_size` ranks whose peers
    hold IDENTICAL shards (same images, same noise).  Every collective first runs through the wrapped one-rank communicator
    `inner` (``Comm`` or ``RcclComm``: the real call sites, RCCL launches and stream ordering of the data-parallel step) and is
    then completed with what the absent peers would have contributed: all-gather -> the shard replicated `world_size`
    times, sum-all-reduce / reduce-scatter -> `world_size` x the local term.  The loss plugins then run the whole sharded
    code path at the shard's real sizes (B local rows of a world_size * B column estimator, packed exchanges, two gradient
    spans) on ONE GPU: ``bench.py``'s shard legs time it, ``tests/test_gpu_ddp.py`` holds it to the single-process step on
    the shard tiled `world_size` times (exact for every term that is symmetric in the ranks: everything except the
    minibatch-stratified weights' one exception cell, math.py:72, and FactorVAE's global permutation)."""

    def __init__(self, inner, world_size, rank=0):
        if inner.world_size != 1:
            raise ValueError("MirroredWorldComm wraps a one-rank communicator")
        if not (0 <= int(rank) < int(world_size)):
            raise ValueError("rank %r outside world_size %r" % (rank, world_size))
        self.inner = inner
        self.group = inner.group
        self.world_size = int(world_size)
        self.rank = int(rank)
        self._bufs = {}

    def all_reduce(self, t):
        self.inner.all_reduce(t)
        record_on_stream(t.mul_, float(self.world_size))
        return t

    def all_reduce_async(self, t):
        h = self.inner.all_reduce_async(t)
        W = float(self.world_size)

        class _Handle:
            def wait(self_inner):
                h.wait()
                record_on_stream(t.mul_, W)

        return _Handle()

    def all_gather_into(self, out, t):
        rows = out.view(self.world_size, -1)
        self.inner.all_gather_into(rows[0], t)
        if self.world_size > 1:
            record_on_stream(rows[1:].copy_, rows[0:1].expand(self.world_size - 1, -1))

    def reduce_scatter_into(self, out, t):
        self.inner.reduce_scatter_into(out, t.view(self.world_size, -1)[self.rank])
        record_on_stream(out.mul_, float(self.world_size))

    def broadcast(self, t, src=0):
        return self.inner.broadcast(t, src=0)

    def group_start(self):
        self.inner.group_start()

    def group_end(self):
        self.inner.group_end()

    def close(self):
        self.inner.close()


def init_process_group_from_env(backend=None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun-style)."""
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group(backend=backend)


def data_parallel(model, loss_f, group=None, estimator="global", transport=None, comm=None):
    """Attach a communicator to a native loss plugin (and make every rank start from rank 0's
    weights).  After this, ``loss_f.fused_step`` / ``call_optimize`` treat their input as this
    rank's shard of a global batch of world_size x B images.

    estimator: scope of the batch-coupled terms (the beta-TCVAE B x B estimator and its minibatch
    weights, FactorVAE's permute_dims).  "global" (default): over the global batch -- equal to the
    single-process step on the concatenated batch, at the price of a latent all-gather, a column-gradient
    reduce-scatter and B x (world B) estimator work per rank.  "local": over each rank's shard -- what the
    reference computes under DistributedDataParallel (gradient all-reduce only); a different estimator.

    transport: "torch" (default; torch.distributed collectives, nccl == RCCL) or "rccl" (the C-ABI's dvae_comm_*);
    DVAE_COMM overrides the default.  comm: a ready communicator object (tests)."""
    if estimator not in ("global", "local"):
        raise ValueError("estimator must be 'global' or 'local'")
    if comm is None:
        transport = transport or os.environ.get("DVAE_COMM", "torch")
        if transport not in ("torch", "rccl"):
            raise ValueError("transport must be 'torch' or 'rccl', got %r" % (transport,))
        comm = RcclComm(group) if transport == "rccl" else Comm(group)
    loss_f.comm = comm
    loss_f.estimator = estimator
    comm.broadcast(model.arena.flat)
    disc = getattr(loss_f, "discriminator", None)
    if disc is not None:
        comm.broadcast(disc.arena.flat)
    return comm
