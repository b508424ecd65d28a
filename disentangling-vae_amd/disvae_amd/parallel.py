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
                     the 128-byte RCCL unique id.  The default on GPU process groups (``transport="auto"``, verified
                     with a round trip); ``data_parallel(..., transport="torch")`` / DVAE_COMM=torch selects the other.
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


_COMM_STREAMS = {}


def _comm_stream(dev):
    """The communication stream of a device, one per process (see engine.device_streams: streams are a scarce resource)."""
    key = (dev.type, dev.index)
    st = _COMM_STREAMS.get(key)
    if st is None:
        st = _COMM_STREAMS[key] = _lib.new_stream(dev)
    return st


def _raw_stream():
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


# Element-wise glue around the collectives.  Device tensors go through the C-ABI (dvae_axpby / dvae_swap_outer): one foreign
# call that lands in a recorded launch plan's C segment -- a torch op costs the host 10-20 us per replay (stream guard +
# dispatch), which at 128 images per GPU is what the step is bound by.  CPU tensors (the gloo tests) use torch.
def scale_(t, alpha):
    """t *= alpha."""
    if t.is_cuda:
        call("dvae_axpby", ptr(t), ptr(t), float(alpha), None, 0.0, t.numel(), _raw_stream())
    else:
        record_on_stream(t.mul_, float(alpha))


def add_scaled_(out, b, beta):
    """out += beta * b (same number of elements, both contiguous)."""
    if out.is_cuda:
        call("dvae_axpby", ptr(out), ptr(out), 1.0, ptr(b), float(beta), out.numel(), _raw_stream())
    else:
        record_on_stream(lambda: out.view(-1).add_(b.reshape(-1), alpha=float(beta)))


def copy_flat_(out, src):
    """out <- src (same number of elements, both contiguous)."""
    if out.is_cuda:
        call("dvae_axpby", ptr(out), ptr(src), 1.0, None, 0.0, out.numel(), _raw_stream())
    else:
        record_on_stream(lambda: out.view(-1).copy_(src.reshape(-1)))


def swap_outer_(dst, src, A, Bn, inner):
    """dst [Bn][A][inner] <- src [A][Bn][inner] (contiguous buffers)."""
    if dst.is_cuda:
        call("dvae_swap_outer", ptr(src), ptr(dst), A, Bn, inner, _raw_stream())
    else:
        record_on_stream(lambda: dst.view(Bn, A, inner).copy_(src.reshape(A, Bn, inner).permute(1, 0, 2)))


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
        swap_outer_(glob, recv, self.world_size, 3, B * D)
        return glob[0], glob[1], glob[2]

    def reduce_scatter_cols(self, dmu_all, dlv_all):
        """Column gradients [B_global, D] summed over ranks -> this rank's rows [B, D] (rows of rank r are contiguous):
        ONE collective on the pair packed as [world, 2, B, D]."""
        W = self.world_size
        B = dmu_all.shape[0] // W
        D = dmu_all.shape[1]
        send = self._buf("cols_send", (W, 2, B, D), dmu_all)
        if (dmu_all.is_contiguous() and dlv_all.is_contiguous()
                and dlv_all.data_ptr() == dmu_all.data_ptr() + dmu_all.numel() * dmu_all.element_size()
                and dmu_all.untyped_storage().data_ptr() == dlv_all.untyped_storage().data_ptr()):
            swap_outer_(send, dmu_all, 2, W, B * D)     # the pair are consecutive slabs of one buffer = [2][W][B*D]
        else:
            record_on_stream(lambda: torch.stack((dmu_all.reshape(W, B, D), dlv_all.reshape(W, B, D)), dim=1, out=send))
        out = self._buf("cols_loc", (2, B, D), dmu_all)
        self.reduce_scatter_into(out, send)
        return out[0], out[1]

    def all_reduce_cols_sums(self, xbuf, B, D, n_tail):
        """ONE sum-all-reduce for the two exchanges that end the beta-TCVAE estimator of a sharded step: xbuf =
        [dmu of ALL world*B columns | dlogvar of all columns | n_tail packed loss sums] -- every rank then reads its own rows
        of the two slabs (a reduce-scatter would move 1/world of the bytes, but at 80 KB both are one latency, and this is
        one collective and no packing pass instead of two and one)."""
        return self.all_reduce(xbuf)

    def close(self):
        pass


class RcclComm(Comm):
    """Same interface, data path through the C-ABI (``dvae_comm_*`` of libdvae_hip.so): RCCL collectives enqueued on the
    current HIP stream.  The torch.distributed group is used once, to broadcast rank 0's RCCL unique id."""

    @staticmethod
    def prepare(rank):
        """The part of the set-up that involves NO other rank: load RCCL into libdvae_hip.so and, on rank 0, draw the unique
        id.  Everything that can fail for local reasons (a missing library, a missing symbol) fails here, BEFORE the first
        collective a peer could be left waiting in (_auto_comm agrees on the outcome across ranks in between)."""
        _lib.lib()
        path = os.environ.get("DVAE_RCCL_LIB")
        if path is None:                      # prefer the RCCL torch itself is linked with (one RCCL per process)
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            path = cand if os.path.exists(cand) else None
        call("dvae_comm_load", path.encode() if path else None)
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (ctypes.c_char * 128)()
            call("dvae_comm_unique_id", ctypes.addressof(buf))
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        return uid

    def __init__(self, group=None, prepared=None):
        super().__init__(group)
        h = _lib.lib()
        uid = prepared if prepared is not None else self.prepare(self.rank)
        dev = torch.device("cuda", torch.cuda.current_device())
        if self.world_size > 1:
            backend = dist.get_backend(group)
            t = uid.to(dev) if backend == "nccl" else uid
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = t.cpu()
        self._uid = bytes(uid.numpy().tobytes())
        handle = ctypes.c_void_p()
        call("dvae_comm_init", ctypes.addressof(handle), self._uid, self.world_size, self.rank)
        self._h = handle.value
        self._side = _comm_stream(dev)
        assert h.dvae_comm_world(self._h) == self.world_size

    @staticmethod
    def _s():
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())

    def all_reduce(self, t):
        call("dvae_comm_allreduce", self._h, ptr(t), t.numel(), self._s())
        return t

    def all_reduce_async(self, t):
        """The collective goes to a communication stream forked from the current one; wait() joins it back."""
        side = self._side.cuda_stream
        call("dvae_stream_order", self._s(), side)
        call("dvae_comm_allreduce", self._h, ptr(t), t.numel(), side)
        s_ = self._s

        class _Handle:
            def wait(self_inner):
                call("dvae_stream_order", side, s_())

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

    def self_test(self):
        """One round trip through the communicator: a sum-all-reduce of ones must come back as world_size."""
        t = torch.ones(64, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
        self.all_reduce(t)
        torch.cuda.synchronize()
        if not bool((t == float(self.world_size)).all()):
            raise _lib.DvaeHipError("dvae_comm self test: all-reduce of ones returned %r" % t[:4].tolist())

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
    times, sum-all-reduce -> `world_size` x the local term, reduce-scatter -> the local chunk + (`world_size` - 1) x a
    peer-bound chunk.  The loss plugins then run the whole sharded
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
        scale_(t, self.world_size)
        return t

    def all_reduce_async(self, t):
        h = self.inner.all_reduce_async(t)
        W = self.world_size

        class _Handle:
            def wait(self_inner):
                h.wait()
                scale_(t, W)

        return _Handle()

    def all_gather_into(self, out, t):
        rows = out.view(self.world_size, -1)
        self.inner.all_gather_into(rows[0], t)
        k = 1
        while k < self.world_size:               # replicate by doubling: log2(world) copies
            m = min(k, self.world_size - k)
            copy_flat_(rows[k:k + m], rows[0:m])
            k += m

    def reduce_scatter_into(self, out, t):
        # this rank's chunk of its OWN send buffer carries terms no peer holds (the row-role gradients of its rows); every
        # peer's contribution to the chunk equals what this rank sends to a peer's chunk
        chunks = t.view(self.world_size, -1)
        self.inner.reduce_scatter_into(out, chunks[self.rank])
        if self.world_size > 1:
            add_scaled_(out, chunks[(self.rank + 1) % self.world_size], self.world_size - 1)

    def all_reduce_cols_sums(self, xbuf, B, D, n_tail):
        # column slabs: this rank's rows = own contribution (with the row-role terms) + (world - 1) x what it sends to a
        # peer's rows; loss sums: world x the local ones.  The other ranks' rows of the slabs are not read.
        W, r = self.world_size, self.rank
        self.inner.all_reduce(xbuf)
        n = B * D
        if W > 1:
            o = (r + 1) % W
            for slab in (0, 1):
                base = slab * W * n
                add_scaled_(xbuf[base + r * n:base + (r + 1) * n], xbuf[base + o * n:base + (o + 1) * n], W - 1)
        scale_(xbuf[2 * W * n:2 * W * n + n_tail], W)
        return xbuf

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


def _auto_comm(group):
    """RcclComm where it demonstrably works on every rank, else Comm (see data_parallel)."""
    if not torch.cuda.is_available() or dist.get_backend(group) != "nccl":
        return Comm(group)
    dev = torch.device("cuda", torch.cuda.current_device())

    def all_ranks_ok(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    comm, err, uid = None, None, None
    # stage 1, local: RCCL loaded into the C-ABI (+ rank 0's unique id).  The ranks agree on it BEFORE the unique-id broadcast:
    # a rank that failed here never enters that broadcast, and its peers must not be left blocked in it
    try:
        uid = RcclComm.prepare(dist.get_rank(group))
    except Exception as e:       # noqa: a missing librccl, a missing symbol
        err = e
    ok = all_ranks_ok(err is None)
    if ok:
        # stage 2, collective on every rank: unique-id broadcast, ncclCommInitRank, one all-reduce round trip
        try:
            comm = RcclComm(group, prepared=uid)
            comm.self_test()
        except Exception as e:   # noqa: a failed ncclCommInitRank, a wrong sum
            err = e
        ok = all_ranks_ok(err is None)
    if ok:
        return comm
    import warnings
    warnings.warn("disvae_amd.parallel: the C-ABI RCCL transport failed its round trip on at least one rank (%s): every rank "
                  "falls back to torch.distributed collectives" % (err if err is not None else "another rank"), RuntimeWarning)
    if comm is not None:
        try:
            comm.close()
        except Exception:   # noqa
            pass
    return Comm(group)


def data_parallel(model, loss_f, group=None, estimator="global", transport=None, comm=None):
    """Attach a communicator to a native loss plugin (and make every rank start from rank 0's
    weights).  After this, ``loss_f.fused_step`` / ``call_optimize`` treat their input as this
    rank's shard of a global batch of world_size x B images.

    estimator: scope of the batch-coupled terms (the beta-TCVAE B x B estimator and its minibatch
    weights, FactorVAE's permute_dims).  "global" (default): over the global batch -- equal to the
    single-process step on the concatenated batch, at the price of a latent all-gather, a column-gradient
    reduce-scatter and B x (world B) estimator work per rank.  "local": over each rank's shard -- what the
    reference computes under DistributedDataParallel (gradient all-reduce only); a different estimator.

    transport: "rccl" (the C-ABI's dvae_comm_*: every collective is an entry of the recorded launch plan, no Python
    between the launches of a step), "torch" (torch.distributed collectives, nccl == RCCL) or "auto" (default; DVAE_COMM
    overrides): "rccl" where the process group runs on GPUs, after a verified round trip on EVERY rank -- one rank's failure
    sends all of them to "torch" with a warning; "torch" on CPU groups (gloo).  comm: a ready communicator object (tests)."""
    if estimator not in ("global", "local"):
        raise ValueError("estimator must be 'global' or 'local'")
    if comm is None:
        transport = transport or os.environ.get("DVAE_COMM", "auto")
        if transport not in ("torch", "rccl", "auto"):
            raise ValueError("transport must be 'auto', 'torch' or 'rccl', got %r" % (transport,))
        if transport == "auto":
            comm = _auto_comm(group)
        else:
            comm = RcclComm(group) if transport == "rccl" else Comm(group)
    loss_f.comm = comm
    loss_f.estimator = estimator
    comm.broadcast(model.arena.flat)
    disc = getattr(loss_f, "discriminator", None)
    if disc is not None:
        comm.broadcast(disc.arena.flat)
    return comm
