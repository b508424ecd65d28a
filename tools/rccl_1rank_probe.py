"""Does a ONE-rank RCCL collective block the host until the stream reaches it?  (The mirrored-world shard legs of bench.py
issue the data-parallel step's collectives through a one-rank communicator: if these calls synchronise, the emulation
overstates the cost of the C-ABI transport -- a real N-rank collective is an asynchronous kernel launch.)
Enqueues ~3 ms of GPU work on a stream, then times the host side of each collective issued behind it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "disentangling-vae_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from disvae_amd import parallel  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
parallel.init_process_group_from_env("nccl")
dev = torch.device("cuda", 0)
big = torch.randn(8192, 8192, device=dev)
x = torch.randn(30720, device=dev)
y = torch.empty(30720, device=dev)
arena = torch.randn(504056, device=dev)


def busy():
    for _ in range(3):
        torch.mm(big, big)


def probe(name, fn):
    fn()
    torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-46s host %8.1f us   (GPU drained %8.1f us later)" % (name, (t1 - t0) * 1e6, (t2 - t1) * 1e6))


for transport in ("rccl", "torch"):
    comm = parallel.RcclComm() if transport == "rccl" else parallel.Comm()
    probe(transport + " all_reduce in place (2 MB)", lambda: comm.all_reduce(arena))
    probe(transport + " all_gather_into (120 KB)", lambda: comm.all_gather_into(y, x))
    probe(transport + " reduce_scatter_into (120 KB)", lambda: comm.reduce_scatter_into(y, x))
    h = [None]

    def asy():
        h[0] = comm.all_reduce_async(arena)
        h[0].wait()
    probe(transport + " all_reduce_async + wait", asy)
    comm.close()
dist.destroy_process_group()
