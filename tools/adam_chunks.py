"""torch's fused Adam over the flat parameter arena (504 056 floats) split into equal chunks: GPU time per step (HIP events)
and host time per optimizer.step() for several chunk sizes (models/vae.py FLAT_CHUNK), next to Adam over the 28 state_dict
shapes (fused and foreach).  python tools/adam_chunks.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "disentangling-vae_amd"))
import torch  # noqa: E402
from disvae_amd.engine import vae_param_shapes  # noqa: E402


def bench(params, n=200, **kw):
    opt = torch.optim.Adam(params, lr=5e-4, **kw)
    for p in params:
        p.grad = torch.randn_like(p)
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        opt.step()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, host


def main():
    dev = "cuda"
    n = 504056
    for chunk in (4096, 8192, 14336, 16384, 32768, 65536, n):
        flat = torch.randn(n, device=dev)
        params = [torch.nn.Parameter(c) for c in torch.split(flat, chunk)]
        g, h = bench(params, fused=True)
        print("flat chunks of %6d: %3d tensors  gpu %.1f us/step (back-to-back)  host issue %.1f us/step" % (chunk, len(params), g, h))
    shapes = vae_param_shapes((3, 64, 64), 10)
    for kw in (dict(fused=True), dict(foreach=True)):
        params = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes.values()]
        g, h = bench(params, **kw)
        print("28 state_dict tensors %s: gpu %.1f us/step  host issue %.1f us/step" % (kw, g, h))


if __name__ == "__main__":
    main()
