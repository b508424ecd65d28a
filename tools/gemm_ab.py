"""The discriminator-sized GEMMs alone: dvae_linear_fwd / _dgrad / _wgrad at M x 1000 x 1000 (discriminator.py:51-56 and its
backward), checked against fp64 and timed with HIP events; prints us per launch and the fraction of the fp32 MFMA peak.
    python tools/gemm_ab.py [M ...]            (default 2048 1024 256 128)
Debug builds: DVAE_GEMM_DMA=0 selects the round-1 kernels (k_gemm_big / k_fcw32) for an A/B in separate processes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "disentangling-vae_amd")]
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr

PEAK = 157.3
NARROW = "--narrow" in sys.argv          # the discriminator's narrow layers instead: M x 1000 -> 2 and M x 10 -> 1000
Ms = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [2048, 1024, 256, 128]
SHAPES = [(1000, 2), (10, 1000)] if NARROW else [(1000, 1000)]
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=dev)
tag = "DVAE_GEMM_DMA=%s" % os.environ.get("DVAE_GEMM_DMA", "default")
if NARROW:
    tag = "DVAE_NARROW=%s" % os.environ.get("DVAE_NARROW", "default")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def err(got, ref):
    ref = ref.to(got.device)
    return float(((got.double() - ref).abs().max() / ref.abs().max()))


for M, (K, N) in [(m_, kn) for m_ in Ms for kn in SHAPES]:
    g = torch.Generator().manual_seed(M)
    x = (torch.rand(M, K, generator=g) - 0.5).to(dev)
    w = ((torch.rand(N, K, generator=g) - 0.5) * 0.1).to(dev)
    b = (torch.rand(N, generator=g) - 0.5).to(dev)
    dy = (torch.rand(M, N, generator=g) - 0.5).to(dev)
    y, dx, dw, db = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    fwd = lambda: call("dvae_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(y), M, K, N, 2, ptr(ws), s)
    dgr = lambda: call("dvae_linear_dgrad", ptr(dy), ptr(w), ptr(x), 2, ptr(dx), M, K, N, ptr(ws), s)
    wgr = lambda: call("dvae_linear_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(db), M, K, N, ptr(ws), s)
    fwd(); dgr(); wgr()
    torch.cuda.synchronize()
    xd, wd, dyd = x.double(), w.double(), dy.double()
    e_f = err(y, torch.nn.functional.leaky_relu(xd @ wd.t() + b.double(), 0.2))
    e_d = err(dx, (dyd @ wd) * torch.where(xd > 0, 1.0, 0.2))
    e_w = err(dw, dyd.t() @ xd)
    e_b = err(db, dyd.sum(0))
    fl = 2.0 * M * K * N
    for name, fn, e in (("fwd", fwd, e_f), ("dgrad", dgr, e_d), ("wgrad", wgr, max(e_w, e_b))):
        us = timeit(fn)
        print("%s M=%-5d %4dx%-4d %-5s %7.1f us  %6.1f TFLOP/s  %.3f of peak   max rel err %.1e %s" % (
            tag, M, K, N, name, us, fl / us / 1e6, fl / us / 1e6 / PEAK, e, "OK" if e < 2e-6 else "MISMATCH"))
