"""The thin-end kernels alone (conv1 forward / convT3 input gradient / the two weight gradients / convT3 forward + likelihood) at
one or more batch sizes, HIP-event timed through the C-ABI: us per launch and the algorithmic TB/s.
    python tools/thin_ab.py [B ...] [--c1]        (default 1024 256 128; --c1: one-channel images)
Debug builds: DVAE_THIN_WS=0 selects k_down_thin instead of k_down_thin_ws (A/B in separate processes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "disentangling-vae_amd")]
import ctypes
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr, NCHW, NHWC

C = 1 if "--c1" in sys.argv else 3
Bs = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [1024, 256, 128]
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=dev)
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("DVAE_THIN")) or "default"


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


for B in Bs:
    x = torch.rand(B, C, 64, 64, device=dev)
    a1 = torch.rand(B, 32, 32, 32, device=dev) - 0.5
    w = torch.rand(32, C, 4, 4, device=dev) - 0.5
    b = torch.zeros(32, device=dev)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * 1024,), dtype=torch.int32, device=dev)
    dw, db = torch.empty(32, C, 4, 4, device=dev), torch.empty(32, device=dev)
    dbc = torch.empty(C, device=dev)
    nx, na, nb = x.numel() * 4, a1.numel() * 4, bits.numel() * 4
    # convT3 forward + likelihood on the staged pair records (what the step launches)
    pairs = torch.empty(32 * _lib.thin_pair_floats(C), device=dev)
    td = _lib.ThinImageDesc()
    td.w, td.img_pairs, td.C = ptr(w), ptr(pairs), C
    call("dvae_stage_weights", None, 0, None, 0, ctypes.addressof(td), None, None, s)
    rec, g = torch.empty_like(x), torch.empty_like(x)
    coef = torch.full((8,), 1.0 / B, device=dev)
    parts = torch.empty(_lib.REC_NPART, device=dev)
    bc = torch.zeros(C, device=dev)
    rows = [
        ("convT3 fwd + likelihood", lambda: call("dvae_convT3_fwd_staged", ptr(a1), ptr(pairs), ptr(bc), ptr(x), 0, ptr(rec), ptr(g), 0, ptr(coef), ptr(parts), B, C, s), na + 3 * nx),
        ("conv1 fwd", lambda: call("dvae_conv4s2_fwd", ptr(x), NCHW, ptr(w), ptr(b), ptr(a1), NHWC, B, C, 64, 64, 32, 1, s), nx + na),
        ("conv1 fwd + bits", lambda: call("dvae_conv1_fwd_bits", ptr(x), 0, ptr(w), ptr(b), ptr(a1), ptr(bits), B, C, s), nx + na + nb),
        ("convT3 dgrad (bit mask)", lambda: call("dvae_convT3_dgrad_bits", ptr(x), ptr(w), ptr(bits), ptr(a1), B, C, s), nx + na + nb),
        ("conv1 wgrad (+reduce)", lambda: call("dvae_conv4s2_wgrad", ptr(x), NCHW, ptr(a1), NHWC, ptr(dw), ptr(db), B, C, 64, 64, 32, ptr(ws), s), nx + na),
        ("convT3 wgrad (+reduce)", lambda: call("dvae_convT4s2_wgrad", ptr(a1), NHWC, ptr(x), NCHW, ptr(dw), ptr(dbc), B, 32, 32, 32, C, ptr(ws), s), nx + na),
    ]
    for name, fn, nbytes in rows:
        us = timeit(fn)
        print("[%s] B=%-5d C=%d %-26s %7.1f us  %5.2f TB/s  %.3f of 8 TB/s" % (tag, B, C, name, us, nbytes / us / 1e6, nbytes / us / 8e6))
