"""Does a low-register kernel on a second stream run 'for free' beside the persistent 32-channel kernels (whose 2 waves per SIMD
hold 430-500 of the 512 VGPRs)?  A = 20 x k_down32dma<16> (or k_up32ws<16>) at 1024 images on stream 1; B = a VALU-bound
elementwise chain (torch: ~20-40 VGPRs) sized to about the same time on stream 2; prints A alone, B alone, A || B."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    sys.path.insert(0, p)
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr, NHWC

dev = "cuda"
B = 1024
f = lambda *sh: torch.rand(*sh, device=dev)
w = f(32, 32, 4, 4) - 0.5
b32 = torch.zeros(32, device=dev)
imd, imu = torch.empty(16384, device=dev), torch.empty(16384, device=dev)
cd = (_lib.ConvImageDesc * 1)()
cd[0].w, cd[0].img_down, cd[0].img_up = ptr(w), ptr(imd), ptr(imu)
s0 = torch.cuda.current_stream().cuda_stream
call("dvae_stage_weights", ctypes.addressof(cd), 1, None, 0, None, None, None, s0)
big, small = f(B, 32, 32, 32), f(B, 16, 16, 32)
obig, osmall = torch.empty_like(big), torch.empty_like(small)
sA, sB = _lib.new_stream(torch.device(dev)), _lib.new_stream(torch.device(dev))
kernels = {"down16": lambda: call("dvae_conv32_down", ptr(big), ptr(imd), ptr(b32), None, ptr(osmall), NHWC, B, 16, _lib.ACT_RELU, sA.cuda_stream),
           "up16": lambda: call("dvae_conv32_up", ptr(small), NHWC, ptr(imu), ptr(b32), None, ptr(obig), B, 16, _lib.ACT_RELU, sA.cuda_stream)}
x = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, device=dev)


def valu():
    with torch.cuda.stream(sB):
        y = x
        for _ in range(6):
            y = torch.sin(y) * 1.0001 + 0.1
        return y


def timed(fa, fb, n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
    for _ in range(n):
        if fa: fa()
        if fb: fb()
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, fa in kernels.items():
    for _ in range(3):
        timed(fa, valu)
    a, b, ab = timed(fa, None), timed(None, valu), timed(fa, valu)
    print("%-7s alone %.1f us | VALU chain alone %.1f us | together %.1f us (sum %.1f, max %.1f)" % (name, a, b, ab, a + b, max(a, b)), flush=True)
