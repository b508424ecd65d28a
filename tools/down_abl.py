"""Timing ablations of k_down32dma (debug build; results of ablated launches are invalid): where the time above the bare
MFMA issue goes.  Usage: python tools/down_abl.py [B]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "disentangling-vae_amd"))
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr, NHWC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


NAMES = {1: "no stores", 2: "no mask loads", 4: "no DMA", 8: "no LDS reads", 16: "no barriers", 32: "no MFMA"}
for hs in (16, 8):
    H = 2 * hs
    big = torch.rand(B, H, H, 32, device=dev)
    small = torch.rand(B, hs, hs, 32, device=dev)
    w = torch.rand(32, 32, 4, 4, device=dev) - 0.5
    b = torch.zeros(32, device=dev)
    imd, imu = torch.empty(16384, device=dev), torch.empty(16384, device=dev)
    cd = (_lib.ConvImageDesc * 1)()
    cd[0].w, cd[0].img_down, cd[0].img_up = ptr(w), ptr(imd), ptr(imu)
    call("dvae_stage_weights", ctypes.addressof(cd), 1, None, 0, None, None, None, s)
    flops = 2.0 * B * hs * hs * 32 * 512
    for masked in (0, 1):
        for abl in (0, 1, 2, 4, 8, 16, 32, 1 | 2, 4 | 16, 8 | 4 | 16, 1 | 2 | 4 | 8 | 16, 32 | 8, 32 | 8 | 1 | 2):
            if not masked and abl & 2 and abl != (1 | 2 | 4 | 8 | 16) and abl != (32 | 8 | 1 | 2) and abl != 3:
                continue
            os.environ["DVAE_DMA_ABLATE"] = str(abl)
            fn = (lambda: call("dvae_conv32_down", ptr(big), ptr(imd), None, ptr(small), ptr(small), NHWC, B, hs, 0, s)) if masked else \
                 (lambda: call("dvae_conv32_down", ptr(big), ptr(imd), ptr(b), None, ptr(small), NHWC, B, hs, 1, s))
            us = timeit(fn)
            what = " + ".join(NAMES[k] for k in NAMES if abl & k) or "complete"
            print("hs=%2d %s abl=%2d %-50s %7.1f us  %6.1f TFLOP/s" % (hs, "masked" if masked else "plain ", abl, what, us, flops / us / 1e6))
os.environ["DVAE_DMA_ABLATE"] = "0"
