"""time one conv2-forward-shaped launch (down kernel, 32x32x32 -> 16x16x32, B=1024); KONE_DATA=zero|rand"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    sys.path.insert(0, p)
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr, NHWC
B = 1024
mode = os.environ.get("KONE_DATA", "rand")
if mode == "zero":
    x = torch.zeros(B, 32, 32, 32, device="cuda"); w = torch.zeros(32, 32, 4, 4, device="cuda")
else:
    x = torch.rand(B, 32, 32, 32, device="cuda"); w = torch.rand(32, 32, 4, 4, device="cuda") - 0.5
b = torch.zeros(32, device="cuda"); y = torch.empty(B, 16, 16, 32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
f = lambda: call("dvae_conv4s2_fwd", ptr(x), NHWC, ptr(w), ptr(b), ptr(y), NHWC, B, 32, 32, 32, 32, 1, s)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("data=%s DVAE_ABLATE=%s V2=%s V1=%s : %.1f us" % (mode, os.environ.get("DVAE_ABLATE", "0"), os.environ.get("DVAE_DOWN_V2"), os.environ.get("DVAE_DOWN_V1"), e0.elapsed_time(e1) / 50 * 1e3))
