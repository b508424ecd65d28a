"""time conv2's weight gradient (k_wgrad32ws<16> + its fixed-order reduction, B = 1024) alone; DVAE_WGWS_ABLATE (debug builds) selects timing ablations"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    sys.path.insert(0, p)
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr, NHWC
B = 1024
big, small = torch.rand(B, 32, 32, 32, device="cuda"), torch.rand(B, 16, 16, 32, device="cuda")
ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device="cuda")
s = torch.cuda.current_stream().cuda_stream
dw, db = torch.empty(32, 32, 4, 4, device="cuda"), torch.empty(32, device="cuda")
f = lambda: call("dvae_conv4s2_wgrad", ptr(big), NHWC, ptr(small), NHWC, ptr(dw), ptr(db), B, 32, 32, 32, 32, ptr(ws), s)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("k_wgrad32ws<16> + reduce, DVAE_WGWS_ABLATE=%s : %.1f us" % (os.environ.get("DVAE_WGWS_ABLATE", "0"), e0.elapsed_time(e1) / 50 * 1e3))
