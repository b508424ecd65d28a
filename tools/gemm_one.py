"""Launch dvae_linear_fwd / _dgrad / _wgrad (M x K x N, default the discriminator's 2048 x 1000 x 1000) a few times: target
for rocprofv3 --pmc / --kernel-trace passes on the large-GEMM kernels alone.
    python tools/gemm_one.py [M K N [reps [fwd|dgrad|wgrad]]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "disentangling-vae_amd")]
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr
M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2048, 1000, 1000)))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
form = sys.argv[5] if len(sys.argv) > 5 else "fwd"
x = torch.rand(M, K, device="cuda") - 0.5
w = torch.rand(N, K, device="cuda") - 0.5
b = torch.zeros(N, device="cuda")
y = torch.empty(M, N, device="cuda")
ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device="cuda")
s = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dy = torch.rand(M, N, device="cuda") - 0.5
dx = torch.empty(M, K, device="cuda")
dw, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
launch = {"fwd": lambda: call("dvae_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(y), M, K, N, 2, ptr(ws), s),
          "dgrad": lambda: call("dvae_linear_dgrad", ptr(dy), ptr(w), ptr(x), 2, ptr(dx), M, K, N, ptr(ws), s),
          "wgrad": lambda: call("dvae_linear_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(db), M, K, N, ptr(ws), s)}[form]
for _ in range(3):
    launch()
e0.record()
for _ in range(reps):
    launch()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print("linear %s %dx%dx%d: %.1f us/launch, %.1f TFLOP/s" % (form, M, K, N, us, 2.0 * M * K * N / us / 1e6))
