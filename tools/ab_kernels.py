"""A/B timing of single kernels through the C-ABI (select the library with DVAE_HIP_LIB): python tools/ab_kernels.py [B] [which,..]
which: wg16 (conv2 wgrad + reduce), wg8 (conv3 wgrad + reduce), utm (convT3 fused forward, fp32 targets), up16 / up16m (convT2 fwd emitting bits /
conv2 dgrad masked by bits), down16 / down16m.  Long warm-up (clocks), 200 timed launches, three repeats."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    sys.path.insert(0, p)
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr, NHWC, NCHW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["wg16", "wg8", "utm"]
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=dev)
f = lambda *sh: torch.rand(*sh, device=dev)
fns = {}
big, small = f(B, 32, 32, 32), f(B, 16, 16, 32)
dw, db = torch.empty(32, 32, 4, 4, device=dev), torch.empty(32, device=dev)
fns["wg16"] = lambda: call("dvae_conv4s2_wgrad", ptr(big), NHWC, ptr(small), NHWC, ptr(dw), ptr(db), B, 32, 32, 32, 32, ptr(ws), s)
big8, small8 = f(B, 16, 16, 32), f(B, 8, 8, 32)
fns["wg8"] = lambda: call("dvae_conv4s2_wgrad", ptr(big8), NHWC, ptr(small8), NHWC, ptr(dw), ptr(db), B, 32, 16, 16, 32, ptr(ws), s)
x, a1 = f(B, 3, 64, 64), f(B, 32, 32, 32)
rec, g = torch.empty_like(x), torch.empty_like(x)
wt, bc = f(32, 3, 4, 4) - 0.5, torch.zeros(3, device=dev)
coef = torch.full((8,), 1.0 / B, device=dev)
parts = torch.empty(_lib.REC_NPART, device=dev)
pairs = torch.empty(32 * _lib.thin_pair_floats(3), device=dev)
td = _lib.ThinImageDesc()
td.w, td.img_pairs, td.C = ptr(wt), ptr(pairs), 3
call("dvae_stage_weights", None, 0, None, 0, ctypes.addressof(td), None, None, s)
fns["utm"] = lambda: call("dvae_convT3_fwd_staged", ptr(a1), ptr(pairs), ptr(bc), ptr(x), 0, ptr(rec), ptr(g), 0, ptr(coef), ptr(parts), B, 3, s)
w = f(32, 32, 4, 4) - 0.5
b32 = torch.zeros(32, device=dev)
imd, imu = torch.empty(16384, device=dev), torch.empty(16384, device=dev)
cd = (_lib.ConvImageDesc * 1)()
cd[0].w, cd[0].img_down, cd[0].img_up = ptr(w), ptr(imd), ptr(imu)
call("dvae_stage_weights", ctypes.addressof(cd), 1, None, 0, None, None, None, s)
obig, osmall = torch.empty_like(big), torch.empty_like(small)
bits = torch.empty(B * 1024, dtype=torch.int32, device=dev)
fns["up16"] = lambda: call("dvae_conv32_up_bits", ptr(small), ptr(imu), ptr(b32), None, ptr(obig), ptr(bits), B, _lib.ACT_RELU, s)
fns["up16m"] = lambda: call("dvae_conv32_up_bits", ptr(small), ptr(imu), None, ptr(bits), ptr(obig), None, B, _lib.ACT_NONE, s)
fns["down16"] = lambda: call("dvae_conv32_down", ptr(big), ptr(imd), ptr(b32), None, ptr(osmall), NHWC, B, 16, _lib.ACT_RELU, s)
fns["down16m"] = lambda: call("dvae_conv32_down", ptr(big), ptr(imd), None, ptr(small), ptr(osmall), NHWC, B, 16, _lib.ACT_NONE, s)
fns["up16"]()
for name in which:
    fn = fns[name]
    for _ in range(150):
        fn()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 200 * 1e3)
    print("%-8s B=%d lib=%s : %s us" % (name, B, os.path.basename(os.environ.get("DVAE_HIP_LIB", "libdvae_hip.so")), " ".join("%.1f" % r for r in res)), flush=True)
