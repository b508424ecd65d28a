"""k_up32ws<16> alone (convT2 forward = plain, conv2 dgrad = masked) at B images: us per launch.  With a debug build,
DVAE_UPWS_ABLATE=<bits> removes parts of the kernel (timing only, results invalid): 1 output stores, 2 mask loads,
4 input-tile loads, 8 MFMAs, 16 epilogue LDS writes, 32 the whole drain."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from disvae_amd import _lib  # noqa: E402
from disvae_amd._lib import call, ptr, NHWC  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
big, small = torch.rand(B, 32, 32, 32, device=dev), torch.rand(B, 16, 16, 32, device=dev)
out = torch.empty_like(big)
w = torch.rand(32, 32, 4, 4, device=dev) - 0.5
b = torch.zeros(32, device=dev)


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


t_plain = timeit(lambda: call("dvae_convT4s2_fwd", ptr(small), NHWC, ptr(w), ptr(b), ptr(out), NHWC, B, 32, 16, 16, 32, 1, s))
t_mask = timeit(lambda: call("dvae_conv4s2_dgrad", ptr(small), NHWC, ptr(w), ptr(big), ptr(out), NHWC, B, 32, 32, 32, 32, s))
print("DVAE_UPWS_ABLATE=%-3s plain %.1f us   masked %.1f us" % (os.environ.get("DVAE_UPWS_ABLATE", "0"), t_plain, t_mask))
