"""Per-kernel micro-benchmark through the C-ABI (HIP events on torch's current stream).
usage: python tools/kbench.py [B]   -> one line per kernel: us per launch, TFLOP/s (algorithmic), GB/s (algorithmic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    sys.path.insert(0, p)
import ctypes
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr, NCHW, NHWC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def report(name, us, flops, bytes_):
    print("%-34s %8.1f us  %7.1f TFLOP/s  %7.0f GB/s" % (name, us, flops / us / 1e6, bytes_ / us / 1e3))


for H in (32, 16, 8):        # big side H x H x 32 <-> small side H/2
    hs = H // 2
    big = torch.rand(B, H, H, 32, device=dev)
    small = torch.rand(B, hs, hs, 32, device=dev)
    w = torch.rand(32, 32, 4, 4, device=dev) - 0.5
    b = torch.zeros(32, device=dev)
    dw, db = torch.empty(32, 32, 4, 4, device=dev), torch.empty(32, device=dev)
    macs = B * hs * hs * 32 * 512
    nb, ns = big.numel() * 4, small.numel() * 4
    report("conv fwd   %dx%d->%dx%d" % (H, H, hs, hs), timeit(lambda: call("dvae_conv4s2_fwd", ptr(big), NHWC, ptr(w), ptr(b), ptr(small), NHWC, B, 32, H, H, 32, 1, s)), 2 * macs, nb + ns)
    report("convT dgrad (down+mask)", timeit(lambda: call("dvae_convT4s2_dgrad", ptr(big), NHWC, ptr(w), ptr(small), ptr(small), NHWC, B, 32, hs, hs, 32, s)), 2 * macs, nb + 2 * ns)
    report("convT fwd  %dx%d->%dx%d" % (hs, hs, H, H), timeit(lambda: call("dvae_convT4s2_fwd", ptr(small), NHWC, ptr(w), ptr(b), ptr(big), NHWC, B, 32, hs, hs, 32, 1, s)), 2 * macs, nb + ns)
    report("conv dgrad (up+mask)", timeit(lambda: call("dvae_conv4s2_dgrad", ptr(small), NHWC, ptr(w), ptr(big), ptr(big), NHWC, B, 32, H, H, 32, s)), 2 * macs, 2 * nb + ns)
    report("conv wgrad (+reduce)", timeit(lambda: call("dvae_conv4s2_wgrad", ptr(big), NHWC, ptr(small), NHWC, ptr(dw), ptr(db), B, 32, H, H, 32, ptr(ws), s)), 2 * macs, nb + ns)
    # the same kernels on pre-staged weight images (dvae_stage_weights): what the training step launches
    imd, imu = torch.empty(16384, device=dev), torch.empty(16384, device=dev)
    cd = (_lib.ConvImageDesc * 1)()
    cd[0].w, cd[0].img_down, cd[0].img_up = ptr(w), ptr(imd), ptr(imu)
    call("dvae_stage_weights", ctypes.addressof(cd), 1, None, 0, None, None, None, s)
    report("  staged: conv fwd (down)", timeit(lambda: call("dvae_conv32_down", ptr(big), ptr(imd), ptr(b), None, ptr(small), NHWC, B, hs, 1, s)), 2 * macs, nb + ns)
    report("  staged: convT dgrad (down+mask)", timeit(lambda: call("dvae_conv32_down", ptr(big), ptr(imd), None, ptr(small), ptr(small), NHWC, B, hs, 0, s)), 2 * macs, nb + 2 * ns)
    report("  staged: convT fwd (up)", timeit(lambda: call("dvae_conv32_up", ptr(small), NHWC, ptr(imu), ptr(b), None, ptr(big), B, hs, 1, s)), 2 * macs, nb + ns)
    report("  staged: conv dgrad (up+mask)", timeit(lambda: call("dvae_conv32_up", ptr(small), NHWC, ptr(imu), None, ptr(big), ptr(big), B, hs, 0, s)), 2 * macs, 2 * nb + ns)
# ReLU masks as bit planes (dvae_*_bits): the 16 -> 32 "up" kernel and the thin "down" kernel, emitting / consuming one uint32 per pixel
small = torch.rand(B, 16, 16, 32, device=dev); big = torch.rand(B, 32, 32, 32, device=dev) - 0.5
w = torch.rand(32, 32, 4, 4, device=dev) - 0.5; b = torch.zeros(32, device=dev)
imd, imu = torch.empty(16384, device=dev), torch.empty(16384, device=dev)
cd = (_lib.ConvImageDesc * 1)()
cd[0].w, cd[0].img_down, cd[0].img_up = ptr(w), ptr(imd), ptr(imu)
call("dvae_stage_weights", ctypes.addressof(cd), 1, None, 0, None, None, None, s)
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * 1024,), dtype=torch.int32, device=dev)
macs = B * 256 * 32 * 512
report("  bits: convT2 fwd (up, emits bits)", timeit(lambda: call("dvae_conv32_up_bits", ptr(small), ptr(imu), ptr(b), None, ptr(big), ptr(bits), B, 1, s)), 2 * macs, big.numel() * 4 + small.numel() * 4 + bits.numel() * 4)
report("  bits: conv2 dgrad (up, bit mask)", timeit(lambda: call("dvae_conv32_up_bits", ptr(small), ptr(imu), None, ptr(bits), ptr(big), None, B, 0, s)), 2 * macs, big.numel() * 4 + small.numel() * 4 + bits.numel() * 4)
for C in (3,):
    x = torch.rand(B, C, 64, 64, device=dev)
    a1 = torch.rand(B, 32, 32, 32, device=dev)
    w = torch.rand(32, C, 4, 4, device=dev) - 0.5
    b = torch.zeros(32, device=dev)
    bc = torch.zeros(C, device=dev)
    dw, db = torch.empty(32, C, 4, 4, device=dev), torch.empty(32, device=dev)
    macs = B * 1024 * 32 * 16 * C
    nx, na = x.numel() * 4, a1.numel() * 4
    report("conv1 fwd (down_thin)", timeit(lambda: call("dvae_conv4s2_fwd", ptr(x), NCHW, ptr(w), ptr(b), ptr(a1), NHWC, B, C, 64, 64, 32, 1, s)), 2 * macs, nx + na)
    report("convT3 dgrad (down_thin+mask)", timeit(lambda: call("dvae_convT4s2_dgrad", ptr(x), NCHW, ptr(w), ptr(a1), ptr(a1), NHWC, B, 32, 32, 32, C, s)), 2 * macs, nx + 2 * na)
    report("convT3 fwd (up_thin+sigmoid)", timeit(lambda: call("dvae_convT4s2_fwd", ptr(a1), NHWC, ptr(w), ptr(bc), ptr(x), NCHW, B, 32, 32, 32, C, 3, s)), 2 * macs, nx + na)
    report("conv1 wgrad (wgrad_thin+reduce)", timeit(lambda: call("dvae_conv4s2_wgrad", ptr(x), NCHW, ptr(a1), NHWC, ptr(dw), ptr(db), B, C, 64, 64, 32, ptr(ws), s)), 2 * macs, nx + na)
    report("  bits: conv1 fwd (down_thin, emits bits)", timeit(lambda: call("dvae_conv1_fwd_bits", ptr(x), 0, ptr(w), ptr(b), ptr(a1), ptr(bits), B, C, s)), 2 * macs, nx + na + bits.numel() * 4)
    report("  bits: convT3 dgrad (down_thin, bit mask)", timeit(lambda: call("dvae_convT3_dgrad_bits", ptr(x), ptr(w), ptr(bits), ptr(a1), B, C, s)), 2 * macs, nx + na + bits.numel() * 4)
    g = torch.empty_like(x)
    coef = torch.full((32,), 1.0 / B, device=dev)
    parts = torch.empty(2048, device=dev)      # DVAE_REC_NPART
    tgt, rec = torch.rand_like(x), torch.empty_like(x)
    report("convT3 fwd + likelihood (up_thin fused)", timeit(lambda: call("dvae_convT4s2_sigmoid_recon_fwd", ptr(a1), NHWC, ptr(w), ptr(bc), ptr(tgt), ptr(rec), ptr(g), 0, ptr(coef), ptr(parts), B, 32, 32, 32, C, s)), 2 * macs, 3 * nx + na)
    pairs = torch.empty(32 * _lib.thin_pair_floats(C), device=dev)
    td = _lib.ThinImageDesc()
    td.w, td.img_pairs, td.C = ptr(w), ptr(pairs), C
    call("dvae_stage_weights", None, 0, None, 0, ctypes.addressof(td), None, None, s)
    report("  staged: convT3 fwd (up_thin_pk+sigmoid)", timeit(lambda: call("dvae_convT3_fwd_staged", ptr(a1), ptr(pairs), ptr(bc), None, 0, ptr(rec), None, 0, None, None, B, C, s)), 2 * macs, nx + na)
    report("  staged: convT3 fwd + likelihood (up_thin_pk fused)", timeit(lambda: call("dvae_convT3_fwd_staged", ptr(a1), ptr(pairs), ptr(bc), ptr(tgt), 0, ptr(rec), ptr(g), 0, ptr(coef), ptr(parts), B, C, s)), 2 * macs, 3 * nx + na)
    t8 = (tgt * 255).to(torch.uint8)
    report("  staged: convT3 fwd + likelihood, uint8 target", timeit(lambda: call("dvae_convT3_fwd_staged", ptr(a1), ptr(pairs), ptr(bc), ptr(t8), 1, ptr(rec), ptr(g), 0, ptr(coef), ptr(parts), B, C, s)), 2 * macs, 2.25 * nx + na)
    dbc = torch.empty(C, device=dev)
    report("convT3 wgrad (wgrad_thin+reduce)", timeit(lambda: call("dvae_convT4s2_wgrad", ptr(a1), NHWC, ptr(g), NCHW, ptr(dw), ptr(dbc), B, 32, 32, 32, C, ptr(ws), s)), 2 * macs, nx + na)
    report("recon_loss (bernoulli)", timeit(lambda: call("dvae_recon_loss", ptr(x), ptr(x), x.numel(), 0, ptr(coef), ptr(parts), ptr(g), 1, s)), 0, 3 * nx)
for (M, K, N) in ((B, 512, 256), (B, 256, 256), (B, 256, 20), (B, 10, 256), (B, 256, 512), (B, 1000, 1000)):
    x = torch.rand(M, K, device=dev); w = torch.rand(N, K, device=dev); b = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dwt = torch.empty(N, K, device=dev)
    fl = 2.0 * M * K * N
    report("linear fwd   %dx%dx%d" % (M, K, N), timeit(lambda: call("dvae_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(y), M, K, N, 1, ptr(ws), s)), fl, 4 * (M * K + N * K + M * N))
    report("linear dgrad", timeit(lambda: call("dvae_linear_dgrad", ptr(y), ptr(w), ptr(x), 1, ptr(dx), M, K, N, ptr(ws), s)), fl, 4 * (M * K + N * K + M * N))
    report("linear wgrad", timeit(lambda: call("dvae_linear_wgrad", ptr(x), ptr(y), ptr(dwt), ptr(b), M, K, N, ptr(ws), s)), fl, 4 * (M * K + N * K + M * N))
# ---- the FC core as one launch per direction + the per-step staging launch
from disvae_amd.models.vae import init_specific_model
model = init_specific_model("Burgess", (3, 64, 64), 10).to(dev)
eng = model.engine
buf = eng.buffers(B)
buf.a_flat.uniform_(0, 1)
coefd = torch.full((8,), 1.0 / B, device=dev)
scal = torch.zeros(32, device=dev)
eps = torch.randn(B, 10, device=dev)
kl = torch.zeros(_lib.KL_FLOATS, device=dev)
report("stage_weights (6 conv + 6 fc + coef)", timeit(lambda: eng.stage(coefd, [1.0 / B] * 8)), 0, 0)
fl = 2.0 * B * 400896
report("fc_chain_fwd  (7 launches before)", timeit(lambda: eng.fc_chain_fwd(buf, eps, kl, B)), fl, 1.6e6)
report("fc_chain_bwd  (7 launches before)", timeit(lambda: eng.fc_chain_bwd(buf, eps, None, None, None, None, scal, coefd, B)), fl, 1.6e6)
D = 10
z = torch.randn(B, D, device=dev); mu = torch.randn(B, D, device=dev); lv = torch.randn(B, D, device=dev) * 0.5
lw = torch.tensor([-12.0, -7.0, -6.9, 0.0], device=dev); rs = torch.empty(B, _lib.ROWSTATS, device=dev); tmp = torch.empty(3 * D, B, device=dev)
coef = torch.tensor([1.0 / B, 0.5, 6.4, 1.0, 1.0, 0, 0, 0], device=dev)
dz, dm, dl = (torch.empty(B, D, device=dev) for _ in range(3))
report("btcvae fwd B=%d" % B, timeit(lambda: call("dvae_btcvae_fwd", ptr(z), ptr(mu), ptr(lv), B, D, 0, B, 1, ptr(lw), ptr(tmp), ptr(rs), s)), 30.0 * B * B * D, 0)
report("btcvae bwd (rows+cols)", timeit(lambda: call("dvae_btcvae_bwd", ptr(z), ptr(mu), ptr(lv), ptr(rs), B, D, 0, B, 1, ptr(lw), ptr(coef), ptr(tmp), ptr(dz), ptr(dm), ptr(dl), s)), 60.0 * B * B * D, 0)

# the estimator as ONE RANK OF 8 sees it (weak scaling, global estimator): its B rows against 8B columns
Bg = 8 * B
zg, mug, lvg = (torch.randn(Bg, D, device=dev) for _ in range(3))
tmpg = torch.empty(3 * D, Bg, device=dev)
dmg, dlg = torch.empty(Bg, D, device=dev), torch.empty(Bg, D, device=dev)
report("btcvae fwd rows=%d cols=%d (rank 3 of 8)" % (B, Bg), timeit(lambda: call("dvae_btcvae_fwd", ptr(zg), ptr(mug), ptr(lvg), Bg, D, 3 * B, B, 1, ptr(lw), ptr(tmpg), ptr(rs), s)), 30.0 * B * Bg * D, 0)
report("btcvae bwd rows=%d cols=%d" % (B, Bg), timeit(lambda: call("dvae_btcvae_bwd", ptr(zg), ptr(mug), ptr(lvg), ptr(rs), Bg, D, 3 * B, B, 1, ptr(lw), ptr(coef), ptr(tmpg), ptr(dz), ptr(dmg), ptr(dlg), s)), 60.0 * B * Bg * D, 0)
