"""Benchmark of the hot path: Trainer._train_iteration-equivalent steps (forward + loss +
backward + Adam [+ RCCL collectives]) of the native HIP engine on synthetic batches resident in HBM.

    python bench.py [--config NAME] --gpus N --steps K --warmup W           (N > 1 without a launcher: the script starts
                                                                             its N ranks itself, self_launch below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads = BASELINE.json configs[1..4] (hyper-parameters: /root/reference/hyperparam.ini:6,76-81,123-137,
main.py:190-193; dataset sizes: utils/datasets.py:148,223):

    --config btcvae_celeba   (default; the config BASELINE.json's metric is quoted on)
             64x64x3, btcvae (alpha 1, beta 6.4, gamma 1), B = 1024, n_data 202599, Adam lr 5e-4
    --config factor_celeba   64x64x3, factor (gamma 6.4), tensor 2048 = 1024 + 1024, lr 1e-4, lr_disc 1e-5
    --config btcvae_dsprites 64x64x1, btcvae, B = 256, n_data 737280, lr 5e-4
    --config factor_dsprites 64x64x1, factor, tensor 256 = 128 + 128, lr 1e-4, lr_disc 1e-4

B always denotes the tensor handed to `_train_iteration`.  With N > 1 GPUs the default is
`--scaling strong`: the config's batch IS the global batch (configs[3]: "b=1024 ... DDP over 8xMI355X"), every
rank gets B/N images and the batch-coupled estimators (B x B log-density matrix, permute_dims) run over the
GLOBAL batch; `--scaling weak` keeps B images per GPU (global batch B x N).

One JSON line on rank 0.  `value` = images/s of the whole job over EXACTLY --steps iterations bracketed by
barrier + synchronize on both sides (max over ranks).  Also reported: HIP-event timing of the same iterations on the
compute stream in 5 segments (`hip_event_ms_per_step`: segments + median), `roofline` = algorithmic FLOPs of the
dominant kernel family / its HIP-event-timed duration vs the 157.3 TFLOP/s fp32 MFMA peak, `roofline_kernels` = the other
two 32-channel conv families (bound "mfma") and the five thin (C = 1 / 3) conv launches of the step (bound "hbm":
algorithmic bytes / HIP-event time vs 8 TB/s), `parity_check` = the first iteration of this very workload (same weights,
batch, injected noise, the SAME optimizer construction as the timed loop) against the oracle, `drop_in` = the same
workload driven exactly as INTEGRATION.md tells a reference user to (optim.Adam(model.parameters()), one loss.item() per
iteration as Trainer does under is_progress_bar=True), `configs` = the other three single-GPU BASELINE workloads as short
legs (value, ms_per_step, fraction of the fp32 peak, parity, CPU baseline), `cpu_baseline` = the reference's CPU path
timed on this box's host cores on a bounded sample: kind "reference" = the unmodified /root/reference Trainer (only where
that directory exists: the build container), kind "port" = the oracle's restatement of it (the GPU boxes).
`python bench.py --cpu-reference` (no GPU needed) times both CPU legs side by side.
"""
import argparse
import glob
import json
import ctypes
import os
import re
import sys
import time
from collections import defaultdict

# one hardware queue per stream of the step (the caller's, the weight-gradient stream, the exchange and communication streams of
# a sharded step, RCCL's own): with HIP's default of 4, streams alias onto shared queues and serialise (measured at 128 images
# per GPU: 0.372 -> 0.494 ms once a fifth stream exists, profiles/r05_v14_hw_queues.txt).  Must be set before HIP initialises;
# disvae_amd/__init__.py sets the same default for library users
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25, betaB_G=1000,
          factor_G=6.4, latent_dim=10, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)
CONFIGS = {
    "vae_mnist": dict(loss="VAE", img=(1, 32, 32), batch=64, n_data=60000, lr=5e-4, lr_disc=1e-4, baseline_config=0),
    "btcvae_celeba": dict(loss="btcvae", img=(3, 64, 64), batch=1024, n_data=202599, lr=5e-4, lr_disc=1e-5, baseline_config=3),
    "factor_celeba": dict(loss="factor", img=(3, 64, 64), batch=2048, n_data=202599, lr=1e-4, lr_disc=1e-5, baseline_config=4),
    "btcvae_dsprites": dict(loss="btcvae", img=(1, 64, 64), batch=256, n_data=737280, lr=5e-4, lr_disc=1e-4, baseline_config=1),
    "factor_dsprites": dict(loss="factor", img=(1, 64, 64), batch=256, n_data=737280, lr=1e-4, lr_disc=1e-4, baseline_config=2),
}
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4 at 2.4 GHz
PEAK_HBM_GBS = 8000.0
N_SEGMENTS = 5


def flops_per_image_train(C, H=64):
    """6 * MACs_fwd - 2 * MACs_conv1 (64x64xC; 32x32xC: the stack without conv_64 / convT_64)."""
    conv = [524288 * C, 4194304, 1048576, 262144] if H == 64 else [131072 * C, 1048576, 262144]
    fc = 131072 + 65536 + 5120 + 2560 + 65536 + 131072
    macs = 2 * sum(conv) + fc
    return 6 * macs - 2 * conv[0]


def flops_per_image_factor(C):
    """SURVEY.md 8d, per image of the full tensor B: 0.5 train + 0.5 encoder forward + 28.1 MFLOP discriminator."""
    conv = [524288 * C, 4194304, 1048576, 262144]
    enc_fwd = 2 * (sum(conv) + 131072 + 65536 + 5120)
    disc = 2 * 8.024e6 + 2 * 16.05e6 + 8.024e6
    return 0.5 * flops_per_image_train(C) + 0.5 * enc_fwd + 0.5 * disc


# ---------------------------------------------------------------------------------- roofline
def pmc_file_order(f):
    """Sort key of a profiles/ file name: round first, then the visit tag -- "finalN" visits follow every "runN" / "vN" visit of
    their round (r02_run6 < r02_final < r03_run1; r04_v35 < r04_final < r04_final3), then the tag's own number."""
    b = os.path.basename(f)
    m = re.match(r"r(\d+)_(final|run|v)(\d*)_", b)
    if not m:
        nums = [int(x) for x in re.findall(r"\d+", b)]
        return [nums[0] if nums else 0, 0, 0]
    return [int(m.group(1)), 1 if m.group(2) == "final" else 0, int(m.group(3) or 0)]


def pmc_images_per_launch(path, text=None):
    """Images per launch of the run a PMC summary was collected on: its "images_per_launch: N" header line (written by
    tools/pmc_summary.py since round 6).  Older files carry none: they were collected on the default workload (1024 images
    per launch) -- except the FactorVAE passes ("factor" in the name), whose per-kernel means mix 1024- and 2048-image
    launches and are therefore not used for traffic (None)."""
    text = open(path).read() if text is None else text
    m = re.search(r"images_per_launch:\s*(\d+)", text)
    if m:
        return int(m.group(1))
    return None if "factor" in os.path.basename(path) else 1024


def pmc_traffic(kernel, images):
    """HBM bytes per launch of the kernel row `kernel` (the exact template variant, e.g. "k_up32ws<16, 2, false>") at `images`
    images per launch, from the committed rocprofv3 PMC summaries under profiles/ (tools/pmc_collect.sh ->
    tools/pmc_summary.py: separate --pmc passes for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled as MI355X_MICROARCH.md
    section HBM prescribes for 16-byte coalesced streaming reads on gfx950): the newest summary collected AT that size if
    one lists the row, else the newest one that lists it, scaled by images / its images-per-launch (every kernel here is
    a persistent loop over per-image units).  Read from a file committed by an earlier GPU visit, not measured in this run.
    Returns (bytes, file, images per launch of the file) or (None, None, None)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.md")), key=pmc_file_order)
    best = None
    for f in reversed(files):
        text = open(f).read()
        ipl = pmc_images_per_launch(f, text)
        if ipl is None:
            continue
        for l in text.splitlines():
            cells = [c.strip() for c in l.strip("|").split("|")]
            if l.startswith("|") and cells and cells[0] == kernel:
                try:
                    hit = ((float(cells[-3]) + float(cells[-2])) * 1e6 * images / ipl, os.path.relpath(f, ROOT), ipl)
                except ValueError:
                    continue
                if ipl == images:
                    return hit
                best = best or hit
    return best or (None, None, None)


def _time_launch(fn, n=50, warm=3, settle_ms=30.0, max_rounds=8):
    """ms per launch of `fn` alone, back-to-back, HIP events on torch's current stream (= the stream the launches go to).
    Untimed first: `warm` launches, then rounds of >= `settle_ms` of launches until two consecutive rounds agree within 2 %
    (a roofline leg follows seconds of host work: twenty launches from an idle-clocked GPU under-read a 80 us kernel by
    8-12 %, profiles/r06_v7_ab2.txt vs the r05 bench lines).  Then `n` timed launches."""
    def run(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    last = run(5)
    for _ in range(max_rounds):
        cur = run(max(5, min(400, int(settle_ms / max(last, 1e-3)))))
        done = abs(cur - last) <= 0.02 * cur
        last = cur
        if done:
            break
    return run(n)


def kernel_rooflines(B, device):
    """HIP-event timing (events on torch's current stream = the stream the launches go to) of the three
    32 <-> 32 channel MFMA conv families at their largest geometry (32x32 <-> 16x16, 8.59 GFLOP per 1024 images:
    2 x 4.19 M MACs per image), launched through the C-ABI.  Per training step each family runs twice
    at this geometry: k_up32ws<16> = convT2 fwd + conv2 dgrad (masked), k_down32dma<16> = conv2 fwd + convT2 dgrad
    (masked), k_wgrad32ws<16> = conv2 wgrad + convT2 wgrad."""
    import ctypes
    from disvae_amd import _lib
    from disvae_amd._lib import call, ptr
    f = lambda *s: torch.rand(*s, device=device)
    big, small = f(B, 32, 32, 32), f(B, 16, 16, 32)
    obig, osmall = torch.empty_like(big), torch.empty_like(small)
    w = f(32, 32, 4, 4) - 0.5
    b = torch.zeros(32, device=device)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=device)
    s = torch.cuda.current_stream().cuda_stream
    NH, RELU, NONE = _lib.NHWC, _lib.ACT_RELU, _lib.ACT_NONE
    # exactly what the training step launches: the kernels on pre-staged weight images (dvae_stage_weights, once per step)
    imd, imu = torch.empty(16384, device=device), torch.empty(16384, device=device)
    cd = (_lib.ConvImageDesc * 1)()
    cd[0].w, cd[0].img_down, cd[0].img_up = ptr(w), ptr(imd), ptr(imu)
    call("dvae_stage_weights", ctypes.addressof(cd), 1, None, 0, None, None, None, s)
    bits = torch.empty(B * 1024, dtype=torch.int32, device=device)
    call("dvae_conv32_up_bits", ptr(small), ptr(imu), ptr(b), None, ptr(obig), ptr(bits), B, RELU, s)   # a real bit plane
    big_b, small_b, bits_b = 32 * 32 * 32 * 4.0, 16 * 16 * 32 * 4.0, 32 * 32 * 4.0          # bytes per image
    # (launch, kernel row of the PMC summary, algorithmic HBM bytes per image: every tensor moved once, call)
    fams = {
        "k_up32ws<16>": [
            ("convT2 fwd (emits the bit plane)", "k_up32ws<16, 0, true>", big_b + small_b + bits_b,
             lambda: call("dvae_conv32_up_bits", ptr(small), ptr(imu), ptr(b), None, ptr(obig), ptr(bits), B, RELU, s)),
            ("conv2 dgrad (masked by the bit plane)", "k_up32ws<16, 2, false>", big_b + small_b + bits_b,
             lambda: call("dvae_conv32_up_bits", ptr(small), ptr(imu), None, ptr(bits), ptr(obig), None, B, NONE, s))],
        "k_down32dma<16>": [
            ("conv2 fwd", "k_down32dma<16, false>", big_b + small_b,
             lambda: call("dvae_conv32_down", ptr(big), ptr(imd), ptr(b), None, ptr(osmall), NH, B, 16, RELU, s)),
            ("convT2 dgrad (masked, fp32 activation)", "k_down32dma<16, true>", big_b + 2 * small_b,
             lambda: call("dvae_conv32_down", ptr(big), ptr(imd), None, ptr(small), ptr(osmall), NH, B, 16, NONE, s))],
        "k_wgrad32ws<16>": [
            ("conv2 wgrad (+reduce)", "k_wgrad32ws<16>", big_b + small_b,
             lambda: call("dvae_conv4s2_wgrad", ptr(big), NH, ptr(small), NH, ptr(dw), ptr(db), B, 32, 32, 32, 32, ptr(ws), s))],
    }
    flops = 2.0 * 4194304 * B              # algorithmic FLOPs per launch: 2 x MACs/img x images per launch
    out = []
    for name, launches in fams.items():
        rows = []
        for what, krow, bpi, fn in launches:
            traffic, src, ipl = pmc_traffic(krow, B)
            rows.append({"launch": what, "kernel": krow, "us": round(_time_launch(fn) * 1e3, 2), "algorithmic_bytes": bpi * B,
                         "traffic": round(traffic) if traffic is not None else None, "traffic_source": src,
                         "traffic_measured_at_images": ipl})
        tot = sum(r["us"] for r in rows) * 1e-3
        achieved = flops * len(rows) / (tot * 1e-3) / 1e12
        tr = [r["traffic"] for r in rows]
        out.append({"bound": "mfma", "kernel": name, "launches": {r["launch"]: r["us"] for r in rows}, "variants": rows,
                    "us_per_launch": round(tot / len(rows) * 1e3, 2), "us_per_step": round(tot * 1e3 * (2 // len(rows)), 2),
                    "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "images_per_launch": B,
                    "algorithmic_bytes": sum(r["algorithmic_bytes"] for r in rows) / len(rows),
                    "traffic": round(sum(tr) / len(tr)) if all(t is not None for t in tr) else None,
                    "traffic_source": rows[0]["traffic_source"],
                    "timing": "kernel alone, back-to-back launches (HIP events on the launch stream)"})
    out.sort(key=lambda r: -r["us_per_step"])
    return out


def in_step_durations(step_fn, n_steps=6):
    """The roofline launches INSIDE the training step: every call of their entry points is bracketed by HIP events on the
    stream it is issued to (disvae_amd._lib.TRACE) over `n_steps` iterations of the timed loop's own step function (eager
    issue); returns {family or thin launch: mean in-step duration in us} -- the three 32-channel families at 32x32 <-> 16x16
    and the five launches that touch the C-channel image.  Inside a step a kernel shares the chip with the other stream's work:
    these are the durations rocprofv3 --kernel-trace reports for the same command (profiles/r*_kernel_stats.md), not the
    kernel's best case."""
    from disvae_amd import _lib

    def family(name, a):
        if name == "dvae_conv32_up_bits" or (name == "dvae_conv32_up" and a[7] == 16):
            return "k_up32ws<16>"
        if name == "dvae_conv32_down" and a[7] == 16:
            return "k_down32dma<16>"
        if name == "dvae_conv4s2_wgrad" and a[7] == 32 and a[8] == 32:
            return "k_wgrad32ws<16>"              # conv2: 32 input channels at 32x32 (+ its reduction launch)
        if name == "dvae_convT4s2_wgrad" and a[7] == 32 and a[8] == 16 and a[10] == 32:
            return "k_wgrad32ws<16>"              # convT2: 16x16 input, 32 output channels
        if name == "dvae_conv1_fwd_bits":
            return "conv1 fwd (emits the bit plane)"
        if name == "dvae_convT3_fwd_staged":
            return "convT3 fwd + sigmoid + likelihood + dL/dlogit"
        if name == "dvae_convT3_dgrad_bits":
            return "convT3 dgrad (masked by the bit plane)"
        if name == "dvae_convT4s2_wgrad" and a[8] == 32 and a[10] in (1, 3):
            return "convT3 wgrad (+reduce)"
        if name == "dvae_conv4s2_wgrad" and a[7] in (1, 3) and a[8] == 64:
            return "conv1 wgrad (+reduce)"
        return None
    trace = {"names": {"dvae_conv32_up_bits", "dvae_conv32_up", "dvae_conv32_down", "dvae_conv4s2_wgrad", "dvae_convT4s2_wgrad",
                       "dvae_conv1_fwd_bits", "dvae_convT3_fwd_staged", "dvae_convT3_dgrad_bits"},
             "out": []}
    _lib.TRACE = trace
    try:
        for _ in range(n_steps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        _lib.TRACE = None
    acc = defaultdict(list)
    for name, a, e0, e1 in trace["out"]:
        f = family(name, a)
        if f:
            acc[f].append(e0.elapsed_time(e1) * 1e3)
    return {f: round(sum(v) / len(v), 2) for f, v in acc.items()}


def apply_in_step(rows, ins):
    """`achieved` / `frac` of a roofline entry = its ALGORITHMIC work over the launch's mean duration INSIDE the timed step
    (in_step_durations; what rocprofv3 reports for the same command); the kernel alone, back to back, moves to `alone`."""
    for r in rows:
        key = r["kernel"] if r["kernel"] in ins else next((k for k in ins if r.get("launch", "").startswith(k)), None)
        if key is None:
            continue
        alone_us = r["us_per_launch"]
        r["alone"] = {"us_per_launch": alone_us, "achieved": r["achieved"], "frac": r["frac"],
                      "timing": "the kernel alone, back-to-back launches after a settle phase (HIP events on the launch stream)"}
        scale = alone_us / ins[key]
        r["us_per_launch"] = ins[key]
        r["achieved"] = round(r["achieved"] * scale, 2)
        r["frac"] = round(r["achieved"] / r["peak"], 4)
        r["timing"] = ("mean duration of the launch inside the timed training step (HIP events on the stream it is issued to, the "
                       "other stream's kernels sharing the chip): agrees with rocprofv3 --kernel-trace of the same command")
        r.pop("in_step_us", None)
    return rows


def disc_kernel_rooflines(M, device):
    """The FactorVAE discriminator's 1000 x 1000 layers (discriminator.py:52-55) through the C-ABI at the step's own row
    counts: forward and the first input-gradient chain + weight gradient at M rows (both halves of the batch), the second
    input-gradient chain at M / 2.  k_gdma / k_gdma_wg (csrc/gemm_dma.hip); 2 M K N FLOP per launch against the fp32 MFMA peak."""
    from disvae_amd import _lib
    from disvae_amd._lib import call, ptr
    K = N = 1000
    s = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=device)
    out = []
    for rows, forms in ((M, ("fwd", "dgrad", "wgrad")), (M // 2, ("dgrad",))):
        x = torch.rand(rows, K, device=device) - 0.5
        w = (torch.rand(N, K, device=device) - 0.5) * 0.1
        b = torch.zeros(N, device=device)
        dy = torch.rand(rows, N, device=device) - 0.5
        y, dx = torch.empty(rows, N, device=device), torch.empty(rows, K, device=device)
        dw, db = torch.empty(N, K, device=device), torch.empty(N, device=device)
        fns = {"fwd": lambda: call("dvae_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(y), rows, K, N, _lib.ACT_LEAKY02, ptr(ws), s),
               "dgrad": lambda: call("dvae_linear_dgrad", ptr(dy), ptr(w), ptr(x), _lib.ACT_LEAKY02, ptr(dx), rows, K, N, ptr(ws), s),
               "wgrad": lambda: call("dvae_linear_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(db), rows, K, N, ptr(ws), s)}
        tile = "128x64" if (rows + 127) // 128 * 16 >= 224 else ("64x64" if (rows + 63) // 64 * 16 >= 192 else "32x32, contraction split over 4 waves")
        for form in forms:
            us = _time_launch(fns[form]) * 1e3
            tf = 2.0 * rows * K * N / us / 1e6
            kern = "k_gdma_wg<64, 3>" if form == "wgrad" else "k_gdma (%s tiles, %s)" % (tile, "w^T k-contiguous" if form == "fwd" else "w contraction-slow")
            out.append({"bound": "mfma", "kernel": kern, "launch": "discriminator %s, %d x 1000 x 1000" % (form, rows),
                        "launches_per_step": 4, "us_per_launch": round(us, 2), "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None})
    return out


def thin_kernel_rooflines(B, C, device):
    """The five launches of a training step that touch the C-channel image (conv1 / convT3 + likelihood are
    bandwidth-bound): HIP-event time vs ALGORITHMIC bytes (every tensor the launch must read or write, once) against the
    8 TB/s HBM peak.  Through the C-ABI, at the step's own sizes."""
    from disvae_amd import _lib
    from disvae_amd._lib import call, ptr
    f = lambda *s: torch.rand(*s, device=device)
    x, a1 = f(B, C, 64, 64), f(B, 32, 32, 32)
    g, rec, ga1 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(a1)
    w, wt = f(32, C, 4, 4) - 0.5, f(32, C, 4, 4) - 0.5
    b32, bc = torch.zeros(32, device=device), torch.zeros(C, device=device)
    dw, db, dbc = torch.empty_like(w), torch.empty_like(b32), torch.empty_like(bc)
    coef = torch.full((8,), 1.0 / B, device=device)
    parts = torch.empty(_lib.REC_NPART, device=device)
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=device)
    s = torch.cuda.current_stream().cuda_stream
    NC, NH = _lib.NCHW, _lib.NHWC
    nx, na = x.numel() * 4.0, a1.numel() * 4.0
    pairs = torch.empty(32 * _lib.thin_pair_floats(C), device=device)
    td = _lib.ThinImageDesc()
    td.w, td.img_pairs, td.C = ptr(wt), ptr(pairs), C
    call("dvae_stage_weights", None, 0, None, 0, ctypes.addressof(td), None, None, s)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * 1024,), dtype=torch.int32, device=device)
    nb = bits.numel() * 4.0
    # (kernel row of the PMC summary, launch, algorithmic bytes, call): what the step launches -- conv1's forward emits the bit
    # plane of its output, convT3's input gradient is masked by convT2's
    # batches >= 192 images take the wave-specialised kernels of conv_thin_ws.hip
    big = B >= 192
    k_fwd = ("k_down_thin_ws<%d, 3, 0>" if big else "k_down_thin<%d, 3, float>") % C
    k_dg = ("k_down_thin_ws<%d, 2, 0>" if big else "k_down_thin<%d, 2, float>") % C
    k_wgT = ("k_wgrad_thin_ws<%d, true, 0>" if big else "k_wgrad_thin<%d, float>") % C
    k_wg1 = ("k_wgrad_thin_ws<%d, false, 0>" if big else "k_wgrad_thin<%d, float>") % C
    launches = [
        (k_fwd, "conv1 fwd (emits the bit plane)", nx + na + nb,
         lambda: call("dvae_conv1_fwd_bits", ptr(x), 0, ptr(w), ptr(b32), ptr(ga1), ptr(bits), B, C, s)),
        ("k_up_thin_mm<true, 0, float>" if C == 3 else "k_up_thin_pk<1, true, float>",
         "convT3 fwd + sigmoid + likelihood + dL/dlogit (%s)" % ("matrix cores, 2 x 2-window form" if C == 3 else "packed FMAs"), na + 3 * nx,
         lambda: call("dvae_convT3_fwd_staged", ptr(a1), ptr(pairs), ptr(bc), ptr(x), 0, ptr(rec), ptr(g), 0, ptr(coef),
                      ptr(parts), B, C, s)),
        (k_dg, "convT3 dgrad (masked by the bit plane)", nx + na + nb,
         lambda: call("dvae_convT3_dgrad_bits", ptr(x), ptr(wt), ptr(bits), ptr(ga1), B, C, s)),
        (k_wgT, "convT3 wgrad (+reduce)", nx + na,
         lambda: call("dvae_convT4s2_wgrad", ptr(a1), NH, ptr(x), NC, ptr(dw), ptr(dbc), B, 32, 32, 32, C, ptr(ws), s)),
        (k_wg1, "conv1 wgrad (+reduce)", nx + na,
         lambda: call("dvae_conv4s2_wgrad", ptr(x), NC, ptr(a1), NH, ptr(dw), ptr(db), B, C, 64, 64, 32, ptr(ws), s)),
    ]
    out = []
    for kern, what, nbytes, fn in launches:
        ms = _time_launch(fn)
        gbs = nbytes / (ms * 1e-3) / 1e9
        traffic, src, ipl = pmc_traffic(kern, B)
        out.append({"bound": "hbm", "kernel": kern, "launch": what, "us_per_launch": round(ms * 1e3, 2),
                    "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "images_per_launch": B, "algorithmic_bytes": nbytes,
                    "traffic": round(traffic) if traffic is not None else None, "traffic_source": src,
                    "traffic_measured_at_images": ipl})
    return out


# ---------------------------------------------------------------------------------- CPU legs
def _oracle_hp(cfg):
    return dict(HP, n_data=cfg["n_data"], lr_disc=cfg["lr_disc"])


_THREADS = {}


def _calibrate_threads(make_step, img):
    """torch's default of one thread per logical CPU oversubscribes large hosts badly (a 256-thread probe alone takes tens
    of seconds on a 256-CPU box): the fastest of {8, 16, 32, 64} (or all CPUs of a smaller host) on a B = 128 probe is used.
    Calibrated ONCE per process: the default bench line times four workloads on the CPU."""
    ncpu = os.cpu_count() or 1
    if "n" in _THREADS:
        torch.set_num_threads(_THREADS["n"])
        return _THREADS["n"], ncpu
    probe = torch.rand((128,) + tuple(img))
    best_t, best_n = None, 1
    cands = [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]
    for n in cands:
        torch.set_num_threads(n)
        step = make_step()
        step(probe)
        t0 = time.perf_counter()
        step(probe)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, n
    torch.set_num_threads(best_n)
    _THREADS["n"] = best_n
    return best_n, ncpu


def _time_cpu(make_step, img, B, iters, warm):
    threads, ncpu = _calibrate_threads(make_step, img)
    step = make_step()
    data = torch.rand((B,) + tuple(img))
    ts = []
    for _ in range(warm + iters):
        t0 = time.perf_counter()
        step(data)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[warm:])
    return ts[len(ts) // 2], threads, ncpu


def cpu_baseline_port(cfg, B, iters=6, warm=2):
    """CPU oracle (port of the reference Trainer iteration, incl. the wasted full-batch forward of
    training.py:153 for factor) on this box's host cores."""
    from oracle import disvae_oracle as O
    loss, img = cfg["loss"], cfg["img"]
    hp = _oracle_hp(cfg)

    def make():
        torch.manual_seed(1234)
        tr = O.OracleTrainer(loss, hp, img, 10, lr=cfg["lr"], lr_disc=cfg["lr_disc"], steps_anneal=HP["reg_anneal"])
        return tr.train_iteration

    med, threads, ncpu = _time_cpu(make, img, B, iters, warm)
    return {"value": round(B / med, 1), "unit": "images/s", "cores": threads, "kind": "port",
            "batch": B,
            "sample": "oracle (torch-CPU restatement of reference Trainer._train_iteration), %s %dx%dx%d B=%d, "
                      "median of %d iterations after %d warm-ups, %.0f ms/iter, %d threads (best of a sweep) on a "
                      "%d-CPU host" % (loss, img[1], img[2], img[0], B, iters, warm, med * 1e3, threads, ncpu)}


REFERENCE_DIR = os.environ.get("DVAE_REFERENCE", "/root/reference")


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE_DIR, "disvae"))


def cpu_baseline_reference(cfg, B, iters=6, warm=2):
    """The UNMODIFIED reference (read-only checkout, imported with the two shims of SURVEY.md 8c: a stub `imageio`
    module and numpy.product) timed on this box's host cores: init_specific_model + optim.Adam + get_loss_f +
    Trainer._train_iteration (training.py:137-164) on the same synthetic batch.  Only where the checkout exists."""
    import logging
    import types
    import numpy as np
    sys.modules.setdefault("imageio", types.ModuleType("imageio"))
    if not hasattr(np, "product"):
        np.product = np.prod
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    from disvae.models.vae import init_specific_model as ref_model
    from disvae.models.losses import get_loss_f as ref_loss_f
    from disvae.training import Trainer as RefTrainer
    loss, img = cfg["loss"], cfg["img"]
    dev = torch.device("cpu")

    os.makedirs("/tmp/dvae_bench_ref", exist_ok=True)      # the reference's LossesLogger opens a file there

    def make():
        torch.manual_seed(1234)
        model = ref_model("Burgess", img, 10)
        opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])
        loss_f = ref_loss_f(loss, n_data=cfg["n_data"], device=dev, lr_disc=cfg["lr_disc"], **HP)
        tr = RefTrainer(model, opt, loss_f, device=dev, logger=logging.getLogger("bench.ref"),
                        save_dir="/tmp/dvae_bench_ref", is_progress_bar=False)
        model.train()
        storer = defaultdict(list)
        return lambda data: tr._train_iteration(data, storer)

    med, threads, ncpu = _time_cpu(make, img, B, iters, warm)
    return {"value": round(B / med, 1), "unit": "images/s", "cores": threads, "kind": "reference",
            "batch": B,
            "sample": "the reference's own Trainer._train_iteration (%s, unmodified, torch CPU), %s %dx%dx%d B=%d, "
                      "median of %d iterations after %d warm-ups, %.0f ms/iter, %d threads (best of a sweep) on a "
                      "%d-CPU host" % (REFERENCE_DIR, loss, img[1], img[2], img[0], B, iters, warm, med * 1e3, threads, ncpu)}


def cpu_baseline(cfg, B, iters=6, warm=2):
    """kind "reference" wherever the reference checkout exists (the build container), kind "port" elsewhere (the GPU
    boxes receive the repository only)."""
    if have_reference():
        return cpu_baseline_reference(cfg, B, iters, warm)
    return cpu_baseline_port(cfg, B, iters, warm)


def parity_check(cfg, B, device):
    """First training iteration of THIS workload (fresh seed-1234 weights, the same kind of synthetic batch,
    injected noise) on the HIP engine vs the oracle: loss vs the fp32 oracle (= the reference's arithmetic),
    gradients vs the fp64 oracle evaluated with the engine's ReLU / LeakyReLU on/off pattern (oracle/gate_match.py: a unit
    whose pre-activation is within rounding of zero is gated differently by any two arithmetics -- the pattern itself is
    checked: such units must sit within 1e-5 of the layer scale of zero).  Outside the timed region; the oracle is the
    checker, never the measured path."""
    from oracle import disvae_oracle as O
    from oracle import gate_match as GM
    from disvae_amd.models.vae import init_specific_model
    from disvae_amd.models.losses import get_loss_f
    loss, img = cfg["loss"], cfg["img"]
    hp = _oracle_hp(cfg)
    torch.manual_seed(1234)
    model = init_specific_model("Burgess", img, 10).to(device)
    opt = make_optimizer(model, cfg["lr"])          # exactly what the timed loop steps with
    loss_f = get_loss_f(loss, n_data=cfg["n_data"], device=device, lr_disc=cfg["lr_disc"], **HP)
    loss_f.replay = None
    model.train()
    torch.manual_seed(1234)
    p0 = O.init_vae_params(img, 10)
    gen = torch.Generator().manual_seed(4321)
    data = torch.rand((B,) + tuple(img), generator=gen)
    st = lambda: O.LossState(steps_anneal=HP["reg_anneal"])
    c64 = lambda p: O.clone_params(p, dtype=torch.float64, requires_grad=True)
    t0 = time.perf_counter()
    log = []
    if loss == "factor":
        d0 = O.init_disc_params(10)
        Bh = B // 2
        eps1, eps2 = torch.randn(Bh, 10, generator=gen), torch.randn(Bh, 10, generator=gen)
        perms = torch.stack([torch.randperm(Bh, generator=gen) for _ in range(10)])
        ref_loss = O.factor_iteration_grads(hp, st(), O.clone_params(p0, requires_grad=True),
                                            O.clone_params(d0, requires_grad=True), data, eps1, eps2, list(perms))[0]
        out = loss_f.call_optimize(data.to(device), model, opt, None, noise=(eps1.to(device), eps2.to(device), perms))
        gates = GM.engine_gates(model, B, splits=[slice(0, Bh), slice(Bh, 2 * Bh)], dec_rows=slice(0, Bh))
        gates.update(GM.discriminator_gates(loss_f.discriminator, 2 * Bh, Bh))
        args64 = (hp, st(), c64(p0), c64(d0), data.double(), eps1.double(), eps2.double(), list(perms))
        with O.gates(None, record=log):
            O.factor_iteration_grads(hp, st(), c64(p0), c64(d0), data.double(), eps1.double(), eps2.double(), list(perms))
        with O.gates(gates):
            _, _, g64, gd64, _ = O.factor_iteration_grads(*args64)
        grads = [(k, p.grad, g64[k]) for k, p in model.named_parameters()]
        grads += [("disc." + k, p.grad, gd64[k]) for k, p in loss_f.discriminator.named_parameters()]
    else:
        eps = torch.randn(B, 10, generator=gen)
        ref_loss = O.train_iteration_grads(loss, hp, st(), O.clone_params(p0, requires_grad=True), data, eps)[0]
        out = loss_f.fused_step(data.to(device), model, opt, None, eps=eps.to(device))
        gates = GM.engine_gates(model, B)
        with torch.no_grad(), O.gates(None, record=log):
            O.vae_forward(O.clone_params(p0, dtype=torch.float64), data.double(), eps.double())
        with O.gates(gates):
            _, _, g64, _ = O.train_iteration_grads(loss, hp, st(), c64(p0), data.double(), eps.double())
        grads = [(k, p.grad, g64[k]) for k, p in model.named_parameters()]
    n_diff, gate_worst, gates_ok = GM.gate_mismatches(gates, log)
    got = float(out.item())
    worst, worst_name = 0.0, ""
    for k, g, r in grads:
        r = r.double()
        e = ((g.detach().cpu().double() - r).abs().max() / (r.abs().max() + 1e-300)).item()
        if e > worst:
            worst, worst_name = e, k
    loss_err = abs(got - float(ref_loss)) / abs(float(ref_loss))
    return {"ok": bool(loss_err <= 1e-5 and worst <= 1e-5 and gates_ok), "loss": got, "oracle_fp32_loss": float(ref_loss),
            "loss_rel_err": loss_err, "loss_rtol": 1e-5, "worst_grad_err_over_max_abs_grad_vs_gate_matched_fp64": worst,
            "worst_grad_tensor": worst_name, "grad_gate": 1e-5, "units_gated_differently_than_fp64": n_diff,
            "worst_such_preactivation_over_layer_scale": gate_worst, "batch": B, "seconds": round(time.perf_counter() - t0, 1)}


def make_optimizer(model, lr):
    """torch.optim.Adam as in main.py:208 on the parameter arena viewed as equal chunks (element-wise identical to Adam
    over the 28 state_dict views: tests/test_gpu_step.py::test_flat_fused_adam_equals_adam_over_the_state_dict_views),
    fused=True = torch's single-kernel multi-tensor variant."""
    return torch.optim.Adam(model.flat_parameters(), lr=lr, fused=True)


def time_leg(cfg, B, device, steps, warmup, drop_in=False, replay=None):
    """One single-GPU leg: fresh seed-1234 model, resident synthetic batch, `warmup` untimed + `steps` timed iterations
    between two synchronisations.  drop_in: driven the way INTEGRATION.md tells a reference user to -- optim.Adam over
    model.parameters() (main.py:208 verbatim); "epoch": through Trainer._train_epoch (what Trainer.__call__ runs, one host
    sync per epoch without a progress bar), True: Trainer._train_iteration, i.e. one loss.item() host sync per iteration
    (training.py:164; what is_progress_bar=True needs).  Returns (ms per step, final loss)."""
    import logging
    from disvae_amd.models.vae import init_specific_model
    from disvae_amd.models.losses import get_loss_f
    from disvae_amd.training import Trainer
    torch.manual_seed(1234)
    model = init_specific_model("Burgess", cfg["img"], 10).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"]) if drop_in else make_optimizer(model, cfg["lr"])
    loss_f = get_loss_f(cfg["loss"], n_data=cfg["n_data"], device=device, lr_disc=cfg["lr_disc"], **HP)
    trainer = Trainer(model, opt, loss_f, device=device, logger=logging.getLogger("bench"), save_dir="/tmp/dvae_bench_leg",
                      is_progress_bar=False, replay=replay)
    model.train()
    gen = torch.Generator(device=device).manual_seed(1234)
    data = torch.rand((B,) + tuple(cfg["img"]), device=device, generator=gen)
    torch.cuda.manual_seed(1234)
    storer = defaultdict(list)
    settle_rounds(lambda: trainer._train_iteration_async(data, storer))       # untimed: the leg may follow seconds of CPU work
    if drop_in == "epoch":
        # Trainer.__call__'s inner loop (training.py:104-135) over a loader of resident batches, no progress bar
        # (main.py --no-progress-bar): the mean epoch loss is the one host sync
        class Resident:
            def __init__(self, n):
                self.n = n

            def __len__(self):
                return self.n

            def __iter__(self):
                return iter([(data, 0)] * self.n)
        trainer._train_epoch(Resident(warmup), storer, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = trainer._train_epoch(Resident(steps), storer, 1)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, float(loss)
    step = trainer._train_iteration if drop_in else trainer._train_iteration_async
    for _ in range(warmup):
        step(data, storer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(data, storer)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / steps * 1e3, float(loss if drop_in else loss.item())


def extra_config(name, device, steps, warmup, with_cpu, with_parity):
    """A BASELINE workload other than the headline one as a short single-GPU leg (outside the headline timed region)."""
    cfg = dict(CONFIGS[name])
    B, C = cfg["batch"], cfg["img"][0]
    mark("configs:%s:gpu" % name)
    ms, final_loss = time_leg(cfg, B, device, steps, warmup)
    flops_img = flops_per_image_factor(C) if cfg["loss"] == "factor" else flops_per_image_train(C, cfg["img"][1])
    tf = flops_img * B / (ms * 1e-3) / 1e12
    out = {"name": name, "baseline_config": cfg["baseline_config"], "loss": cfg["loss"], "img": list(cfg["img"]), "batch": B,
           "value": round(B / (ms * 1e-3), 1), "unit": "images/s", "ms_per_step": round(ms, 4), "steps": steps, "warmup": warmup,
           "step_tflops": round(tf, 2), "step_frac_of_fp32_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
           "final_loss": round(final_loss, 4)}
    mark("configs:%s:parity" % name)
    if with_parity:
        pc = parity_check(cfg, B, device)
        out["parity_check"] = {k: pc[k] for k in ("ok", "loss_rel_err", "worst_grad_err_over_max_abs_grad_vs_gate_matched_fp64",
                                                  "units_gated_differently_than_fp64", "seconds")}
    if cfg["loss"] == "factor":
        out["roofline_kernels"] = disc_kernel_rooflines(B, device)     # the discriminator's GEMMs at this config's row counts
    mark("configs:%s:cpu" % name)
    if with_cpu:
        # the workload's own tensor (factor_celeba: 2048 images per CPU iteration, 1 warm-up + 2 timed; "batch" says so)
        out["cpu_baseline"] = cpu_baseline(cfg, B, iters=3 if B <= 256 else 2, warm=1)
    return out


DEFAULT_TRANSPORT = {"auto": "rccl"}.get(os.environ.get("DVAE_COMM", "auto"), os.environ.get("DVAE_COMM", "auto"))
SHARD_WORLD = 8       # BASELINE configs[3]: "b=1024 ... DDP over 8xMI355X" = 128 images per GPU


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def settle_rounds(step_fn, max_rounds=12, tol=0.015, min_round_ms=40.0):
    """Untimed settle phase: rounds of >= 5 steps and >= `min_round_ms` each until two consecutive rounds agree within `tol`
    (clocks, caches and the allocator reach their steady state: a leg that follows seconds of CPU work starts on an idle-clocked
    GPU).  Returns the ms per step of each round."""
    rounds, n = [], 5
    for _ in range(max_rounds):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(n):
            step_fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - ts) * 1e3
        rounds.append(dt / n)
        if dt < min_round_ms:
            n = min(int(n * min_round_ms / max(dt, 1e-3)) + 1, 400)
            continue
        if len(rounds) >= 2 and abs(rounds[-1] - rounds[-2]) <= tol * rounds[-1]:
            break
    return rounds


def host_issue_ms(step_fn, n=30):
    """Host time to ISSUE one iteration (the GPU idle at the start, no synchronisation inside): when this exceeds the GPU's
    time per step the iteration is bound by the host, whatever the kernels do.  Median of n."""
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step_fn()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return sorted(ts)[len(ts) // 2] * 1e3


def shard_legs(device, steps, warmup, with_parity, with_roofline, which=("single", "rccl", "torch"), name="btcvae_celeba"):
    """BASELINE configs[3] (name = "btcvae_celeba") or configs[4] ("factor_celeba") as ONE of its eight ranks runs it, on this
    one GPU, through the SHARDED code path of the step (disvae_amd.parallel):
      btcvae: 128 images -- packed latent all-gather, the rank's 128 rows of the 1024-column B x B estimator, column gradients
              + loss sums in one all-reduce, the gradient arena all-reduced under the backward pass;
      factor: tensor 256 = 128 + 128 -- all-gather of the second half's latents (permute_dims over the GLOBAL 1024-row half
              batch, shared-seed permutations), the 16 MB discriminator gradient arena all-reduced on the communication stream
              under the whole VAE backward pass, the 2 MB VAE arena behind it, two optimizers (losses.py:281-308).
    The seven absent peers are stood in for by parallel.MirroredWorldComm (identical shards: every collective goes through a
    real one-rank RCCL communicator -- torch.distributed's, then the C-ABI's dvae_comm_* -- and is completed by replication /
    scaling), so the leg times the rank's own kernels, launches and RCCL call sites; what it cannot contain is the xGMI
    transfer time of the bytes a real step exchanges (btcvae 2 MB + 120 KB + 80 KB; factor 16 MB + 2 MB + 40 KB).  Beside
    them: the same tensor as a single-process step (no communicator)."""
    import logging
    import torch.distributed as dist
    from disvae_amd.models.vae import init_specific_model
    from disvae_amd.models.losses import get_loss_f
    from disvae_amd.training import Trainer
    from disvae_amd import parallel
    cfg = dict(CONFIGS[name])
    loss_name = cfg["loss"]
    Bg = cfg["batch"]
    B = Bg // SHARD_WORLD
    C = cfg["img"][0]
    flops = (flops_per_image_factor(C) if loss_name == "factor" else flops_per_image_train(C)) * B
    out = {"name": name + "_shard", "baseline_config": cfg["baseline_config"], "loss": loss_name, "img": list(cfg["img"]),
           "global_batch": Bg, "batch_per_gpu": B, "world_emulated": SHARD_WORLD, "steps": steps, "warmup": warmup,
           "unit": "images/s",
           "what": ("one rank of eight: the sharded btcvae step at 128 images per GPU (global 1024-column estimator, packed "
                    "collectives through a one-rank RCCL communicator completed by MirroredWorldComm); value = images/s of "
                    "THIS rank, xGMI transfer time not included") if loss_name != "factor" else
                   ("one rank of eight: the sharded FactorVAE step at tensor 256 = 128 + 128 per GPU (permute_dims over the "
                    "global half batch, the 16 MB discriminator gradient all-reduce in flight under the VAE backward pass, two "
                    "optimizers; collectives through a one-rank RCCL communicator completed by MirroredWorldComm); value = "
                    "images/s of THIS rank, xGMI transfer time not included")}

    def leg(transport):
        torch.manual_seed(1234)
        model = init_specific_model("Burgess", cfg["img"], 10).to(device)
        opt = make_optimizer(model, cfg["lr"])
        loss_f = get_loss_f(loss_name, n_data=cfg["n_data"], device=device, lr_disc=cfg["lr_disc"], **HP)
        trainer = Trainer(model, opt, loss_f, device=device, logger=logging.getLogger("bench"), save_dir="/tmp/dvae_bench_shard",
                          is_progress_bar=False)
        model.train()
        comm = None
        if transport is not None:
            inner = parallel.RcclComm() if transport == "rccl" else parallel.Comm()
            comm = parallel.data_parallel(model, loss_f, comm=parallel.MirroredWorldComm(inner, SHARD_WORLD, 0))
        gen = torch.Generator(device=device).manual_seed(1234)
        data = torch.rand((B,) + tuple(cfg["img"]), device=device, generator=gen)
        torch.cuda.manual_seed(1234)
        storer = defaultdict(list)
        rounds = settle_rounds(lambda: trainer._train_iteration_async(data, storer))
        for _ in range(warmup):
            trainer._train_iteration_async(data, storer)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = trainer._train_iteration_async(data, storer)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        final = float(loss.item())
        host = host_issue_ms(lambda: trainer._train_iteration_async(data, storer))
        if comm is not None:
            comm.close()
        tf = flops / (ms * 1e-3) / 1e12
        return {"ms_per_step": round(ms, 4), "value": round(B / (ms * 1e-3), 1), "step_tflops": round(tf, 2),
                "step_frac_of_fp32_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "final_loss": round(final, 4),
                "replay": loss_f._replay_mode(True, data) or "eager", "host_issue_ms_per_step": round(host, 4),
                "settle_ms_per_step": [round(x, 4) for x in rounds]}

    mark("configs:%s_shard:single" % name)
    if "single" in which:
        out["single_process"] = leg(None)
    mark("configs:%s_shard:ddp" % name)
    own_group = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            parallel.init_process_group_from_env("nccl")
            own_group = True
        out["transports"] = {t: leg(t) for t in ("rccl", "torch") if t in which}
        if len(out["transports"]) == 2:
            a, b = out["transports"]["torch"]["ms_per_step"], out["transports"]["rccl"]["ms_per_step"]
            out["transports_rel_diff"] = round(abs(a - b) / min(a, b), 4)
        if out["transports"]:
            # the headline numbers of the leg = the default transport's
            first = out["transports"].get(DEFAULT_TRANSPORT) or list(out["transports"].values())[0]
            out.update({k: first[k] for k in ("ms_per_step", "value", "step_tflops", "step_frac_of_fp32_peak")})
            out["node_value_at_this_rate"] = round(out["value"] * SHARD_WORLD, 1)
    except Exception as e:                       # a box whose RCCL cannot initialise still gets the rest of the line
        out["error"] = "%s: %s" % (type(e).__name__, e)
    finally:
        if own_group and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass
    mark("configs:%s_shard:parity" % name)
    if with_parity:
        # the kernels of the shard's size (batch-sized variants: the thin convolutions, the FC chain, replayed launch plan)
        # against the oracle: the first iteration of a single-process step on the shard's tensor
        pc = parity_check(dict(cfg, batch=B), B, device)
        out["parity_check"] = {k: pc[k] for k in ("ok", "loss_rel_err", "worst_grad_err_over_max_abs_grad_vs_gate_matched_fp64",
                                                  "units_gated_differently_than_fp64", "batch", "seconds")}
        out["parity_check"]["sharded_path"] = ("tests/test_gpu_ddp.py: sharded == global-batch step (2 ranks), mirrored world == "
                                               "tiled single-process step at these sizes")
    mark("configs:%s_shard:rooflines" % name)
    if with_roofline:
        nimg = B // 2 if loss_name == "factor" else B        # FactorVAE: the decoder and the backward pass see the first half
        out["roofline_kernels"] = kernel_rooflines(nimg, device) + thin_kernel_rooflines(nimg, C, device)
        if loss_name == "factor":
            out["roofline_kernels"] += disc_kernel_rooflines(B, device)
    return out


# ---------------------------------------------------------------------------------- N ranks from one command
def self_launch_cmd(argv, n, port):
    """`python bench.py --gpus N ...` typed without a launcher: the command that runs the same arguments as N ranks of
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, n):
    """Re-execute under torch.distributed.run and pass the job's output through: rank 0's JSON line stays the LAST line
    of stdout (the launcher's own chatter goes to stderr).  Returns the job's exit code."""
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit("--gpus %d but this node shows %d GPU(s)" % (n, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = self_launch_cmd(argv, n, _free_port())
    print("[bench] --gpus %d without a launcher: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    last = None
    for line in proc.stdout:
        if last is not None:
            sys.stdout.write(last)
        last = line
    rc = proc.wait()
    if last is not None:
        sys.stdout.write(last)
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------- main
_MARKS = []


def mark(what):
    """progress on stderr (a run killed by a time limit still shows where the time went) + the `timing_s` entry of the line"""
    now = time.time()
    _MARKS.append((what, now))
    if len(_MARKS) > 1 and os.environ.get("RANK", "0") == "0":
        print("[bench] %-28s %6.1f s" % (_MARKS[-2][0], now - _MARKS[-2][1]), file=sys.stderr, flush=True)


def main():
    t_main = time.time()
    mark("setup")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="BASELINE.json workload (default btcvae_celeba)")
    ap.add_argument("--loss", default=None, choices=["btcvae", "factor", "VAE", "betaH", "betaB"],
                    help="override the config's loss (legacy: --loss factor == --config factor_celeba)")
    ap.add_argument("--batch", type=int, default=None, help="override the tensor handed to _train_iteration (the GLOBAL "
                    "batch under --scaling strong, the per-GPU batch under --scaling weak)")
    ap.add_argument("--channels", type=int, default=None)
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="N > 1: strong (default) = the config's batch is the global batch, split over the ranks; "
                         "weak = the config's batch per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--force-ddp", action="store_true", help="run the data-parallel code path (RCCL process group, "
                    "collectives, barriers) even with ONE rank: exercises the N>1 path of this script on a single GPU")
    ap.add_argument("--estimator", default="global", choices=["global", "local"],
                    help="scope of the batch-coupled estimators under data parallelism (disvae_amd.parallel.data_parallel)")
    ap.add_argument("--transport", default=None, choices=["auto", "torch", "rccl"],
                    help="collectives of the data-parallel step: the C-ABI's dvae_comm_* (RCCL enqueued by libdvae_hip.so: "
                         "rccl), torch.distributed (nccl = RCCL: torch) or auto (default: rccl after a verified round trip on "
                         "every rank, else torch); DVAE_COMM sets the default")
    ap.add_argument("--replay", default=None, choices=["auto", "eager", "plan", "graph"],
                    help="how the launches of an iteration are issued (disvae_amd/graph.py)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the short legs of the other three BASELINE "
                    "workloads (they run by default with N = 1 and the default workload)")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the drop-in leg (Adam(model.parameters()) + a host "
                    "sync per iteration)")
    ap.add_argument("--shard-which", default="single,torch,rccl", help="subset of the shard legs (profiling)")
    ap.add_argument("--shard-world", type=int, default=None, help="with --shard-legs: one rank of THIS many instead of eight "
                    "(profiling the 2- and 4-GPU shards: 512 / 256 images per rank)")
    ap.add_argument("--shard-legs", action="store_true", help="only the shard legs of --config (btcvae_celeba, default, or "
                    "factor_celeba: one rank of eight -- single process, torch transport, rccl transport), print them, exit")
    ap.add_argument("--cpu-reference", action="store_true", help="no GPU needed: time the unmodified reference Trainer "
                    "(where /root/reference exists) and the oracle's port of it on this host, print both, exit")
    args = ap.parse_args()

    if args.cpu_reference:
        name = args.config or "btcvae_celeba"
        cfg = dict(CONFIGS[name])
        B = args.batch or min(cfg["batch"], 256)
        out = {"config": name, "batch": B, "port": cpu_baseline_port(cfg, B, iters=4, warm=1)}
        if have_reference():
            out["reference"] = cpu_baseline_reference(cfg, B, iters=4, warm=1)
            out["port_over_reference"] = round(out["port"]["value"] / out["reference"]["value"], 3)
        print(json.dumps(out), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))       # `python bench.py --gpus N`: start the N ranks ourselves
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=1" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.shard_legs:
        if args.shard_world:
            global SHARD_WORLD
            SHARD_WORLD = int(args.shard_world)
        res = shard_legs(device, steps=args.steps, warmup=args.warmup, with_parity=not args.no_parity_check,
                         with_roofline=not args.no_roofline, which=tuple(args.shard_which.split(",")),
                         name=args.config or "btcvae_celeba")
        try:                                     # RCCL's banner goes through C stdio: out before the result line
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
        return

    name = args.config or ("factor_celeba" if args.loss == "factor" else "btcvae_celeba")
    cfg = dict(CONFIGS[name])
    if args.loss:
        cfg["loss"] = args.loss
        if args.loss == "factor" and CONFIGS[name]["loss"] != "factor":
            cfg["lr"] = 1e-4
    if args.channels:
        cfg["img"] = (args.channels, 64, 64)
    if args.batch:
        cfg["batch"] = args.batch
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    if scaling == "strong" and world > 1:
        if cfg["batch"] % (world * (2 if cfg["loss"] == "factor" else 1)):
            raise SystemExit("global batch %d does not split over %d ranks" % (cfg["batch"], world))
        B = cfg["batch"] // world
    else:
        B = cfg["batch"]
    B_global = B * world
    loss_name, img, lr = cfg["loss"], cfg["img"], cfg["lr"]

    from disvae_amd.models.vae import init_specific_model
    from disvae_amd.models.losses import get_loss_f
    from disvae_amd.training import Trainer
    from disvae_amd import parallel

    parity = None
    mark("parity_check")
    if world == 1 and not args.no_parity_check:
        parity = parity_check(cfg, B, device)
    mark("timed_leg")

    torch.manual_seed(1234)
    model = init_specific_model("Burgess", img, 10).to(device)
    optimizer = make_optimizer(model, lr)
    loss_f = get_loss_f(loss_name, n_data=cfg["n_data"], device=device, lr_disc=cfg["lr_disc"], **HP)
    import logging
    trainer = Trainer(model, optimizer, loss_f, device=device, logger=logging.getLogger("bench"),
                      save_dir="/tmp/dvae_bench_%d" % rank, is_progress_bar=False,
                      replay=None if args.replay is None else (False if args.replay == "eager" else args.replay))
    model.train()
    ddp = world > 1 or args.force_ddp
    if ddp:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        parallel.init_process_group_from_env("nccl")
        comm = parallel.data_parallel(model, loss_f, estimator=args.estimator, transport=args.transport)
    # synthetic batch, resident in HBM; every rank draws its own shard and its own device noise, while
    # the CPU generator (FactorVAE permutations, losses.py:505) stays identical on all ranks
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    data = torch.rand((B,) + tuple(img), device=device, generator=gen)
    torch.cuda.manual_seed(1234 + rank)
    storer = defaultdict(list)

    def barrier():
        if ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # settle phase (before the W warm-up steps, untimed): clocks, caches and the allocator reach their steady state.  Five
    # warm-up steps alone left the first timed segment 7 % slower than the last (BENCH_r03: 1.314 -> 1.222 ms).  Single
    # process: rounds of 5 steps until two consecutive rounds agree within 1.5 % (at most 12 rounds); data parallel: a fixed 20
    # steps (every rank must issue the same collectives).
    settle = []
    if ddp:
        for _ in range(20):
            trainer._train_iteration_async(data, storer)
    else:
        settle = settle_rounds(lambda: trainer._train_iteration_async(data, storer), min_round_ms=0.0)
    for _ in range(args.warmup):
        trainer._train_iteration_async(data, storer)
    # HIP events on the compute stream (torch's current stream = the stream the engine launches on) at the
    # boundaries of N_SEGMENTS equal segments of the timed region
    nseg = N_SEGMENTS if args.steps >= N_SEGMENTS else 1
    bounds = [round(i * args.steps / nseg) for i in range(nseg + 1)]
    events = [torch.cuda.Event(enable_timing=True) for _ in range(nseg + 1)]
    barrier()
    t0 = time.perf_counter()
    events[0].record()
    nxt = 1
    for i in range(args.steps):
        loss = trainer._train_iteration_async(data, storer)
        if i + 1 == bounds[nxt]:
            events[nxt].record()
            nxt += 1
    barrier()
    dt = time.perf_counter() - t0
    seg_ms = [events[i].elapsed_time(events[i + 1]) / max(bounds[i + 1] - bounds[i], 1) for i in range(nseg)]
    if ddp:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    host_ms = host_issue_ms(lambda: trainer._train_iteration_async(data, storer), n=20) if not ddp else None

    def flush_c_stdio():
        # RCCL in this image prints a banner ("Hostname", "Librccl path") through C stdio, block-buffered on a pipe:
        # it would otherwise surface at process exit, AFTER the result line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    if ddp:
        flush_c_stdio()
        comm.close()
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        flush_c_stdio()
    if rank != 0:
        return
    if world > 1:
        time.sleep(1.0)      # (outside the timed region) let the other ranks exit: the JSON line is the LAST line of the job
    ms = dt / args.steps * 1e3
    value = B_global * args.steps / dt
    C = img[0]
    flops_img = flops_per_image_factor(C) if loss_name == "factor" else flops_per_image_train(C)
    step_tf = flops_img * B_global / (ms * 1e-3) / 1e12
    dset = "celeba" if C == 3 else "dsprites"
    out = {
        "metric": "images/sec (whole node) at 64x64x%d, %s loss" % (C, loss_name),
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": scaling if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s (BASELINE.json configs[%d]): Burgess VAE 64x64x%d, %s loss, z=10, tensor handed to "
                               "_train_iteration B=%d global (%d per GPU), Adam lr %g%s, n_data=%d, fwd+loss+bwd+Adam%s"
                               % (name, cfg["baseline_config"], C, loss_name, B_global, B, lr,
                                  (", lr_disc %g" % cfg["lr_disc"]) if loss_name == "factor" else "", cfg["n_data"],
                                  "+RCCL collectives" if world > 1 else ""),
                   "name": name, "dataset_shape": dset, "batch_per_gpu": B, "global_batch": B_global,
                   "parallelism": "dp%d" % world, "estimator": args.estimator if ddp else None,
                   "transport": ("rccl" if type(comm).__name__ == "RcclComm" else "torch") if ddp else None,
                   "final_loss": round(final_loss, 4)},
        "hip_event_ms_per_step": {"segments": [round(x, 4) for x in seg_ms], "median": round(sorted(seg_ms)[len(seg_ms) // 2], 4),
                                  "note": "HIP events on the compute stream of rank 0 around %d equal segments of the timed "
                                          "iterations" % nseg},
        "step_tflops": round(step_tf, 2),
        "step_frac_of_fp32_peak": round(step_tf / world / PEAK_FP32_MFMA_TFLOPS, 4),
        "host_issue_ms_per_step": None if host_ms is None else round(host_ms, 4),
        "settle": {"untimed_steps_before_warmup": 20 if ddp else 5 * len(settle), "ms_per_step_rounds_of_5": [round(x, 4) for x in settle]},
    }
    if parity is not None:
        out["parity_check"] = parity
    mark("rooflines")
    if not args.no_roofline:
        nimg = B if loss_name != "factor" else B // 2
        fams = kernel_rooflines(nimg, device)
        thin = thin_kernel_rooflines(nimg, C, device)
        if not ddp and loss_name != "factor":
            # the durations the launches get INSIDE the timed step are what `achieved` / `frac` are computed from (FactorVAE: the
            # encoder launches run over both halves, the decoder's over one -- its entries stay kernel-alone).  The brackets sit
            # in the host-side call path, so these iterations are issued eagerly (the same launches on the same streams as the
            # recorded plan the timed loop replays up to 1024 images)
            saved_replay, loss_f.replay = loss_f.replay, None
            try:
                ins = in_step_durations(lambda: trainer._train_iteration_async(data, storer))
            finally:
                loss_f.replay = saved_replay
            apply_in_step(fams, ins)
            apply_in_step(thin, ins)
            fams.sort(key=lambda r: -r["us_per_launch"] * len(r.get("launches", [1, 1])))
        out["roofline"] = fams[0]           # the family with the largest share of the step
        out["roofline_kernels"] = fams[1:] + thin
        if loss_name == "factor":
            out["roofline_kernels"] += disc_kernel_rooflines(B, device)
    mark("drop_in")
    if world == 1 and not args.no_drop_in:
        d_steps = min(args.steps, 50)
        d_ms, _ = time_leg(cfg, B, device, d_steps, min(args.warmup, 10), drop_in="epoch")
        p_ms, _ = time_leg(cfg, B, device, d_steps, min(args.warmup, 10), drop_in=True)
        out["drop_in"] = {"ms_per_step": round(d_ms, 4), "value": round(B / (d_ms * 1e-3), 1), "steps": d_steps,
                          "driven_as": "INTEGRATION.md / main.py:204-224 verbatim: optim.Adam(model.parameters(), lr) handed to "
                                       "Trainer(..., is_progress_bar=False), one epoch of Trainer._train_epoch over a loader of "
                                       "resident batches",
                          "optimizer": "torch.optim.Adam(model.parameters(), lr) (main.py:208 verbatim, 28 tensors); the Trainer "
                                       "switches a stock Adam to torch's fused multi-tensor kernel (training.fuse_plain_adam)",
                          "host_sync": "one per epoch (the mean epoch loss, training.py:135)",
                          "over_timed_configuration": round(d_ms / ms, 3),
                          "with_progress_bar": {"ms_per_step": round(p_ms, 4), "value": round(B / (p_ms * 1e-3), 1),
                                                "host_sync": "loss.item() every iteration (Trainer._train_iteration, "
                                                             "training.py:164: what tqdm's postfix needs)",
                                                "over_timed_configuration": round(p_ms / ms, 3)}}
    mark("cpu_baseline")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, B)
    mark("configs")
    if world == 1 and not args.force_ddp and not args.no_extra_configs and name == "btcvae_celeba" and not (
            args.batch or args.channels or args.loss):
        out["configs"] = [extra_config(n, device, steps=min(args.steps, 30), warmup=min(args.warmup, 10),
                                       with_cpu=not args.no_cpu_baseline, with_parity=not args.no_parity_check)
                          for n in ("vae_mnist", "btcvae_dsprites", "factor_dsprites", "factor_celeba")]
        for shard_of in ("btcvae_celeba", "factor_celeba"):
            out["configs"].append(shard_legs(device, steps=max(min(args.steps, 100), 30), warmup=min(args.warmup, 10),
                                             with_parity=not args.no_parity_check, with_roofline=not args.no_roofline,
                                             name=shard_of))
    mark("end")
    out["timing_s"] = {a[0]: round(b[1] - a[1], 1) for a, b in zip(_MARKS[:-1], _MARKS[1:])}
    out["bench_wall_s"] = round(time.time() - t_main, 1)     # this process, main() entry to the line below (imports excluded)
    flush_c_stdio()
    print(json.dumps(out), flush=True)
    if parity is not None and not parity["ok"]:
        sys.exit(3)


if __name__ == "__main__":
    main()
