"""FC-chain kernels alone at batch B (debug builds: DVAE_FCC_VARIANT = 10 * ring depth + contraction split selects the
instantiation).  usage: DVAE_FCC_VARIANT=162 python tools/fcc_ab.py [B ...]
FCC_COLD=1: every timed launch follows a 1 GB streaming write, i.e. the weight images come from HBM as they do inside a
training step on a box whose caches do not retain them (profiles/r03_final_timeline.md: 61 us in the step, 26 us here)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd")):
    sys.path.insert(0, p)
import torch
from disvae_amd import _lib
from disvae_amd.models.vae import init_specific_model

dev = "cuda"
for B in [int(a) for a in sys.argv[1:]] or [128, 1024]:
    torch.manual_seed(0)
    model = init_specific_model("Burgess", (3, 64, 64), 10).to(dev)
    eng = model.engine
    buf = eng.buffers(B)
    g = torch.Generator(device=dev).manual_seed(1)
    buf.a_flat.uniform_(0, 1, generator=g)
    buf.gd3.uniform_(-1, 1, generator=g)
    coefd = torch.full((8,), 1.0 / B, device=dev)
    scal = torch.zeros(32, device=dev)
    eps = torch.randn(B, 10, device=dev, generator=g)
    kl = torch.zeros(_lib.KL_FLOATS, device=dev)
    eng.stage(coefd, [1.0 / B] * 8)

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    if os.environ.get("FCC_COLD") == "1":
        junk = torch.empty(256 * 1024 * 1024, device=dev)

        def timeit(fn, n=12):          # noqa: F811
            tot = 0.0
            for it in range(n + 2):
                junk.fill_(float(it))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                if it >= 2:
                    tot += e0.elapsed_time(e1)
            return tot / n * 1e3
    f = timeit(lambda: eng.fc_chain_fwd(buf, eps, kl, B))
    b = timeit(lambda: eng.fc_chain_bwd(buf, eps, None, None, None, None, scal, coefd, B))
    print(("cold " if os.environ.get("FCC_COLD") == "1" else "") + "variant %s B=%d: fc_chain_fwd %.1f us, fc_chain_bwd %.1f us; checksums %.9e %.9e" % (
        os.environ.get("DVAE_FCC_VARIANT", "default"), B, f, b, buf.d3.double().sum().item(), buf.ga_flat.double().sum().item()))
