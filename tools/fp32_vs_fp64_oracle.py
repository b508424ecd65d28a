"""CPU only: how far the reference's OWN fp32 arithmetic (the torch-CPU oracle) is from fp64 on the gradients of one
training iteration, and how much of that is ReLU gating at rounding level.

    python tools/fp32_vs_fp64_oracle.py [btcvae_celeba|btcvae_dsprites]

Prints, per parameter tensor, max|g32 - g64| / max|g64| for (a) the plain fp64 oracle and (b) the fp64 oracle
evaluated with the fp32 run's ReLU on/off pattern (oracle.gates).  (a) is 1e-5 .. 3e-4, (b) ~1e-6: the difference between
two correct arithmetics is dominated by units whose pre-activation is within rounding of zero.  This is why the
whole-step parity tests compare the HIP engine with the gate-matched fp64 oracle (tests/test_gpu_bench_sizes.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import disvae_oracle as O  # noqa: E402

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25,
          betaB_G=1000, factor_G=6.4, latent_dim=10, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)
CFG = {"btcvae_dsprites": ((1, 64, 64), 256, 737280), "btcvae_celeba": ((3, 64, 64), 1024, 202599)}


def main(name):
    img, B, n_data = CFG[name]
    torch.manual_seed(1234)
    p0 = O.init_vae_params(img, 10)
    hp = dict(HP, n_data=n_data)
    gen = torch.Generator().manual_seed(1235)
    data = torch.rand((B,) + img, generator=gen)
    eps = torch.randn(B, 10, generator=gen)
    st = lambda: O.LossState(steps_anneal=10000)
    log = []
    with O.gates(None, record=log):
        _, _, g32, _ = O.train_iteration_grads("btcvae", hp, st(), O.clone_params(p0, requires_grad=True), data, eps)
    gates = {}
    for n, pre in log:
        gates.setdefault(n, []).append(pre > 0)
    c64 = lambda: O.clone_params(p0, dtype=torch.float64, requires_grad=True)
    _, _, g64, _ = O.train_iteration_grads("btcvae", hp, st(), c64(), data.double(), eps.double())
    with O.gates(gates):
        _, _, g64g, _ = O.train_iteration_grads("btcvae", hp, st(), c64(), data.double(), eps.double())
    print("%s: max|g_fp32 - g_fp64| / max|g_fp64|   plain fp64    gate-matched fp64" % name)
    for k in g64:
        a = ((g32[k].double() - g64[k]).abs().max() / g64[k].abs().max()).item()
        b = ((g32[k].double() - g64g[k]).abs().max() / g64g[k].abs().max()).item()
        print("  %-30s %12.2e %16.2e" % (k, a, b))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "btcvae_celeba")
