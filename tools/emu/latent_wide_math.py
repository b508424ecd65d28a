"""No GPU needed: the arithmetic and the buffer layouts of csrc/latent_wide.hip (latent dimensions above 16) restated kernel by
kernel in fp64 torch and held to the oracle (oracle/disvae_oracle.py, itself pinned to the real reference at 32 / 24 / 20
latents: tests/golden/*_z32_*, *_z24_*, *_z20_*).

  k_btcvae_prep (loss.hip)   tmp = [3][D][Bg]: mu^T, -0.5 (log 2pi + logvar)^T, exp(-logvar)^T
  k_tcw_joint                S[il][j] = sum_d (log N(z_i[d]; mu_j[d], var_j[d]) + log W[i][j])     behind tmp, [Bl][Bg]
  k_tcw_rowstats             rowstats[il] = {log_pz, logsumexp_j S, sum_d lse_d, log q(z_i|x_i), lse_d[0..D-1]}, stride ROWSTATS_STRIDE(D)
  k_tcw_bwd_rows             dz[il][d]   (a wave per (row, dimension))
  k_tcw_bwd_cols             dmu[j][d], dlogvar[j][d] over the local rows (a thread per (column, dimension))
  loss_pack_body / loss_finalize_body (loss.hip), wide layout: packed[32 + d], scal[32 + d]

Checked: whole batch == the oracle's densities / gradients; two row shards reproduce the whole (rows bit for bit, column
gradients add up); the packed / scalar layouts.  Run by tests/test_host_logic.py."""
import math
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "disentangling-vae_amd"))
from oracle import disvae_oracle as O                      # noqa: E402
from disvae_amd import _lib                                # noqa: E402
from disvae_amd.utils.math import log_importance_weights   # noqa: E402

L2PI = math.log(2 * math.pi)


def log_w_ij(i, j, Bg, lN, lS, lM):
    if j == 0:
        return lS if i == Bg - 2 else lN
    if j == 1:
        return lS
    return lM


def forward(z, mu, lv, row0, Bl, lw, is_mss):
    Bg, D = z.shape
    lN, lS, lM = (lw[0], lw[1], lw[2]) if is_mss else (0.0, 0.0, 0.0)
    W = torch.tensor([[log_w_ij(row0 + il, j, Bg, lN, lS, lM) for j in range(Bg)] for il in range(Bl)], dtype=torch.float64)
    tmp = torch.empty(_lib.btcvae_tmp_floats(Bg, Bl, D), dtype=torch.float64)
    T = tmp[:3 * D * Bg].view(3, D, Bg)
    T[0], T[1], T[2] = mu.t(), (-0.5 * (L2PI + lv)).t(), torch.exp(-lv).t()
    S = tmp[3 * D * Bg:].view(Bl, Bg)
    zi = z[row0:row0 + Bl]
    ld = torch.empty(Bl, Bg, D, dtype=torch.float64)
    for d in range(D):
        diff = zi[:, d, None] - T[0, d][None, :]
        ld[:, :, d] = (T[1, d][None, :] - 0.5 * (diff * diff * T[2, d][None, :])) + W
    S.copy_(ld.sum(2))
    stride = _lib.rowstats_stride(D)
    rs = torch.full((Bl, stride), 7.0, dtype=torch.float64)
    rs[:, 1] = torch.logsumexp(S, 1)
    rs[:, 4:4 + D] = torch.logsumexp(ld, 1)
    rs[:, 2] = rs[:, 4:4 + D].sum(1)
    mi, li = mu[row0:row0 + Bl], lv[row0:row0 + Bl]
    rs[:, 0] = (-0.5 * L2PI - 0.5 * zi * zi).sum(1)
    rs[:, 3] = (-0.5 * (L2PI + li) - 0.5 * ((zi - mi) ** 2 * torch.exp(-li))).sum(1)
    return tmp, rs, W


def backward(z, mu, lv, row0, Bl, tmp, rs, W, coef):
    Bg, D = z.shape
    alpha, beta, gam = coef["alpha"], coef["beta"], coef["gamma"] * coef["anneal"]
    invB = 1.0 / Bg
    cP, cQ = (beta - alpha) * invB, (gam - beta) * invB
    T = tmp[:3 * D * Bg].view(3, D, Bg)
    S = tmp[3 * D * Bg:].view(Bl, Bg)
    zi = z[row0:row0 + Bl]
    P = torch.exp(S - rs[:, 1, None])
    dz = torch.empty(Bl, D, dtype=torch.float64)
    dmu, dlv = torch.empty(Bg, D, dtype=torch.float64), torch.empty(Bg, D, dtype=torch.float64)
    local = torch.zeros(Bg, dtype=torch.bool)
    local[row0:row0 + Bl] = True
    for d in range(D):
        diff = zi[:, d, None] - T[0, d][None, :]
        iv = T[2, d][None, :]
        r = diff * iv
        ld = (T[1, d][None, :] - 0.5 * (diff * diff * iv)) + W
        G = cP * P + cQ * torch.exp(ld - rs[:, 4 + d, None])
        rr = (zi[:, d] - mu[row0:row0 + Bl, d]) * torch.exp(-lv[row0:row0 + Bl, d])
        dz[:, d] = -(G * r).sum(1) - alpha * invB * rr + gam * invB * zi[:, d]
        gm, gl = (G * r).sum(0), (G * (-0.5 + 0.5 * r * diff)).sum(0)
        dj = z[:, d] - mu[:, d]
        rj = dj * torch.exp(-lv[:, d])
        gm = gm + torch.where(local, alpha * invB * rj, torch.zeros_like(rj))
        gl = gl + torch.where(local, alpha * invB * (-0.5 + 0.5 * rj * dj), torch.zeros_like(rj))
        dmu[:, d], dlv[:, d] = gm, gl
    return dz, dmu, dlv


def main():
    torch.manual_seed(0)
    checked = 0
    for B, D, n_data, mss in ((37, 19, 5000, True), (12, 33, 737280, True), (9, 17, 100, False)):
        mu = torch.randn(B, D, dtype=torch.float64)
        lv = torch.randn(B, D, dtype=torch.float64) * 0.7 - 0.5
        z = mu + torch.exp(0.5 * lv) * torch.randn(B, D, dtype=torch.float64)
        lw = log_importance_weights(B, n_data).double()
        tmp, rs, W = forward(z, mu, lv, 0, B, lw, mss)
        ref = O.btcvae_log_densities(z, mu, lv, n_data, mss)
        for k in range(4):
            assert (rs[:, k] - ref[k]).abs().max() < 1e-9, (B, D, k)
        assert _lib.rowstats_stride(D) >= 4 + D and bool((rs[:, 4 + D:] == 7.0).all())
        coef = dict(alpha=1.0, beta=6.4, gamma=1.5, anneal=0.37)
        dz, dmu, dlv = backward(z, mu, lv, 0, B, tmp, rs, W, coef)
        zr, mr, lr = (t.clone().requires_grad_(True) for t in (z, mu, lv))
        mi, tc, dw = O.btcvae_terms(zr, mr, lr, n_data, mss)
        (coef["alpha"] * mi + coef["beta"] * tc + coef["anneal"] * coef["gamma"] * dw).backward()
        for got, want in ((dz, zr.grad), (dmu, mr.grad), (dlv, lr.grad)):
            assert (got - want).abs().max() < 1e-12 + 1e-9 * want.abs().max(), (B, D)
        # two row shards: every call owns its tmp (the joint log-densities of ITS rows); rows reproduce, column gradients add up
        h = B // 2
        ta, ra, Wa = forward(z, mu, lv, 0, h, lw, mss)
        tb, rb, Wb = forward(z, mu, lv, h, B - h, lw, mss)
        assert torch.equal(ra[:, :4 + D], rs[:h, :4 + D]) and torch.equal(rb[:, :4 + D], rs[h:, :4 + D])
        dza, dma, dla = backward(z, mu, lv, 0, h, ta, ra, Wa, coef)
        dzb, dmb, dlb = backward(z, mu, lv, h, B - h, tb, rb, Wb, coef)
        assert (torch.cat((dza, dzb)) - dz).abs().max() < 1e-12
        assert (dma + dmb - dmu).abs().max() < 1e-12 and (dla + dlb - dlv).abs().max() < 1e-12
        # wide packed / scalar layouts (loss_pack_body / loss_finalize_body): KL values behind the 32 fixed slots
        kl = torch.rand(D, dtype=torch.float64)
        packed = torch.zeros(_lib.npack(D), dtype=torch.float64)
        packed[_lib.WIDE_KL0:_lib.WIDE_KL0 + D] = kl
        packed[17:21] = rs[:, :4].sum(0)
        scal = torch.zeros(_lib.nscal(D), dtype=torch.float64)
        scal[_lib.kl0(D):_lib.kl0(D) + D] = packed[_lib.WIDE_KL0:_lib.WIDE_KL0 + D]
        assert _lib.kl0(D) == 32 and _lib.npack(D) == 32 + D and float(scal[32:].sum()) == float(kl.sum())
        mi_p, tc_p, dw_p = (packed[20] - packed[18]) / B, (packed[18] - packed[19]) / B, (packed[19] - packed[17]) / B
        assert abs(mi_p - mi.item()) < 1e-9 and abs(tc_p - tc.item()) < 1e-9 and abs(dw_p - dw.item()) < 1e-9
        checked += 1
    print("latent_wide_math: %d cases OK" % checked)
    return checked


N_CASES = main() if __name__ == "__main__" else 0
