"""No GPU needed: numpy emulation of the ADDRESSING of csrc/gemm_dma.hip (k_gdma forward / input-gradient forms, k_gdma_wg)
-- workgroup -> tile map, the per-lane LDS-DMA source of every transfer (swizzle, clamped rows, zero chunks beyond the
contraction / the matrix), the lane-linear LDS image of a slab, the operand read offsets with the permuted contraction index,
the v_mfma_f32_32x32x2_f32 lane layout and the epilogue's (row, column) of every accumulator register -- against numpy
matmul on awkward shapes.  It does not model the ring / waitcnt protocol (argued in the kernel's comments), only where bytes go.

    python tools/emu/gemm_dma_index_math.py
"""
import numpy as np


def tile_of(L, tiles_m, tiles_n):
    T = tiles_m * tiles_n
    xcd, slot, per, rem = L & 7, L >> 3, T >> 3, T & 7
    t = xcd * per + min(xcd, rem) + slot
    full = (tiles_n >> 3) * tiles_m * 8
    if t < full:
        nb, r = divmod(t, tiles_m * 8)
        return r >> 3, nb * 8 + (r & 7)
    w, r = tiles_n & 7, t - full
    return r // w, (tiles_n & ~7) + r % w


def mfma_32x32x2(acc, a_lane, b_lane):
    """acc[lane][e] += sum_k A[i][k] B[k][j]; a_lane[l] = A[l % 32][l // 32], b_lane[l] = B[l // 32][l % 32];
    acc[l][e] = C[(e & 3) + 8 (e >> 2) + 4 (l // 32)][l % 32]."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    for l in range(64):
        A[l % 32, l // 32] = a_lane[l]
        B[l // 32, l % 32] = b_lane[l]
    C = A @ B
    for l in range(64):
        for e in range(16):
            acc[l, e] += C[(e & 3) + 8 * (e >> 2) + 4 * (l // 32), l % 32]


def emu_gdma(a, b, M, N, Kc, TM, TN, KS, NWK, b_jfast):
    """a [M, Kc]; b [N, Kc] (forward form) or [Kc, N] (input-gradient form) -> C [M, N]."""
    lda, ldb = a.shape[1], b.shape[1]
    af, bf = a.ravel(), b.ravel()
    CPR, RPB, SWD = KS // 4, 64 // (KS // 4), 16 // (KS // 4)
    NBA, NBB = TM * KS // 256, TN * KS // 256
    P = (NBA + NBB) // 4
    NR = KS // 8
    NWN = TN // 32
    NWM = 4 // (NWN * NWK)
    NACC, NRW = TM // NWM // 32, NR // NWK
    JCH, JRPB = TN // 4, 256 // TN
    swz = lambda r: (r // SWD) & (CPR - 1)
    tiles_m, tiles_n = (M + TM - 1) // TM, (N + TN - 1) // TN
    nslab = (Kc + KS - 1) // KS
    klast = (nslab - 1) * KS
    C = np.full((M, N), np.nan)
    seen = set()
    for L in range(tiles_m * tiles_n):
        tm, tn = tile_of(L, tiles_m, tiles_n)
        assert (tm, tn) not in seen and tm < tiles_m and tn < tiles_n
        seen.add((tm, tn))
        m0, n0 = tm * TM, tn * TN
        acc = np.zeros((4, NACC, 64, 16))
        for slab in range(nslab):
            lds = np.full((TM + TN) * KS, np.nan)
            for lw in range(4):
                for p in range(P):
                    g = lw + 4 * p
                    for lane in range(64):
                        if g < NBA or not b_jfast:
                            isA = g < NBA
                            blk = g if isA else g - NBA
                            r, s = blk * RPB + lane // CPR, lane % CPR
                            q = s ^ swz(r)
                            gr = min(m0 + r, M - 1) if isA else min(n0 + r, N - 1)
                            base = (gr * lda if isA else gr * ldb) + 4 * q
                            ok = klast + 4 * q < Kc
                            src, arr = base + slab * KS, (af if isA else bf)
                        else:
                            blk = g - NBA
                            kk, col = blk * JRPB + lane // JCH, n0 + 4 * (lane % JCH)
                            col_ok = col < N
                            ok = klast + kk < Kc
                            src, arr = (kk * ldb + col + slab * KS * ldb, bf) if col_ok else (None, None)
                        dst = g * 256 + lane * 4
                        if src is None or (slab == nslab - 1 and not ok):
                            lds[dst:dst + 4] = 0.0
                        else:
                            assert src + 4 <= arr.size, "out-of-bounds source"
                            lds[dst:dst + 4] = arr[src:src + 4]
            assert not np.isnan(lds).any(), "every LDS byte of the stage is written by exactly the transfers"
            for wv in range(4):
                wj, wi, wk = wv % NWN, (wv // NWN) % NWM, wv // (NWN * NWM)
                for j in range(NRW):
                    rr = wk * NRW + j
                    for u in range(4):
                        for t in range(NACC):
                            al, bl = np.zeros(64), np.zeros(64)
                            for lane in range(64):
                                i, h = lane & 31, lane >> 5
                                offr = i * KS + (((2 * rr + h) ^ swz(i)) << 2)
                                al[lane] = lds[wi * (TM // NWM) * KS + t * 32 * KS + offr + u]
                                if b_jfast:
                                    bl[lane] = lds[TM * KS + (8 * wk * NRW + 4 * h) * TN + wj * 32 + i + (8 * j + u) * TN]
                                else:
                                    bl[lane] = lds[TM * KS + wj * 32 * KS + offr + u]
                            mfma_32x32x2(acc[wv, t], bl, al)       # transposed product: A operand = weights
        if NWK > 1:                                               # the waves hold partial sums of the same block
            tot = (acc[0] + acc[1]) + (acc[2] + acc[3])
        for wv in range(4):
            wj, wi, wk = wv % NWN, (wv // NWN) % NWM, wv // (NWN * NWM)
            for t in range(NACC):
                for lane in range(64):
                    i, h = lane & 31, lane >> 5
                    row = m0 + wi * (TM // NWM) + t * 32 + i
                    for g in (range(4) if NWK == 1 else [wk]):
                        for q in range(4):
                            col = n0 + wj * 32 + 4 * h + 8 * g + q
                            if row < M and col < N:
                                assert np.isnan(C[row, col])
                                C[row, col] = acc[wv, t, lane, 4 * g + q] if NWK == 1 else tot[t, lane, 4 * g + q]
    return C


def emu_wg(dy, x, M, N, K, KS):
    dyf, xf = dy.ravel(), x.ravel()
    NB = KS // 4
    P = 2 * NB // 4
    tiles_n, tiles_k = (N + 63) // 64, (K + 63) // 64
    nslab = (M + KS - 1) // KS
    mlast = (nslab - 1) * KS
    dw = np.full((N, K), np.nan)
    db = np.full(N, np.nan)
    for L in range(tiles_n * tiles_k):
        tnn, tk = tile_of(L, tiles_n, tiles_k)
        n0, k0 = tnn * 64, tk * 64
        acc = np.zeros((4, 2, 2, 64, 16))
        rs = np.zeros((4, 2, 64))
        for slab in range(nslab):
            lds = np.full(2 * KS * 64, np.nan)
            for wv in range(4):
                for p in range(P):
                    g = wv + 4 * p
                    isA = g < NB
                    blk = g if isA else g - NB
                    for lane in range(64):
                        mm = blk * 4 + (lane >> 4)
                        col = (n0 if isA else k0) + 4 * (lane & 15)
                        col_ok = col < (N if isA else K)
                        ld = N if isA else K
                        ok = mlast + mm < M
                        dst = g * 256 + lane * 4
                        if not col_ok or (slab == nslab - 1 and not ok):
                            lds[dst:dst + 4] = 0.0
                        else:
                            src = mm * ld + col + slab * KS * ld
                            arr = dyf if isA else xf
                            assert src + 4 <= arr.size
                            lds[dst:dst + 4] = arr[src:src + 4]
            for wv in range(4):
                for t in range(KS // 8):
                    al, bl = np.zeros((2, 64)), np.zeros((2, 64))
                    for lane in range(64):
                        i, h = lane & 31, lane >> 5
                        a_off = (16 * wv + h) * 64 + 2 * i + t * 128
                        al[:, lane] = lds[a_off:a_off + 2]
                        bl[:, lane] = lds[KS * 64 + a_off:KS * 64 + a_off + 2]
                    assert not (np.isnan(al).any() or np.isnan(bl).any())
                    for q in range(2):
                        for q2 in range(2):
                            mfma_32x32x2(acc[wv, q, q2], al[q], bl[q2])
                        rs[wv, q] += al[q]
        tot = (acc[0] + acc[1]) + (acc[2] + acc[3])
        rst = (rs[0] + rs[1]) + (rs[2] + rs[3])
        for lane in range(64):
            i, h = lane & 31, lane >> 5
            for e in range(16):
                ie = (e & 3) + 8 * (e >> 2) + 4 * h
                for q in range(2):
                    for q2 in range(2):
                        row, col = n0 + 2 * ie + q, k0 + 2 * i + q2
                        if row < N and col < K:
                            assert np.isnan(dw[row, col])
                            dw[row, col] = tot[q, q2, lane, e]
            if tk == 0 and h == 0:
                for q in range(2):
                    if n0 + 2 * i + q < N:
                        db[n0 + 2 * i + q] = rst[q, lane] + rst[q, lane + 32]
    return dw, db


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for (M, N, Kc, TM, TN, KS, NWK) in ((130, 136, 260, 128, 64, 32, 1), (70, 200, 264, 64, 64, 64, 1), (64, 128, 256, 64, 64, 64, 1),
                                        (129, 132, 292, 128, 64, 32, 1), (70, 132, 264, 32, 32, 64, 4), (33, 100, 320, 32, 32, 64, 4)):
        a = rng.standard_normal((M, Kc))
        w = rng.standard_normal((N, Kc))
        C = emu_gdma(a, w, M, N, Kc, TM, TN, KS, NWK, False)
        print("fwd   M=%d N=%d K=%d tile %dx%d KS=%d NWK=%d  max err %.2e" % (M, N, Kc, TM, TN, KS, NWK, np.abs(C - a @ w.T).max()))
        assert np.allclose(C, a @ w.T)
        # input-gradient form: contraction over the rows of w2 [Kc, N]: dx[M, N] = dy[M, Kc] w2[Kc, N]
        dy = rng.standard_normal((M, Kc))
        w2 = rng.standard_normal((Kc, N))
        C = emu_gdma(dy, w2, M, N, Kc, TM, TN, KS, NWK, True)
        print("dgrad M=%d N=%d K=%d tile %dx%d KS=%d NWK=%d  max err %.2e" % (M, N, Kc, TM, TN, KS, NWK, np.abs(C - dy @ w2).max()))
        assert np.allclose(C, dy @ w2)
    for (M, N, K, KS) in ((70, 136, 132, 64), (128, 128, 200, 64), (200, 132, 128, 64)):
        dy = rng.standard_normal((M, N))
        x = rng.standard_normal((M, K))
        dw, db = emu_wg(dy, x, M, N, K, KS)
        print("wgrad M=%d N=%d K=%d KS=%d  max err %.2e  db %.2e" % (M, N, K, KS, np.abs(dw - dy.T @ x).max(), np.abs(db - dy.sum(0)).max()))
        assert np.allclose(dw, dy.T @ x) and np.allclose(db, dy.sum(0))
    print("INDEX MATH OK")
