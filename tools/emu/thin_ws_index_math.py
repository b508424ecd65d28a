"""No GPU needed: numpy emulation of the ADDRESSING of csrc/conv_thin_ws.hip -- k_down_thin_ws (conv1 forward / convT3 input
gradient) and k_wgrad_thin_ws (the two thin weight gradients) -- against a direct numpy convolution:

  * workgroup -> (XCD, image lane, part of the image) and the images a workgroup visits;
  * the loaders' 1 KB LDS-DMA transfers: per-lane source offset, active-lane mask, lane-linear LDS destination; image rows outside
    the image = masked lanes over zero-initialised LDS; the zero zone behind each channel plane;
  * the compute waves' operand addresses (the two lanes that would read columns -1 / 64 read the zero zone), the
    v_mfma_f32_32x32x2_f32 (transposed product) and v_mfma_f32_16x16x4_f32 lane layouts;
  * k_down_thin_ws: the output stage's bank swizzle, the drainers' chunk -> (pixel, channel chunk) map, the bit-plane word;
  * k_wgrad_thin_ws: which D register of which lane is dw[cs][cb][kh][kw], the partial-buffer slot it lands in
    (k_wgrad_thin_reduce's layout), the bias partial sums.
It does not model the ring / barrier protocol (argued in the kernels' comments), only where every byte goes.

    python tools/emu/thin_ws_index_math.py
"""
import numpy as np

PLANE_D, ZZ = 896, 640          # ThinWsGeo
PLANE_W = 768                   # ThinWgGeo (zero zone = rows 10, 11 of the plane: floats 640..767, never written)


def conv_ref(x, w):
    """y[n, cs, sy, sx] = sum x[n, cb, 2 sy - 1 + kh, 2 sx - 1 + kw] w[cs, cb, kh, kw]  (k4 s2 p1, 64 -> 32)."""
    N, C = x.shape[:2]
    xp = np.zeros((N, C, 66, 66))
    xp[:, :, 1:65, 1:65] = x
    y = np.zeros((N, 32, 32, 32))
    for kh in range(4):
        for kw in range(4):
            patch = xp[:, :, kh:kh + 64:2, kw:kw + 64:2]                     # [N, C, 32, 32]
            y += np.einsum("ncyx,oc->noyx", patch, w[:, :, kh, kw])
    return y


def wg_map(b, grid):
    xcd, slot = b & 7, b >> 3
    return xcd + 8 * (slot >> 3), slot & 7, grid >> 3                       # first image, part, images per step


def load_big_tile(x, n, C, sy0, plane):
    """The loaders' transfers of the big tile into a zero-initialised stage: returns the stage (floats)."""
    st = np.zeros(C * plane + 256)
    for d in range(3 * C):
        c, q = divmod(d, 3)
        for lane in range(64):
            r, col = 4 * q + (lane >> 4), 4 * (lane & 15)
            by = 2 * sy0 - 1 + r
            if not (r < 10 and 0 <= by < 64):
                continue                                                      # masked lane: LDS keeps its zeros
            voff = (c * 64 + by) * 64 + col                                   # floats from the image's base
            src = x[n].reshape(-1)[voff:voff + 4]
            dst = c * plane + q * 256 + lane * 4                              # lane-linear LDS side
            st[dst:dst + 4] = src
    return st


def mfma_32x32x2(acc, a_lane, b_lane):
    """acc[l][e] += sum_k A[i][k] B[k][j]: a_lane[l] = A[l % 32][l // 32], b_lane[l] = B[l // 32][l % 32],
    acc[l][e] = D[(e & 3) + 8 (e >> 2) + 4 (l // 32)][l % 32]."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    for l in range(64):
        A[l % 32, l // 32] = a_lane[l]
        B[l // 32, l % 32] = b_lane[l]
    D = A @ B
    for l in range(64):
        for e in range(16):
            acc[l, e] += D[(e & 3) + 8 * (e >> 2) + 4 * (l // 32), l % 32]


def mfma_16x16x4(acc, a_lane, b_lane):
    """a_lane[l] = A[l % 16][l // 16], b_lane[l] = B[l // 16][l % 16], acc[l][r] = D[4 (l // 16) + r][l % 16]."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for l in range(64):
        A[l % 16, l // 16] = a_lane[l]
        B[l // 16, l % 16] = b_lane[l]
    D = A @ B
    for l in range(64):
        for r in range(4):
            acc[l, r] += D[4 * (l // 16) + r, l % 16]


def emu_down(x, w, bias, grid=64, mode=3, mask_bits=None):
    """k_down_thin_ws: returns (out NHWC [N, 32, 32, 32], bits [N * 1024]) as the drainers write them."""
    N, C = x.shape[:2]
    out = np.full((N, 32, 32, 32), np.nan)
    bits = np.zeros(N * 1024, dtype=np.uint64)
    wf = w.reshape(32, 16 * C)
    for b in range(grid):
        n0, part, ipi = wg_map(b, grid)
        sy0 = part * 4
        for n in range(n0, N, ipi):
            st = load_big_tile(x, n, C, sy0, PLANE_D)
            ostage = np.full(128 * 32, np.nan)
            owords = np.zeros(128, dtype=np.uint64)
            for wv in range(4):
                acc = np.zeros((64, 16))
                for kk in range(8 * C):
                    kwb, kh, cb = kk & 1, (kk >> 1) & 3, kk >> 3
                    a_l, b_l = np.zeros(64), np.zeros(64)
                    for lane in range(64):
                        i, h = lane & 31, lane >> 5
                        a_l[lane] = wf[i, 2 * kk + h]                          # A = weights: w[cs = i][k = 2 kk + h]
                        if kwb == 0:
                            base = ZZ if lane == 0 else (2 * wv) * 64 + 2 * i + h - 1
                        else:
                            base = ZZ if lane == 63 else (2 * wv) * 64 + 2 * i + h + 1
                        b_l[lane] = st[base + cb * PLANE_D + kh * 64]          # B = pixels
                    mfma_32x32x2(acc, a_l, b_l)
                # epilogue: register e = 4 g + q of lane (i, h) = channel 8 g + 4 h + q of pixel i
                for lane in range(64):
                    i, h = lane & 31, lane >> 5
                    word = 0
                    for g in range(4):
                        vals = []
                        for q in range(4):
                            ch = 8 * g + 4 * h + q
                            v = acc[lane, 4 * g + q] + (bias[ch] if mode != 2 else 0.0)
                            if mode == 3:
                                if v > 0:
                                    word |= 1 << ch
                                v = max(v, 0.0)
                            if mode == 2:
                                mw = int(mask_bits[(n * 32 + sy0 + wv) * 32 + i])
                                v = v if (mw >> ch) & 1 else 0.0
                            vals.append(v)
                        slot = (2 * g + h) ^ ((i >> 1) & 7)                    # the stage's bank swizzle
                        o = (wv * 32 + i) * 32 + slot * 4
                        assert np.isnan(ostage[o:o + 4]).all()
                        ostage[o:o + 4] = vals
                    owords[wv * 32 + i] |= np.uint64(word)                      # both halves OR into lane (i, 0)'s word
            assert not np.isnan(ostage).any()
            # drainers: chunk X = j * 128 + dl -> pixel X >> 3, LDS slot X & 7, channel chunk = slot ^ (pixel >> 1) & 7
            blk = out[n].reshape(-1)[sy0 * 32 * 32:]                            # the unit's 16 KB output block
            for X in range(1024):
                p, slot = X >> 3, X & 7
                c = slot ^ ((p >> 1) & 7)
                dst = p * 32 + c * 4
                assert np.isnan(blk[dst:dst + 4]).all()
                blk[dst:dst + 4] = ostage[X * 4:X * 4 + 4]
            bits[(n * 32 + sy0) * 32:(n * 32 + sy0) * 32 + 128] = owords
    return out, bits


def emu_wgrad(big, small, grid=64, bias_big=False):
    """k_wgrad_thin_ws: returns the partial buffers [grid, STRIDE] in k_wgrad_thin's layout."""
    N, C = big.shape[:2]
    NT32 = (16 * C + 31) // 32
    STRIDE = NT32 * 1024 + 32 + NT32 * 32
    ws = np.zeros((grid, STRIDE))
    for b in range(grid):
        n0, part, ipi = wg_map(b, grid)
        sy0 = part * 4
        acc = np.zeros((4, 2, C, 64, 4))                                        # [wave][mt][nt][lane][r]
        sumS = np.zeros((4, 64, 2)); sumB = np.zeros((4, C, 64))
        for n in range(n0, N, ipi):
            st_big = load_big_tile(big, n, C, sy0, PLANE_W)
            st_small = small[n, sy0:sy0 + 4].reshape(-1)                        # 128 pixels x 32 channels, contiguous: 16 transfers
            for wv in range(4):
                for t in range(8):
                    a = np.zeros((2, 64)); bv = np.zeros((C, 64))
                    for lane in range(64):
                        i16, kq = lane & 15, lane >> 4
                        kh, kw = i16 >> 2, i16 & 3
                        ao = (wv * 32 + 4 * t + kq) * 32 + 2 * i16
                        a[0, lane], a[1, lane] = st_small[ao], st_small[ao + 1]
                        bmid = (2 * wv + kh) * 64 + 2 * kq - 1 + kw
                        if t == 0:
                            base = ZZ if (kq == 0 and kw == 0) else bmid
                        elif t == 7:
                            base = ZZ if (kq == 3 and kw == 3) else bmid + 56
                        else:
                            base = bmid + 8 * t
                        for nt in range(C):
                            bv[nt, lane] = st_big[base + nt * PLANE_W]
                    for nt in range(C):
                        for mt in range(2):
                            mfma_16x16x4(acc[wv, mt, nt], a[mt], bv[nt])
                        sumB[wv, nt] += bv[nt]
                    sumS[wv, :, 0] += a[0]; sumS[wv, :, 1] += a[1]
        tot = acc.sum(0)                                                        # cross-wave sum
        for idx in range(NT32 * 1024):
            nt32, cs, j = idx >> 10, (idx >> 5) & 31, idx & 31
            nidx = nt32 * 32 + j
            if nidx < 16 * C:
                nt, tap, mt, row = nidx >> 4, nidx & 15, cs & 1, cs >> 1
                ws[b, idx] = tot[mt, nt, (row >> 2) * 16 + tap, row & 3]
        if not bias_big:
            for cs in range(32):
                i, q = cs >> 1, cs & 1
                ws[b, NT32 * 1024 + cs] = sum(sumS[wv, 16 * kq + i, q] for wv in range(4) for kq in range(4))
        else:
            for nidx in range(16 * C):
                nt, tap = nidx >> 4, nidx & 15
                ws[b, NT32 * 1024 + 32 + nidx] = sum(sumB[wv, nt, 16 * kq + tap] for wv in range(4) for kq in range(4))
    return ws


def reduce_ws(ws, C, bias_big):
    """k_wgrad_thin_reduce: dw[cs][cb * 16 + tap] and the bias gradient from the partial buffers."""
    NT32 = (16 * C + 31) // 32
    t = ws.sum(0)
    dw = np.zeros((32, 16 * C))
    for q in range(NT32 * 1024):
        nt, cs, nidx = q >> 10, (q >> 5) & 31, (q >> 10) * 32 + (q & 31)
        if nidx < 16 * C:
            dw[cs, nidx] = t[q]
    if not bias_big:
        return dw, t[NT32 * 1024:NT32 * 1024 + 32]
    fin = t[NT32 * 1024 + 32:]
    return dw, np.array([(fin[16 * cb + 5] + fin[16 * cb + 6]) + (fin[16 * cb + 9] + fin[16 * cb + 10]) for cb in range(C)])


def main():
    rng = np.random.default_rng(0)
    for C, N in ((3, 9), (1, 10)):
        x = rng.standard_normal((N, C, 64, 64))
        w = rng.standard_normal((32, C, 4, 4))
        bias = rng.standard_normal(32)
        ref = conv_ref(x, w) + bias[None, :, None, None]
        out, bits = emu_down(x, w, bias, mode=3)
        got = out.transpose(0, 3, 1, 2)
        assert np.allclose(got, np.maximum(ref, 0)), np.abs(got - np.maximum(ref, 0)).max()
        want_bits = ((ref.transpose(0, 2, 3, 1).reshape(-1, 32) > 0) * (1 << np.arange(32, dtype=np.uint64))).sum(1)
        assert (bits == want_bits.astype(np.uint64)).all()
        print("down  C=%d N=%d  forward + bit plane  max err %.2e" % (C, N, np.abs(got - np.maximum(ref, 0)).max()))
        # input-gradient form: masked by a given bit plane, no bias, no activation
        mb = rng.integers(0, 2 ** 32, size=N * 1024, dtype=np.uint64)
        out2, _ = emu_down(x, w, bias, mode=2, mask_bits=mb)
        mask = ((mb[:, None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)).reshape(N, 32, 32, 32).astype(bool)
        want = np.where(mask, conv_ref(x, w).transpose(0, 2, 3, 1), 0.0)
        assert np.allclose(out2, want), np.abs(out2 - want).max()
        print("down  C=%d N=%d  masked by a bit plane max err %.2e" % (C, N, np.abs(out2 - want).max()))
        # weight gradients: dw[cs][cb][kh][kw] = sum small[n][sy][sx][cs] big[n][cb][2 sy - 1 + kh][2 sx - 1 + kw]
        small = rng.standard_normal((N, 32, 32, 32))                            # NHWC
        xp = np.zeros((N, C, 66, 66)); xp[:, :, 1:65, 1:65] = x
        dw_ref = np.zeros((32, C, 4, 4))
        for kh in range(4):
            for kw in range(4):
                dw_ref[:, :, kh, kw] = np.einsum("nyxo,ncyx->oc", small, xp[:, :, kh:kh + 64:2, kw:kw + 64:2])
        for bias_big in (False, True):
            ws = emu_wgrad(x, small, bias_big=bias_big)
            dw, db = reduce_ws(ws, C, bias_big)
            assert np.allclose(dw.reshape(32, C, 4, 4), dw_ref), np.abs(dw.reshape(32, C, 4, 4) - dw_ref).max()
            db_ref = x.sum((0, 2, 3)) if bias_big else small.sum((0, 1, 2))
            assert np.allclose(db, db_ref), (db, db_ref)
            print("wgrad C=%d N=%d  bias from the %s side  max err %.2e  bias %.2e" % (
                C, N, "big" if bias_big else "small", np.abs(dw.reshape(32, C, 4, 4) - dw_ref).max(), np.abs(db - db_ref).max()))
    print("INDEX MATH OK")


if __name__ == "__main__":
    main()
