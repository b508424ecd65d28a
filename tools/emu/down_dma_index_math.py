"""Index-level emulation (numpy, no GPU) of k_down32dma (disentangling-vae_amd/csrc/conv_down_dma.hip): the loader lanes' LDS-DMA source mapping (swizzle and
zero halo on the source side), the compute lanes' LDS operand reads, the weight-fragment indices of the staged image and the transposed
16x16x4 product -- against torch's conv2d.  Run before spending a GPU visit on a change of the tile layout:
    python tools/emu/down_dma_index_math.py"""
import numpy as np
def swz(HS, r, cw):
    if HS == 16: return (cw >> 1) & 7
    if HS == 8: return ((cw >> 1) & 3) | (((r >> 1) & 1) << 2)
def run(HS, N, grid):
    rng = np.random.default_rng(0)
    HB = 2 * HS; R = 64 // HS if HS * HS >= 64 else HS; CW = HS + 1; BROWS = 2 * R + 2
    BIG_FLOATS = BROWS * 2 * CW * 32; NCHUNK = BIG_FLOATS // 4; UPI = HS * HS // 64
    big = rng.standard_normal((N, HB, HB, 32)).astype(np.float32)
    w = rng.standard_normal((32, 32, 4, 4)).astype(np.float32)     # [cs][cb][kh][kw]
    bigf = big.reshape(-1)
    n_units = N * HS * HS // 64
    grid = min(grid, n_units); grid -= grid % UPI
    out = np.zeros((N * HS * HS, 32), np.float64)
    # staged image wl[tap][kc/4][n][kc%4]: kc = cb, n = cs
    wl = np.zeros((16, 8, 32, 4), np.float32)
    for t in range(16):
        for kc in range(32):
            wl[t, kc // 4, :, kc % 4] = w[:, kc, t // 4, t % 4]
    wlf = wl.reshape(-1)
    for wg in range(grid):
        unit0 = wg
        n0, sy0 = unit0 // UPI, (unit0 % UPI) * R
        base = ((n0 * HB + 2 * sy0 - 1) * HB) * 32
        step = (grid // UPI) * HB * HB * 32
        # loader slots
        src = {}
        for lw in range(4):
            for k in range(11):
                for lane in range(64):
                    c = (k * 4 + lw) * 64 + lane
                    q, j = c >> 3, c & 7
                    cw, rp = q % CW, q // CW
                    par, r = rp & 1, rp >> 1
                    by, bx = 2 * sy0 - 1 + r, 2 * cw + par - 1
                    ok = c < NCHUNK and 0 <= by < HB and 0 <= bx < HB
                    src[c] = (base + (r * HB + bx) * 32 + ((j ^ swz(HS, r, cw)) << 2)) if ok else None
        it = 0
        unit = unit0
        while unit < n_units:
            tile = np.zeros(44 * 256, np.float32)
            for c, s in src.items():
                if s is not None:
                    a = s + it * step
                    tile[c * 4:c * 4 + 4] = bigf[a:a + 4]
            for wv in range(4):
                ch, ph = wv & 1, wv >> 1
                for lane in range(64):
                    i16, kq = lane & 15, lane >> 4
                    for mt in range(2):
                        p = 32 * ph + 16 * mt + i16
                        sy_l, sx = (p // HS) % R, p % HS
                        # this lane = pixel i16 (B operand, k-slot kq): contributes to D[m][n = i16] for all m: emulate the full
                        # product by accumulating per (pixel, cs) with the operands both lanes would hold
                        for t in range(16):
                            kh, kw = t >> 2, t & 3
                            tc = ((kh * 2 + (kw & 1)) * CW) * 32
                            for h in range(2):
                                s_, s2 = kw >> 1, (kh >> 1) if HS == 8 else 0
                                r = 2 * sy_l + 2 * s2; cw = sx + s_
                                vo = ((2 * sy_l * 2) * CW + cw) * 32 + (((4 * h + kq) ^ swz(HS, r, cw)) << 2)
                                P = tile[vo + tc: vo + tc + 4]                      # channels 16h + 4kq + j of the tap's input pixel
                                # expected
                                n_img = unit // UPI
                                sy = (unit % UPI) * R + sy_l
                                by, bx = 2 * sy - 1 + kh, 2 * sx - 1 + kw
                                exp = big[n_img, by, bx, 16 * h + 4 * kq:16 * h + 4 * kq + 4] if (0 <= by < HB and 0 <= bx < HB) else np.zeros(4, np.float32)
                                assert np.array_equal(P, exp), (HS, unit, wv, lane, mt, t, h)
                                # A operand lanes m = 0..15 (cs = 16 ch + m), same kq: W[t][h][j]
                                for m in range(16):
                                    Wv = wlf[(((t * 8 + 4 * h + kq) * 32 + 16 * ch + m) << 2):][:4]
                                    out[unit * 64 + p, 16 * ch + m] += float(np.dot(Wv.astype(np.float64), P.astype(np.float64)))
            unit += grid; it += 1
    # direct conv
    import torch, torch.nn.functional as F
    x = torch.from_numpy(big).permute(0, 3, 1, 2).double()
    ref = F.conv2d(x, torch.from_numpy(w).double(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, 32).numpy()
    err = np.abs(out - ref).max()
    print("HS", HS, "N", N, "grid", grid, "max err", err)
    assert err < 1e-9
run(16, 3, 8)
run(16, 2, 256)
run(8, 5, 3)
