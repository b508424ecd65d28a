"""-m gpu: the round-3 launches that shorten the step's latency chain, through the C-ABI:
  * dvae_stage_weights -- the pre-staged LDS weight images of the 32-channel conv layers and the k-chunked operand
    streams of the FC layers (exact re-layouts: compared bit for bit with the layout definition);
  * dvae_conv32_down / dvae_conv32_up -- the tuned conv kernels on pre-staged images == the same kernels on raw weights,
    bit for bit (only the prologue differs), at sizes where every persistent workgroup loops;
  * dvae_fc_chain_fwd / dvae_fc_chain_bwd -- the FC core in one launch per direction vs fp64 torch (rtol 1e-5 + 2e-6 of the
    tensor's scale) and vs the per-layer entry points it replaces in the training step.
Reference lines: encoders.py:73-87, vae.py:52-71, losses.py:452-480, decoders.py:71-80."""
import ctypes
import os
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib  # noqa


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _stage(conv=(), fc=(), coef=None, coef_vals=None, thin=None):
    """conv: [(w, img_down, img_up)], fc: [(w, img_fwd, img_bwd, N, K)], thin: (w, img_pairs, C) (device tensors / None)."""
    cd = (_lib.ConvImageDesc * max(1, len(conv)))()
    for d, (w, a, b) in zip(cd, conv):
        d.w, d.img_down, d.img_up = ptr(w), ptr(a), ptr(b)
    fd = (_lib.FcImageDesc * max(1, len(fc)))()
    for d, (w, a, b, N, K) in zip(fd, fc):
        d.w, d.img_fwd, d.img_bwd, d.N, d.K = ptr(w), ptr(a), ptr(b), N, K
    cv = None
    if coef_vals is not None:
        arr = (ctypes.c_float * 8)(*coef_vals)
        cv = ctypes.addressof(arr)
    ta = None
    if thin is not None:
        td = _lib.ThinImageDesc()
        td.w, td.img_pairs, td.C = ptr(thin[0]), ptr(thin[1]), thin[2]
        ta = ctypes.addressof(td)
    call("dvae_stage_weights", ctypes.addressof(cd), len(conv), ctypes.addressof(fd), len(fc), ta, ptr(coef), cv, stream())
    torch.cuda.synchronize()


def _conv_image(w, down):
    """wl[tap][kc/4][n][kc%4] of conv_mfma_common.h from w[cs][cb][4][4]: down kc = cb, n = cs; up kc = cs, n = cb."""
    t = w.reshape(32, 32, 16)                       # [cs][cb][tap]
    t = t.permute(2, 1, 0) if down else t.permute(2, 0, 1)     # [tap][kc][n]
    return t.reshape(16, 8, 4, 32).permute(0, 1, 3, 2).contiguous().reshape(-1)   # [tap][kc4][n][r]


def _fc_images(w):
    N, K = w.shape
    kp = (K + 3) // 4 * 4
    wk = torch.zeros(N, kp)
    wk[:, :K] = w
    fwd = wk.reshape(N, kp // 4, 4).permute(1, 0, 2).contiguous().reshape(-1)       # [K/4][N][4]
    npad = (N + 3) // 4 * 4
    wn = torch.zeros(npad, K)
    wn[:N] = w
    bwd = wn.reshape(npad // 4, 4, K).permute(0, 2, 1).contiguous().reshape(-1)     # [N/4][K][4]
    return fwd, bwd


def test_stage_weights_layouts_and_coefficients():
    w1, w2 = _rand(32, 32, 4, 4, seed=1), _rand(32, 32, 4, 4, seed=2)
    f = lambda n: torch.full((n,), 7.0, device=DEV)
    imgs = [(dev(w1), f(16384), f(16384)), (dev(w2), f(16384), None)]
    fcs = []
    shapes = [(256, 512), (20, 256), (256, 10), (512, 256), (2, 256), (256, 1)]
    ws = [_rand(N, K, seed=10 + i) for i, (N, K) in enumerate(shapes)]
    for w, (N, K) in zip(ws, shapes):
        fcs.append((dev(w), f((K + 3) // 4 * N * 4), f((N + 3) // 4 * K * 4), N, K))
    coef = torch.zeros(_lib.NCOEF, device=DEV)
    _stage(imgs, fcs, coef, [0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5])
    assert torch.equal(imgs[0][1].cpu(), _conv_image(w1, True))
    assert torch.equal(imgs[0][2].cpu(), _conv_image(w1, False))
    assert torch.equal(imgs[1][1].cpu(), _conv_image(w2, True))
    for w, (wd, a, b, N, K) in zip(ws, fcs):
        fwd, bwd = _fc_images(w)
        assert torch.equal(a.cpu(), fwd), (N, K)
        assert torch.equal(b.cpu(), bwd), (N, K)
    assert coef.cpu().tolist() == [0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5]
    # coefficients alone (no image): one tiny launch
    _stage((), (), coef, [1.0] * 8)
    assert coef.cpu().tolist() == [1.0] * 8


@pytest.mark.parametrize("N,Hs", [(3, 16), (70, 16), (1030, 16), (5, 8), (1100, 8), (3, 4), (70, 4), (4099, 4)])
def test_conv32_on_staged_images_is_bit_identical_to_the_raw_weight_path(N, Hs):
    """Conv2d forward / ConvT dgrad (down) and ConvT forward / Conv dgrad (up) on pre-staged images vs the raw-weight entry
    points: the same kernels, only the weight prologue differs."""
    C, Hb = 32, 2 * Hs
    w = _rand(C, C, 4, 4, seed=2, scale=0.2)
    b = _rand(C, seed=3, scale=0.1)
    wd, bd = dev(w), dev(b)
    img_d, img_u = torch.empty(16384, device=DEV), torch.empty(16384, device=DEV)
    _stage([(wd, img_d, img_u)])
    big = nhwc(_rand(N, C, Hb, Hb, seed=1))
    small_act = torch.relu(_rand(N, C, Hs, Hs, seed=5))
    layouts = [_lib.NHWC] + ([_lib.NCHW] if Hs == 4 else [])
    for lay in layouts:
        shape = (N, Hs, Hs, C) if lay == _lib.NHWC else (N, C, Hs, Hs)
        # down, forward flavour: bias + ReLU, no mask
        y0, y1 = torch.empty(shape, device=DEV), torch.empty(shape, device=DEV)
        call("dvae_conv4s2_fwd", ptr(big), _lib.NHWC, ptr(wd), ptr(bd), ptr(y0), lay, N, C, Hb, Hb, C, _lib.ACT_RELU, stream())
        call("dvae_conv32_down", ptr(big), ptr(img_d), ptr(bd), None, ptr(y1), lay, N, Hs, _lib.ACT_RELU, stream())
        assert torch.equal(y0, y1), "down fwd layout %d" % lay
        # down, dgrad flavour (ConvT dgrad): mask, no bias
        mask = nhwc(small_act) if lay == _lib.NHWC else dev(small_act)
        call("dvae_convT4s2_dgrad", ptr(big), _lib.NHWC, ptr(wd), ptr(mask), ptr(y0), lay, N, C, Hs, Hs, C, stream())
        call("dvae_conv32_down", ptr(big), ptr(img_d), None, ptr(mask), ptr(y1), lay, N, Hs, _lib.ACT_NONE, stream())
        assert torch.equal(y0, y1), "down dgrad layout %d" % lay
        # up, forward flavour (ConvT forward)
        small = nhwc(small_act) if lay == _lib.NHWC else dev(small_act)
        o0, o1 = torch.empty(N, Hb, Hb, C, device=DEV), torch.empty(N, Hb, Hb, C, device=DEV)
        call("dvae_convT4s2_fwd", ptr(small), lay, ptr(wd), ptr(bd), ptr(o0), _lib.NHWC, N, C, Hs, Hs, C, _lib.ACT_RELU, stream())
        call("dvae_conv32_up", ptr(small), lay, ptr(img_u), ptr(bd), None, ptr(o1), N, Hs, _lib.ACT_RELU, stream())
        assert torch.equal(o0, o1), "up fwd layout %d" % lay
        # up, dgrad flavour (Conv dgrad): mask = the big-side activation
        bmask = nhwc(torch.relu(_rand(N, C, Hb, Hb, seed=7)))
        call("dvae_conv4s2_dgrad", ptr(small), lay, ptr(wd), ptr(bmask), ptr(o0), _lib.NHWC, N, C, Hb, Hb, C, stream())
        call("dvae_conv32_up", ptr(small), lay, ptr(img_u), None, ptr(bmask), ptr(o1), N, Hs, _lib.ACT_NONE, stream())
        assert torch.equal(o0, o1), "up dgrad layout %d" % lay


def _fc_params(D, seed=0):
    shapes = {"e1": (256, 512), "e2": (256, 256), "ml": (2 * D, 256), "d1": (256, D), "d2": (256, 256), "d3": (512, 256)}
    W = {k: _rand(N, K, seed=seed + i, scale=math.sqrt(6.0 / K)) for i, (k, (N, K)) in enumerate(shapes.items())}
    Bv = {k: _rand(N, seed=seed + 20 + i, scale=1.0 / math.sqrt(K)) for i, (k, (N, K)) in enumerate(shapes.items())}
    return shapes, W, Bv


def _fc_stage(shapes, W):
    f = lambda n: torch.empty(n, device=DEV)
    ent = {}
    for k, (N, K) in shapes.items():
        ent[k] = (dev(W[k]), f((K + 3) // 4 * N * 4), f((N + 3) // 4 * K * 4), N, K)
    _stage((), list(ent.values()))
    return ent


@pytest.mark.parametrize("n_enc,n_kl,n_dec,D,noise", [(1, 1, 1, 10, True), (8, 8, 8, 10, True), (13, 13, 13, 10, False),
                                                     (128, 128, 128, 10, True), (1030, 1030, 1030, 10, True),
                                                     (262, 131, 131, 10, True), (40, 20, 20, 1, True), (77, 77, 77, 16, True),
                                                     (2048, 1024, 1024, 10, True), (50, 50, 0, 6, True)])
def test_fc_chain_forward(n_enc, n_kl, n_dec, D, noise):
    shapes, W, Bv = _fc_params(D, seed=3)
    ent = _fc_stage(shapes, W)
    a = torch.relu(_rand(n_enc, 512, seed=1))
    eps = torch.randn(n_enc, D, generator=torch.Generator().manual_seed(2)) if noise else None
    f = lambda *s: torch.full(s, 7.0, device=DEV)
    out = dict(h1=f(n_enc, 256), h2=f(n_enc, 256), ml=f(n_enc, 2 * D), mu=f(n_enc, D), logvar=f(n_enc, D), z=f(n_enc, D),
               d1=f(max(n_dec, 1), 256), d2=f(max(n_dec, 1), 256), d3=f(max(n_dec, 1), 512))
    kl = torch.full((_lib.KL_FLOATS,), 7.0, device=DEV)
    bd = {k: dev(v) for k, v in Bv.items()}
    ad, ed = dev(a), (dev(eps) if noise else None)
    st, addr = _lib.struct_of(_lib.FcChainFwdArgs, a_flat=ptr(ad), eps=ptr(ed), kl_part=ptr(kl) + 64,
                              n_enc=n_enc, n_kl=n_kl, n_dec=n_dec, D=D,
                              **{"w_" + k: ptr(ent[k][1]) for k in shapes}, **{"b_" + k: ptr(bd[k]) for k in shapes},
                              **{k: ptr(v) for k, v in out.items()})
    call("dvae_fc_chain_fwd", addr, stream())
    # fp64 reference (encoders.py:81-87, vae.py:66-68, decoders.py:71-73)
    Wd = {k: v.double() for k, v in W.items()}
    Bd = {k: v.double() for k, v in Bv.items()}
    h1 = torch.relu(F.linear(a.double(), Wd["e1"], Bd["e1"]))
    h2 = torch.relu(F.linear(h1, Wd["e2"], Bd["e2"]))
    ml = F.linear(h2, Wd["ml"], Bd["ml"])
    mu, lv = ml.view(n_enc, D, 2).unbind(-1)
    z = mu + torch.exp(0.5 * lv) * eps.double() if noise else mu
    tol = dict(rtol=1e-5, atol_rel=2e-6)
    check(out["h1"], h1, what="chain h1", **tol)
    check(out["h2"], h2, what="chain h2", **tol)
    check(out["ml"], ml, what="chain ml", **tol)
    check(out["mu"], mu, what="chain mu", **tol)
    check(out["logvar"], lv, what="chain logvar", **tol)
    check(out["z"], z, what="chain z", **tol)
    klref = (0.5 * (-1 - lv + mu * mu + torch.exp(lv)))[:n_kl].sum(0)
    rows = _lib.fc_chain_rows(n_enc)                 # 4 rows per workgroup up to 1024 batch rows, 8 above (dvae_fc_chain_rows)
    assert rows == (4 if n_enc <= 1024 else 8)
    nblk = (n_enc + rows - 1) // rows
    parts = kl[16:16 + nblk * 16].view(nblk, 16).cpu().double()
    check(parts.sum(0)[:D], klref, rtol=1e-5, atol_rel=2e-6, what="chain KL partial blocks")
    assert torch.all(parts[:, D:] == 0)
    if n_dec:
        d1 = torch.relu(F.linear(z[:n_dec], Wd["d1"], Bd["d1"]))
        d2 = torch.relu(F.linear(d1, Wd["d2"], Bd["d2"]))
        d3 = torch.relu(F.linear(d2, Wd["d3"], Bd["d3"]))
        check(out["d1"], d1, what="chain d1", **tol)
        check(out["d2"], d2, what="chain d2", **tol)
        check(out["d3"], d3, what="chain d3", **tol)
    else:
        assert torch.all(out["d3"] == 7.0)
    # the finishing launch and the one-launch loss epilogue add the partial blocks in the same order (bit-identical)
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / max(n_kl, 1); coef[_lib.C_BETA] = 4.0; coef[_lib.C_ANNEAL] = 1.0
    coefd = dev(coef)
    packed, scal = torch.zeros(_lib.NPACK, device=DEV), torch.zeros(_lib.NSCAL, device=DEV)
    partials = torch.zeros(_lib.REC_NPART, device=DEV)
    call("dvae_loss_epilogue", _lib.LOSS_BETAH, ptr(partials), ptr(kl), nblk, D, None, 0, None, max(n_kl, 1), ptr(coefd), ptr(packed),
         ptr(scal), stream())
    call("dvae_kl_finish", ptr(kl), nblk, ptr(coefd), D, stream())
    assert torch.equal(kl[:D].cpu(), packed[1:1 + D].cpu())
    check(kl[:D], klref / max(n_kl, 1), rtol=1e-5, atol_rel=2e-6, what="chain KL finished")
    # vs the per-layer entry points the training step used before (same tolerance, now on fp32 against fp32)
    h1p, mlp = torch.empty(n_enc, 256, device=DEV), torch.empty(n_enc, 2 * D, device=DEV)
    h2p = torch.empty(n_enc, 256, device=DEV)
    call("dvae_linear_fwd", ptr(ad), ptr(ent["e1"][0]), ptr(bd["e1"]), ptr(h1p), n_enc, 512, 256, _lib.ACT_RELU, None, stream())
    call("dvae_linear_fwd", ptr(h1p), ptr(ent["e2"][0]), ptr(bd["e2"]), ptr(h2p), n_enc, 256, 256, _lib.ACT_RELU, None, stream())
    call("dvae_linear_fwd", ptr(h2p), ptr(ent["ml"][0]), ptr(bd["ml"]), ptr(mlp), n_enc, 256, 2 * D, _lib.ACT_NONE, None, stream())
    check(out["ml"], mlp.cpu(), rtol=2e-5, atol_rel=4e-6, what="chain ml vs per-layer kernels")


@pytest.mark.parametrize("n,D,noise,extra", [(1, 10, True, 0), (8, 10, True, 1), (13, 10, False, 0), (128, 10, True, 2),
                                            (1030, 10, True, 1), (77, 16, True, 2), (40, 1, True, 1), (1024, 10, True, 2)])
def test_fc_chain_backward(n, D, noise, extra):
    """extra: 0 = no external latent gradients, 1 = the btcvae set (dz2, dmu_x, dlv_x), 2 = the factor set (dz2, dz3)."""
    shapes, W, Bv = _fc_params(D, seed=5)
    ent = _fc_stage(shapes, W)
    gd3 = _rand(n, 512, seed=1)
    acts = {k: torch.relu(_rand(n, w, seed=10 + i)) for i, (k, w) in enumerate(
        [("d2", 256), ("d1", 256), ("h2", 256), ("h1", 256), ("a_flat", 512)])}
    mu, lv = _rand(n, D, seed=20), _rand(n, D, seed=21, scale=0.7)
    eps = torch.randn(n, D, generator=torch.Generator().manual_seed(3)) if noise else None
    dz2 = _rand(n, D, seed=22) if extra else None
    dz3 = _rand(n, D, seed=23) if extra == 2 else None
    dmu_x = _rand(n, D, seed=24) if extra == 1 else None
    dlv_x = _rand(n, D, seed=25) if extra == 1 else None
    scal = torch.zeros(_lib.NSCAL); scal[_lib.S_KLW] = 1.7
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / n
    f = lambda *s: torch.full(s, 7.0, device=DEV)
    out = dict(gd2=f(n, 256), gd1=f(n, 256), dz=f(n, D), dml=f(n, 2 * D), gh2=f(n, 256), gh1=f(n, 256), ga_flat=f(n, 512))
    dv = lambda t: None if t is None else dev(t)
    ins = dict(gd3=dev(gd3), mu=dev(mu), logvar=dev(lv), eps=dv(eps), dz2=dv(dz2), dz3=dv(dz3), dmu_x=dv(dmu_x), dlv_x=dv(dlv_x),
               scal=dev(scal), coef=dev(coef), **{k: dev(v) for k, v in acts.items()})
    st, addr = _lib.struct_of(_lib.FcChainBwdArgs, n=n, D=D, **{"w_" + k: ptr(ent[k][2]) for k in shapes},
                              **{k: ptr(v) for k, v in ins.items()}, **{k: ptr(v) for k, v in out.items()})
    call("dvae_fc_chain_bwd", addr, stream())
    Wd = {k: v.double() for k, v in W.items()}
    g = gd3.double()
    gd2 = (g @ Wd["d3"]) * (acts["d2"] > 0)
    gd1 = (gd2 @ Wd["d2"]) * (acts["d1"] > 0)
    dz = gd1 @ Wd["d1"]
    gz = dz.clone()
    if dz2 is not None:
        gz = gz + dz2.double()
    if dz3 is not None:
        gz = gz + dz3.double()
    klw = 1.7 / n
    m, l = mu.double(), lv.double()
    dm = gz + klw * m
    dl = klw * 0.5 * (torch.exp(l) - 1)
    if noise:
        dl = dl + gz * eps.double() * 0.5 * torch.exp(0.5 * l)
    if dmu_x is not None:
        dm, dl = dm + dmu_x.double(), dl + dlv_x.double()
    dml = torch.stack((dm, dl), dim=-1).reshape(n, 2 * D)
    gh2 = (dml @ Wd["ml"]) * (acts["h2"] > 0)
    gh1 = (gh2 @ Wd["e2"]) * (acts["h1"] > 0)
    ga = (gh1 @ Wd["e1"]) * (acts["a_flat"] > 0)
    tol = dict(rtol=1e-5, atol_rel=2e-6)
    for k, ref in (("gd2", gd2), ("gd1", gd1), ("dz", dz), ("dml", dml), ("gh2", gh2), ("gh1", gh1), ("ga_flat", ga)):
        check(out[k], ref, what="chain bwd " + k, **tol)
    # vs the per-layer entry points (dvae_linear_dgrad with the fused ReLU mask) on the first two layers
    gd2p, gd1p = torch.empty(n, 256, device=DEV), torch.empty(n, 256, device=DEV)
    call("dvae_linear_dgrad", ptr(ins["gd3"]), ptr(ent["d3"][0]), ptr(ins["d2"]), _lib.ACT_RELU, ptr(gd2p), n, 256, 512, None, stream())
    call("dvae_linear_dgrad", ptr(gd2p), ptr(ent["d2"][0]), ptr(ins["d1"]), _lib.ACT_RELU, ptr(gd1p), n, 256, 256, None, stream())
    check(out["gd1"], gd1p.cpu(), rtol=2e-5, atol_rel=4e-6, what="chain gd1 vs per-layer kernels")


def _thin_records(w):
    """Per contracted channel the operand-pair record of k_up_thin_pk (conv_thin.hip) from the ConvTranspose2d weight
    w[32][C][4][4]: C = 3: [2 (4 t + cls) + {0,1}] = w[cs][{0,1}][kh][kw], cls = 2 py + px, t = 2 ty + tx, kh = 1 - py + 2 ty,
    kw = 1 - px + 2 tx; then plane C-1 in the tap order below (pairs that share their source pixel, then the four corners)."""
    C = w.shape[1]
    plane = [5, 6, 9, 10, 13, 14, 1, 2, 7, 11, 4, 8, 0, 3, 12, 15]
    wt = w.reshape(32, C, 16)
    rec = []
    for cs in range(32):
        r = []
        if C == 3:
            for t in range(4):
                ty, tx = t >> 1, t & 1
                for cls in range(4):
                    py, px = cls >> 1, cls & 1
                    tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx)
                    r += [wt[cs, 0, tap], wt[cs, 1, tap]]
        r += [wt[cs, C - 1, tap] for tap in plane]
        rec.append(torch.stack(r))
    out = torch.stack(rec).reshape(-1)
    if C == 3:
        # behind the records: the A-operand image of the matrix-core kernel (k_up_thin_mm, conv_up_thin_mm.hip): float
        # (tap * 8 + i) * 64 + lane = w[cs = 8 (lane / 16) + i][c][kh = 2 - 2a + dy][kw = 2 - 2b + dx] for output row
        # m = lane % 16 = 4 c + 2 dy + dx (rows 12..15: zero), tap = 2 a + b: the 2 x 2 input window -> 2 x 2 output block form
        img = torch.zeros(32, 64)
        for mf in range(32):
            a, b, i = mf >> 4, (mf >> 3) & 1, mf & 7
            for ln in range(64):
                m, cs = ln & 15, 8 * (ln >> 4) + i
                if m < 12:
                    c, dy, dx = m >> 2, (m >> 1) & 1, m & 1
                    img[mf, ln] = w[cs, c, 2 - 2 * a + dy, 2 - 2 * b + dx]
        out = torch.cat((out, img.reshape(-1)))
    return out


@pytest.mark.parametrize("C", [1, 3])
@pytest.mark.parametrize("N", [1, 3, 86, 300])
def test_convT3_forward_on_staged_pair_records(N, C):
    """The last decoder layer (decoders.py:82) as the packed-FMA kernel on the records of dvae_stage_weights: the records bit
    for bit; reconstruction, likelihood partial sums (losses.py:394-449) and dL/dlogit vs the raw-weight entry points
    (fp32 and uint8 targets; rtol 1e-5 + 2e-6 of the scale: the summation order of the accumulators differs) and vs fp64."""
    w = _rand(32, C, 4, 4, seed=2, scale=0.2)
    b = _rand(C, seed=3, scale=0.1)
    wd, bd = dev(w), dev(b)
    pairs = torch.full((32 * _lib.thin_pair_floats(C),), 7.0, device=DEV)
    _stage(thin=(wd, pairs, C))
    assert torch.equal(pairs.cpu(), _thin_records(w))
    x = nhwc(torch.relu(_rand(N, 32, 32, 32, seed=1)))
    tgt8 = torch.randint(0, 256, (N, C, 64, 64), dtype=torch.uint8, generator=torch.Generator().manual_seed(4))
    tgt = tgt8.float() / 255.0
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / N
    coefd = dev(coef)
    f = lambda: torch.empty(N, C, 64, 64, device=DEV)
    tol = dict(rtol=1e-5, atol_rel=2e-6)
    # plain forward
    r0, r1 = f(), f()
    call("dvae_convT4s2_fwd", ptr(x), _lib.NHWC, ptr(wd), ptr(bd), ptr(r0), _lib.NCHW, N, 32, 32, 32, C, _lib.ACT_SIGMOID, stream())
    call("dvae_convT3_fwd_staged", ptr(x), ptr(pairs), ptr(bd), None, 0, ptr(r1), None, 0, None, None, N, C, stream())
    check(r1, r0.cpu(), what="staged convT3 fwd vs raw", **tol)
    xr = x.cpu().permute(0, 3, 1, 2).double()
    ref = torch.sigmoid(F.conv_transpose2d(xr, w.double(), b.double(), stride=2, padding=1))
    check(r1, ref, what="staged convT3 fwd vs fp64", **tol)
    # fused likelihood, fp32 and uint8 targets
    for dist in (0, 1, 2):
        g0, g1, g2, r2 = f(), f(), f(), f()
        p0, p1, p2 = (torch.full((_lib.REC_NPART,), 7.0, device=DEV) for _ in range(3))
        td = dev(tgt)
        call("dvae_convT4s2_sigmoid_recon_fwd", ptr(x), _lib.NHWC, ptr(wd), ptr(bd), ptr(td), ptr(r0), ptr(g0), dist, ptr(coefd),
             ptr(p0), N, 32, 32, 32, C, stream())
        call("dvae_convT3_fwd_staged", ptr(x), ptr(pairs), ptr(bd), ptr(td), 0, ptr(r1), ptr(g1), dist, ptr(coefd), ptr(p1), N, C,
             stream())
        t8 = tgt8.to(DEV)
        call("dvae_convT3_fwd_staged", ptr(x), ptr(pairs), ptr(bd), ptr(t8), 1, ptr(r2), ptr(g2), dist, ptr(coefd), ptr(p2), N, C,
             stream())
        check(r1, r0.cpu(), what="fused recon dist %d" % dist, **tol)
        check(g1, g0.cpu(), rtol=1e-4, atol_rel=2e-6, what="fused dL/dlogit dist %d" % dist)
        check(p1.sum(), p0.sum().cpu(), rtol=1e-5, what="fused loss sum dist %d" % dist)
        if C == 1:
            assert torch.equal(r2, r1) and torch.equal(g2, g1) and torch.equal(p2, p1), "uint8 target == ToTensor(target), dist %d" % dist
        else:       # fp32 targets: the matrix-core kernel; uint8 targets: the packed-FMA kernel (another summation order)
            check(r2, r1.cpu(), what="uint8-target recon dist %d" % dist, **tol)
            check(g2, g1.cpu(), rtol=1e-4, atol_rel=2e-6, what="uint8-target dL/dlogit dist %d" % dist)
            check(p2.sum(), p1.sum().cpu(), rtol=1e-5, what="uint8-target loss sum dist %d" % dist)
        # against fp64: likelihood sum and dL/dlogit of the fp32-target launch
        xr64 = x.cpu().permute(0, 3, 1, 2).double()
        logit = F.conv_transpose2d(xr64, w.double(), b.double(), stride=2, padding=1)
        pr, t64 = torch.sigmoid(logit), tgt.double()
        if dist == 0:
            tot = F.binary_cross_entropy(pr, t64, reduction="sum"); gref = (pr - t64) / N
        elif dist == 1:
            tot = ((255 * pr - 255 * t64) ** 2).sum() / 255; gref = 2 * 255 * (pr - t64) * pr * (1 - pr) / N
        else:
            tot = 3 * (pr - t64).abs().sum(); gref = 3 * torch.sign(pr - t64) * pr * (1 - pr) / N
        check(p1.sum(), tot, rtol=1e-5, what="fused loss sum vs fp64 dist %d" % dist)
        check(g1, gref, rtol=1e-4, atol_rel=4e-6, what="fused dL/dlogit vs fp64 dist %d" % dist)


# logits at and beyond the points where torch.sigmoid's fp32 result rounds to 1 (v >= ~16.64) or is exactly 0 (exp(-v)
# overflows: v < -88.72), plus the stretch below where 1 - p keeps only a few bits
_SAT = [16.6, 17.0, 20.0, 40.0, 88.0, 90.0, 104.0]
_SAT_TRIPLES = ([(v, -v, v2) for v, v2 in zip(_SAT, _SAT[1:] + _SAT[:1])] + [(-v, v, -v2) for v, v2 in zip(_SAT, _SAT[2:] + _SAT[:2])]
                + [(8.5, -12.0, 3.0), (15.9, 16.2, 16.5), (-87.0, -88.5, -89.0), (0.0, 12.7, -0.5)])


def _sat_targets(N, C, kind):
    """(fp32 target, uint8 target or None): exact 0 / 0.5 / 1 planes, a random plane, or uint8 pixels with 0 and 255 forced in."""
    gen = torch.Generator().manual_seed(11)
    if kind == "u8":
        t8 = torch.randint(0, 256, (N, C, 64, 64), dtype=torch.uint8, generator=gen)
        t8[:, :, ::3, ::2] = 0
        t8[:, :, 1::3, 1::2] = 255
        return t8.float() / 255.0, t8
    t = torch.rand(N, C, 64, 64, generator=gen)
    t[:, :, 0::4, :] = 0.0
    t[:, :, 1::4, :] = 1.0
    t[:, :, 2::4, 0::2] = 0.5
    return t, None


@pytest.mark.parametrize("kind", ["f32", "u8"])
@pytest.mark.parametrize("C", [1, 3])
@pytest.mark.parametrize("N", [2, 70])
def test_convT3_bernoulli_likelihood_on_saturated_logits(N, C, kind):
    """losses.py:430 through the fused last decoder layer (k_up_thin_mm for 3 channels, k_up_thin_pk for 1; fp32 and uint8
    targets) and through the raw-weight entry point, at logits where the reference's fp32 arithmetic decides the value:
    F.binary_cross_entropy clamps log(1 - p) at -100 once sigmoid(v) rounds to 1 (term 100 (1 - x), not v (1 - x)), log p once
    exp(-v) overflows (term 100 x), and below that 1 - p has only a few bits.  Zero weights, bias = the logit, so every path
    sees exactly the logit the fp32 oracle (torch CPU: the reference's own arithmetic) sees.  Per element: reconstruction
    rtol 2e-7 (1-2 ulp) for p >= 0.5, 1e-5 below, + 1e-37; dL/dlogit rtol 2e-5 + 1e-7 of the scale + 1e-20; likelihood sum
    rtol 2e-6."""
    wd = dev(torch.zeros(32, C, 4, 4))
    pairs = torch.empty(32 * _lib.thin_pair_floats(C), device=DEV)
    _stage(thin=(wd, pairs, C))
    x = nhwc(torch.relu(_rand(N, 32, 32, 32, seed=1)))
    tgt, tgt8 = _sat_targets(N, C, kind)
    td = tgt8.to(DEV) if tgt8 is not None else dev(tgt)
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / N
    coefd = dev(coef)
    for trip in _SAT_TRIPLES:
        for rot in range(3 if C == 1 else 1):
            b = torch.tensor(trip[rot:rot + 1] if C == 1 else trip, dtype=torch.float32)
            bd = dev(b)
            v = b.view(1, C, 1, 1).expand(N, C, 64, 64).contiguous().requires_grad_(True)
            pref = torch.sigmoid(v)
            tot = F.binary_cross_entropy(pref, tgt, reduction="sum")             # fp32: what the reference computes
            (tot / N).backward()
            elem = F.binary_cross_entropy(pref.detach(), tgt, reduction="none").double().sum()
            launches = [("staged", lambda r, g, p: call("dvae_convT3_fwd_staged", ptr(x), ptr(pairs), ptr(bd), ptr(td), int(tgt8 is not None),
                                                        ptr(r), ptr(g), 0, ptr(coefd), ptr(p), N, C, stream()))]
            if tgt8 is None:
                launches.append(("raw", lambda r, g, p: call("dvae_convT4s2_sigmoid_recon_fwd", ptr(x), _lib.NHWC, ptr(wd), ptr(bd), ptr(td),
                                                             ptr(r), ptr(g), 0, ptr(coefd), ptr(p), N, 32, 32, 32, C, stream())))
            for name, go in launches:
                r, g = torch.empty(N, C, 64, 64, device=DEV), torch.empty(N, C, 64, 64, device=DEV)
                p = torch.full((_lib.REC_NPART,), 7.0, device=DEV)
                go(r, g, p)
                what = "%s C=%d %s logits %s: " % (name, C, kind, tuple(b.tolist()))
                got = p.double().sum().item()
                assert abs(got - elem.item()) <= 2e-6 * abs(elem.item()) + 1e-30, what + "likelihood sum %r vs fp32 reference %r (ATen: %r)" % (got, elem.item(), tot.item())
                rc, rr = r.cpu().double(), pref.detach().double()
                # p >= 0.5: ATen's rounding of p is what the likelihood sees (1-2 ulp allowed for near-ties); below, v_exp_f32's
                # argument v log2(e) carries |v| 6e-8 of relative error into p (5e-6 at v = -88)
                rtol_p = torch.where(rr >= 0.5, 2e-7, 1e-5)
                assert ((rc - rr).abs() <= rtol_p * rr + 1e-37).all(), what + "reconstruction: %r" % ((rc - rr).abs().max().item(),)
                gerr, gref = (g.cpu().double() - v.grad.double()).abs(), v.grad.double().abs()
                # (+ 1e-20: at v = -88 ATen's p is a denormal 6e-39 and its clamped backward leaves 1e12 p x ~ 3e-27; the
                # hardware exponential flushes that p to 0)
                assert (gerr <= 2e-5 * gref + 1e-7 * gref.max() + 1e-20).all(), what + "dL/dlogit: %r of %r" % (gerr.max().item(), gref.max().item())


def test_convT3_bernoulli_reference_recorded_saturated_values():
    """The same through the values recorded from the REAL reference (tests/golden/kats.npz: losses._reconstruction_loss on
    sigmoid(+-16.6 ... +-104), make_golden.py --kats), fp32 and uint8 targets, both likelihood kernels."""
    import numpy as np
    kat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.npz"))
    N = 2
    ar = torch.arange(N * 3 * 64 * 64)
    xs = ((ar % 5).float() / 4).view(N, 3, 64, 64)
    x8 = (torch.tensor([0, 128, 255], dtype=torch.uint8)[ar % 3]).view(N, 3, 64, 64)
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / N
    coefd = dev(coef)
    x = nhwc(torch.relu(_rand(N, 32, 32, 32, seed=1)))
    for C in (3, 1):
        wd = dev(torch.zeros(32, C, 4, 4))
        pairs = torch.empty(32 * _lib.thin_pair_floats(C), device=DEV)
        _stage(thin=(wd, pairs, C))
        for i, trip in enumerate(kat["kat_rec_sat_logits"]):
            for key, td, u8 in (("kat_rec_sat_loss", dev(xs), 0), ("kat_rec_sat_loss_u8", keep(x8.to(DEV)), 1)):
                want = float(kat[key][i])
                if C == 3:
                    bd = dev(torch.from_numpy(trip))
                    r, g = torch.empty(N, 3, 64, 64, device=DEV), torch.empty(N, 3, 64, 64, device=DEV)
                    p = torch.empty(_lib.REC_NPART, device=DEV)
                    call("dvae_convT3_fwd_staged", ptr(x), ptr(pairs), ptr(bd), ptr(td), u8, ptr(r), ptr(g), 0, ptr(coefd), ptr(p), N, 3, stream())
                    got = p.double().sum().item() / N
                else:       # one channel at a time through the 1-channel kernel
                    got = 0.0
                    tc = (x8 if u8 else xs)
                    for c in range(3):
                        bd = dev(torch.from_numpy(trip[c:c + 1].copy()))
                        tcd = keep(tc[:, c:c + 1].contiguous().to(DEV))
                        r, g = torch.empty(N, 1, 64, 64, device=DEV), torch.empty(N, 1, 64, 64, device=DEV)
                        p = torch.empty(_lib.REC_NPART, device=DEV)
                        call("dvae_convT3_fwd_staged", ptr(x), ptr(pairs), ptr(bd), ptr(tcd), u8, ptr(r), ptr(g), 0, ptr(coefd), ptr(p), N, 1, stream())
                        got += p.double().sum().item() / N
                assert abs(got - want) <= 2e-6 * abs(want), "C=%d %s logits %s: %r vs the reference's %r" % (C, key, trip.tolist(), got, want)


@pytest.mark.parametrize("C", [1, 3])
def test_convT3_bernoulli_likelihood_is_atens_formula_on_the_emitted_reconstruction(C):
    """Random weights at a scale that spreads the logits over +-100: whatever fp32 reconstruction p the fused kernel emits, its
    likelihood sum is F.binary_cross_entropy of THAT p (fp32 torch CPU), clamps included -- rtol 1e-5; p itself vs fp64."""
    N = 40
    w = _rand(32, C, 4, 4, seed=2, scale=8.0)
    b = _rand(C, seed=3, scale=2.0)
    wd, bd = dev(w), dev(b)
    pairs = torch.empty(32 * _lib.thin_pair_floats(C), device=DEV)
    _stage(thin=(wd, pairs, C))
    x = nhwc(torch.relu(_rand(N, 32, 32, 32, seed=1)))
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / N
    coefd = dev(coef)
    logit = F.conv_transpose2d(x.cpu().permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=1)
    assert (logit > 17).float().mean() > 0.05 and (logit < -17).float().mean() > 0.05 and logit.min() < -100.0
    for kind in ("f32", "u8"):
        tgt, tgt8 = _sat_targets(N, C, kind)
        td = tgt8.to(DEV) if tgt8 is not None else dev(tgt)
        r, g = torch.empty(N, C, 64, 64, device=DEV), torch.empty(N, C, 64, 64, device=DEV)
        p = torch.empty(_lib.REC_NPART, device=DEV)
        call("dvae_convT3_fwd_staged", ptr(x), ptr(pairs), ptr(bd), ptr(td), int(tgt8 is not None), ptr(r), ptr(g), 0, ptr(coefd), ptr(p), N, C, stream())
        rc = r.cpu()
        want = F.binary_cross_entropy(rc, tgt, reduction="none").double().sum().item()
        got = p.double().sum().item()
        assert abs(got - want) <= 1e-5 * want, "C=%d %s: likelihood sum %r vs ATen's formula on the emitted reconstruction %r" % (C, kind, got, want)
        check(r, torch.sigmoid(logit), rtol=1e-5, atol_rel=2e-6, what="reconstruction vs fp64 at saturating scale")
        vg = rc.clone().requires_grad_(True)
        (F.binary_cross_entropy(vg, tgt, reduction="sum") / N).backward()
        check(g, vg.grad * rc * (1 - rc), rtol=1e-5, atol_rel=2e-6, what="dL/dlogit = ATen's clamped backward x sigmoid'")


def test_event_slots_order_a_late_consumer_after_marked_work():
    """dvae_event_record / dvae_event_wait (include/dvae_hip.h): work enqueued after the wait runs after everything that was
    enqueued on the other stream before the record -- and NOT after what was enqueued there later.  A chain of dependent adds
    on the side stream (each launch reads the previous result), a mark, MORE side work, then a consumer on the current stream
    that waits for the mark only."""
    n = 1 << 24
    side = torch.cuda.Stream()
    a = torch.zeros(n, device=DEV)
    one = torch.ones(n, device=DEV)
    out = torch.empty(n, device=DEV)
    torch.cuda.synchronize()
    for rep in range(3):
        a.zero_()
        torch.cuda.synchronize()
        for _ in range(20):                                    # a += 1, twenty times, on the side stream
            call("dvae_add", ptr(a), ptr(one), ptr(a), n, side.cuda_stream)
        call("dvae_event_record", 5, side.cuda_stream)
        call("dvae_event_wait", 5, stream())
        call("dvae_add", ptr(a), ptr(one), ptr(out), n, stream())       # must see all twenty additions
        torch.cuda.synchronize()
        assert torch.equal(out, torch.full_like(out, 21.0)), (rep, out[:4], out[-4:])
    with pytest.raises(_lib.DvaeHipError):
        call("dvae_event_record", 99, stream())


@pytest.mark.parametrize("n_enc,n_dec,D", [(1, 1, 10), (4, 4, 10), (7, 7, 6), (128, 128, 10), (1024, 1024, 10), (1030, 1030, 10),
                                          (262, 131, 10), (2048, 1024, 10), (50, 0, 10)])
def test_fc_chain_with_conv_ends(n_enc, n_dec, D):
    """dvae_fc_chain_fwd / dvae_fc_chain_bwd with the 4x4 end of the conv stacks in the same launch (conv_in / convT_w,
    convT_gout / conv_w: encoders.py:76-81, decoders.py:73-76 and their input gradients) == the three-launch sequences
    dvae_conv32_down -> chain -> dvae_conv32_up they replace in the training step, bit for bit, every output."""
    shapes, W, Bv = _fc_params(D, seed=7)
    ent = _fc_stage(shapes, W)
    wc, wt = dev(_rand(32, 32, 4, 4, seed=31, scale=0.2)), dev(_rand(32, 32, 4, 4, seed=32, scale=0.2))
    bc, bt_ = dev(_rand(32, seed=33, scale=0.1)), dev(_rand(32, seed=34, scale=0.1))
    f = lambda *s: torch.full(s, 7.0, device=DEV)
    img = {k: f(16384) for k in ("c_down", "c_up", "t_down", "t_up")}
    _stage([(wc, img["c_down"], img["c_up"]), (wt, img["t_down"], img["t_up"])])
    bd = {k: dev(v) for k, v in Bv.items()}
    eps = dev(torch.randn(n_enc, D, generator=torch.Generator().manual_seed(2)))
    conv_in = dev(torch.relu(_rand(n_enc, 8, 8, 32, seed=41)))
    nd = max(n_dec, 1)

    def fwd(fused):
        out = dict(h1=f(n_enc, 256), h2=f(n_enc, 256), ml=f(n_enc, 2 * D), mu=f(n_enc, D), logvar=f(n_enc, D), z=f(n_enc, D),
                   d1=f(nd, 256), d2=f(nd, 256), d3=f(nd, 512))
        a_flat, up = f(n_enc, 512), f(nd, 8, 8, 32)
        kl = torch.full((_lib.KL_FLOATS,), 7.0, device=DEV)
        extra = {}
        if fused:
            extra = dict(conv_in=ptr(conv_in), conv_w=ptr(img["c_down"]), conv_b=ptr(bc))
            if n_dec:
                extra.update(convT_w=ptr(img["t_up"]), convT_b=ptr(bt_), convT_out=ptr(up))
        else:
            call("dvae_conv32_down", ptr(conv_in), ptr(img["c_down"]), ptr(bc), None, ptr(a_flat), _lib.NCHW, n_enc, 4,
                 _lib.ACT_RELU, stream())
        st, addr = _lib.struct_of(_lib.FcChainFwdArgs, a_flat=ptr(a_flat), eps=ptr(eps), kl_part=ptr(kl) + 64,
                                  n_enc=n_enc, n_kl=n_enc, n_dec=n_dec, D=D,
                                  **{"w_" + k: ptr(ent[k][1]) for k in shapes}, **{"b_" + k: ptr(bd[k]) for k in shapes},
                                  **{k: ptr(v) for k, v in out.items()}, **extra)
        call("dvae_fc_chain_fwd", addr, stream())
        if not fused and n_dec:
            call("dvae_conv32_up", ptr(out["d3"]), _lib.NCHW, ptr(img["t_up"]), ptr(bt_), None, ptr(up), n_dec, 4, _lib.ACT_RELU,
                 stream())
        torch.cuda.synchronize()
        out.update(a_flat=a_flat, up=up, kl=kl)
        return out

    ref, got = fwd(False), fwd(True)
    for k in ref:
        assert torch.equal(ref[k], got[k]), "forward %s" % k
    assert float(ref["a_flat"].abs().max()) > 0 and (n_dec == 0 or float(ref["up"][:n_dec].abs().max()) > 0)
    if n_dec == 0:
        return
    # ---- backward over n = n_dec rows
    n = n_dec
    gout = dev(_rand(n, 8, 8, 32, seed=51))
    acts = {k: dev(torch.relu(_rand(n, w, seed=60 + i))) for i, (k, w) in enumerate(
        [("d2", 256), ("d1", 256), ("h2", 256), ("h1", 256), ("a_flat", 512), ("d3", 512)])}
    conv_act = dev(torch.relu(_rand(n, 8, 8, 32, seed=71)))
    mu, lv = dev(_rand(n, D, seed=20)), dev(_rand(n, D, seed=21, scale=0.7))
    dz2 = dev(_rand(n, D, seed=22))
    scal = torch.zeros(_lib.NSCAL); scal[_lib.S_KLW] = 1.7
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / n
    scal, coef = dev(scal), dev(coef)

    def bwd(fused):
        out = dict(gd2=f(n, 256), gd1=f(n, 256), dz=f(n, D), dml=f(n, 2 * D), gh2=f(n, 256), gh1=f(n, 256), ga_flat=f(n, 512))
        gd3, gin = f(n, 512), f(n, 8, 8, 32)
        extra = {}
        if fused:
            extra = dict(convT_gout=ptr(gout), convT_w=ptr(img["t_down"]), d3=ptr(acts["d3"]), conv_w=ptr(img["c_up"]),
                         conv_act=ptr(conv_act), conv_gin=ptr(gin))
        else:
            call("dvae_conv32_down", ptr(gout), ptr(img["t_down"]), None, ptr(acts["d3"]), ptr(gd3), _lib.NCHW, n, 4,
                 _lib.ACT_NONE, stream())
        ins = dict(gd3=gd3, mu=mu, logvar=lv, eps=eps[:n].contiguous(), dz2=dz2, scal=scal, coef=coef,
                   **{k: v for k, v in acts.items() if k != "d3"})
        st, addr = _lib.struct_of(_lib.FcChainBwdArgs, n=n, D=D, **{"w_" + k: ptr(ent[k][2]) for k in shapes},
                                  **{k: ptr(v) for k, v in ins.items()}, **{k: ptr(v) for k, v in out.items()}, **extra)
        call("dvae_fc_chain_bwd", addr, stream())
        if not fused:
            call("dvae_conv32_up", ptr(out["ga_flat"]), _lib.NCHW, ptr(img["c_up"]), None, ptr(conv_act), ptr(gin), n, 4,
                 _lib.ACT_NONE, stream())
        torch.cuda.synchronize()
        out.update(gd3=gd3, gin=gin)
        return out

    ref, got = bwd(False), bwd(True)
    for k in ref:
        assert torch.equal(ref[k], got[k]), "backward %s" % k
    assert float(ref["gd3"].abs().max()) > 0 and float(ref["gin"].abs().max()) > 0
