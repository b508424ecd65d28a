"""-m gpu: every HIP kernel through the C-ABI vs torch CPU (fp64 accumulation) on the same
seeded inputs.  Tolerance: rtol 1e-4 plus 2e-5 x max|ref| (fp32 accumulation-order noise)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib  # noqa
from oracle import disvae_oracle as O


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


CONV_CASES = [
    # N, Cin, H, x_layout                 (Cout = 32, y NHWC)
    (3, 1, 64, _lib.NCHW), (5, 3, 64, _lib.NCHW),          # conv1 (thin, MFMA K=16C)
    (200, 3, 64, _lib.NCHW), (193, 1, 64, _lib.NCHW),      # conv1 at batches that take the wave-specialised kernel (conv_thin_ws.hip)
    (3, 32, 32, _lib.NHWC), (70, 32, 32, _lib.NHWC),       # conv2 (MFMA HS=16; 280 units > 256 workgroups)
    (5, 32, 16, _lib.NHWC), (6, 32, 8, _lib.NHWC), (9, 32, 8, _lib.NHWC),   # conv3 / conv_64 (tails)
    (2, 1, 32, _lib.NCHW),                                  # MNIST geometry -> generic kernel
    (64, 1, 32, _lib.NCHW), (37, 3, 32, _lib.NCHW),         # ... at batches where its weight gradient splits the positions over chunks
]


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("N,Cin,H,xl", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(N, Cin, H, xl, generic):
    if generic and N > 9:
        pytest.skip("large case only for the tuned path")
    Cout = 32
    x = _rand(N, Cin, H, H, seed=1)
    w = _rand(Cout, Cin, 4, 4, seed=2, scale=0.2)
    b = _rand(Cout, seed=3, scale=0.1)
    xd = nhwc(x) if xl == _lib.NHWC else dev(x)
    wd, bd = dev(w), dev(b)
    y = torch.empty(N, H // 2, H // 2, Cout, device=DEV)
    with force_generic(generic):
        call("dvae_conv4s2_fwd", ptr(xd), xl, ptr(wd), ptr(bd), ptr(y), _lib.NHWC, N, Cin, H, H, Cout, _lib.ACT_RELU, stream())
        ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
        check(from_nhwc(y, N, Cout, H // 2, H // 2), ref, what="conv fwd")
        # wgrad + bias grad
        dy = _rand(N, Cout, H // 2, H // 2, seed=4)
        dyd = nhwc(dy)
        ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)
        dw, db = torch.full((Cout, Cin, 4, 4), 7.0, device=DEV), torch.full((Cout,), 7.0, device=DEV)
        call("dvae_conv4s2_wgrad", ptr(xd), xl, ptr(dyd), _lib.NHWC, ptr(dw), ptr(db), N, Cin, H, H, Cout, ptr(ws), stream())
        xr = x.double().requires_grad_(True)
        wr = w.double().requires_grad_(True)
        br = b.double().requires_grad_(True)
        out = F.conv2d(xr, wr, br, stride=2, padding=1)
        out.backward(dy.double())
        check(dw, wr.grad, what="conv wgrad")
        check(db, br.grad, what="conv bias grad")
        # dgrad with fused ReLU mask of the producing layer
        if Cin == 32:
            xact = torch.relu(_rand(N, Cin, H, H, seed=5))
            dx = torch.empty(N, H, H, Cin, device=DEV)
            call("dvae_conv4s2_dgrad", ptr(dyd), _lib.NHWC, ptr(wd), ptr(nhwc(xact)), ptr(dx), _lib.NHWC, N, Cin, H, H, Cout, stream())
            check(from_nhwc(dx, N, Cin, H, H), xr.grad * (xact > 0), what="conv dgrad")


CONVT_CASES = [
    # N, H(in), Cout, y_layout, act
    (3, 4, 32, _lib.NHWC, _lib.ACT_RELU), (6, 4, 32, _lib.NHWC, _lib.ACT_RELU), (5, 8, 32, _lib.NHWC, _lib.ACT_RELU),
    (3, 16, 32, _lib.NHWC, _lib.ACT_RELU), (70, 16, 32, _lib.NHWC, _lib.ACT_RELU),
    (3, 32, 1, _lib.NCHW, _lib.ACT_SIGMOID), (5, 32, 3, _lib.NCHW, _lib.ACT_SIGMOID),
    (200, 32, 3, _lib.NCHW, _lib.ACT_SIGMOID), (193, 32, 1, _lib.NCHW, _lib.ACT_SIGMOID),   # convT3: its input gradient on conv_thin_ws.hip
    (2, 16, 1, _lib.NCHW, _lib.ACT_SIGMOID),
    (64, 16, 1, _lib.NCHW, _lib.ACT_SIGMOID), (37, 16, 3, _lib.NCHW, _lib.ACT_SIGMOID),   # MNIST geometry: split generic weight gradient, bias from the big side
]


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("N,H,Cout,yl,act", CONVT_CASES)
def test_convT_fwd_dgrad_wgrad(N, H, Cout, yl, act, generic):
    if generic and N > 9:
        pytest.skip("large case only for the tuned path")
    Cin = 32
    x = torch.relu(_rand(N, Cin, H, H, seed=1))
    w = _rand(Cin, Cout, 4, 4, seed=2, scale=0.2)
    b = _rand(Cout, seed=3, scale=0.1)
    xd, wd, bd = nhwc(x), dev(w), dev(b)
    H2 = 2 * H
    y = torch.empty((N, H2, H2, Cout) if yl == _lib.NHWC else (N, Cout, H2, H2), device=DEV)
    with force_generic(generic):
        call("dvae_convT4s2_fwd", ptr(xd), _lib.NHWC, ptr(wd), ptr(bd), ptr(y), yl, N, Cin, H, H, Cout, act, stream())
        xr = x.double().requires_grad_(True)
        wr = w.double().requires_grad_(True)
        br = b.double().requires_grad_(True)
        pre = F.conv_transpose2d(xr, wr, br, stride=2, padding=1)
        ref = torch.relu(pre) if act == _lib.ACT_RELU else torch.sigmoid(pre)
        got = from_nhwc(y, N, Cout, H2, H2) if yl == _lib.NHWC else y
        check(got, ref, what="convT fwd")
        dy = _rand(N, Cout, H2, H2, seed=4)
        pre.backward(dy.double())
        dyd = nhwc(dy) if yl == _lib.NHWC else dev(dy)
        ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)
        dw, db = torch.full((Cin, Cout, 4, 4), 7.0, device=DEV), torch.full((Cout,), 7.0, device=DEV)
        call("dvae_convT4s2_wgrad", ptr(xd), _lib.NHWC, ptr(dyd), yl, ptr(dw), ptr(db), N, Cin, H, H, Cout, ptr(ws), stream())
        check(dw, wr.grad, what="convT wgrad")
        check(db, br.grad, what="convT bias grad")
        dx = torch.empty(N, H, H, Cin, device=DEV)
        call("dvae_convT4s2_dgrad", ptr(dyd), yl, ptr(wd), ptr(xd), ptr(dx), _lib.NHWC, N, Cin, H, H, Cout, stream())
        check(from_nhwc(dx, N, Cin, H, H), xr.grad * (x > 0), what="convT dgrad")


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("N,C,H,dist", [(5, 3, 32, "bernoulli"), (3, 1, 32, "gaussian"), (300, 3, 32, "bernoulli"),
                                         (4, 1, 16, "laplace")])
def test_convT_sigmoid_recon_fused(N, C, H, dist, generic):
    """last decoder layer fused with the likelihood: recon, loss partial sums and dL/dlogit in one pass."""
    if generic and N > 9:
        pytest.skip("large case only for the tuned path")
    x = torch.relu(_rand(N, 32, H, H, seed=1))
    w = _rand(32, C, 4, 4, seed=2, scale=0.2)
    b = _rand(C, seed=3, scale=0.1)
    tgt = torch.rand(N, C, 2 * H, 2 * H, generator=torch.Generator().manual_seed(4))
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / N
    recon = torch.empty(N, C, 2 * H, 2 * H, device=DEV)
    g = torch.empty_like(recon)
    parts = torch.full((_lib.REC_NPART,), 7.0, device=DEV)
    with force_generic(generic):
        call("dvae_convT4s2_sigmoid_recon_fwd", ptr(nhwc(x)), _lib.NHWC, ptr(dev(w)), ptr(dev(b)), ptr(dev(tgt)), ptr(recon),
             ptr(g), _lib.REC[dist], ptr(dev(coef)), ptr(parts), N, 32, H, H, C, stream())
    lr = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1).requires_grad_(True)
    ref_recon = torch.sigmoid(lr)
    loss = O.reconstruction_loss(tgt.double(), ref_recon, dist)
    loss.backward()
    check(recon, ref_recon, what="fused recon")
    check(parts.sum() / N, loss, rtol=2e-5, what="fused loss")
    check(g, lr.grad, rtol=2e-4, atol_rel=2e-5, what="fused dL/dlogit")


def test_relayout():
    x = _rand(5, 32, 4, 4)
    a = nhwc(x)
    out = torch.empty(5, 512, device=DEV)
    call("dvae_relayout", ptr(a), _lib.NHWC, ptr(out), 5, 32, 4, 4, stream())
    assert torch.equal(out.cpu(), x.reshape(5, 512))
    back = torch.empty(5, 4, 4, 32, device=DEV)
    call("dvae_relayout", ptr(out), _lib.NCHW, ptr(back), 5, 32, 4, 4, stream())
    assert torch.equal(back.cpu(), x.permute(0, 2, 3, 1))


@pytest.mark.parametrize("M,K,N,act", [(7, 10, 256, _lib.ACT_RELU), (130, 512, 256, _lib.ACT_RELU),
                                        (33, 256, 20, _lib.ACT_NONE), (64, 1000, 1000, _lib.ACT_LEAKY02),
                                        (256, 10, 1000, _lib.ACT_LEAKY02), (100, 1000, 2, _lib.ACT_NONE),
                                        (1, 256, 512, _lib.ACT_RELU), (1024, 512, 256, _lib.ACT_RELU), (700, 256, 20, _lib.ACT_NONE),
                                        (5, 64, 48, _lib.ACT_RELU), (40, 100, 36, _lib.ACT_LEAKY02), (300, 128, 128, _lib.ACT_NONE),
                                        (1800, 1000, 1000, _lib.ACT_LEAKY02), (200, 300, 260, _lib.ACT_RELU),
                                        (2050, 1000, 2, _lib.ACT_NONE), (2050, 10, 1000, _lib.ACT_LEAKY02), (130, 16, 600, _lib.ACT_NONE),
                                        (300, 1000, 1000, _lib.ACT_LEAKY02), (1030, 1000, 1000, _lib.ACT_LEAKY02)])
def test_linear(M, K, N, act):
    x = _rand(M, K, seed=1)
    w = _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    b = _rand(N, seed=3, scale=0.1)
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.empty(M, N, device=DEV)
    call("dvae_linear_fwd", ptr(xd), ptr(wd), ptr(bd), ptr(y), M, K, N, act, None, stream())
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = F.linear(xr, wr, br)
    ref = {0: pre, 1: torch.relu(pre), 2: F.leaky_relu(pre, 0.2)}[act]
    check(y, ref, what="linear fwd")
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)      # split-contraction schedule
    y2 = torch.empty(M, N, device=DEV)
    call("dvae_linear_fwd", ptr(xd), ptr(wd), ptr(bd), ptr(y2), M, K, N, act, ptr(ws), stream())
    check(y2, ref, what="linear fwd (split)")
    dy = _rand(M, N, seed=4)
    pre.backward(dy.double())
    dyd = dev(dy)
    dw, db = torch.full((N, K), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
    call("dvae_linear_wgrad", ptr(xd), ptr(dyd), ptr(dw), ptr(db), M, K, N, None, stream())
    check(dw, wr.grad, what="linear wgrad")
    check(db, br.grad, what="linear bias grad")
    dw2, db2 = torch.full((N, K), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
    call("dvae_linear_wgrad", ptr(xd), ptr(dyd), ptr(dw2), ptr(db2), M, K, N, ptr(ws), stream())
    check(dw2, wr.grad, what="linear wgrad (split)")
    check(db2, br.grad, what="linear bias grad (split)")
    xact = _rand(M, K, seed=5)
    for mact in (_lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_LEAKY02):
        mult = {0: torch.ones_like(xact), 1: (xact > 0).float(), 2: torch.where(xact > 0, 1.0, 0.2)}[mact]
        for wsp in (None, ptr(ws)):
            dx = torch.empty(M, K, device=DEV)
            call("dvae_linear_dgrad", ptr(dyd), ptr(wd), ptr(dev(xact)) if mact else None, mact, ptr(dx), M, K, N, wsp, stream())
            check(dx, xr.grad * mult.double(), what="linear dgrad act=%d split=%s" % (mact, wsp is not None))


@pytest.mark.parametrize("B", [2, 8, 200, 1500])
def test_reparam_kl(B):
    D = 10
    ml = _rand(B, 2 * D, seed=1, scale=1.5)
    eps = torch.randn(B, D, generator=torch.Generator().manual_seed(2))
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / B
    mld, epsd, coefd = dev(ml), dev(eps), dev(coef)
    mu, lv, z = (torch.empty(B, D, device=DEV) for _ in range(3))
    kl = torch.zeros(16 + 64 * 16, device=DEV)
    call("dvae_reparam_kl_fwd", ptr(mld), ptr(epsd), ptr(mu), ptr(lv), ptr(z), ptr(kl), ptr(coefd), B, D, stream())
    m_ref, l_ref = ml.view(B, D, 2).unbind(-1)
    assert torch.equal(mu.cpu(), m_ref) and torch.equal(lv.cpu(), l_ref)
    check(z, O.reparameterize(m_ref.double(), l_ref.double(), eps.double()), what="z")
    check(kl[:D], O.kl_normal_loss(m_ref.double(), l_ref.double())[1], what="kl_dim")
    call("dvae_reparam_kl_fwd", ptr(mld), None, ptr(mu), ptr(lv), ptr(z), None, None, B, D, stream())
    assert torch.equal(z.cpu(), m_ref)      # eval mode: z = mean (vae.py:69-71)
    # backward
    dz, dmx, dlx = _rand(B, D, seed=3), _rand(B, D, seed=4), _rand(B, D, seed=5)
    scal = torch.zeros(_lib.NSCAL); scal[_lib.S_KLW] = 2.5
    dml = torch.empty(B, 2 * D, device=DEV)
    dz2, dz3 = _rand(B, D, seed=6), _rand(B, D, seed=7)          # gradients reaching z by other routes
    call("dvae_reparam_kl_bwd", ptr(dev(dz - dz2 - dz3)), ptr(dev(dz2)), ptr(dev(dz3)), ptr(dev(dmx)), ptr(dev(dlx)),
         ptr(dev(m_ref)), ptr(dev(l_ref)), ptr(epsd), ptr(dev(scal)), ptr(coefd), ptr(dml), B, D, stream())
    mr, lr = m_ref.double().requires_grad_(True), l_ref.double().requires_grad_(True)
    zz = O.reparameterize(mr, lr, eps.double())
    obj = (zz * dz.double()).sum() + (mr * dmx.double()).sum() + (lr * dlx.double()).sum() + 2.5 * O.kl_normal_loss(mr, lr)[0]
    obj.backward()
    ref = torch.stack((mr.grad, lr.grad), -1).reshape(B, 2 * D)
    check(dml, ref, what="dml")


@pytest.mark.parametrize("dist", ["bernoulli", "gaussian", "laplace"])
def test_recon_loss(dist):
    B, C, H = 6, 3, 64
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(B, C, H, H, generator=g) * 3
    logits.view(-1)[:8] = torch.tensor([40., -40., 20., -20., 17., -17., 90., -90.])   # saturated sigmoids / clamps
    x = torch.rand(B, C, H, H, generator=g)
    x.view(-1)[:4] = torch.tensor([0., 1., 1., 0.])
    recon = torch.sigmoid(logits)
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / B
    parts = torch.empty(_lib.REC_NPART, device=DEV)
    gl = torch.empty_like(recon, device=DEV)
    call("dvae_recon_loss", ptr(dev(recon)), ptr(dev(x)), recon.numel(), _lib.REC[dist], ptr(dev(coef)), ptr(parts), ptr(gl), 1, stream())
    ref = O.reconstruction_loss(x, recon, dist)        # fp32, like the reference
    check(parts.sum() / B, ref, rtol=2e-5, what="recon loss " + dist)
    lr = logits.clone().requires_grad_(True)
    O.reconstruction_loss(x, torch.sigmoid(lr), dist).backward()
    check(gl, lr.grad, rtol=1e-4, atol_rel=1e-6, what="dL/dlogit " + dist)
    gr = torch.empty_like(recon, device=DEV)
    call("dvae_recon_loss", ptr(dev(recon)), ptr(dev(x)), recon.numel(), _lib.REC[dist], ptr(dev(coef)), ptr(parts), ptr(gr), 0, stream())
    rr = recon.clone().requires_grad_(True)
    O.reconstruction_loss(x, rr, dist).backward()
    check(gr, rr.grad, rtol=1e-4, atol_rel=1e-6, what="dL/drecon " + dist)
    out = torch.empty_like(recon, device=DEV)
    call("dvae_sigmoid_bwd", ptr(gr), ptr(dev(recon)), ptr(out), recon.numel(), stream())
    check(out, lr.grad, rtol=1e-4, atol_rel=1e-6, what="sigmoid bwd")


def test_btcvae_kat_reference_values():
    """The RNG-free known-answer vectors recorded from the REAL reference (tests/golden/kats.npz; SURVEY 8c:
    B=4, D=3, i=arange(12).view(4,3), mu=sin(i), logvar=0.5cos(i), eps=cos(2i+1), n_data=100) through the HIP
    estimator kernel (run-time latent dimension 3), with and without minibatch-stratified weights."""
    from golden_util import load
    from disvae_amd.utils.math import log_importance_weights
    kat = load("kats")
    B, D = 4, 3
    i = torch.arange(12, dtype=torch.float32).view(B, D)
    mu, lv, eps = torch.sin(i), 0.5 * torch.cos(i), torch.cos(2 * i + 1)
    z = mu + torch.exp(0.5 * lv) * eps
    lw = torch.zeros(4); lw[:3] = log_importance_weights(B, 100)
    for mss in (1, 0):
        rs = torch.empty(B, _lib.ROWSTATS, device=DEV)
        tmp = torch.empty(3 * D, B, device=DEV)
        call("dvae_btcvae_fwd", ptr(dev(z)), ptr(dev(mu)), ptr(dev(lv)), B, D, 0, B, mss, ptr(dev(lw)), ptr(tmp), ptr(rs), stream())
        for k, nm in enumerate(["log_pz", "log_qz", "log_prod_qzi", "log_q_zCx"]):
            np.testing.assert_allclose(rs[:, k].cpu().numpy(), kat["kat_%s_mss%d" % (nm, mss)], rtol=1e-5, atol=1e-6, err_msg=nm)


@pytest.mark.parametrize("B,n_data,mss,D", [(4, 100, True, 10), (4, 100, False, 10), (8, 737280, True, 10), (64, 202599, True, 10),
                                            (256, 737280, True, 10), (1024, 202599, True, 10), (100, 5000, True, 10),
                                            (70, 5000, True, 1), (300, 202599, True, 6), (256, 737280, True, 12), (130, 5000, True, 16)])
def test_btcvae_fwd_bwd(B, n_data, mss, D):
    g = torch.Generator().manual_seed(B)
    mu = torch.randn(B, D, generator=g)
    lv = torch.randn(B, D, generator=g) * 0.7 - 0.5
    eps = torch.randn(B, D, generator=g)
    z = mu + torch.exp(0.5 * lv) * eps
    from disvae_amd.utils.math import log_importance_weights
    lw = torch.zeros(4); lw[:3] = log_importance_weights(B, n_data)
    rs = torch.empty(B, _lib.ROWSTATS, device=DEV)
    tmp = torch.empty(3 * D, B, device=DEV)
    zd, mud, lvd, lwd = dev(z), dev(mu), dev(lv), dev(lw)
    call("dvae_btcvae_fwd", ptr(zd), ptr(mud), ptr(lvd), B, D, 0, B, int(mss), ptr(lwd), ptr(tmp), ptr(rs), stream())
    ref = O.btcvae_log_densities(z.double(), mu.double(), lv.double(), n_data, mss)
    for k, nm in enumerate(["log_pz", "log_qz", "log_prod_qzi", "log_q_zCx"]):
        check(rs[:, k], ref[k], rtol=2e-6, atol_rel=2e-6, what=nm)
    ref32 = O.btcvae_log_densities(z, mu, lv, n_data, mss)     # and against the fp32 reference arithmetic
    for k in range(4):
        check(rs[:, k], ref32[k], rtol=1e-5, atol_rel=1e-5, what="fp32 col %d" % k)
    # row-sharded evaluation gives the same rows (data-parallel path)
    half = B // 2
    rs2 = torch.empty(B - half, _lib.ROWSTATS, device=DEV)
    call("dvae_btcvae_fwd", ptr(zd), ptr(mud), ptr(lvd), B, D, half, B - half, int(mss), ptr(lwd), ptr(tmp), ptr(rs2), stream())
    assert torch.equal(rs2.cpu()[:, :4 + D], rs[half:].cpu()[:, :4 + D])      # 4 sums + D per-dimension logsumexps
    # backward of alpha*mi + beta*tc + anneal*gamma*dw
    alpha, beta, gamma, anneal = 1.0, 6.4, 1.5, 0.37
    coef = torch.zeros(_lib.NCOEF)
    coef[_lib.C_ALPHA], coef[_lib.C_BETA], coef[_lib.C_GAMMA], coef[_lib.C_ANNEAL] = alpha, beta, gamma, anneal
    dz, dmu, dlv = (torch.empty(B, D, device=DEV) for _ in range(3))
    call("dvae_btcvae_bwd", ptr(zd), ptr(mud), ptr(lvd), ptr(rs), B, D, 0, B, int(mss), ptr(lwd), ptr(dev(coef)), ptr(tmp),
         ptr(dz), ptr(dmu), ptr(dlv), stream())
    zr, mr, lr = (t.double().requires_grad_(True) for t in (z, mu, lv))
    mi, tc, dw = O.btcvae_terms(zr, mr, lr, n_data, mss)
    (alpha * mi + beta * tc + anneal * gamma * dw).backward()
    check(dz, zr.grad, rtol=2e-4, atol_rel=1e-5, what="dz")
    check(dmu, mr.grad, rtol=2e-4, atol_rel=1e-5, what="dmu")
    check(dlv, lr.grad, rtol=2e-4, atol_rel=1e-5, what="dlv")
    # sharded backward: column sums add up, row grads are the local rows
    dza, dma, dla = (torch.empty(half, D, device=DEV), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV))
    dzb, dmb, dlb = (torch.empty(B - half, D, device=DEV), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV))
    call("dvae_btcvae_bwd", ptr(zd), ptr(mud), ptr(lvd), ptr(rs), B, D, 0, half, int(mss), ptr(lwd), ptr(dev(coef)), ptr(tmp),
         ptr(dza), ptr(dma), ptr(dla), stream())
    call("dvae_btcvae_bwd", ptr(zd), ptr(mud), ptr(lvd), ptr(rs2), B, D, half, B - half, int(mss), ptr(lwd), ptr(dev(coef)), ptr(tmp),
         ptr(dzb), ptr(dmb), ptr(dlb), stream())
    check(torch.cat((dza, dzb)), zr.grad, rtol=2e-4, atol_rel=1e-5, what="sharded dz")
    check(dma + dmb, mr.grad, rtol=2e-4, atol_rel=1e-5, what="sharded dmu")
    check(dla + dlb, lr.grad, rtol=2e-4, atol_rel=1e-5, what="sharded dlv")


def test_permute_dims_and_disc_losses():
    B, D = 37, 10
    z = _rand(B, D, seed=1)
    perms = torch.stack([torch.randperm(B, generator=torch.Generator().manual_seed(d)) for d in range(D)])
    out = torch.empty(B, D, device=DEV)
    call("dvae_permute_dims", ptr(dev(z)), ptr(keep(perms.to(DEV))), ptr(out), B, D, stream())
    assert torch.equal(out.cpu(), O.permute_dims(z, list(perms)))
    Bh = 50
    lg = _rand(2 * Bh, 2, seed=2, scale=3)
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_ANNEAL], coef[_lib.C_BETA] = 0.3, 6.4
    sums, g_dtc, g_tc = torch.empty(4, device=DEV), torch.empty(2 * Bh, 2, device=DEV), torch.empty(Bh, 2, device=DEV)
    call("dvae_disc_losses", ptr(dev(lg)), Bh, ptr(dev(coef)), ptr(sums), ptr(g_dtc), ptr(g_tc), stream())
    lr = lg.double().requires_grad_(True)
    d_z, d_zp = lr[:Bh], lr[Bh:]
    tc = (d_z[:, 0] - d_z[:, 1]).mean()
    ce0 = F.cross_entropy(d_z, torch.zeros(Bh, dtype=torch.long))
    ce1 = F.cross_entropy(d_zp, torch.ones(Bh, dtype=torch.long))
    check(sums[:3], torch.stack((tc * Bh, ce0 * Bh, ce1 * Bh)), what="disc sums")
    (0.5 * (ce0 + ce1)).backward()
    check(g_dtc, lr.grad, what="g_dtc")
    ref_tc = torch.zeros(Bh, 2, dtype=torch.double); ref_tc[:, 0] = 0.3 * 6.4 / Bh; ref_tc[:, 1] = -0.3 * 6.4 / Bh
    check(g_tc, ref_tc, what="g_tc")


@pytest.mark.parametrize("kind,B", [(_lib.LOSS_BETAH, 8), (_lib.LOSS_BETAB, 700), (_lib.LOSS_BTCVAE, 300), (_lib.LOSS_FACTOR, 20000)])
def test_loss_epilogue_equals_pack_then_finalize(kind, B):
    """dvae_loss_epilogue (one launch: finish the KL partials + pack + finalize) == the sharded
    sequence dvae_reparam_kl_fwd(coef) -> dvae_loss_pack -> dvae_loss_finalize, bit for bit."""
    D = 10
    ml = dev(_rand(B, 2 * D, seed=1, scale=0.5))
    eps = dev(_rand(B, D, seed=2))
    coef = torch.zeros(_lib.NCOEF)
    coef[_lib.C_INV_B], coef[_lib.C_ANNEAL], coef[_lib.C_BETA] = 1.0 / B, 0.3, 4.0
    coef[_lib.C_ALPHA], coef[_lib.C_GAMMA], coef[_lib.C_CAP] = 1.0, 2.0, 7.0
    coefd = dev(coef)
    partials = dev(_rand(_lib.REC_NPART, seed=3).abs())
    rowstats = dev(_rand(B, _lib.ROWSTATS, seed=4)) if kind == _lib.LOSS_BTCVAE else None
    disc = dev(_rand(4, seed=5)) if kind == _lib.LOSS_FACTOR else None
    f = lambda *s: torch.empty(*s, device=DEV)
    mu, lv, z = f(B, D), f(B, D), f(B, D)
    out = []
    for fused in (False, True):
        kl = torch.zeros(_lib.KL_FLOATS, device=DEV)
        packed, scal = torch.zeros(_lib.NPACK, device=DEV), torch.zeros(_lib.NSCAL, device=DEV)
        call("dvae_reparam_kl_fwd", ptr(ml), ptr(eps), ptr(mu), ptr(lv), ptr(z), ptr(kl), None if fused else ptr(coefd), B, D, stream())
        if fused:
            call("dvae_loss_epilogue", kind, ptr(partials), ptr(kl), _lib.lib().dvae_reparam_kl_blocks(B), D, ptr(rowstats), B if rowstats is not None else 0,
                 ptr(disc), B, ptr(coefd), ptr(packed), ptr(scal), stream())
        else:
            call("dvae_loss_pack", ptr(partials), ptr(kl), D, ptr(rowstats), B if rowstats is not None else 0, ptr(disc), ptr(packed), stream())
            call("dvae_loss_finalize", kind, ptr(packed), D, B, ptr(coefd), ptr(scal), stream())
        out.append((packed.cpu(), scal.cpu()))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    assert out[0][1][_lib.S_LOSS].abs() > 0


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("N", [3, 7, 70])
def test_4x4_end_of_the_conv_stack_writes_nchw(N, generic):
    """The last encoder conv (8x8 -> 4x4) writes its output NCHW = the (c,h,w) flatten order of encoders.py:80,
    and the first decoder convT's dgrad (8x8 -> 4x4) writes dx NCHW with an NCHW ReLU mask (decoders.py:74):
    the tuned MFMA kernel and the generic kernel, vs torch."""
    C = 32
    x = _rand(N, C, 8, 8, seed=1)
    w = _rand(C, C, 4, 4, seed=2, scale=0.2)
    b = _rand(C, seed=3, scale=0.1)
    with force_generic(generic):
        y = torch.empty(N, C, 4, 4, device=DEV)
        call("dvae_conv4s2_fwd", ptr(nhwc(x)), _lib.NHWC, ptr(dev(w)), ptr(dev(b)), ptr(y), _lib.NCHW, N, C, 8, 8, C, _lib.ACT_RELU, stream())
        check(y, torch.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)), what="conv fwd -> NCHW")
        # convT 4x4 -> 8x8: dgrad takes dy[N,8,8,32] back to dx[N,32,4,4] (NCHW), masked by the NCHW activation
        xa = torch.relu(_rand(N, C, 4, 4, seed=5))
        wt = _rand(C, C, 4, 4, seed=6, scale=0.2)
        dy = _rand(N, C, 8, 8, seed=7)
        xr = xa.double().requires_grad_(True)
        F.conv_transpose2d(xr, wt.double(), None, stride=2, padding=1).backward(dy.double())
        dx = torch.empty(N, C, 4, 4, device=DEV)
        call("dvae_convT4s2_dgrad", ptr(nhwc(dy)), _lib.NHWC, ptr(dev(wt)), ptr(dev(xa)), ptr(dx), _lib.NCHW, N, C, 4, 4, C, stream())
        check(dx, xr.grad * (xa > 0), what="convT dgrad -> NCHW")


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("N", [3, 7, 70])
def test_4x4_end_of_the_conv_stack_reads_nchw(N, generic):
    """The first decoder convT reads its 4x4x32 input NCHW (= lin3's output, decoders.py:74), and the last
    encoder conv's backward reads its 4x4x32 output gradient NCHW (= lin1's input gradient): forward, dgrad and
    both weight gradients, tuned MFMA kernels and generic kernels, vs torch."""
    C = 32
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)
    with force_generic(generic):
        # ---- convT 4x4 -> 8x8 with NCHW input
        x = torch.relu(_rand(N, C, 4, 4, seed=1))
        w = _rand(C, C, 4, 4, seed=2, scale=0.2)
        b = _rand(C, seed=3, scale=0.1)
        y = torch.empty(N, 8, 8, C, device=DEV)
        call("dvae_convT4s2_fwd", ptr(dev(x)), _lib.NCHW, ptr(dev(w)), ptr(dev(b)), ptr(y), _lib.NHWC, N, C, 4, 4, C, _lib.ACT_RELU, stream())
        xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
        pre = F.conv_transpose2d(xr, wr, br, stride=2, padding=1)
        check(from_nhwc(y, N, C, 8, 8), torch.relu(pre), what="convT fwd <- NCHW")
        dy = _rand(N, C, 8, 8, seed=4)
        pre.backward(dy.double())
        dw, db = torch.full((C, C, 4, 4), 7.0, device=DEV), torch.full((C,), 7.0, device=DEV)
        call("dvae_convT4s2_wgrad", ptr(dev(x)), _lib.NCHW, ptr(nhwc(dy)), _lib.NHWC, ptr(dw), ptr(db), N, C, 4, 4, C, ptr(ws), stream())
        check(dw, wr.grad, what="convT wgrad <- NCHW x")
        check(db, br.grad, what="convT bias grad")
        # ---- conv 8x8 -> 4x4 backward with the NCHW output gradient
        xin = torch.relu(_rand(N, C, 8, 8, seed=5))
        w2 = _rand(C, C, 4, 4, seed=6, scale=0.2)
        g = _rand(N, C, 4, 4, seed=7)
        xr2, wr2, br2 = xin.double().requires_grad_(True), w2.double().requires_grad_(True), torch.zeros(C, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr2, wr2, br2, stride=2, padding=1).backward(g.double())
        dx = torch.empty(N, 8, 8, C, device=DEV)
        call("dvae_conv4s2_dgrad", ptr(dev(g)), _lib.NCHW, ptr(dev(w2)), ptr(nhwc(xin)), ptr(dx), _lib.NHWC, N, C, 8, 8, C, stream())
        check(from_nhwc(dx, N, C, 8, 8), xr2.grad * (xin > 0), what="conv dgrad <- NCHW dy")
        dw2, db2 = torch.full((C, C, 4, 4), 7.0, device=DEV), torch.full((C,), 7.0, device=DEV)
        call("dvae_conv4s2_wgrad", ptr(nhwc(xin)), _lib.NHWC, ptr(dev(g)), _lib.NCHW, ptr(dw2), ptr(db2), N, C, 8, 8, C, ptr(ws), stream())
        check(dw2, wr2.grad, what="conv wgrad <- NCHW dy")
        check(db2, br2.grad, what="conv bias grad <- NCHW dy")


@pytest.mark.parametrize("M", [3, 64, 100, 128, 1024, 1500])
def test_linear_wgrad_grouped(M):
    """dvae_linear_wgrad_grouped: the six FC weight gradients of a training step (encoder lin1/lin2/mu_logvar_gen,
    decoder lin1/lin2/lin3: encoders.py:63-67, decoders.py:53-55) + ragged shapes in ONE launch, vs fp64."""
    shapes = [(512, 256), (256, 512), (256, 256), (10, 256), (256, 20), (7, 33), (100, 36), (256, 256)]   # (K, N)
    probs, refs, outs = [], [], []
    for q, (K, N) in enumerate(shapes):
        x, dy = _rand(M, K, seed=10 + q), _rand(M, N, seed=30 + q)
        xd, dyd = dev(x), dev(dy)
        dw, db = torch.full((N, K), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
        if q == 5:
            db = None                                  # db is optional
        probs.append((ptr(xd), ptr(dyd), ptr(dw), ptr(db), M, K, N))
        refs.append((dy.double().t() @ x.double(), dy.double().sum(0)))
        outs.append((dw, db))
    arr, addr = _lib.wgrad_descs(probs)
    call("dvae_linear_wgrad_grouped", addr, len(probs), stream())
    for q, ((dw, db), (rw, rb)) in enumerate(zip(outs, refs)):
        check(dw, rw, rtol=1e-5, atol_rel=2e-6, what="grouped wgrad dw[%d] M=%d" % (q, M))
        if db is not None:
            check(db, rb, rtol=1e-5, atol_rel=2e-6, what="grouped wgrad db[%d] M=%d" % (q, M))
    # a sub-group (3 problems) gives bit-identical results: the decomposition of a problem does not depend on its neighbours
    dw2, db2 = torch.empty_like(outs[0][0]), torch.empty_like(outs[0][1])
    p0 = probs[0]
    arr2, addr2 = _lib.wgrad_descs([probs[3], (p0[0], p0[1], ptr(dw2), ptr(db2), M, p0[5], p0[6]), probs[4]])
    call("dvae_linear_wgrad_grouped", addr2, 3, stream())
    assert torch.equal(dw2, outs[0][0]) and torch.equal(db2, outs[0][1])


def test_data_parallel_glue_kernels():
    """dvae_axpby (out = alpha a [+ beta b], in place allowed) and dvae_swap_outer (src [A][Bn][inner] -> dst [Bn][A][inner]):
    the element-wise glue of the sharded step (disvae_amd/parallel.py) against torch."""
    g = torch.Generator().manual_seed(3)
    for n in (1, 7, 1280, 65537):
        a, b = torch.randn(n, generator=g), torch.randn(n, generator=g)
        ad, bd, od = dev(a), dev(b), dev(torch.zeros(n))
        call("dvae_axpby", ptr(od), ptr(ad), 0.125, ptr(bd), 7.0, n, stream())
        torch.testing.assert_close(od.cpu(), 0.125 * a + 7.0 * b, rtol=2e-7, atol=1e-6)     # (the device contracts to an fma)
        call("dvae_axpby", ptr(od), ptr(ad), 3.0, None, 0.0, n, stream())
        assert torch.equal(od.cpu(), 3.0 * a)
        call("dvae_axpby", ptr(ad), ptr(ad), 8.0, None, 0.0, n, stream())             # in place: the mirrored world's x world_size
        assert torch.equal(ad.cpu(), 8.0 * a)
        call("dvae_axpby", ptr(bd), ptr(bd), 1.0, ptr(od), 7.0, n, stream())          # out aliases a: out += beta * b
        torch.testing.assert_close(bd.cpu(), b + 7.0 * (3.0 * a), rtol=2e-7, atol=1e-6)
    for A, Bn, inner in ((8, 3, 1280), (2, 8, 1280), (3, 5, 7), (1, 4, 33), (4, 1, 10)):
        src = torch.randn(A, Bn, inner, generator=g)
        sd, dd = dev(src), dev(torch.zeros(Bn, A, inner))
        call("dvae_swap_outer", ptr(sd), ptr(dd), A, Bn, inner, stream())
        assert torch.equal(dd.cpu(), src.permute(1, 0, 2).contiguous()), (A, Bn, inner)


@pytest.mark.parametrize("N,C,H", [(3, 1, 32), (64, 1, 32), (5, 3, 32), (9, 1, 16), (2, 3, 24), (70, 3, 32)])
def test_thin_ends_at_any_size_match_the_plain_generic_kernels(N, C, H):
    """The thin ends of images the tuned 64x64 kernels do not cover (BASELINE configs[0]: 32x32) run on k_down_thin_px /
    k_up_thin_px (conv_generic.hip, round 6: weights in LDS, 16-byte accesses); every output is the same fmaf chain as in the
    plain shape-generic kernels (DVAE_FORCE_GENERIC=1 selects those): conv1 forward (encoders.py:54,73), convT3's input
    gradient and convT3 forward + sigmoid (decoders.py:65,82) bit for bit."""
    Hs = H // 2
    w1, b1 = dev(_rand(32, C, 4, 4, seed=1, scale=0.3)), dev(_rand(32, seed=2, scale=0.1))
    wt, bt = dev(_rand(32, C, 4, 4, seed=3, scale=0.3)), dev(_rand(C, seed=4, scale=0.1))
    x = dev(torch.rand(N, C, H, H, generator=torch.Generator().manual_seed(5)))
    a = dev(torch.relu(_rand(N, Hs, Hs, 32, seed=6)))            # NHWC 32-channel activation (input of convT3 / mask of its dgrad)
    dy = dev(_rand(N, C, H, H, seed=7))
    outs = []
    for generic in (True, False):
        y = torch.full((N, Hs, Hs, 32), 7.0, device=DEV)
        dx = torch.full((N, Hs, Hs, 32), 7.0, device=DEV)
        rec = torch.full((N, C, H, H), 7.0, device=DEV)
        with force_generic(generic):
            call("dvae_conv4s2_fwd", ptr(x), _lib.NCHW, ptr(w1), ptr(b1), ptr(y), _lib.NHWC, N, C, H, H, 32, _lib.ACT_RELU, stream())
            call("dvae_convT4s2_dgrad", ptr(dy), _lib.NCHW, ptr(wt), ptr(a), ptr(dx), _lib.NHWC, N, 32, Hs, Hs, C, stream())
            call("dvae_convT4s2_fwd", ptr(a), _lib.NHWC, ptr(wt), ptr(bt), ptr(rec), _lib.NCHW, N, 32, Hs, Hs, C, _lib.ACT_SIGMOID,
                 stream())
            torch.cuda.synchronize()
        outs.append((y, dx, rec))
    for g, t, what in zip(outs[0], outs[1], ("conv1 forward", "convT3 input gradient", "convT3 forward + sigmoid")):
        assert torch.equal(g, t), what
    assert float(outs[0][0].abs().max()) > 0 and float(outs[0][1].abs().max()) > 0
    ref = torch.sigmoid(torch.nn.functional.conv_transpose2d(a.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), bt.double().cpu(),
                                                             stride=2, padding=1))
    check(outs[1][2], ref, rtol=1e-5, atol_rel=2e-6, what="convT3 forward vs fp64")
