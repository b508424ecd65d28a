"""-m gpu: parity AT THE SIZES bench.py RUNS (BASELINE.json configs[1..4]).

* whole training iterations at btcvae 64x64x1 B=256, factor 64x64x1 B=256 (tensor 128+128), btcvae
  64x64x3 B=1024, factor 64x64x3 B=2048 (tensor 1024+1024): loss / logged scalars vs the fp32 oracle (= the
  reference's arithmetic, training.py:137-164, losses.py:243-313,356-391), gradients and activations vs
  the fp64 oracle;
* every persistent conv / convT / wgrad kernel with enough units that each workgroup runs >= 4 iterations
  of its steady-state loop (grids are min(n_units, 256) workgroups of 64-pixel units: HS=16 needs
  N >= 260 images, HS=8 N >= 1100, HS=4 N >= 4200; thin kernels: 8 units per image), tuned path only,
  plus odd image counts that leave a ragged last unit.

Stated fp32 tolerances here: kernels rtol 1e-5 + 2e-6 max|ref| vs fp64; loss scalars rtol 1e-5 vs the fp32
oracle; whole-step gradients rtol 1e-5 + 2e-6 max|g| vs the fp64 oracle EVALUATED WITH THE ENGINE'S ReLU
ON/OFF PATTERN (oracle.gates): ReLU' is discontinuous at 0, so a unit whose pre-activation is within fp32
rounding of zero is gated differently by ANY two arithmetics -- the reference's own fp32 torch-CPU gradients
differ from fp64 by 1e-5 .. 3e-4 of max|g| for that reason (tools/fp32_vs_fp64_oracle.py), 100x the error of the
engine's kernels.  The pattern itself is checked: wherever the engine's gate differs from sign(fp64
pre-activation), that pre-activation must be within 1e-5 of the layer's scale of zero.  The measured errors
behind these numbers are dumped with DVAE_PARITY_STATS=<file> (profiles/r02_parity_stats.json)."""
from collections import defaultdict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import *  # noqa
from gpu_util import _lib, record_stat  # noqa
from oracle import disvae_oracle as O
from oracle.gate_match import engine_gates, discriminator_gates
from disvae_amd.models.vae import init_specific_model
from disvae_amd.models.losses import get_loss_f

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25,
          betaB_G=1000, factor_G=6.4, latent_dim=10, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)

K_RTOL, K_ATOL = 1e-5, 2e-6       # kernels vs fp64
G_RTOL, G_ATOL = 1e-5, 2e-6       # whole-step gradients vs the gate-matched fp64 oracle
GATE_EPS = 1e-5                   # |fp64 pre-activation| / layer scale of a unit the engine gates differently


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def check_grad(got, ref, what):
    check(got, ref, rtol=G_RTOL, atol_rel=G_ATOL, what=what)


def check_gate_pattern(gates, log, what):
    """log: [(layer, fp64 pre-activation)] of the UN-gated fp64 oracle in call order.  Units gated differently by
    the engine must sit within GATE_EPS x the layer's scale of zero."""
    seen = defaultdict(int)
    n_diff = 0
    for name, pre in log:
        if name not in gates:
            continue
        gate = gates[name][seen[name]]
        seen[name] += 1
        diff = gate != (pre > 0)
        if diff.any():
            n_diff += int(diff.sum())
            worst = (pre.abs()[diff].max() / pre.abs().max()).item()
            record_stat(what + " gate-mismatch |pre|/scale " + name, worst, worst / GATE_EPS)
            assert worst <= GATE_EPS, "%s: %s gated differently at |pre-activation| = %.2e of the layer scale" % (what, name, worst)
    record_stat(what + " units gated differently than fp64 (count)", n_diff, 0.0)


def _native(loss, img, seed, n_data, lr, lr_disc):
    torch.manual_seed(seed)
    model = init_specific_model("Burgess", img, 10)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    loss_f = get_loss_f(loss, n_data=n_data, device=torch.device(DEV), lr_disc=lr_disc, **HP)
    loss_f.replay = None            # the eager launch sequence = what bench.py times at these sizes
    model.to(DEV)
    model.train()
    return model, opt, loss_f


# BASELINE.json configs[1] and configs[3] (hyperparam.ini:123-137): dataset size, lr
@pytest.mark.parametrize("name,img,B,n_data", [("btcvae_dsprites", (1, 64, 64), 256, 737280),
                                               ("btcvae_celeba", (3, 64, 64), 1024, 202599)])
def test_btcvae_step_at_baseline_batch(name, img, B, n_data):
    seed, lr = 1234, 5e-4
    model, opt, loss_f = _native("btcvae", img, seed, n_data, lr, 1e-4)
    torch.manual_seed(seed)
    p0 = O.init_vae_params(img, 10)
    hp = dict(HP, n_data=n_data)
    gen = torch.Generator().manual_seed(seed + 1)
    data = torch.rand((B,) + tuple(img), generator=gen)
    eps = torch.randn(B, 10, generator=gen)
    st = lambda: O.LossState(steps_anneal=HP["reg_anneal"])
    ref_loss, ref_logs, _, _ = O.train_iteration_grads("btcvae", hp, st(), O.clone_params(p0, requires_grad=True), data, eps)
    storer = defaultdict(list)
    out = loss_f.fused_step(dev(data), model, opt, storer, eps=dev(eps))
    buf = model.engine.buffers(B)
    np.testing.assert_allclose(out.item(), ref_loss.item(), rtol=1e-5, err_msg="loss")
    assert list(storer.keys()) == list(ref_logs.keys())
    for k in ref_logs:
        np.testing.assert_allclose(storer[k][0], ref_logs[k].item(), rtol=2e-5, atol=2e-6, err_msg=k)
    gates = engine_gates(model, B)
    log = []
    p64 = O.clone_params(p0, dtype=torch.float64)
    with torch.no_grad(), O.gates(None, record=log):
        O.vae_forward(p64, data.double(), eps.double())
    check_gate_pattern(gates, log, name)
    with O.gates(gates):
        _, _, g64, outs64 = O.train_iteration_grads("btcvae", hp, st(), O.clone_params(p0, dtype=torch.float64, requires_grad=True),
                                                    data.double(), eps.double())
    for k in ("mu", "logvar", "z", "recon"):
        check(getattr(buf, k), outs64[k], rtol=K_RTOL, atol_rel=K_ATOL, what="%s act %s" % (name, k))
    for k, p in model.named_parameters():
        check_grad(p.grad, g64[k], "%s grad %s" % (name, k))


# BASELINE.json configs[2] and configs[4]: the tensor handed to _train_iteration is the doubled batch (main.py:190-193)
@pytest.mark.parametrize("name,img,B,n_data,lr_disc", [("factor_dsprites", (1, 64, 64), 256, 737280, 1e-4),
                                                       ("factor_celeba", (3, 64, 64), 2048, 202599, 1e-5)])
def test_factor_step_at_baseline_batch(name, img, B, n_data, lr_disc):
    seed, lr = 1234, 1e-4
    model, opt, loss_f = _native("factor", img, seed, n_data, lr, lr_disc)
    torch.manual_seed(seed)
    p0 = O.init_vae_params(img, 10)
    d0 = O.init_disc_params(10)
    hp = dict(HP, n_data=n_data, lr_disc=lr_disc)
    gen = torch.Generator().manual_seed(seed + 1)
    Bh = B // 2
    data = torch.rand((B,) + tuple(img), generator=gen)
    eps1, eps2 = torch.randn(Bh, 10, generator=gen), torch.randn(Bh, 10, generator=gen)
    perms = torch.stack([torch.randperm(Bh, generator=gen) for _ in range(10)])
    st = lambda: O.LossState(steps_anneal=HP["reg_anneal"])
    ref_loss, ref_logs, _, _, _ = O.factor_iteration_grads(hp, st(), O.clone_params(p0, requires_grad=True),
                                                           O.clone_params(d0, requires_grad=True), data, eps1, eps2, list(perms))
    storer = defaultdict(list)
    out = loss_f.call_optimize(dev(data), model, opt, storer, noise=(dev(eps1), dev(eps2), perms))
    buf = model.engine.buffers(B)
    np.testing.assert_allclose(out.item(), ref_loss.item(), rtol=1e-5, err_msg="loss")
    assert list(storer.keys()) == list(ref_logs.keys())
    for k in ref_logs:
        np.testing.assert_allclose(storer[k][0], ref_logs[k].item(), rtol=2e-5, atol=2e-6, err_msg=k)
    # the encoder ran on both halves (data1 then data2 in the oracle's call order), the decoder on data1, the
    # discriminator on [z1; z_perm]
    gates = engine_gates(model, B, splits=[slice(0, Bh), slice(Bh, 2 * Bh)], dec_rows=slice(0, Bh))
    gates.update(discriminator_gates(loss_f.discriminator, 2 * Bh, Bh))
    c64 = lambda p, rg: O.clone_params(p, dtype=torch.float64, requires_grad=rg)
    log = []
    with O.gates(None, record=log):
        O.factor_iteration_grads(hp, st(), c64(p0, True), c64(d0, True), data.double(), eps1.double(), eps2.double(), list(perms))
    check_gate_pattern(gates, log, name)
    with O.gates(gates):
        _, _, g64, gd64, outs64 = O.factor_iteration_grads(hp, st(), c64(p0, True), c64(d0, True), data.double(),
                                                            eps1.double(), eps2.double(), list(perms))
    check(buf.z[:Bh], outs64["z1"], rtol=K_RTOL, atol_rel=K_ATOL, what=name + " act z1")
    check(buf.z[Bh:2 * Bh], outs64["z2"], rtol=K_RTOL, atol_rel=K_ATOL, what=name + " act z2")
    for k, p in model.named_parameters():
        check_grad(p.grad, g64[k], "%s grad %s" % (name, k))
    for k, p in loss_f.discriminator.named_parameters():
        check_grad(p.grad, gd64[k], "%s disc grad %s" % (name, k))


# ---- kernels: >= 4 steady-state iterations per persistent workgroup ---------------------------
@pytest.mark.parametrize("N,Cin,H", [
    (1024, 32, 32), (261, 32, 32),          # HS=16: 4096 / 1044 units over 256 workgroups
    (1100, 32, 16), (1027, 32, 16),         # HS=8
    (4200, 32, 8), (4099, 32, 8),           # HS=4 (ragged last unit: 4099 % 4 = 3 images)
    (1024, 3, 64), (263, 1, 64),            # conv1 (thin)
])
def test_conv_persistent_loops(N, Cin, H):
    Cout = 32
    xl = _lib.NHWC if Cin == 32 else _lib.NCHW
    x = _rand(N, Cin, H, H, seed=1)
    w = _rand(Cout, Cin, 4, 4, seed=2, scale=0.2)
    b = _rand(Cout, seed=3, scale=0.1)
    xd = nhwc(x) if xl == _lib.NHWC else dev(x)
    wd, bd = dev(w), dev(b)
    tag = "conv N=%d C=%d H=%d " % (N, Cin, H)
    y = torch.empty(N, H // 2, H // 2, Cout, device=DEV)
    call("dvae_conv4s2_fwd", ptr(xd), xl, ptr(wd), ptr(bd), ptr(y), _lib.NHWC, N, Cin, H, H, Cout, _lib.ACT_RELU, stream())
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, stride=2, padding=1)
    check(from_nhwc(y, N, Cout, H // 2, H // 2), torch.relu(pre), rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "fwd")
    dy = _rand(N, Cout, H // 2, H // 2, seed=4)
    dyd = nhwc(dy)
    pre.backward(dy.double())
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)
    dw, db = torch.full((Cout, Cin, 4, 4), 7.0, device=DEV), torch.full((Cout,), 7.0, device=DEV)
    call("dvae_conv4s2_wgrad", ptr(xd), xl, ptr(dyd), _lib.NHWC, ptr(dw), ptr(db), N, Cin, H, H, Cout, ptr(ws), stream())
    check(dw, wr.grad, rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "wgrad")
    check(db, br.grad, rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "bias grad")
    if Cin == 32:
        xact = torch.relu(_rand(N, Cin, H, H, seed=5))
        dx = torch.empty(N, H, H, Cin, device=DEV)
        call("dvae_conv4s2_dgrad", ptr(dyd), _lib.NHWC, ptr(wd), ptr(nhwc(xact)), ptr(dx), _lib.NHWC, N, Cin, H, H, Cout, stream())
        check(from_nhwc(dx, N, Cin, H, H), xr.grad * (xact > 0), rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "dgrad")


@pytest.mark.parametrize("N,H,Cout", [
    (1024, 16, 32), (261, 16, 32), (1100, 8, 32), (1027, 8, 32), (4200, 4, 32), (4099, 4, 32),
    (1024, 32, 3), (263, 32, 1),            # convT3 (thin), sigmoid epilogue
])
def test_convT_persistent_loops(N, H, Cout):
    Cin = 32
    yl, act = (_lib.NHWC, _lib.ACT_RELU) if Cout == 32 else (_lib.NCHW, _lib.ACT_SIGMOID)
    x = torch.relu(_rand(N, Cin, H, H, seed=1))
    w = _rand(Cin, Cout, 4, 4, seed=2, scale=0.2)
    b = _rand(Cout, seed=3, scale=0.1)
    xd, wd, bd = nhwc(x), dev(w), dev(b)
    H2 = 2 * H
    tag = "convT N=%d H=%d C=%d " % (N, H, Cout)
    y = torch.empty((N, H2, H2, Cout) if yl == _lib.NHWC else (N, Cout, H2, H2), device=DEV)
    call("dvae_convT4s2_fwd", ptr(xd), _lib.NHWC, ptr(wd), ptr(bd), ptr(y), yl, N, Cin, H, H, Cout, act, stream())
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = F.conv_transpose2d(xr, wr, br, stride=2, padding=1)
    ref = torch.relu(pre) if act == _lib.ACT_RELU else torch.sigmoid(pre)
    check(from_nhwc(y, N, Cout, H2, H2) if yl == _lib.NHWC else y, ref, rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "fwd")
    dy = _rand(N, Cout, H2, H2, seed=4)
    pre.backward(dy.double())
    dyd = nhwc(dy) if yl == _lib.NHWC else dev(dy)
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)
    dw, db = torch.full((Cin, Cout, 4, 4), 7.0, device=DEV), torch.full((Cout,), 7.0, device=DEV)
    call("dvae_convT4s2_wgrad", ptr(xd), _lib.NHWC, ptr(dyd), yl, ptr(dw), ptr(db), N, Cin, H, H, Cout, ptr(ws), stream())
    check(dw, wr.grad, rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "wgrad")
    check(db, br.grad, rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "bias grad")
    dx = torch.empty(N, H, H, Cin, device=DEV)
    call("dvae_convT4s2_dgrad", ptr(dyd), yl, ptr(wd), ptr(xd), ptr(dx), _lib.NHWC, N, Cin, H, H, Cout, stream())
    check(from_nhwc(dx, N, Cin, H, H), xr.grad * (x > 0), rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "dgrad")


@pytest.mark.parametrize("N,C", [(1024, 3), (263, 1)])
def test_convT_sigmoid_recon_fused_persistent(N, C):
    H = 32
    x = torch.relu(_rand(N, 32, H, H, seed=1))
    w = _rand(32, C, 4, 4, seed=2, scale=0.2)
    b = _rand(C, seed=3, scale=0.1)
    tgt = torch.rand(N, C, 2 * H, 2 * H, generator=torch.Generator().manual_seed(4))
    coef = torch.zeros(_lib.NCOEF); coef[_lib.C_INV_B] = 1.0 / N
    recon = torch.empty(N, C, 2 * H, 2 * H, device=DEV)
    g = torch.empty_like(recon)
    parts = torch.full((_lib.REC_NPART,), 7.0, device=DEV)
    call("dvae_convT4s2_sigmoid_recon_fwd", ptr(nhwc(x)), _lib.NHWC, ptr(dev(w)), ptr(dev(b)), ptr(dev(tgt)), ptr(recon),
         ptr(g), _lib.REC["bernoulli"], ptr(dev(coef)), ptr(parts), N, 32, H, H, C, stream())
    lr = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1).requires_grad_(True)
    ref_recon = torch.sigmoid(lr)
    loss = O.reconstruction_loss(tgt.double(), ref_recon, "bernoulli")
    loss.backward()
    tag = "fused recon N=%d C=%d " % (N, C)
    check(recon, ref_recon, rtol=K_RTOL, atol_rel=K_ATOL, what=tag + "recon")
    check(parts.sum() / N, loss, rtol=1e-5, what=tag + "loss")
    check(g, lr.grad, rtol=2e-5, atol_rel=K_ATOL, what=tag + "dL/dlogit")


@pytest.mark.parametrize("N", [1024, 4099])
def test_4x4_end_nchw_persistent(N):
    """the NCHW-writing / NCHW-reading variants of the HS=4 kernels at bench scale (k_down32<4> out_nchw,
    k_up32<4> / k_wgrad32<4> small_nchw)."""
    C = 32
    ws = torch.empty(_lib.lib().dvae_conv_wgrad_ws_floats(), device=DEV)
    x = _rand(N, C, 8, 8, seed=1)
    w = _rand(C, C, 4, 4, seed=2, scale=0.2)
    b = _rand(C, seed=3, scale=0.1)
    y = torch.empty(N, C, 4, 4, device=DEV)
    call("dvae_conv4s2_fwd", ptr(nhwc(x)), _lib.NHWC, ptr(dev(w)), ptr(dev(b)), ptr(y), _lib.NCHW, N, C, 8, 8, C, _lib.ACT_RELU, stream())
    check(y, torch.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)), rtol=K_RTOL, atol_rel=K_ATOL,
          what="N=%d conv fwd -> NCHW" % N)
    xa = torch.relu(_rand(N, C, 4, 4, seed=5))
    wt = _rand(C, C, 4, 4, seed=6, scale=0.2)
    bt = _rand(C, seed=8, scale=0.1)
    dy = _rand(N, C, 8, 8, seed=7)
    xr, wr, br = xa.double().requires_grad_(True), wt.double().requires_grad_(True), bt.double().requires_grad_(True)
    pre = F.conv_transpose2d(xr, wr, br, stride=2, padding=1)
    yT = torch.empty(N, 8, 8, C, device=DEV)
    call("dvae_convT4s2_fwd", ptr(dev(xa)), _lib.NCHW, ptr(dev(wt)), ptr(dev(bt)), ptr(yT), _lib.NHWC, N, C, 4, 4, C, _lib.ACT_RELU, stream())
    check(from_nhwc(yT, N, C, 8, 8), torch.relu(pre), rtol=K_RTOL, atol_rel=K_ATOL, what="N=%d convT fwd <- NCHW" % N)
    pre.backward(dy.double())
    dx = torch.empty(N, C, 4, 4, device=DEV)
    call("dvae_convT4s2_dgrad", ptr(nhwc(dy)), _lib.NHWC, ptr(dev(wt)), ptr(dev(xa)), ptr(dx), _lib.NCHW, N, C, 4, 4, C, stream())
    check(dx, xr.grad * (xa > 0), rtol=K_RTOL, atol_rel=K_ATOL, what="N=%d convT dgrad -> NCHW" % N)
    dw, db = torch.full((C, C, 4, 4), 7.0, device=DEV), torch.full((C,), 7.0, device=DEV)
    call("dvae_convT4s2_wgrad", ptr(dev(xa)), _lib.NCHW, ptr(nhwc(dy)), _lib.NHWC, ptr(dw), ptr(db), N, C, 4, 4, C, ptr(ws), stream())
    check(dw, wr.grad, rtol=K_RTOL, atol_rel=K_ATOL, what="N=%d convT wgrad <- NCHW x" % N)
    check(db, br.grad, rtol=K_RTOL, atol_rel=K_ATOL, what="N=%d convT bias grad" % N)


# ---------------------------------------------------------------------------------------------------------------------------
# against numbers RECORDED FROM THE REAL REFERENCE at a bench size (tests/golden/make_golden.py --bench-size)
REF_REC_GRAD_BOUND = 6e-4     # two fp32 arithmetics on the same step differ by up to 3e-4 of max|g| EACH from fp64 (ReLU units
                              # within rounding of zero gate differently: tools/fp32_vs_fp64_oracle.py, decoder.convT2.weight)


@pytest.mark.parametrize("name,loss", [("btcvae_dsprites_b256", "btcvae"), ("factor_dsprites_b256", "factor")])
def test_step0_vs_reference_recorded_numbers_at_batch_256(name, loss):
    """BASELINE configs[1] / configs[2] at their own batch: the first training iteration of the native engine against what the
    unmodified reference's Trainer._train_iteration produced on the same seed, batch and recorded noise -- initial weights bit
    for bit, loss and logged scalars to 1e-5, every gradient tensor's digest (24 sampled entries, sum, abs-sum) to the bound
    two fp32 arithmetics can be held to: 6e-4 of the tensor's max|g| per entry (REF_REC_GRAD_BOUND), 1e-3 of the abs-sum for
    the sums.  The 1e-5 gradient bar proper is the gate-matched fp64 oracle's (tests above)."""
    from golden_util import load, tensor_digest
    g = load(name)
    img, B = (1, 64, 64), 256
    seed = int(g["seed"])
    model, opt, loss_f = _native(loss, img, seed, int(g["n_data"]), float(g["lr"]), 1e-4)
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(tensor_digest(v)[2:], g["init_digest/" + k][2:], err_msg=k)
    data = torch.rand((B,) + img, generator=torch.Generator().manual_seed(seed + 1))
    storer = defaultdict(list)
    if loss == "factor":
        for k, v in loss_f.discriminator.state_dict().items():
            np.testing.assert_array_equal(tensor_digest(v)[2:], g["dinit_digest/" + k][2:], err_msg=k)
        noise = (dev(torch.from_numpy(g["step0/randn1"])), dev(torch.from_numpy(g["step0/randn2"])), torch.from_numpy(g["step0/perms"]))
        out = loss_f.call_optimize(dev(data), model, opt, storer, noise=noise)
    else:
        out = loss_f.fused_step(dev(data), model, opt, storer, eps=dev(torch.from_numpy(g["step0/randn0"])))
    np.testing.assert_allclose(out.item(), g["step0/loss"], rtol=1e-5)
    assert set(storer) == {k.split("/")[-1] for k in g if k.startswith("step0/storer/")}
    for k, v in storer.items():
        np.testing.assert_allclose(v[0], g["step0/storer/" + k], rtol=1e-5, atol=1e-6, err_msg=k)

    def digests(named, prefix):
        for k, p in named:
            got, want = tensor_digest(p.grad), g["step0/%s_digest/%s" % (prefix, k)]
            amax = float(g["step0/%s_absmax/%s" % (prefix, k)])
            err = np.abs(got[2:] - want[2:]).max()
            record_stat("%s %s vs reference-recorded digest %s" % (name, prefix, k), err / amax, err / (REF_REC_GRAD_BOUND * amax))
            assert err <= REF_REC_GRAD_BOUND * amax, (name, k, err, amax)
            assert abs(got[0] - want[0]) <= 1e-3 * want[1] and abs(got[1] - want[1]) <= 1e-3 * want[1], (name, k, got[:2], want[:2])

    digests(model.named_parameters(), "grad")
    if loss == "factor":
        digests(loss_f.discriminator.named_parameters(), "dgrad")
