"""Timing of the FC chain with / without the 4x4 conv ends in the same launch (dvae_fc_chain_fwd / _bwd, conv_in / convT_gout
fields) against the three-launch sequences they replace: python tools/ab_chain_ends.py [rows ...]"""
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "disentangling-vae_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from disvae_amd import _lib
from disvae_amd._lib import call, ptr
import test_gpu_fused_core as T

DEV = "cuda"
D = 10
s = torch.cuda.current_stream().cuda_stream
for n in [int(a) for a in sys.argv[1:]] or [128, 1024]:
    shapes, W, Bv = T._fc_params(D, seed=7)
    ent = T._fc_stage(shapes, W)
    f = lambda *sh: torch.rand(*sh, device=DEV) - 0.3
    wc, wt = f(32, 32, 4, 4) * 0.2, f(32, 32, 4, 4) * 0.2
    bc, bt_ = f(32), f(32)
    img = {k: torch.empty(16384, device=DEV) for k in ("c_down", "c_up", "t_down", "t_up")}
    T._stage([(wc, img["c_down"], img["c_up"]), (wt, img["t_down"], img["t_up"])])
    bd = {k: v.to(DEV) for k, v in Bv.items()}
    eps = torch.randn(n, D, device=DEV)
    conv_in = torch.relu(f(n, 8, 8, 32))
    out = dict(h1=f(n, 256), h2=f(n, 256), ml=f(n, 2 * D), mu=f(n, D), logvar=f(n, D), z=f(n, D), d1=f(n, 256), d2=f(n, 256), d3=f(n, 512))
    a_flat, up = f(n, 512), f(n, 8, 8, 32)
    kl = torch.empty(_lib.KL_FLOATS, device=DEV)
    base = dict(a_flat=ptr(a_flat), eps=ptr(eps), kl_part=ptr(kl) + 64, n_enc=n, n_kl=n, n_dec=n, D=D,
                **{"w_" + k: ptr(ent[k][1]) for k in shapes}, **{"b_" + k: ptr(bd[k]) for k in shapes}, **{k: ptr(v) for k, v in out.items()})
    st0, a0 = _lib.struct_of(_lib.FcChainFwdArgs, **base)
    st1, a1 = _lib.struct_of(_lib.FcChainFwdArgs, conv_in=ptr(conv_in), conv_w=ptr(img["c_down"]), conv_b=ptr(bc),
                             convT_w=ptr(img["t_up"]), convT_b=ptr(bt_), convT_out=ptr(up), **base)

    def fwd_seq():
        call("dvae_conv32_down", ptr(conv_in), ptr(img["c_down"]), ptr(bc), None, ptr(a_flat), _lib.NCHW, n, 4, _lib.ACT_RELU, s)
        call("dvae_fc_chain_fwd", a0, s)
        call("dvae_conv32_up", ptr(out["d3"]), _lib.NCHW, ptr(img["t_up"]), ptr(bt_), None, ptr(up), n, 4, _lib.ACT_RELU, s)

    gout, conv_act = f(n, 8, 8, 32), torch.relu(f(n, 8, 8, 32))
    acts = {k: torch.relu(f(n, w)) for k, w in [("d2", 256), ("d1", 256), ("h2", 256), ("h1", 256), ("a_flat", 512), ("d3", 512)]}
    mu, lv, dz2 = f(n, D), f(n, D), f(n, D)
    scal = torch.zeros(_lib.NSCAL, device=DEV); scal[_lib.S_KLW] = 1.7
    coef = torch.zeros(_lib.NCOEF, device=DEV); coef[_lib.C_INV_B] = 1.0 / n
    bo = dict(gd2=f(n, 256), gd1=f(n, 256), dz=f(n, D), dml=f(n, 2 * D), gh2=f(n, 256), gh1=f(n, 256), ga_flat=f(n, 512))
    gd3, gin = f(n, 512), f(n, 8, 8, 32)
    ins = dict(gd3=gd3, mu=mu, logvar=lv, eps=eps, dz2=dz2, scal=scal, coef=coef, **{k: v for k, v in acts.items() if k != "d3"})
    bbase = dict(n=n, D=D, **{"w_" + k: ptr(ent[k][2]) for k in shapes}, **{k: ptr(v) for k, v in ins.items()}, **{k: ptr(v) for k, v in bo.items()})
    sb0, b0 = _lib.struct_of(_lib.FcChainBwdArgs, **bbase)
    sb1, b1 = _lib.struct_of(_lib.FcChainBwdArgs, convT_gout=ptr(gout), convT_w=ptr(img["t_down"]), d3=ptr(acts["d3"]),
                             conv_w=ptr(img["c_up"]), conv_act=ptr(conv_act), conv_gin=ptr(gin), **bbase)

    def bwd_seq():
        call("dvae_conv32_down", ptr(gout), ptr(img["t_down"]), None, ptr(acts["d3"]), ptr(gd3), _lib.NCHW, n, 4, _lib.ACT_NONE, s)
        call("dvae_fc_chain_bwd", b0, s)
        call("dvae_conv32_up", ptr(bo["ga_flat"]), _lib.NCHW, ptr(img["c_up"]), None, ptr(conv_act), ptr(gin), n, 4, _lib.ACT_NONE, s)

    fns = {"fwd 3 launches": fwd_seq, "fwd chain alone": lambda: call("dvae_fc_chain_fwd", a0, s), "fwd fused": lambda: call("dvae_fc_chain_fwd", a1, s),
           "bwd 3 launches": bwd_seq, "bwd chain alone": lambda: call("dvae_fc_chain_bwd", b0, s), "bwd fused": lambda: call("dvae_fc_chain_bwd", b1, s)}
    for name, fn in fns.items():
        for _ in range(100):
            fn()
        res = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 200 * 1e3)
        print("rows %5d  %-16s : %s us" % (n, name, " ".join("%.1f" % r for r in res)), flush=True)
