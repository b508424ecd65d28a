"""Test infrastructure (like everything under oracle/): gate-matched comparison of the HIP engine with the fp64 oracle.

ReLU' / LeakyReLU' are discontinuous at 0: a unit whose pre-activation is within fp32 rounding of zero is gated
differently by any two arithmetics, which changes gradients by a finite amount (the reference's own fp32 torch-CPU
gradients differ from fp64 by 1e-5 .. 3e-4 of max|g| for that reason, tools/fp32_vs_fp64_oracle.py).  The whole-step
checks therefore evaluate the fp64 oracle with the ENGINE's on/off pattern (oracle.gates) and check the pattern itself
separately.  Used by tests/, __graft_entry__.smoke() and bench.py's parity_check -- never by the product path.
"""
from collections import defaultdict


def engine_gates(model, B, splits=None, dec_rows=None):
    """ReLU on/off pattern of the native forward that just ran (engine workspace), in reference layout (NCHW).
    splits: row ranges in the oracle's call order (factor: data1 then data2); dec_rows: rows the decoder ran on."""
    eng = model.engine
    buf = eng.buffers(B)
    splits = splits or [slice(0, B)]
    dec = dec_rows or slice(0, B)
    g = {}
    last = len(eng.enc_names) - 1
    for k, (n, act) in enumerate(zip(eng.enc_names, buf.enc_act)):
        t = buf.a_flat.view(B, 32, 4, 4) if k == last else act.permute(0, 3, 1, 2)
        g["encoder." + n] = [(t[sl] > 0).cpu() for sl in splits]
    g["encoder.lin1"] = [(buf.h1[sl] > 0).cpu() for sl in splits]
    g["encoder.lin2"] = [(buf.h2[sl] > 0).cpu() for sl in splits]
    for n, t in (("lin1", buf.d1), ("lin2", buf.d2), ("lin3", buf.d3)):
        g["decoder." + n] = [(t[dec] > 0).cpu()]
    for n, act in zip(eng.dec_names, buf.dec_act):
        g["decoder." + n] = [(act.permute(0, 3, 1, 2)[dec] > 0).cpu()]
    return g


def discriminator_gates(disc, M, Bh):
    """LeakyReLU on/off pattern of Discriminator.forward_raw(zin, M = 2 Bh): D(z1) first, then D(z_perm)."""
    hs = disc._acts[M]["h"]
    return {"disc.lin%d" % (i + 1): [(hs[i][:Bh] > 0).cpu(), (hs[i][Bh:2 * Bh] > 0).cpu()] for i in range(5)}


def gate_mismatches(gates, log, eps=1e-5):
    """log: [(layer, fp64 pre-activation)] of the UN-gated fp64 oracle in call order.  Returns (number of units the engine
    gates differently, worst |pre-activation| / layer scale among them, ok = all within eps of zero)."""
    seen = defaultdict(int)
    n_diff, worst = 0, 0.0
    for name, pre in log:
        if name not in gates:
            continue
        gate = gates[name][seen[name]]
        seen[name] += 1
        diff = gate != (pre > 0)
        if diff.any():
            n_diff += int(diff.sum())
            worst = max(worst, (pre.abs()[diff].max() / pre.abs().max()).item())
    return n_diff, worst, worst <= eps
