"""Helpers for the -m gpu parity tests: every call goes through the C-ABI (ctypes)."""
import os
import contextlib

import numpy as np
import torch

from disvae_amd import _lib
from disvae_amd._lib import call, ptr

DEV = "cuda"


def stream():
    return torch.cuda.current_stream().cuda_stream


_KEEP = []   # device tensors created by dev() stay alive until the test ends: a temporary freed
             # right after ptr() would be re-used by the next allocation before the launch


def dev(t):
    d = t.detach().to(DEV, torch.float32).contiguous()
    _KEEP.append(d)
    return d


def keep(t):
    _KEEP.append(t)
    return t


def nhwc(t):
    """NCHW torch tensor -> NHWC-contiguous device buffer."""
    return dev(t.permute(0, 2, 3, 1).contiguous())


def clear_keep():
    del _KEEP[:]


def from_nhwc(t, N, C, H, W):
    return t.view(N, H, W, C).permute(0, 3, 1, 2).cpu()


@contextlib.contextmanager
def force_generic(flag):
    old = os.environ.get("DVAE_FORCE_GENERIC")
    os.environ["DVAE_FORCE_GENERIC"] = "1" if flag else "0"
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("DVAE_FORCE_GENERIC", None)
        else:
            os.environ["DVAE_FORCE_GENERIC"] = old


STATS = {}   # what -> worst (max abs err / max |ref|, err / tolerance) seen; dumped by conftest.py (DVAE_PARITY_STATS)


def record_stat(what, rel_to_max, frac_of_tol):
    cur = STATS.get(what, (0.0, 0.0))
    STATS[what] = (max(cur[0], float(rel_to_max)), max(cur[1], float(frac_of_tol)))


def check(got, ref, rtol=1e-4, atol_rel=2e-5, what=""):
    """got: device/cpu fp32 tensor; ref: cpu tensor (fp64 preferred).
    |got - ref| <= rtol * |ref| + atol_rel * max|ref|, element-wise."""
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite values"
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    tol = atol_rel * scale + rtol * ref.abs()
    record_stat(what, err.max().item() / (scale + 1e-300), (err / (tol + 1e-300)).max().item())
    bad = err > tol
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError("%s: %d/%d mismatches, max abs err %.3e (scale %.3e), first at %s: got %.6e ref %.6e"
                             % (what, int(bad.sum()), bad.numel(), err.max().item(), scale, idx,
                                got[tuple(idx)].item(), ref[tuple(idx)].item()))
