"""Helpers shared by the golden-vector tests (digest = sum, abs-sum, sampled entries)."""
import os
import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def tensor_digest(t, n_samples=24):
    f = t.detach().cpu().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n_samples, f.numel())).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def assert_digest_close(got, want, rtol, atol_scale=1e-6, what=""):
    """Compare digests: sampled entries element-wise, sums with an abs-sum-scaled atol."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    abssum = max(abs(want[1]), 1e-30)
    n = max(len(want) - 2, 1)
    # sum / abs-sum: error budget relative to the abs-sum
    assert abs(got[0] - want[0]) <= rtol * abssum + 1e-12, (what, "sum", got[0], want[0], abssum)
    assert abs(got[1] - want[1]) <= rtol * abssum + 1e-12, (what, "abssum", got[1], want[1])
    scale = abssum / n  # typical magnitude... only used for atol
    mx = np.max(np.abs(want[2:])) if len(want) > 2 else 0.0
    atol = atol_scale * max(mx, scale) + 1e-12
    np.testing.assert_allclose(got[2:], want[2:], rtol=rtol, atol=atol, err_msg=what)
