"""Host-side scalars of the beta-TCVAE estimator (disvae/utils/math.py:54-73).

The reference rebuilds a B x B log-importance-weight matrix on the CPU every call; with
M + 1 == B its strided writes only ever produce three distinct values (column 0 = 1/N,
column 1 = strat, the rest 1/M, plus W[M-1,0] = strat), so the HIP kernel needs just their
fp32 logs.  They are computed here in fp32 exactly like ``torch.Tensor(...).fill_().log()``.
"""
import torch


def log_importance_weights(batch_size, dataset_size):
    """-> float32 tensor [log(1/N), log(strat), log(1/M)] (math.py:66-73)."""
    N = dataset_size
    M = batch_size - 1
    strat_weight = (N - M) / (N * M)
    w = torch.tensor([1 / N, strat_weight, 1 / M], dtype=torch.float32)
    return w.log()
