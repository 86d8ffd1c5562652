"""kNN near-tie margin of a molecule (shared by the windowed chain test and tools/chain_divergence.py)."""
import numpy as np


def knn_margin_rel(x, k=8):
    """min over atoms of (d2_{k+1} - d2_k) / d2_k in float64; inf if the molecule has <= k + 1 atoms
    (every other atom is then a neighbour: nothing to choose)."""
    x = np.asarray(x, np.float64)
    n = len(x)
    if n <= k + 1:
        return np.inf
    d = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    s = np.sort(d, 1)
    return float(((s[:, k] - s[:, k - 1]) / s[:, k - 1]).min())
