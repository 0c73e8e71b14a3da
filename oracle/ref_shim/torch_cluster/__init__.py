"""Test-only stand-in for torch_cluster.knn via scipy cKDTree.

knn(x, y, k) -> LongTensor [2, |y|*k]: row 0 = index into y (query), row 1 = index into x
(neighbour), grouped by query (torch_cluster documentation). Neighbour order
within a query and tie-breaking are implementation-defined upstream; fixtures use distinct points.
"""
import numpy as np
import torch
from scipy.spatial import cKDTree


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1):
    assert batch_x is None and batch_y is None and not cosine
    xn = x.detach().cpu().double().numpy()
    yn = y.detach().cpu().double().numpy()
    k_eff = min(k, xn.shape[0])
    _, idx = cKDTree(xn).query(yn, k=k_eff)
    idx = np.asarray(idx).reshape(yn.shape[0], k_eff)
    row = np.repeat(np.arange(yn.shape[0]), k_eff)
    return torch.from_numpy(np.stack([row, idx.reshape(-1)], axis=0)).long().to(x.device)
