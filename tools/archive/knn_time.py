#!/usr/bin/env python
"""genie_knn timing: the refine pass's fresh 112 000-point query set against a 10 000-node grid (process_continuous_days.py:926-980)
and the config-4 base graph (50 000 nodes with themselves), next to the fp64 torch search it replaces."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import engine  # noqa: E402


def torch_knn(xc, xq, k):
    xc, xq = xc.double() / 1000.0, xq.double() / 1000.0
    idx = torch.empty((xq.shape[0], k), dtype=torch.long, device=xq.device)
    chunk = max(1, min(xq.shape[0], int(4e7 // max(1, xc.shape[0]))))
    for a in range(0, xq.shape[0], chunk):
        d = ((xq[a:a + chunk, None, :] - xc[None, :, :]) ** 2).sum(-1)
        idx[a:a + chunk] = torch.topk(d, k, dim=1, largest=False, sorted=True)[1]
    return idx


def main():
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    for nc, nq, k, self_ in ((10000, 10000, 10, False), (10000, 112000, 10, False), (50000, 50000, 15, True), (2000, 2000, 8, True)):
        xc = torch.from_numpy(rng.uniform(0, 300e3, (nc, 3)).astype(np.float32)).to(dev)
        xq = xc if self_ else torch.from_numpy(rng.uniform(0, 300e3, (nq, 3)).astype(np.float32)).to(dev)
        for f, name in ((lambda: engine.knn_device(xc, xq, k, exclude_self=self_), "genie_knn"),
                        (lambda: torch_knn(xc, xq, k + (1 if self_ else 0)), "torch fp64 topk")):
            f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            print("%-16s context %6d query %6d k %2d: %.3f ms" % (name, nc, nq, k, (time.perf_counter() - t0) / 3 * 1e3))


if __name__ == "__main__":
    main()
