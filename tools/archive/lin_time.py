#!/usr/bin/env python
"""Time genie_linear_bwd_wb against the library GEMM + column sum it replaces, at the product-sized shapes of config 2."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genie_amd import engine
dev = "cuda:0"
e = torch.zeros((2, 0), dtype=torch.long)
hp = engine.HipPath(3, 4, engine.csr_from_edges(e, 3), engine.csr_from_edges(e, 4), device=dev)
N = 2_000_000
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for K, M in ((8, 30), (33, 30), (60, 30), (64, 30), (94, 15)):
    x, dy = torch.randn((N, K), device=dev), torch.randn((N, M), device=dev)
    W = torch.randn((M, K), device=dev)
    t_lib = tm(lambda: (dy.t() @ x, dy.sum(0)))
    t_dx = tm(lambda: dy @ W)
    t_hip = tm(lambda: hp.linear_bwd_wb(x, dy))
    print("K=%3d M=%2d: torch dW+db %.3f ms, torch dX %.3f ms, genie_linear_bwd_wb %.3f ms (%.0f GB/s)" % (
        K, M, t_lib, t_dx, t_hip, N * (K + M) * 4 / t_hip / 1e6))
