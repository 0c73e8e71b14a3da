"""The arithmetic of the f16x2 stage-1 kernel (k_stage1_h2) restated in numpy and run inside the oracle's
DataAggregation on a golden fixture: every Linear of the stage with its operands split into fp16 pieces (round to nearest),
the kernel's partial products in the kernel's order, fp32 accumulation per K = 16 step. This is how the form was chosen
before the kernel existed; the test keeps the three facts the design rests on:

* two fp16 pieces per operand and three products (W0 x1, W1' x0 / 16, W0 x0) are as accurate as the exact three-piece bf16 form
  with six products that rounds 1-2 used, and more accurate than plain fp32 evaluation (the reference's own arithmetic);
* without the factor 16 on the second weight piece (an fp16 SUBNORMAL for weights of size 0.1) the error is 1.6x larger;
* truncated pieces (what a first estimate assumed) would not do.
"""
import numpy as np
import pytest
import torch

from oracle import genie_oracle as O
from tests.util import Case

STAGE1 = ("init_trns", "l1_t1_2", "l1_t2_2", "l2_t1_1", "l2_t2_1", "l2_t1_2", "l2_t2_2")


def split_bf16_trunc(x, n):
    out, r = [], x.astype(np.float32).copy()
    for _ in range(n):
        p = (r.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
        out.append(p.astype(np.float64))
        r = (r - p).astype(np.float32)
    return out


def split_f16(x, n, scale=1.0, trunc=False):
    """pieces of x * scale in fp16 (value returned unscaled, as float64): rn16, or truncation toward zero."""
    out, r = [], (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    for _ in range(n):
        p = r.astype(np.float16)
        if trunc:
            over = np.abs(p.astype(np.float32)) > np.abs(r)
            p = np.where(over, np.nextafter(p, np.float16(0)), p).astype(np.float16)
        pf = p.astype(np.float32)
        out.append(pf.astype(np.float64) / scale)
        r = (r - pf).astype(np.float32)
    return out


def split_linear(x, W, b, scheme):
    xn, Wn = x.numpy(), W.numpy()
    if scheme == "bf16x3":          # rounds 1-2: exact operands, six of nine products
        xp, wp, prods = split_bf16_trunc(xn, 3), split_bf16_trunc(Wn, 3), [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]
    elif scheme == "h2":            # the kernel: W1' = rn16(16 (W - W0)) against x0 / 16
        xp = split_f16(xn, 2)
        w0 = split_f16(Wn, 1)[0]
        w1 = split_f16((Wn.astype(np.float64) - w0).astype(np.float32), 1, scale=16.0)[0]
        wp, prods = [w0, w1], [(0, 1), (1, 0), (0, 0)]
    elif scheme == "h2_unscaled":
        xp, wp, prods = split_f16(xn, 2), split_f16(Wn, 2), [(0, 1), (1, 0), (0, 0)]
    elif scheme == "h2_trunc":
        xp, wp, prods = split_f16(xn, 2, trunc=True), split_f16(Wn, 2, scale=16.0, trunc=True), [(0, 1), (1, 0), (0, 0)]
    else:
        raise ValueError(scheme)
    acc = np.broadcast_to(b.numpy().astype(np.float32), (xn.shape[0], Wn.shape[0])).copy()
    for wi, xi in prods:
        for k0 in range(0, xn.shape[1], 16):
            acc = (acc.astype(np.float64) + xp[xi][:, k0:k0 + 16] @ wp[wi][:, k0:k0 + 16].T).astype(np.float32)
    return torch.from_numpy(acc)


def x_latent_rms_error(case, scheme, ref):
    orig = O.linear

    def lin(x, w, prefix):
        parts = prefix.split(".")
        if scheme != "f32" and parts[0] == "DataAggregation" and parts[1] in STAGE1:
            return split_linear(x, w[prefix + ".weight"], w[prefix + ".bias"], scheme)
        return orig(x, w, prefix)

    O.linear = lin
    try:
        out = case.oracle_forward(torch.float32, structured=False)
    finally:
        O.linear = orig
    d = out["x_latent"].double() - ref["x_latent"]
    return float((d ** 2).mean().sqrt())


def test_two_fp16_pieces_are_as_accurate_as_three_exact_bf16_pieces():
    case = Case("o1_20x500")            # every Linear weight x 2.1: outputs O(1), the fixture on which 1e-5 absolute is a real bound
    ref = case.oracle_forward(torch.float64, structured=False)
    err = {s: x_latent_rms_error(case, s, ref) for s in ("f32", "bf16x3", "h2", "h2_unscaled", "h2_trunc")}
    print(err)
    assert err["h2"] < err["f32"], err                       # better than the reference's own fp32 evaluation
    assert err["h2"] < 1.35 * err["bf16x3"], err             # fp32-class like the exact form (observed 0.81e-7 vs 0.68e-7)
    assert err["h2_unscaled"] > 1.4 * err["h2"], err         # why the second weight piece is scaled (observed 1.30e-7)
    assert err["h2_trunc"] > 3.0 * err["h2"], err            # truncated pieces lose two bits
