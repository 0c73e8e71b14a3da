#!/usr/bin/env python
"""Feasibility probe for CU partitioning: a HIP stream restricted to a CU mask (hipExtStreamCreateWithCUMask) wrapped as a
torch ExternalStream; (1) does a bandwidth-bound torch op slow down in proportion to the mask, (2) the standalone G-sized tail
on masks of 8 / 16 / 32 / 64 CUs (is a small dedicated partition enough for one tail per window?), (3) which XCDs the enabled
CUs of a mask belong to (genie_debug_xcc_map needs a tuning build; skipped otherwise)."""
import ctypes, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def spread_mask(n_cu, total=256):
    """n_cu CUs spread evenly over the CU index range."""
    bits = 0
    for k in range(n_cu):
        bits |= 1 << ((k * total) // n_cu)
    return bits


def main():
    dev = "cuda:0"
    x = torch.empty(1 << 28, device=dev)            # 1 GiB
    y = torch.empty_like(x)
    for n in (256, 128, 64, 16):
        s = masked_stream(spread_mask(n))
        with torch.cuda.stream(s):
            for _ in range(2): y.copy_(x)
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(5): y.copy_(x)
            s.synchronize()
        print("copy 1 GiB on %3d CUs: %.3f ms" % (n, (time.perf_counter() - t0) / 5 * 1e3))
    from genie_amd import _lib, module, synthetic
    from genie_amd.engine import _ptr
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
    torch.manual_seed(0)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev).eval()
    locs, xg = torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(dev), locs, xg)
    win = synthetic.make_window(geom, n_picks, seed=2)
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    xq, tq = torch.from_numpy(geom.x_query).float().to(dev), torch.from_numpy(geom.t_query).float().to(dev)
    hp = net._hip
    with torch.no_grad():
        net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
        torch.cuda.synchronize()
        knn = net.SpatialAttention.query_table(xq, xg, 10)
        bip = torch.empty((G, 15), device=dev); xs1 = torch.empty((G, 30), device=dev)
        for n in (256, 64, 32, 16, 8):
            s = masked_stream(spread_mask(n))
            with torch.cuda.stream(s):
                st = ctypes.c_void_p(s.cuda_stream)
                def single():
                    _lib.check(hp.lib.genie_bipartite_readout(hp.ctx, _ptr(bip), hp._ws_ptr, st), "bip")
                    _lib.check(hp.lib.genie_spatial_agg3_fwd(hp.ctx, _ptr(bip), _ptr(xg), _ptr(xs1), hp._ws_ptr, st), "sa")
                    hp.readout_grid(xs1, tq); hp.readout_query(xs1, xg, xq, knn, tq)
                for _ in range(3): single()
                s.synchronize(); t0 = time.perf_counter()
                for _ in range(20): single()
                s.synchronize()
            print("per-window tail on %3d CUs [TAIL_RO=%s TAIL_SA=%s]: %.1f us" % (n, os.environ.get("GENIE_TAIL_RO"), os.environ.get("GENIE_TAIL_SA"), (time.perf_counter() - t0) / 20 * 1e6))


if __name__ == "__main__":
    main()
