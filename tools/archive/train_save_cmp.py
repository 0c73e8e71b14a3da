#!/usr/bin/env python
"""Saved pre-activations of the training forward: f16x2 stage 1 vs the fp32-MFMA stage 1 (GENIE_S1=f32), block by block."""
import os, sys, subprocess
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
if len(sys.argv) > 1:
    from genie_amd import module, synthetic
    from tests.util import Case
    DEV = "cuda:0"
    S, G, Q = 33, 257, 100
    geom = synthetic.Geometry(S, G, L=200e3, n_query=Q, seed=3)
    win = synthetic.make_window(geom, 700, seed=4)
    w0 = Case("o1_20x500").weights
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), t(geom.locs), t(geom.x_grid))
    net._hip.sync_weights(net._path_params)
    r, xl, save = net._hip.train_fwd(t(win["Slice"]), t(win["Mask"]), net._edge_attr)
    torch.cuda.synchronize()
    np.save(sys.argv[1], save.cpu().numpy())
    sys.exit(0)
for mode in ("h2", "f32"):
    env = dict(os.environ)
    if mode == "f32":
        env["GENIE_S1"] = "f32"
    subprocess.run([sys.executable, __file__, "/tmp/save_%s.npy" % mode], env=env, check=True)
a, b = np.load("/tmp/save_h2.npy"), np.load("/tmp/save_f32.npy")
P = 33 * 257
a, b = a.reshape(14, P, 16), b.reshape(14, P, 16)
names = ["z0a", "z0b", "t1a", "t1b", "t2a", "t2b", "upa", "upb", "vpa", "vpb", "o1", "o2", "zba", "zbb"]
for k in range(14):
    d = np.abs(a[k] - b[k])
    i = np.unravel_index(np.argmax(d), d.shape)
    flips = int(((a[k] > 0) != (b[k] > 0)).sum())
    print("%-4s max|diff| %.3e at node %d ch %d (h2 %.6e f32 %.6e)  max|ref| %.3e  sign flips %d" % (names[k], d.max(), i[0], i[1], a[k][i], b[k][i], np.abs(b[k]).max(), flips))
