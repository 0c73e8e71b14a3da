import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genie_amd import engine
from tests.util import Case
DEV="cuda:0"
c = Case("cfg1_20x500")
S, G, w = c.S, c.G, c.weights
sta_nbr, src_nbr = c.tables()
def run(env):
    for k in ("GENIE_S1", "GENIE_S2", "GENIE_NOFAST", "GENIE_NOFAST2"):
        os.environ.pop(k, None)
    os.environ.update(env)
    hp = engine.HipPath(S, G, engine.csr_from_table(sta_nbr), engine.csr_from_table(src_nbr), grid_order=engine.morton_order(c.x_grid.numpy()), device=DEV)
    hp.set_weights({k: v.to(DEV) for k, v in w.items()})
    hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV))
    cc, wv = hp.export(0).cpu().numpy(), hp.export(2).cpu().numpy()
    xl, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    return xl.cpu().numpy(), cc, wv
a2 = float(w["DataAggregation.activate2.weight"])
nbr = src_nbr.numpy()
for name, env in (("generic", {"GENIE_S1": "f32", "GENIE_NOFAST": "1", "GENIE_NOFAST2": "1"}), ("fast", {"GENIE_S1": "f32"})):
    xl, cc, wv = run(env)
    wv3 = wv.reshape(G, S, 15)
    s = np.zeros((G, S, 15), np.float32)
    for k in range(15):
        s = (s + wv3[nbr[:, k]]).astype(np.float32)
    c2 = cc.reshape(G, S, 30)[:, :, 15:]
    inv = np.float32(1.0) / np.float32(15.0)
    sep = (c2 + (s * inv).astype(np.float32)).astype(np.float32)
    fma = (c2.astype(np.float64) + s.astype(np.float64) * np.float64(inv)).astype(np.float32)
    pre = lambda o: np.where(o > 0, o, np.float32(a2) * o).astype(np.float32)
    got = xl.reshape(G, S, 30)[:, :, 15:]
    print(name, "matches separate mul+add:", int((pre(sep) != got).sum()), "mismatches; matches fma:", int((pre(fma) != got).sum()))
