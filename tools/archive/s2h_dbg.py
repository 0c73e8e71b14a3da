import sys, torch
sys.path.insert(0, '/root/repo')
from genie_amd import engine
from tests.util import Case, max_abs
DEV = "cuda:0"
c = Case("cfg1_20x500")
sta_nbr, src_nbr = c.tables()
hp = engine.HipPath(c.S, c.G, engine.csr_from_table(sta_nbr), engine.csr_from_table(src_nbr), grid_order=engine.morton_order(c.x_grid.numpy()), device=DEV)
hp.set_weights({k: v.to(DEV) for k, v in c.weights.items()})
print(hp.stage_precision())
hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV))
xl, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
o = c.oracle_forward(torch.float32, structured=True)
d = (xl.cpu() - o["x_latent"]).abs()
print("per-channel max err:", [round(float(v), 5) for v in d.max(0)[0]])
print("row0 got", xl[0].cpu().numpy().round(4)); print("row0 ref", o["x_latent"][0].numpy().round(4))
print("bip err", max_abs(bip.cpu(), o["bip"]))
