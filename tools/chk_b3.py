import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from genie_amd import engine, synthetic
from tests.util import Case
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg1_20x500"
S, G, n_picks, L, nq = synthetic.CONFIGS[cfg]
geom = synthetic.Geometry(S, G, L=L, n_query=10, seed=1)
win = synthetic.make_window(geom, n_picks, seed=2)
dev = "cuda:0"
Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
ea, pos = torch.from_numpy(geom.edge_attr()).to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
w = {k: v.to(dev) for k, v in Case("cfg1_20x500").weights.items()}
sta = engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S)
src = engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G)
order = engine.morton_order(geom.x_grid)
res = {}
for mode in ("f32", "b3"):
    os.environ["GENIE_S1"] = mode
    hp = engine.HipPath(S, G, sta, src, grid_order=order, device=dev)
    hp.set_weights(w)
    dbg = hp.da_stage1(Slice, Mask, debug=True)
    xl, bip = hp.da_stage2_bipartite(Mask, ea, want_x_latent=True)
    res[mode] = (dbg, xl, bip)
    ts = []
    for i in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hp.da_stage1(Slice, Mask); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(mode, "stage1 ms", np.median(ts), flush=True)
def flat(x):
    out = []
    if isinstance(x, (tuple, list)):
        for y in x: out += flat(y)
    elif isinstance(x, dict):
        for k in sorted(x): out += flat(x[k])
    elif torch.is_tensor(x) and x.dtype == torch.float32: out.append(x)
    return out
for a_, b_ in zip(flat(res["f32"]), flat(res["b3"])):
    print(tuple(a_.shape), "max|diff|", float((a_ - b_).abs().max()), "max|ref|", float(a_.abs().max()), "equal", bool(torch.equal(a_, b_)))
