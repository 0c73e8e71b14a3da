import sys, warnings, traceback
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from genie_amd import graph, module, synthetic
dev = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev); net.train()
def sample(i):
    S, G = 60, 800
    geom = synthetic.Geometry(S - (i % 3), G, L=300e3, n_query=200, seed=100 + i)
    smp = synthetic.training_sample(geom, 600, n_src=4, seed=3 + i)
    A1, A2, A3, A4 = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, geom.n_sta, G, device=dev)
    ea = graph.GraphEdges(x=t(geom.edge_attr()), edge_index=A3)
    eaf = graph.GraphEdges(x=ea.x, edge_index=A3.flip(0).contiguous())
    return (t(smp["Slice"]), t(smp["Mask"]), A1, A2, ea, eaf, A4, torch.from_numpy(geom.A_src_src).to(dev), t(smp["A_edges_p"]).long(),
            t(smp["A_edges_s"]).long(), t(smp["dt_partition"]), t(smp["tlatent"]), t(smp["tpick"]), t(smp["ipick"]).long(), t(smp["phase_label"]),
            t(geom.locs), t(geom.x_grid), t(geom.x_query), t(smp["x_query_src"]), t(geom.t_query), t(smp["tq_sample"]), t(smp["trv_out_q"]))
ss = [sample(i) for i in range(3)]
for s in ss:
    out = net(*s); sum(o.sum() for o in out).backward()
torch.cuda.synchronize()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "genie_amd" in f.filename]
    print("SYNC:", str(message)[:60], "<-", " | ".join("%s:%d" % (f.filename.split("/")[-1], f.lineno) for f in st[-4:]))
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
out = net(*ss[0])
loss = sum(o.sum() for o in out)
loss.backward()
torch.cuda.set_sync_debug_mode("default")
print("done")
