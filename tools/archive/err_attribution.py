"""Where does the fp32 error of (y, x) come from? Oracle-level attribution on a golden fixture: every stage of forward_fixed_source
once in fp64 with all others in fp32, and once in fp32 with all others in fp64, against the reference fp64 run (CPU only).
Usage: python tools/err_attribution.py"""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from tests.util import Case, max_abs
from oracle import genie_oracle as O
torch.set_num_threads(8)
c = Case("o1_20x500")
y64, x64 = c.ref("y64"), c.ref("x64")
D, F = torch.float64, torch.float32
def run(stage_dtype):
    # stage_dtype: dict name -> dtype for stages: da, bip, sa1, sa2, sa3, ro_grid(y_latent+TA), sat, ta_q
    w64 = {k: v.to(D) for k, v in c.weights.items()}
    w32 = c.weights
    W = lambda t: w64 if t == D else w32
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    t = stage_dtype['da']
    da = O.data_aggregation(W(t), c.Slice.to(t), c.Mask.to(t), A_in_sta, A_in_src, full=True)
    xl = da["x_latent"]
    t = stage_dtype['bip']; bip = O.bipartite_read_in(W(t), xl.to(t), c.edge_attr.to(t), A_src_in_prod, c.Mask.to(t))
    t = stage_dtype['sa1']; sa1 = O.spatial_aggregation(W(t), bip.to(t), c.A_src_src, c.x_grid.to(t), "SpatialAggregation1")
    t = stage_dtype['sa2']; sa2 = O.spatial_aggregation(W(t), sa1.to(t), c.A_src_src, c.x_grid.to(t), "SpatialAggregation2")
    t = stage_dtype['sa3']; sa3 = O.spatial_aggregation(W(t), sa2.to(t), c.A_src_src, c.x_grid.to(t), "SpatialAggregation3")
    t = stage_dtype['rog']; yl = O.spatial_direct(W(t), sa3.to(t)); y = O.temporal_attention(W(t), yl, c.t_query.to(t))
    t = stage_dtype['sat']; xq = O.spatial_attention(W(t), sa3.to(t), c.x_query.to(t), c.x_grid.to(t))
    t = stage_dtype['taq']; x = O.temporal_attention(W(t), xq.to(t), c.t_query.to(t))
    return max_abs(y, y64), max_abs(x, x64), float((y.double()-y64).pow(2).mean().sqrt())
names = ['da','bip','sa1','sa2','sa3','rog','sat','taq']
print("all f32:", run({n: F for n in names}))
print("all f64:", run({n: D for n in names}))
for n in names:
    d = {m: F for m in names}; d[n] = D
    print("f64 only in %s:" % n, run(d))
for n in names:
    d = {m: D for m in names}; d[n] = F
    print("f32 only in %s:" % n, run(d))
