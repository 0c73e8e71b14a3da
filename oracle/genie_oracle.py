"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (PyTorch CPU tensors, fp32 or fp64, no PyG / torch_scatter / torch_cluster) of the
reference's station<->source-grid message-passing path and of the read-out heads needed to return
`(y, x)` from `GCN_Detection_Network_extended.forward_fixed_source`
(`/root/reference/Code/module.py:999-1020`). Every function cites the reference lines it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module,
and only as the checker / reported baseline. `genie_amd/` never imports it.

Pinning: the reference has no tests and no golden vectors, and the arithmetic of
`propagate` / `scatter` / `knn` / `softmax` lives in un-vendored, un-pinned third-party packages
(`module.py:14-20`; `Code/install_dependencies.txt:7-17`) -> at that boundary the reference itself
leaves parity unpinned. This oracle is pinned against outputs of the reference's own `module.py`
executed in the build container (PyG replaced by the documented-semantics shim in `oracle/ref_shim`)
and committed as fixtures under `tests/golden/` by `oracle/make_golden.py`
(`tests/test_oracle_golden.py` checks every one of them).

Two formulations are provided and checked against each other:
* "literal": explicit `[2, E]` product edge lists, gather + scatter-mean, exactly the reference dataflow;
* "structured": base kNN tables on a `[G, S, C]` view (no product edge lists), used at sizes where the
  literal form does not fit in memory.

Weights are a dict keyed by the reference's `state_dict` names (e.g. `DataAggregation.init_trns.weight`).
"""
import math

import numpy as np
import torch
from scipy.spatial import cKDTree

SCALE_REL = 30000.0          # config.yaml:73
KERNEL_SIG_T = 3.0           # train_config.yaml:17
SCALE_T = 3.0 * KERNEL_SIG_T  # module.py:40


def prelu(x, a):
    """nn.PReLU with a single scalar slope: max(0,x) + a*min(0,x)."""
    return torch.where(x >= 0, x, a * x)


def linear(x, w, prefix):
    """nn.Linear `prefix` (weight [out,in], bias [out])."""
    return x @ w[prefix + ".weight"].T + w[prefix + ".bias"]


def act(x, w, prefix):
    return prelu(x, w[prefix + ".weight"])


def scatter_sum(msg, index, n):
    out = torch.zeros((n,) + tuple(msg.shape[1:]), dtype=msg.dtype)
    return out.index_add_(0, index, msg)


def scatter_mean(msg, index, n):
    """torch_scatter 'mean': sum / clamp(count, 1); empty segment -> 0."""
    s = scatter_sum(msg, index, n)
    cnt = torch.zeros(n, dtype=msg.dtype).index_add_(0, index, torch.ones(index.shape[0], dtype=msg.dtype))
    return s / cnt.clamp(min=1).view((-1,) + (1,) * (msg.dim() - 1))


def propagate_mean(x, edge_index):
    """MessagePassing('mean').propagate(edge_index, x=x) with the default message x_j
    (module.py:54,90): out[i] = mean_{(j,i) in E} x[j]."""
    return scatter_mean(x[edge_index[0]], edge_index[1], x.shape[0])


# ----------------------------------------------------------------------------------------------
# a-1  DataAggregation.forward  (module.py:85-98)
# ----------------------------------------------------------------------------------------------
def data_aggregation(w, Slice, Mask, A_in_sta, A_in_src, pre="DataAggregation", full=False):
    tr = torch.cat((Slice, Mask), dim=-1)                                        # :87
    h0 = act(linear(tr, w, pre + ".init_trns"), w, pre + ".activate")            # :88
    n1 = propagate_mean(act(h0, w, pre + ".activate11"), A_in_sta)               # :90
    n2 = propagate_mean(act(h0, w, pre + ".activate12"), A_in_src)               # :91
    tr1 = linear(torch.cat((h0, n1, Mask), dim=1), w, pre + ".l1_t1_2")          # :90
    tr2 = linear(torch.cat((h0, n2, Mask), dim=1), w, pre + ".l1_t2_2")          # :91
    h1 = act(torch.cat((tr1, tr2), dim=1), w, pre + ".activate1")                # :92
    u = act(linear(h1, w, pre + ".l2_t1_1"), w, pre + ".activate21")             # :94
    v = act(linear(h1, w, pre + ".l2_t2_1"), w, pre + ".activate22")             # :95
    o1 = linear(torch.cat((h1, propagate_mean(u, A_in_sta), Mask), dim=1), w, pre + ".l2_t1_2")  # :94
    o2 = linear(torch.cat((h1, propagate_mean(v, A_in_src), Mask), dim=1), w, pre + ".l2_t2_2")  # :95
    x_latent = act(torch.cat((o1, o2), dim=1), w, pre + ".activate2")            # :96
    if full:
        return {"h0": h0, "h1": h1, "u": u, "v": v, "x_latent": x_latent}
    return x_latent


def absolute_pos_inputs(Slice, locs, x_grid, A_src_in_sta, scale_rel=SCALE_REL):
    """`use_absolute_pos: True` (config.yaml:92): module.py:1007 appends the station and the source position of every product
    node, divided by 3 * scale_rel, to Slice (in_channels 4 -> 10)."""
    return torch.cat((Slice, locs[A_src_in_sta[0]] / (3.0 * scale_rel), x_grid[A_src_in_sta[1]] / (3.0 * scale_rel)), dim=1)


def edge_pos_features(pos, edge_index, node_of=None, scale_rel=SCALE_REL):
    """Edge features of the `use_updated_model_definition` variant (module.py:1059-1072 / :1102-1111): for an edge j -> i,
    d = pos[j] - pos[i] (3), |d| (1), each through phi(d) = sign(d) exp(-d^2 / (2 scale_rel^2)). `node_of` maps product
    nodes to base nodes (A_src_in_sta[0] for stations, [1] for source nodes)."""
    j, i = edge_index[0], edge_index[1]
    if node_of is not None:
        j, i = node_of[j], node_of[i]
    rel = pos[j] - pos[i]
    rel = torch.cat((rel, torch.norm(rel, dim=1, keepdim=True)), dim=1)
    return torch.sign(rel) * torch.exp(-0.5 * (rel ** 2) / (scale_rel ** 2))


def propagate_mean_edges(x, edge_attr, edge_index):
    """MessagePassing('mean') whose message is cat(x_j, edge_attr) (module.py:162-172)."""
    return scatter_mean(torch.cat((x[edge_index[0]], edge_attr), dim=1), edge_index[1], x.shape[0])


def data_aggregation_edges(w, Slice, Mask, A_in_sta, A_in_src, pos_rel_sta, pos_rel_src, pre="DataAggregation", full=False):
    """DataAggregationEdges.forward (module.py:141-160): l1_t?_2 [30,68] and l2_t?_2 [15,98] see
    cat(node, mean_j cat(x_j, edge features), Mask)."""
    tr = torch.cat((Slice, Mask), dim=-1)                                                          # :143
    h0 = act(linear(tr, w, pre + ".init_trns"), w, pre + ".activate")                              # :144
    n1 = propagate_mean_edges(act(h0, w, pre + ".activate11"), pos_rel_sta, A_in_sta)              # :157
    n2 = propagate_mean_edges(act(h0, w, pre + ".activate12"), pos_rel_src, A_in_src)              # :158
    tr1 = linear(torch.cat((h0, n1, Mask), dim=1), w, pre + ".l1_t1_2")
    tr2 = linear(torch.cat((h0, n2, Mask), dim=1), w, pre + ".l1_t2_2")
    h1 = act(torch.cat((tr1, tr2), dim=1), w, pre + ".activate1")                                  # :159
    u = act(linear(h1, w, pre + ".l2_t1_1"), w, pre + ".activate21")
    v = act(linear(h1, w, pre + ".l2_t2_1"), w, pre + ".activate22")
    o1 = linear(torch.cat((h1, propagate_mean_edges(u, pos_rel_sta, A_in_sta), Mask), dim=1), w, pre + ".l2_t1_2")   # :161
    o2 = linear(torch.cat((h1, propagate_mean_edges(v, pos_rel_src, A_in_src), Mask), dim=1), w, pre + ".l2_t2_2")   # :162
    x_latent = act(torch.cat((o1, o2), dim=1), w, pre + ".activate2")                              # :163
    if full:
        return {"h0": h0, "h1": h1, "u": u, "v": v, "x_latent": x_latent}
    return x_latent


def _gather_mean_sta(x3, sta_nbr):
    """x3 [G,S,C]; mean over station neighbours within the same source node: [G,S,C]."""
    if sta_nbr.shape[1] == 0:
        return torch.zeros_like(x3)
    return x3[:, sta_nbr.long(), :].mean(dim=2)


def _gather_mean_src(x3, src_nbr):
    """x3 [G,S,C]; mean over source-node neighbours for the same station: [G,S,C]."""
    if src_nbr.shape[1] == 0:
        return torch.zeros_like(x3)
    out = torch.zeros_like(x3)
    for k in range(src_nbr.shape[1]):          # accumulate in edge order
        out += x3[src_nbr[:, k].long()]
    return out / src_nbr.shape[1]


def data_aggregation_structured(w, Slice, Mask, sta_nbr, src_nbr, n_sta, n_grid,
                                pre="DataAggregation", full=False):
    """Same as `data_aggregation` on the full Cartesian product graph, but from the base kNN tables
    `sta_nbr[S,ks]`, `src_nbr[G,kp]` (identity (4) of SURVEY.md Appendix A); p = g*S + s."""
    S, G = n_sta, n_grid

    def v3(t):
        return t.view(G, S, -1)

    def v2(t):
        return t.reshape(G * S, -1)

    tr = torch.cat((Slice, Mask), dim=-1)
    h0 = act(linear(tr, w, pre + ".init_trns"), w, pre + ".activate")
    n1 = v2(_gather_mean_sta(v3(act(h0, w, pre + ".activate11")), sta_nbr))
    n2 = v2(_gather_mean_src(v3(act(h0, w, pre + ".activate12")), src_nbr))
    tr1 = linear(torch.cat((h0, n1, Mask), dim=1), w, pre + ".l1_t1_2")
    tr2 = linear(torch.cat((h0, n2, Mask), dim=1), w, pre + ".l1_t2_2")
    h1 = act(torch.cat((tr1, tr2), dim=1), w, pre + ".activate1")
    u = act(linear(h1, w, pre + ".l2_t1_1"), w, pre + ".activate21")
    v = act(linear(h1, w, pre + ".l2_t2_1"), w, pre + ".activate22")
    n1b = v2(_gather_mean_sta(v3(u), sta_nbr))
    n2b = v2(_gather_mean_src(v3(v), src_nbr))
    o1 = linear(torch.cat((h1, n1b, Mask), dim=1), w, pre + ".l2_t1_2")
    o2 = linear(torch.cat((h1, n2b, Mask), dim=1), w, pre + ".l2_t2_2")
    x_latent = act(torch.cat((o1, o2), dim=1), w, pre + ".activate2")
    if full:
        return {"h0": h0, "h1": h1, "u": u, "v": v, "x_latent": x_latent}
    return x_latent


# ----------------------------------------------------------------------------------------------
# a-2  BipartiteGraphOperator.forward  (module.py:224-229)
# ----------------------------------------------------------------------------------------------
def bipartite_read_in(w, x_latent, edge_attr, edge_index, Mask, pre="Bipartite_ReadIn"):
    N = int(edge_index[0].max().item()) + 1                                      # :226
    M = int(edge_index[1].max().item()) + 1                                      # :227
    assert N == x_latent.shape[0]
    m = Mask.max(1, keepdim=True)[0]                                             # :229
    msg = m * act(linear(torch.cat((x_latent, edge_attr), dim=-1), w, pre + ".fc1"), w, pre + ".activate1")
    r = scatter_sum(msg[edge_index[0]], edge_index[1], M)                        # propagate 'add', size=(N,M)
    return act(linear(r, w, pre + ".fc2"), w, pre + ".activate2")


def bipartite_read_in_structured(w, x_latent, edge_attr, Mask, n_sta, n_grid, pre="Bipartite_ReadIn"):
    m = Mask.max(1, keepdim=True)[0]
    msg = m * act(linear(torch.cat((x_latent, edge_attr), dim=-1), w, pre + ".fc1"), w, pre + ".activate1")
    r = msg.view(n_grid, n_sta, -1).sum(dim=1)
    return act(linear(r, w, pre + ".fc2"), w, pre + ".activate2")


# ----------------------------------------------------------------------------------------------
# a-3  SpatialAggregation.forward / message  (module.py:243-249)
# ----------------------------------------------------------------------------------------------
def spatial_aggregation(w, tr, A_src, pos, pre, scale_rel=SCALE_REL):
    j, i = A_src[0], A_src[1]
    p = pos / scale_rel                                                          # :245
    x_j = tr[j]
    c = act(linear(x_j, w, pre + ".fglobal"), w, pre + ".activate3").mean(0, keepdim=True)   # :249 mean over ALL edges
    msg = act(linear(torch.cat((x_j, p[i] - p[j], c.repeat(x_j.shape[0], 1)), dim=-1), w, pre + ".fc1"),
              w, pre + ".activate1")                                             # :249
    a = scatter_mean(msg, i, tr.shape[0])                                        # 'mean' aggregation :233
    return act(linear(torch.cat((tr, a), dim=-1), w, pre + ".fc2"), w, pre + ".activate2")   # :245


# ----------------------------------------------------------------------------------------------
# a-6  read-out heads needed for (y, x)
# ----------------------------------------------------------------------------------------------
def spatial_direct(w, x, pre="SpatialDirect"):
    """module.py:251-260."""
    return act(linear(x, w, pre + ".f_direct"), w, pre + ".activate")


def temporal_attention(w, inpts, t_query, pre="TemporalAttention", n_heads=5, n_latent=15, scale_t=SCALE_T):
    """module.py:325-331 (no softmax: score * value, mean over heads)."""
    H, L = n_heads, n_latent
    context = linear(act(linear(inpts, w, pre + ".f_context_1"), w, pre + ".activate1"), w, pre + ".f_context_2").view(-1, H, L)
    values = linear(act(linear(inpts, w, pre + ".f_values_1"), w, pre + ".activate2"), w, pre + ".f_values_2").view(-1, H, L)
    query = linear(act(linear(t_query / scale_t, w, pre + ".temporal_query_1"), w, pre + ".activate3"),
                   w, pre + ".temporal_query_2").view(-1, H, L)
    score = (context.unsqueeze(1) * query.unsqueeze(0)).sum(-1, keepdim=True) / math.sqrt(L)   # [N,T,H,1]
    z = (score * values.unsqueeze(1)).mean(2)                                                  # [N,T,L]
    return linear(act(linear(act(z, w, pre + ".activate4"), w, pre + ".proj_1"), w, pre + ".activate5"), w, pre + ".proj_2")


def knn_edges(x_context, x_query, k):
    """`knn(x_context/1000, x_query/1000, k).flip(0)` (module.py:282): row0 = context j, row1 = query i."""
    xc = x_context.detach().double().numpy() / 1000.0
    xq = x_query.detach().double().numpy() / 1000.0
    k = min(k, xc.shape[0])
    _, idx = cKDTree(xc).query(xq, k=k)
    idx = np.asarray(idx).reshape(xq.shape[0], k)
    row_q = np.repeat(np.arange(xq.shape[0]), k)
    return torch.from_numpy(np.stack([idx.reshape(-1), row_q], axis=0)).long()


def segment_softmax(src, index, n):
    """torch_geometric.utils.softmax."""
    idx = index.view(-1, 1).expand_as(src)
    mx = torch.full((n, src.shape[1]), float("-inf"), dtype=src.dtype).scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    den = torch.zeros((n, src.shape[1]), dtype=src.dtype).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def spatial_attention(w, inpts, x_query, x_context, k=10, pre="SpatialAttention", n_heads=5, n_latent=15,
                      scale_rel=SCALE_REL, edge_index=None):
    """module.py:280-297."""
    H, L = n_heads, n_latent
    if edge_index is None:
        edge_index = knn_edges(x_context, x_query, k)                            # :282
    j, i = edge_index[0], edge_index[1]
    edge_attr = (x_query[i] - x_context[j]) / scale_rel                          # :283
    x_j = inpts[j]
    q = linear(edge_attr, w, pre + ".f_queries").view(-1, H, L)                  # :289
    c = linear(torch.cat((x_j, edge_attr), dim=-1), w, pre + ".f_context").view(-1, H, L)   # :290
    v = linear(torch.cat((x_j, edge_attr), dim=-1), w, pre + ".f_values").view(-1, H, L)    # :291
    alpha = act((q * c).sum(-1) / math.sqrt(L), w, pre + ".activate1")           # :293
    alpha = segment_softmax(alpha, i, x_query.shape[0])                          # :295
    agg = scatter_sum(alpha.unsqueeze(-1) * v, i, x_query.shape[0])              # 'add' :264, size=(ctx, query)
    return act(linear(agg.mean(1), w, pre + ".proj"), w, pre + ".activate2")     # :285


# ----------------------------------------------------------------------------------------------
# a-4  forward_fixed_source  (module.py:999-1020)
# ----------------------------------------------------------------------------------------------
def forward_fixed_source(w, Slice, Mask, A_in_sta, A_in_src, edge_attr, A_src_in_prod, A_src,
                         x_grid_cart, x_query_cart, t_query, full=False, query_edges=None, pos_rel=None):
    """Literal formulation. `use_absolute_pos=False` (config.yaml:92). Returns (y, x) or a dict. `pos_rel` =
    (pos_rel_sta, pos_rel_src) selects the DataAggregationEdges variant (module.py:1163-1185)."""
    if w["DataAggregation.init_trns.weight"].shape[1] == 14 and Slice.shape[1] == 4:
        raise ValueError("use_absolute_pos weights: append the scaled positions to Slice first (absolute_pos_inputs)")
    if pos_rel is not None:
        da = data_aggregation_edges(w, Slice, Mask, A_in_sta, A_in_src, pos_rel[0], pos_rel[1], full=True)
    else:
        da = data_aggregation(w, Slice, Mask, A_in_sta, A_in_src, full=True)                     # :1010
    bip = bipartite_read_in(w, da["x_latent"], edge_attr, A_src_in_prod, Mask)                   # :1011
    sa1 = spatial_aggregation(w, bip, A_src, x_grid_cart, "SpatialAggregation1")                 # :1012
    sa2 = spatial_aggregation(w, sa1, A_src, x_grid_cart, "SpatialAggregation2")                 # :1013
    sa3 = spatial_aggregation(w, sa2, A_src, x_grid_cart, "SpatialAggregation3")                 # :1014
    y_latent = spatial_direct(w, sa3)                                                            # :1015
    y = temporal_attention(w, y_latent, t_query)                                                 # :1016
    xq = spatial_attention(w, sa3, x_query_cart, x_grid_cart, edge_index=query_edges)            # :1017
    x = temporal_attention(w, xq, t_query)                                                       # :1018
    if full:
        out = dict(da)
        out.update({"bip": bip, "sa1": sa1, "sa2": sa2, "sa3": sa3, "y_latent": y_latent, "xq": xq, "y": y, "x": x})
        return out
    return y, x


def forward_fixed_source_structured(w, Slice, Mask, sta_nbr, src_nbr, edge_attr, A_src, x_grid_cart,
                                    x_query_cart, t_query, n_sta, n_grid, full=False, query_edges=None):
    """Structured formulation (no product edge lists); identical math, different summation grouping."""
    da = data_aggregation_structured(w, Slice, Mask, sta_nbr, src_nbr, n_sta, n_grid, full=True)
    bip = bipartite_read_in_structured(w, da["x_latent"], edge_attr, Mask, n_sta, n_grid)
    sa1 = spatial_aggregation(w, bip, A_src, x_grid_cart, "SpatialAggregation1")
    sa2 = spatial_aggregation(w, sa1, A_src, x_grid_cart, "SpatialAggregation2")
    sa3 = spatial_aggregation(w, sa2, A_src, x_grid_cart, "SpatialAggregation3")
    y_latent = spatial_direct(w, sa3)
    y = temporal_attention(w, y_latent, t_query)
    xq = spatial_attention(w, sa3, x_query_cart, x_grid_cart, edge_index=query_edges)
    x = temporal_attention(w, xq, t_query)
    if full:
        out = dict(da)
        out.update({"bip": bip, "sa1": sa1, "sa2": sa2, "sa3": sa3, "y_latent": y_latent, "xq": xq, "y": y, "x": x})
        return out
    return y, x


def weights_from_npz(z, dtype=torch.float32, prefix="w/"):
    """Load a weight dict from an npz whose keys are `w/<state_dict name>`."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(z[k])).to(dtype) for k in z.files if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------------
# f-2  association heads of forward / forward_fixed  (module.py:928-937, :985-995)
# ----------------------------------------------------------------------------------------------
EPS = 5.0 * KERNEL_SIG_T     # module.py:41


def bipartite_read_out(w, y_latent, edge_attr, mask_src, n_sta, pre="BipartiteGraphReadOutOperator", src_of=None):
    """module.py:343-352 with A_Lg_in_src.edge_index = [g(p); p] (one edge per product node, in product order):
    s_p = PReLU2(fc2(mask[g] * PReLU1(fc1([y_latent[g] || edge_attr[p]])))); second output mask[g(p)]. `src_of` = g(p) of an
    irregular product graph (`use_subgraph`); default: the Cartesian numbering p = g * n_sta + s."""
    P = edge_attr.shape[0]
    g = torch.arange(P) // n_sta if src_of is None else src_of.long()
    msg = mask_src[g] * act(linear(torch.cat((y_latent[g], edge_attr), dim=-1), w, pre + ".fc1"), w, pre + ".activate1")
    return act(linear(msg, w, pre + ".fc2"), w, pre + ".activate2"), mask_src[g]


def data_aggregation_association(w, s, latent, mask1, mask2, A_in_sta, A_in_src, pre="DataAggregationAssociationPhase", pos_rel=None):
    """module.py:389-403 (unlike DataAggregation, l1_t1_1 / l1_t2_1 ARE applied before the layer-1 activations). `pos_rel` =
    (pos_rel_sta, pos_rel_src): DataAggregationAssociationPhaseEdges (module.py:444-480), every message carries its edge's
    position features (l?_t?_2 have 4 more input columns, between the mean and the mask)."""
    if pos_rel is None:
        mean1 = lambda x: propagate_mean(x, A_in_sta)
        mean2 = lambda x: propagate_mean(x, A_in_src)
    else:
        mean1 = lambda x: propagate_mean_edges(x, pos_rel[0], A_in_sta)                              # :462, :476
        mean2 = lambda x: propagate_mean_edges(x, pos_rel[1], A_in_src)                              # :463, :480
    mask = torch.cat((mask1, mask2), dim=-1)                                                          # :391
    tr = act(linear(torch.cat((s, latent, mask), dim=-1), w, pre + ".init_trns"), w, pre + ".activate")   # :392-393
    a1 = mean1(act(linear(tr, w, pre + ".l1_t1_1"), w, pre + ".activate11"))                          # :395
    a2 = mean2(act(linear(tr, w, pre + ".l1_t2_1"), w, pre + ".activate12"))                          # :396
    tr1 = linear(torch.cat((tr, a1, mask), dim=1), w, pre + ".l1_t1_2")
    tr2 = linear(torch.cat((tr, a2, mask), dim=1), w, pre + ".l1_t2_2")
    tr = act(torch.cat((tr1, tr2), dim=1), w, pre + ".activate1")                                     # :397
    b1 = mean1(act(linear(tr, w, pre + ".l2_t1_1"), w, pre + ".activate21"))                          # :399
    b2 = mean2(act(linear(tr, w, pre + ".l2_t2_1"), w, pre + ".activate22"))                          # :400
    tr1 = linear(torch.cat((tr, b1, mask), dim=1), w, pre + ".l2_t1_2")
    tr2 = linear(torch.cat((tr, b2, mask), dim=1), w, pre + ".l2_t2_2")
    return act(torch.cat((tr1, tr2), dim=1), w, pre + ".activate2")                                   # :401


def local_slice_collapse(w, A_edges, dt_partition, tpick, ipick, phase_label, inpt, tlatent, pre, k_infer=10, eps=EPS):
    """LocalSliceLgCollapse.forward / message (module.py:623-659), aggr 'mean', use_phase_types = True (config.yaml:91)."""
    n_arvs, l_dt = len(tpick), len(dt_partition)
    dt = dt_partition[1] - dt_partition[0]
    t_index = torch.floor((tpick - dt_partition[0]) / dt).long()                                       # :635
    t_index = ((ipick * l_dt * k_infer + t_index * k_infer).view(-1, 1) + torch.arange(k_infer).view(1, -1)).reshape(-1).long()
    src_index = torch.arange(n_arvs).view(-1, 1).repeat(1, k_infer).view(-1)                           # :638
    e0, e1 = A_edges[t_index].long(), src_index                                                        # :640
    t_rel = tpick[e1] - tlatent[e0, 0]                                                                 # :642
    keep = torch.where(t_rel.abs() < 2.0 * eps)[0]                                                     # :645
    e0, e1 = e0[keep], e1[keep]
    msg = act(linear(torch.cat((inpt[e0], (tpick.view(-1, 1)[e1] - tlatent[e0]) / eps, phase_label[e1]), dim=-1), w, pre + ".fc1"),
              w, pre + ".activate1")                                                                   # :659
    agg = scatter_mean(msg, e1, n_arvs)                                                                # 'mean', size=(N, M=n_arvs)
    return act(linear(agg, w, pre + ".fc2"), w, pre + ".activate2")                                    # :652


def station_source_attention(w, n_src, stime, src_embed, trv_src, arrival_p, arrival_s, tpick, ipick, phase_label,
                             pre="Arrivals", n_heads=3, n_latent=15, eps=EPS):
    """StationSourceAttentionMergedPhases.forward / message (module.py:698-775), use_sparse = True,
    use_neighbor_assoc_edges = False, use_phase_types = True."""
    import itertools
    n_sta, n_arv = trv_src.shape[1], len(tpick)
    H, L = n_heads, n_latent
    ip = ipick.numpy()
    lists = [np.where(ip == u)[0] for u in np.unique(ip)]                                              # :703-705 (same stations, sorted)
    arrival = torch.cat((torch.cat((arrival_p, torch.zeros(1, arrival_p.shape[1])), 0),
                         torch.cat((arrival_s, torch.zeros(1, arrival_s.shape[1])), 0)), dim=1)        # :709-711
    pairs = [np.array(list(itertools.product(l, np.concatenate((l, [n_arv]))))).T for l in lists]      # rows (a; b)
    edges = torch.from_numpy(np.flip(np.hstack(pairs), axis=0).copy()).long()                          # :713 -> rows (b; a)
    n_edge = edges.shape[1]
    edges = edges.repeat(1, n_src) + torch.cat((torch.zeros(1, n_src * n_edge, dtype=torch.long),
                                                (torch.arange(n_src) * n_arv).repeat_interleave(n_edge).view(1, -1)), dim=0)   # :717
    src_index = torch.arange(n_src).repeat_interleave(n_edge)                                          # :718
    atime = torch.cat((tpick, torch.tensor([-eps], dtype=tpick.dtype)))
    stindex = torch.cat((ipick, torch.tensor([n_sta])))
    tsrc_p = torch.cat((trv_src[:, :, 0], -eps * torch.ones(n_src, 1, dtype=trv_src.dtype)), dim=1)
    tsrc_s = torch.cat((trv_src[:, :, 1], -eps * torch.ones(n_src, 1, dtype=trv_src.dtype)), dim=1)
    phase = torch.cat((phase_label, torch.tensor([[-1.0]], dtype=phase_label.dtype)), dim=0)
    rel_p = atime[edges[0]] - (tsrc_p[src_index, stindex[edges[0]]] + stime[src_index])                # :724
    rel_s = atime[edges[0]] - (tsrc_s[src_index, stindex[edges[0]]] + stime[src_index])                # :725
    keep = torch.where((rel_p.abs() < 2.0 * eps) | (rel_s.abs() < 2.0 * eps))[0]                       # :726
    edges, src_index = edges[:, keep].contiguous(), src_index[keep]
    e0, e1 = edges[0], edges[1]
    e0max = int(e0.max().item())
    rel_p = (atime[e0] - (tsrc_p[src_index, stindex[e0]] + stime[src_index])).view(-1, 1)             # :755
    rel_s = (atime[e0] - (tsrc_s[src_index, stindex[e0]] + stime[src_index])).view(-1, 1)             # :758
    t2 = eps ** 2
    fp = torch.cat((torch.exp(-0.5 * rel_p ** 2 / t2), torch.sign(rel_p), phase[e0]), dim=1)           # :756
    fs = torch.cat((torch.exp(-0.5 * rel_s ** 2 / t2), torch.sign(rel_s), phase[e0]), dim=1)           # :759
    self_link = (e0 == torch.remainder(e1, e0max)).view(-1, 1).to(tpick.dtype)                        # :762
    null_link = (e0 == e0max).view(-1, 1).to(tpick.dtype)                                              # :763
    x_j = arrival[e0]
    ctx = linear(act(linear(torch.cat((src_embed[src_index], stime[src_index].view(-1, 1), self_link, null_link), dim=1),
                            w, pre + ".f_src_context_1"), w, pre + ".activate1"), w, pre + ".f_src_context_2").view(-1, H, L)   # :764
    qry = linear(act(linear(torch.cat((x_j, fp, fs), dim=1), w, pre + ".f_arrival_query_1"), w, pre + ".activate2"),
                 w, pre + ".f_arrival_query_2").view(-1, H, L)                                          # :765
    val = linear(act(linear(torch.cat((x_j, fp, fs, self_link, null_link), dim=1), w, pre + ".f_values_1"), w, pre + ".activate3"),
                 w, pre + ".f_values_2").view(-1, H, L)                                                 # :766
    scores = (qry * ctx).sum(-1) / math.sqrt(L)                                                        # :772
    alpha = segment_softmax(scores, e1, n_arv * n_src)                                                 # :773
    agg = scatter_sum(alpha.unsqueeze(-1) * val, e1, n_arv * n_src)                                    # 'add', size=(N, M)
    out = linear(act(linear(agg.mean(1), w, pre + ".proj_1"), w, pre + ".activate4"), w, pre + ".proj_2")   # :745
    return out.view(n_src, n_arv, -1)                                                                  # :747


def forward_fixed(w, Slice, Mask, A_in_sta, A_in_src, edge_attr, A_src_in_prod, A_src, A_edges_p, A_edges_s, dt_partition,
                  tlatent, tpick, ipick, phase_label, x_grid_cart, x_query_cart, x_query_src_cart, t_query, tq_sample, trv_out_q,
                  n_sta, pos_rel=None, abs_pos=None, use_phase_types=True):
    """forward_fixed (module.py:963-997): (y, x, arv_p, arv_s). `use_phase_types=False` (config.yaml:91): the two pick-sized heads
    see phase_label * 0 (module.py:632-633, :706-707). The product graph may be irregular (`use_subgraph`: A_src_in_prod[1] is the
    source node of every product node). `pos_rel` = (pos_rel_sta, pos_rel_src) per product edge: the
    use_updated_model_definition class (module.py:1128-1161); `abs_pos` = (locs, A_src_in_sta): use_absolute_pos (the scaled
    station / source positions appended to Slice, :969-970, and to the association embedding, :987-988)."""
    scaled = None
    if abs_pos is not None:
        locs, A_src_in_sta = abs_pos
        scaled = torch.cat((locs[A_src_in_sta[0]] / (3.0 * SCALE_REL), x_grid_cart[A_src_in_sta[1]] / (3.0 * SCALE_REL)), dim=1)
        Slice = torch.cat((Slice, scaled), dim=1)                                                      # :969-970
    o = forward_fixed_source(w, Slice, Mask, A_in_sta, A_in_src, edge_attr, A_src_in_prod, A_src, x_grid_cart, x_query_cart,
                             t_query, full=True, pos_rel=pos_rel)
    x_src = spatial_attention(w, o["sa3"], x_query_src_cart, x_grid_cart)                              # :981
    mask_out = 1.0 * (o["y"][:, :, 0].max(1, keepdim=True)[0] > 0.01)                                  # :985
    if not use_phase_types:
        phase_label = phase_label * 0.0
    s, m1 = bipartite_read_out(w, o["y_latent"], edge_attr, mask_out.to(Slice.dtype), n_sta, src_of=A_src_in_prod[1])   # :986
    if scaled is not None:
        s = torch.cat((s, scaled), dim=1)                                                              # :987-988
    s = data_aggregation_association(w, s, o["x_latent"].detach(), m1, Mask, A_in_sta, A_in_src, pos_rel=pos_rel)   # :990 (x_latent.detach())
    arv_p = local_slice_collapse(w, A_edges_p, dt_partition, tpick, ipick, phase_label, s, tlatent[:, 0:1], "LocalSliceLgCollapseP")
    arv_s = local_slice_collapse(w, A_edges_s, dt_partition, tpick, ipick, phase_label, s, tlatent[:, 1:2], "LocalSliceLgCollapseS")
    arv = station_source_attention(w, x_query_src_cart.shape[0], tq_sample, x_src, trv_out_q, arv_p, arv_s, tpick, ipick,
                                   phase_label)                                                         # :993
    return o["y"], o["x"], arv[:, :, 0:1], arv[:, :, 1:2]
