"""TEST INFRASTRUCTURE, NOT PRODUCT CODE: plain-PyTorch restatements of the read-out and association heads
(`/root/reference/Code/module.py:251-352, :356-403, :610-775`) on the parameters of the state_dict-compatible classes of
`genie_amd/module.py`, which themselves hold parameters only -- every head is computed by libgenie_hip on the GPU, in eval mode and
in training steps. The CPU tests pin these restatements to the reference's fixtures (tests/test_host_cpu.py, tests/test_assoc_cpu.py)
and the GPU tests compare the HIP kernels with them on random inputs (tests/test_hip_parity.py). Until round 4 this code sat inside
the product module as `forward` methods no GPU call reached; `attach(net)` binds them onto a model object for the tests that call
`net.<Head>(...)`.
"""
import math
import types

import torch

from genie_amd import module as _m


def _scatter_mean_rows(msg, index, n):
    out = torch.zeros((n, msg.shape[1]), dtype=msg.dtype, device=msg.device).index_add_(0, index, msg)
    cnt = torch.zeros(n, dtype=msg.dtype, device=msg.device).index_add_(0, index, torch.ones_like(index, dtype=msg.dtype))
    return out / cnt.clamp(min=1).view(-1, 1)


def spatial_direct(self, inpts):
    return self.activate(self.f_direct(inpts))


def knn_query_edges(x_context, x_query, k):
    """Exact kNN of each query in the context set on `x/1000` (module.py:282) by a dense fp64 distance matrix + top-k, on whatever
    device the tensors live. Returns LongTensor [2, Q*k]: row 0 = context j, row 1 = query i (the `.flip(0)` layout)."""
    xc = (x_context.double() / 1000.0)
    xq = (x_query.double() / 1000.0)
    k = min(k, xc.shape[0])
    idx = torch.empty((xq.shape[0], k), dtype=torch.long, device=xq.device)
    chunk = max(1, min(xq.shape[0], int(4e7 // max(1, xc.shape[0]))))
    for a in range(0, xq.shape[0], chunk):
        d = ((xq[a:a + chunk, None, :] - xc[None, :, :]) ** 2).sum(-1)
        idx[a:a + chunk] = torch.topk(d, k, dim=1, largest=False, sorted=True)[1]
    row_q = torch.arange(xq.shape[0], device=xq.device).repeat_interleave(k)
    return torch.stack([idx.reshape(-1), row_q], dim=0)


def spatial_attention(self, inpts, x_query, x_context, k=10):
    H, L = self.n_heads, self.n_latent
    edge_index = knn_query_edges(x_context, x_query, k)
    j, i = edge_index[0], edge_index[1]
    kk = edge_index.shape[1] // x_query.shape[0]
    edge_attr = (x_query[i] - x_context[j]) / self.scale_rel
    x_j = inpts[j]
    cat = torch.cat((x_j, edge_attr), dim=-1)
    q = self.f_queries(edge_attr).view(-1, H, L)
    c = self.f_context(cat).view(-1, H, L)
    v = self.f_values(cat).view(-1, H, L)
    alpha = self.activate1((q * c).sum(-1) / self.scale)                       # [E, H]
    # segment softmax over the k edges of each query (edges are grouped by query, k each)
    alpha = alpha.view(-1, kk, H)
    alpha = alpha - alpha.max(dim=1, keepdim=True)[0]
    alpha = alpha.exp()
    alpha = alpha / (alpha.sum(dim=1, keepdim=True) + 1e-16)
    agg = (alpha.unsqueeze(-1) * v.view(-1, kk, H, L)).sum(dim=1)               # [Q, H, L]
    return self.activate2(self.proj(agg.mean(1)))


def temporal_attention(self, inpts, t_query):
    H, L = self.n_heads, self.n_latent
    context = self.f_context_2(self.activate1(self.f_context_1(inpts))).view(-1, H, L)
    values = self.f_values_2(self.activate2(self.f_values_1(inpts))).view(-1, H, L)
    query = self.temporal_query_2(self.activate3(self.temporal_query_1(t_query / self.scale_t))).view(-1, H, L)
    score = torch.einsum("nhl,thl->nth", context, query) / self.scale           # [N, T, H]
    z = torch.einsum("nth,nhl->ntl", score, values) / H                         # mean over heads
    return self.proj_2(self.activate5(self.proj_1(self.activate4(z))))


def _mean_over_sta(x, sta_nbr, n_sta, n_grid):
    """mean over the station neighbours inside the same source node; x [P,C], sta_nbr Long [S,ks]."""
    if sta_nbr.shape[1] == 0:
        return torch.zeros_like(x)
    return x.view(n_grid, n_sta, -1)[:, sta_nbr, :].mean(dim=2).reshape(n_grid * n_sta, -1)


def _mean_over_src(x, src_nbr, n_sta, n_grid):
    """mean over the source-node neighbours for the same station; x [P,C], src_nbr Long [G,kp] (summed in edge order)."""
    if src_nbr.shape[1] == 0:
        return torch.zeros_like(x)
    x3 = x.view(n_grid, n_sta, -1)
    out = torch.zeros_like(x3)
    for k in range(src_nbr.shape[1]):
        out += x3[src_nbr[:, k]]
    return (out / src_nbr.shape[1]).reshape(n_grid * n_sta, -1)


def bipartite_read_out(self, inpt, edge_attr, mask, n_sta):
    g = torch.arange(edge_attr.shape[0], device=edge_attr.device) // n_sta
    msg = mask[g] * self.activate1(self.fc1(torch.cat((inpt[g], edge_attr), dim=-1)))            # :352
    return self.activate2(self.fc2(msg)), mask[g]                                                # :348


def association_phase(self, tr, latent, mask1, mask2, sta_nbr, src_nbr, n_sta, n_grid, edge_means=None):
    """`edge_means` = (m_sta [S, 4], m_src [G, 4]): the mean edge position feature of every base node's in-neighbourhood; on the
    product graph the mean over a node's messages of cat(x_j, e_ij) is cat(mean x_j, m[node]) (module.py:462-480)."""
    def means(x1, x2):
        a, b = _mean_over_sta(x1, sta_nbr, n_sta, n_grid), _mean_over_src(x2, src_nbr, n_sta, n_grid)
        if edge_means is not None:
            a = torch.cat((a, edge_means[0].repeat(n_grid, 1)), dim=1)
            b = torch.cat((b, edge_means[1].repeat_interleave(n_sta, dim=0)), dim=1)
        return a, b

    mask = torch.cat((mask1, mask2), dim=-1)
    tr = self.activate(self.init_trns(torch.cat((tr, latent, mask), dim=-1)))
    a1, a2 = means(self.activate11(self.l1_t1_1(tr)), self.activate12(self.l1_t2_1(tr)))
    tr = self.activate1(torch.cat((self.l1_t1_2(torch.cat((tr, a1, mask), dim=1)),
                                   self.l1_t2_2(torch.cat((tr, a2, mask), dim=1))), dim=1))
    b1, b2 = means(self.activate21(self.l2_t1_1(tr)), self.activate22(self.l2_t2_1(tr)))
    return self.activate2(torch.cat((self.l2_t1_2(torch.cat((tr, b1, mask), dim=1)),
                                     self.l2_t2_2(torch.cat((tr, b2, mask), dim=1))), dim=1))


def _segment_softmax(src, index, n):
    """torch_geometric.utils.softmax semantics: per-segment max subtraction, exp, / (sum + 1e-16)."""
    idx = index.view(-1, 1).expand_as(src)
    mx = torch.full((n, src.shape[1]), float("-inf"), dtype=src.dtype, device=src.device).scatter_reduce(
        0, idx, src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    den = torch.zeros((n, src.shape[1]), dtype=src.dtype, device=src.device).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def local_slice_collapse(self, A_edges, dt_partition, tpick, ipick, phase_label, inpt, tlatent, k_infer=10):
    dev = inpt.device
    n_arvs, l_dt = len(tpick), len(dt_partition)
    dt = dt_partition[1] - dt_partition[0]
    t_index = torch.floor((tpick - dt_partition[0]) / dt).long()                                           # :635
    t_index = ((ipick * l_dt * k_infer + t_index * k_infer).view(-1, 1) + torch.arange(k_infer, device=dev).view(1, -1)).reshape(-1)
    e1 = torch.arange(n_arvs, device=dev).view(-1, 1).repeat(1, k_infer).view(-1)                         # :638
    e0 = A_edges[t_index].long()
    keep = torch.where((tpick[e1] - tlatent[e0, 0]).abs() < 2.0 * self.eps)[0]                             # :642-645
    e0, e1 = e0[keep], e1[keep]
    msg = self.activate1(self.fc1(torch.cat((inpt[e0], (tpick.view(-1, 1)[e1] - tlatent[e0]) / self.eps, phase_label[e1]), dim=-1)))
    agg = torch.zeros((n_arvs, msg.shape[1]), dtype=msg.dtype, device=dev).index_add_(0, e1, msg)
    cnt = torch.zeros(n_arvs, dtype=msg.dtype, device=dev).index_add_(0, e1, torch.ones_like(e1, dtype=msg.dtype))
    return self.activate2(self.fc2(agg / cnt.clamp(min=1).view(-1, 1)))                                    # 'mean' :612


def station_pick_pairs(ipick):
    """Pick x pick edge list of the arrival-association head, built where `ipick` lives (module.py:703-713 does it on the host
    with cKDTree / itertools per call): for every station u with picks l_u (in pick order) all pairs (a, b), a in l_u,
    b in l_u + [n_arv] (the null pick), a-major, stations ascending. Returns LongTensor [2, sum n_u (n_u + 1)], rows (b; a)."""
    n = int(ipick.shape[0])
    dev = ipick.device
    order = torch.sort(ipick, stable=True)[1]
    _, inv, counts = torch.unique_consecutive(ipick[order], return_inverse=True, return_counts=True)
    seg_len = counts[inv]
    seg_start = (torch.cumsum(counts, 0) - counts)[inv]
    cnt = seg_len + 1
    a_rep = torch.repeat_interleave(torch.arange(n, device=dev), cnt)
    pos = torch.arange(a_rep.shape[0], device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
    real = pos < seg_len[a_rep]
    b = torch.where(real, order[(seg_start[a_rep] + pos).clamp(max=max(n - 1, 0))], torch.full_like(pos, n))
    return torch.stack((b, order[a_rep]), dim=0)


def arrivals(self, n_src, stime, src_embed, trv_src, arrival_p, arrival_s, tpick, ipick, phase_label):
    dev, dt_ = tpick.device, tpick.dtype
    n_sta, n_arv, H, L, eps = trv_src.shape[1], len(tpick), self.n_heads, self.n_latent, self.eps
    edges = station_pick_pairs(ipick)                                                                       # rows (b; a)  :703-713
    n_edge = edges.shape[1]
    edges = edges.repeat(1, n_src) + torch.cat((torch.zeros(1, n_src * n_edge, dtype=torch.long, device=dev),
                                                (torch.arange(n_src, device=dev) * n_arv).repeat_interleave(n_edge).view(1, -1)), 0)
    sidx = torch.arange(n_src, device=dev).repeat_interleave(n_edge)
    arrival = torch.cat((torch.cat((arrival_p, arrival_p.new_zeros(1, arrival_p.shape[1])), 0),
                         torch.cat((arrival_s, arrival_s.new_zeros(1, arrival_s.shape[1])), 0)), dim=1)
    atime = torch.cat((tpick, tpick.new_full((1,), -eps)))
    stindex = torch.cat((ipick, ipick.new_full((1,), n_sta)))
    tsrc_p = torch.cat((trv_src[:, :, 0], trv_src.new_full((n_src, 1), -eps)), dim=1)
    tsrc_s = torch.cat((trv_src[:, :, 1], trv_src.new_full((n_src, 1), -eps)), dim=1)
    phase = torch.cat((phase_label, phase_label.new_full((1, 1), -1.0)), dim=0)

    def rel(e0, si, tsrc):
        return atime[e0] - (tsrc[si, stindex[e0]] + stime[si])
    keep = torch.where((rel(edges[0], sidx, tsrc_p).abs() < 2.0 * eps) | (rel(edges[0], sidx, tsrc_s).abs() < 2.0 * eps))[0]
    edges, sidx = edges[:, keep], sidx[keep]
    e0, e1 = edges[0], edges[1]
    e0max = int(e0.max().item())                                                                            # :762-763
    rp, rs = rel(e0, sidx, tsrc_p).view(-1, 1), rel(e0, sidx, tsrc_s).view(-1, 1)
    fp = torch.cat((torch.exp(-0.5 * rp ** 2 / eps ** 2), torch.sign(rp), phase[e0]), dim=1)
    fs = torch.cat((torch.exp(-0.5 * rs ** 2 / eps ** 2), torch.sign(rs), phase[e0]), dim=1)
    self_link = (e0 == torch.remainder(e1, e0max)).view(-1, 1).to(dt_)
    null_link = (e0 == e0max).view(-1, 1).to(dt_)
    x_j = arrival[e0]
    d = lambda lin, x: lin(x)
    ctx = d(self.f_src_context_2, self.activate1(d(self.f_src_context_1,
        torch.cat((src_embed[sidx], stime[sidx].view(-1, 1), self_link, null_link), dim=1)))).view(-1, H, L)
    qry = d(self.f_arrival_query_2, self.activate2(d(self.f_arrival_query_1, torch.cat((x_j, fp, fs), dim=1)))).view(-1, H, L)
    val = d(self.f_values_2, self.activate3(d(self.f_values_1, torch.cat((x_j, fp, fs, self_link, null_link), dim=1)))).view(-1, H, L)
    alpha = _segment_softmax((qry * ctx).sum(-1) / math.sqrt(L), e1, n_arv * n_src)
    agg = torch.zeros((n_arv * n_src, H, L), dtype=dt_, device=dev).index_add_(0, e1, alpha.unsqueeze(-1) * val)
    return self.proj_2(self.activate4(self.proj_1(agg.mean(1)))).view(n_src, n_arv, -1)


def attach(net):
    """Bind the restatements as `forward` of the head modules of one model object (tests only)."""
    pairs = ((net.SpatialDirect, spatial_direct), (net.SpatialAttention, spatial_attention), (net.TemporalAttention, temporal_attention),
             (net.BipartiteGraphReadOutOperator, bipartite_read_out), (net.DataAggregationAssociationPhase, association_phase),
             (net.LocalSliceLgCollapseP, local_slice_collapse), (net.LocalSliceLgCollapseS, local_slice_collapse), (net.Arrivals, arrivals))
    for mod, fn in pairs:
        mod.forward = types.MethodType(fn, mod)
    return net
