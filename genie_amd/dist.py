"""Source-node sharding of the product graph across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm). Product nodes are `(g, s)`; rank r
OWNS a contiguous block of a space-filling-curve order of the source nodes and therefore the rows `[g*S, (g+1)*S)` of
every `[P, .]` tensor. Station-neighbour gathers and the Bipartite station sum are local. Source-neighbour gathers need
rows of neighbour source nodes owned elsewhere (the HALO):

* layer 1 recomputes every neighbour's hidden state from its raw 8 input floats, so the halo of layer 1 is just the
  halo nodes' `Slice/Mask` rows, which every rank loads/embeds itself (input distribution, no collective);
* layer 2 gathers the projected operand `wv` (15 channels, 64-B rows): ONE all-to-all of halo rows per window between
  `genie_da_stage1` and `genie_da_stage2_bipartite`;
* the `[G, 15]` Bipartite output is all-gathered and the G-sized SpatialAggregation / read-out kernels run replicated.

The reference has no multi-GPU code at all; this is new.
"""
import numpy as np
import torch


class ShardPlan(object):
    """Static plan for one rank (pure numpy, deterministic and identical logic on every rank).

    A_src_src: int [2, E] global source graph (row 0 = neighbour j, row 1 = centre i, in-edges grouped by centre).
    order:     permutation of the G source nodes (space-filling-curve order); rank r owns order[lo_r:hi_r].
    Local numbering on a rank: owned nodes 0..n_own-1 (in `order` order), then halo nodes grouped by owner rank.
    """

    def __init__(self, A_src_src, n_grid, world, rank, order=None):
        A = np.asarray(A_src_src)
        G, W = int(n_grid), int(world)
        order = np.arange(G) if order is None else np.asarray(order, dtype=np.int64)
        assert sorted(order.tolist()) == list(range(G)), "order must be a permutation of the source nodes"
        self.n_grid, self.world, self.rank = G, W, int(rank)
        bounds = [(G * r) // W for r in range(W + 1)]
        self.bounds = bounds
        owner = np.empty(G, dtype=np.int64)
        for r in range(W):
            owner[order[bounds[r]:bounds[r + 1]]] = r
        self.owner = owner
        self.owned = [order[bounds[r]:bounds[r + 1]].copy() for r in range(W)]          # global ids per rank
        j, i = A[0].astype(np.int64), A[1].astype(np.int64)
        # need[r][q] = sorted global ids owned by q that appear as neighbours of nodes owned by r
        self.need = [[None] * W for _ in range(W)]
        for r in range(W):
            mine = owner[i] == r
            nb = np.unique(j[mine])
            for q in range(W):
                self.need[r][q] = nb[owner[nb] == q] if q != r else np.zeros(0, dtype=np.int64)
        me = self.rank
        self.own_global = self.owned[me]
        self.n_own = int(self.own_global.size)
        self.halo_global = np.concatenate([self.need[me][q] for q in range(W)]) if W > 1 else np.zeros(0, dtype=np.int64)
        self.n_halo = int(self.halo_global.size)
        self.n_ext = self.n_own + self.n_halo
        self.ext_global = np.concatenate([self.own_global, self.halo_global])
        g2l = -np.ones(G, dtype=np.int64)
        g2l[self.ext_global] = np.arange(self.n_ext)
        self.global_to_local = g2l
        # local CSR of the owned nodes (in-edges in the original edge order), columns in local numbering
        order_e = np.argsort(i, kind="stable")
        js, is_ = j[order_e], i[order_e]
        start = np.searchsorted(is_, np.arange(G), side="left")
        stop = np.searchsorted(is_, np.arange(G), side="right")
        rowptr = np.zeros(self.n_own + 1, dtype=np.int64)
        cols = []
        for k, g in enumerate(self.own_global):
            c = g2l[js[start[g]:stop[g]]]
            assert (c >= 0).all(), "halo is incomplete"
            cols.append(c)
            rowptr[k + 1] = rowptr[k] + c.size
        self.src_rowptr = rowptr.astype(np.int32)
        self.src_col = (np.concatenate(cols) if cols else np.zeros(0)).astype(np.int32)
        # what this rank sends to q = the rows q needs from it, as LOCAL owned indices, in q's halo order
        self.send_local = [g2l[self.need[q][me]] if q != me else np.zeros(0, dtype=np.int64) for q in range(W)]
        self.send_counts = [int(x.size) for x in self.send_local]
        self.recv_counts = [int(self.need[me][q].size) for q in range(W)]

    def halo_fraction(self):
        return self.n_halo / max(1, self.n_own)


def exchange_halo_rows(rows_own, plan, n_sta, group=None):
    """All-to-all of per-source-node row blocks: `rows_own` [n_own*S, C] -> halo rows [n_halo*S, C] in halo order.

    One `all_to_all_single` (RCCL on GPUs; gloo in the CPU tests): rank r sends, to every peer q, the S-row blocks of
    its owned nodes that q lists in `need[q][r]`."""
    import torch.distributed as dist
    C = rows_own.shape[1]
    S = int(n_sta)
    blocks = rows_own.view(plan.n_own, S * C)
    send_idx = torch.as_tensor(np.concatenate(plan.send_local).astype(np.int64), device=rows_own.device)
    send = blocks.index_select(0, send_idx).contiguous() if send_idx.numel() else blocks.new_zeros((0, S * C))
    recv = blocks.new_empty((plan.n_halo, S * C))
    if plan.world == 1:
        return recv.view(-1, C)
    dist.all_to_all_single(recv, send, output_split_sizes=plan.recv_counts, input_split_sizes=plan.send_counts, group=group)
    return recv.view(plan.n_halo * S, C)


def allgather_owned(x_own, plan, group=None):
    """All-gather a per-owned-source-node tensor `[n_own, C]` into global order `[G, C]`."""
    import torch.distributed as dist
    if plan.world == 1:
        out = x_own.new_empty((plan.n_grid, x_own.shape[1]))
        out[torch.as_tensor(plan.own_global, device=x_own.device)] = x_own
        return out
    n_max = max(len(o) for o in plan.owned)
    pad = x_own.new_zeros((n_max, x_own.shape[1]))
    pad[: plan.n_own] = x_own
    parts = [torch.empty_like(pad) for _ in range(plan.world)]
    dist.all_gather(parts, pad, group=group)
    out = x_own.new_empty((plan.n_grid, x_own.shape[1]))
    for r in range(plan.world):
        out[torch.as_tensor(plan.owned[r], device=x_own.device)] = parts[r][: len(plan.owned[r])]
    return out


class ShardedPath(object):
    """Sharded DataAggregation + Bipartite on this rank's GPU, replicated SpatialAggregation / read-out.

    sta_csr: (rowptr, col) of the station graph; A_src_src: global [2, E]; edge_attr_own: [n_own*S, 3] rows of the
    owned nodes in local order; pos_global: [G, 3]; pos_sta: optional [S, 3] station positions -> station processing order
    (the same on every rank: the halo rows of `wv` travel in that order)."""

    def __init__(self, n_sta, n_grid, sta_csr, A_src_src, pos_global, world, rank, device, group=None, scale_rel=30000.0,
                 pos_sta=None):
        from . import engine
        self.group = group
        self.n_sta, self.n_grid = int(n_sta), int(n_grid)
        order = engine.sfc_order(np.asarray(pos_global))
        self.plan = ShardPlan(A_src_src, n_grid, world, rank, order)
        p = self.plan
        self.local = engine.HipPath(n_sta, p.n_own, sta_csr, (torch.from_numpy(p.src_rowptr), torch.from_numpy(p.src_col)),
                                    n_grid_ext=p.n_ext, grid_order=None, scale_rel=scale_rel, device=device,
                                    sta_order=engine.sfc_order(np.asarray(pos_sta)) if pos_sta is not None else None)
        self.full = engine.HipPath(1, n_grid, (torch.zeros(2, dtype=torch.int32), torch.zeros(0, dtype=torch.int32)),
                                   engine.csr_from_edges(torch.as_tensor(A_src_src), n_grid), grid_order=None,
                                   scale_rel=scale_rel, device=device)
        self.device = self.local.device

    def set_weights(self, named):
        self.local.set_weights(named)
        self.full.set_weights(named)

    def wv_view(self):
        """Float view [n_ext*S, 16] of the projected operand `wv` inside the local workspace."""
        import ctypes
        lp = self.local
        ptr = lp.lib.genie_ws_v_ptr(lp.ctx, lp._ws_ptr)
        off = int(ptr) - lp.ws.data_ptr()
        pitch = int(lp.lib.genie_ws_v_pitch(lp.ctx))
        n = self.plan.n_ext * self.n_sta * pitch
        return lp.ws[off: off + 4 * n].view(torch.float32).view(self.plan.n_ext * self.n_sta, pitch)

    def path_fwd(self, Slice_ext, Mask_ext, edge_attr_own, pos_global):
        """Slice_ext / Mask_ext: [n_ext*S, 4] rows of owned then halo source nodes (local order). Returns x_spatial [G,30]."""
        p, S = self.plan, self.n_sta
        lp = self.local
        Slice_ext, Mask_ext = lp.da_stage1(Slice_ext, Mask_ext)
        wv = self.wv_view()
        if p.n_halo:
            wv[p.n_own * S:] = exchange_halo_rows(wv[: p.n_own * S], p, S, self.group)
        Mask_own = Mask_ext[: p.n_own * S]
        _, bip_own = lp.da_stage2_bipartite(Mask_own, edge_attr_own)
        bip = allgather_owned(bip_own, p, self.group)
        o = bip
        for layer in (1, 2, 3):
            o = self.full.spatial_agg(layer, o, pos_global)
        return o
