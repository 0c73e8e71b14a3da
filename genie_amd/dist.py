"""Source-node sharding of the product graph across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm). Product nodes are `(g, s)`; rank r
OWNS a contiguous block of a space-filling-curve order of the source nodes and therefore the rows `[g*S, (g+1)*S)` of
every `[P, .]` tensor. Station-neighbour gathers and the Bipartite station sum are local. Source-neighbour gathers need
rows of neighbour source nodes owned elsewhere (the HALO):

* layer 1 recomputes every neighbour's hidden state from its raw 8 input floats, so the halo of layer 1 is just the
  halo nodes' `Slice/Mask` rows, which every rank loads/embeds itself (input distribution, no collective);
* layer 2 gathers the projected operand `wv` (15 channels, 64-B rows): ONE all-to-all of halo rows per window, issued on a
  communication stream as soon as stage 1 has produced the rows other ranks need, and overlapped with the rest of stage 1 and
  with stage 2 of the source nodes that have no halo neighbour;
* the `[G, 15]` Bipartite output is all-gathered (one `all_gather_into_tensor` + one index op) and the G-sized
  SpatialAggregation / read-out kernels run replicated, on a tail stream under the next window's P-sized kernels.

How the overlap is arranged: a rank's owned source nodes fall into the classes SEND (some other rank lists them as a
neighbour) and NEED (they have a neighbour owned elsewhere). The local processing order is
`[SEND only | SEND and NEED | NEED only | interior]`, each class in space-filling-curve order, so that
  stage 1 = range [0, n_send) -> all-to-all starts -> range [n_send, n_own)
  stage 2 = the two ranges without NEED nodes -> wait for the halo rows -> the NEED range
are contiguous sub-ranges of that order (`genie_da_stage1_range`, `genie_da_stage2_partials_range`).

The reference has no multi-GPU code at all (`process_config.yaml:66 parallel_processing: False`); the layout
sharded here is its product-node numbering `p = g * n_sta + s` (`process_utils.py:720-722`).
"""
import numpy as np
import torch


class ShardRows(object):
    """Rows of a per-product-node tensor in a rank's LOCAL order: the S-row blocks of its owned source nodes (space-filling-curve
    order), then of its halo nodes (`ShardPlan.ext_global`), or of the owned nodes only (`own_only`). What the sharded drop-in model
    produces (`embed_window`, `node_rows`) and accepts in place of the reference's full `[P, C]` tensors -- a full tensor and a
    local one can have the same shape (world size 1), so the local form is a type of its own, never guessed from a shape."""
    __slots__ = ("t", "own_only")

    def __init__(self, t, own_only=False):
        self.t, self.own_only = t, bool(own_only)

    @property
    def shape(self):
        return self.t.shape


def resolve_shard(process_group=None, shard=None):
    """(rank, world, group) of a sharded model, or None for an unsharded one. `process_group`: a `torch.distributed` group, or True
    for the default group; `shard` = (rank, world) overrides what the group reports (and is all there is without a group: a virtual
    rank, whose collectives the caller replaces -- `bench.py --emulate-world`)."""
    if process_group is None and shard is None:
        return None
    import torch.distributed as dist
    group = None
    if process_group is not None and process_group is not True:
        group = process_group
    if shard is not None:
        rank, world = int(shard[0]), int(shard[1])
    else:
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("process_group given but torch.distributed is not initialised (init_process_group first)")
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    if not 0 <= rank < world:
        raise ValueError("shard = (rank, world) needs 0 <= rank < world")
    return rank, world, group


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
        assert order.shape == (G,) and np.array_equal(np.sort(order), np.arange(G)), "order must be a permutation of the source nodes"
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
        cross = owner[i] != owner[j]
        ci, cj = owner[i[cross]], j[cross]
        self.need = [[np.zeros(0, dtype=np.int64)] * W for _ in range(W)]
        for r in range(W):
            nb = np.unique(cj[ci == r])
            onb = owner[nb]
            self.need[r] = [nb[onb == q] if q != r else np.zeros(0, dtype=np.int64) for q in range(W)]
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
        start = np.searchsorted(is_, self.own_global, side="left")
        stop = np.searchsorted(is_, self.own_global, side="right")
        deg = stop - start
        rowptr = np.zeros(self.n_own + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(deg)
        take = np.repeat(start - rowptr[:-1], deg) + np.arange(int(rowptr[-1]))        # edge positions, row by row
        col = g2l[js[take]] if take.size else np.zeros(0, dtype=np.int64)
        assert (col >= 0).all(), "halo is incomplete"
        self.src_rowptr = rowptr.astype(np.int32)
        self.src_col = col.astype(np.int32)
        # what this rank sends to q = the rows q needs from it, as LOCAL owned indices, in q's halo order
        self.send_local = [g2l[self.need[q][me]] if q != me else np.zeros(0, dtype=np.int64) for q in range(W)]
        self.send_counts = [int(x.size) for x in self.send_local]
        self.recv_counts = [int(self.need[me][q].size) for q in range(W)]
        # classes of owned nodes and the local processing order [SEND only | SEND and NEED | NEED only | interior]
        send = np.zeros(self.n_own, dtype=bool)
        if W > 1:
            send[np.concatenate(self.send_local).astype(np.int64)] = True
        need = np.zeros(self.n_own, dtype=bool)
        if col.size:
            row_of = np.repeat(np.arange(self.n_own), deg)
            need[row_of[col >= self.n_own]] = True
        loc = np.arange(self.n_own)
        parts = [loc[send & ~need], loc[send & need], loc[~send & need], loc[~send & ~need]]
        self.proc_order = np.concatenate(parts).astype(np.int32)
        c = np.cumsum([0] + [p.size for p in parts])
        self.r_send = (int(c[0]), int(c[2]))          # positions of the SEND nodes in proc_order
        self.r_need = (int(c[1]), int(c[3]))          # positions of the NEED nodes
        self.n_send_nodes, self.n_need_nodes = int(send.sum()), int(need.sum())

    def halo_fraction(self):
        return self.n_halo / max(1, self.n_own)


class Transport(object):
    """The two collectives of the sharded path. RCCL ("nccl") takes device buffers directly on the current stream; a process
    group without device collectives (gloo: the CPU tests and the several-processes-on-one-GPU test) is served by staging
    through host memory, which synchronises the calling stream -- test plumbing, not a data path."""

    def __init__(self, group=None, world=None, emulate=False):
        """`world`: the rank count of the plan the collectives serve. A process group of another size cannot carry them: a one-rank
        plan inside an N-rank job (a single-GPU leg of a multi-rank run) then runs without collectives -- pass an explicit one-rank
        `group` to exercise the RCCL path there --, any other mismatch is an error.
        `emulate`: ONE virtual rank of a `world`-rank plan alone on its GPU (`bench.py --emulate-world`): no peer exists, every
        collective is replaced by a device-to-device copy of the same number of bytes (this rank's own rows stand in for the peers'), so
        the kernels, sub-range launches, stream dependencies and buffer sizes are those of rank r at N = world and the RESULTS ARE NOT
        the model's output. For timing only."""
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.emulate = bool(emulate)
        self.world = int(world or 1)
        if self.emulate:
            self.on, self.device_collectives = False, False
            return
        self.on = dist.is_available() and dist.is_initialized()
        if self.on and world is not None and dist.get_world_size(group) != int(world):
            if int(world) != 1:
                raise ValueError("process group of %d ranks cannot serve a %d-rank shard plan" % (dist.get_world_size(group), int(world)))
            self.on = False
        self.device_collectives = self.on and dist.get_backend(group) == "nccl"

    def _no_group(self):
        """A collective of a plan of several ranks without a process group: the result would silently miss the peers' rows."""
        if self.world > 1:
            raise RuntimeError("a %d-rank shard plan needs an initialised torch.distributed process group for its collectives (or "
                               "emulate=True: one virtual rank timed alone, results meaningless)" % self.world)

    def all_to_all_rows(self, recv, send, recv_counts, send_counts):
        """recv [sum(recv_counts), C] <- send [sum(send_counts), C], row blocks grouped by peer."""
        if self.emulate:
            if recv.shape[0] and send.shape[0]:
                torch.index_select(send, 0, torch.arange(recv.shape[0], device=send.device) % send.shape[0], out=recv)
            return
        if not self.on:
            return self._no_group()
        if self.device_collectives or not send.is_cuda:
            self.dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=self.group)
            return
        r = torch.empty(tuple(recv.shape), dtype=recv.dtype)
        self.dist.all_to_all_single(r, send.cpu(), output_split_sizes=recv_counts, input_split_sizes=send_counts, group=self.group)
        recv.copy_(r)

    def halo_p2p_rows(self, recv, send, recv_counts, send_counts, rank, pack=None):
        """The halo exchange bucketed by destination: one (send, receive) pair per peer this rank really exchanges rows with, the
        largest transfer first, each pair its own group (`batch_isend_irecv`: on RCCL one fused send / receive per peer = per
        xGMI link, the pairs independent of each other). `pack(q, view)` fills the send rows of peer q right before that peer's
        transfer is posted, so the packing of peer q + 1 runs under the transfer of peer q. Same rows in the same places as
        `all_to_all_rows` (tests/test_dist_cpu.py: both forms over gloo)."""
        if self.emulate:
            so = np.concatenate(([0], np.cumsum(send_counts))).astype(np.int64)
            for q in range(len(send_counts)):
                if pack is not None and q != rank and send_counts[q]:
                    pack(q, send[so[q]:so[q + 1]])
            return self.all_to_all_rows(recv, send, recv_counts, send_counts)
        if not self.on:
            return self._no_group()
        dist = self.dist
        so = np.concatenate(([0], np.cumsum(send_counts))).astype(np.int64)
        ro = np.concatenate(([0], np.cumsum(recv_counts))).astype(np.int64)
        peers = [q for q in range(len(send_counts)) if q != rank and (send_counts[q] or recv_counts[q])]
        peers.sort(key=lambda q: -(send_counts[q] + recv_counts[q]))
        device = self.device_collectives or not send.is_cuda
        works, staged = [], []
        for q in peers:
            sv, rv = send[so[q]:so[q + 1]], recv[ro[q]:ro[q + 1]]
            if pack is not None and send_counts[q]:
                pack(q, sv)
            ops = []
            if not device:               # process group without device collectives: staged through host memory (test plumbing)
                sv, rh = sv.cpu(), torch.empty(tuple(rv.shape), dtype=rv.dtype)
                staged.append((rv, rh))
                rv = rh
            if send_counts[q]:
                ops.append(dist.P2POp(dist.isend, sv, q if self.group is None else dist.get_global_rank(self.group, q), self.group))
            if recv_counts[q]:
                ops.append(dist.P2POp(dist.irecv, rv, q if self.group is None else dist.get_global_rank(self.group, q), self.group))
            works += dist.batch_isend_irecv(ops)
        for w in works:
            w.wait()
        for rv, rh in staged:
            rv.copy_(rh)

    def all_gather_rows(self, out, x):
        """out [world, n, C] <- x [n, C] of every rank."""
        if self.emulate:
            out.copy_(x.unsqueeze(0).expand_as(out))
            return
        if not self.on:
            self._no_group()
            out[0].copy_(x)
            return
        if self.device_collectives:
            self.dist.all_gather_into_tensor(out.view(-1, x.shape[1]), x, group=self.group)
            return
        parts = [torch.empty(tuple(x.shape), dtype=x.dtype) for _ in range(out.shape[0])]
        self.dist.all_gather(parts, x.cpu(), group=self.group)
        out.copy_(torch.stack(parts))


def exchange_halo_rows(rows_own, plan, n_sta, group=None, mode="a2a"):
    """Exchange of per-source-node row blocks: `rows_own` [n_own*S, C] -> halo rows [n_halo*S, C] in halo order: rank r sends, to
    every peer q, the S-row blocks of its owned nodes that q lists in `need[q][r]`. mode "a2a": one `all_to_all_single` with split
    sizes (RCCL on GPUs; gloo in the CPU tests); "p2p": one send / receive pair per peer, packed per destination
    (`Transport.halo_p2p_rows`)."""
    C = rows_own.shape[1]
    S = int(n_sta)
    blocks = rows_own.view(plan.n_own, S * C)
    recv = blocks.new_empty((plan.n_halo, S * C))
    if mode == "p2p":
        send = blocks.new_empty((int(sum(plan.send_counts)), S * C))
        if plan.world > 1:
            idx = [torch.as_tensor(x.astype(np.int64), device=rows_own.device) for x in plan.send_local]
            Transport(group, plan.world).halo_p2p_rows(recv, send, plan.recv_counts, plan.send_counts, plan.rank,
                                                       pack=lambda q, view: torch.index_select(blocks, 0, idx[q], out=view))
        return recv.view(plan.n_halo * S, C)
    send_idx = torch.as_tensor(np.concatenate(plan.send_local).astype(np.int64), device=rows_own.device)
    send = blocks.index_select(0, send_idx).contiguous() if send_idx.numel() else blocks.new_zeros((0, S * C))
    if plan.world > 1:
        Transport(group, plan.world).all_to_all_rows(recv, send, plan.recv_counts, plan.send_counts)
    return recv.view(plan.n_halo * S, C)


def gather_index(plan):
    """int64 [G]: position of global source node g inside the all-gathered `[world, n_max, C]` buffer."""
    n_max = max(len(o) for o in plan.owned)
    idx = np.empty(plan.n_grid, dtype=np.int64)
    for r in range(plan.world):
        idx[plan.owned[r]] = r * n_max + np.arange(len(plan.owned[r]))
    return idx, n_max


def allgather_owned(x_own, plan, group=None, index=None):
    """All-gather a per-owned-source-node tensor `[n_own, C]` into global order `[G, C]`: one collective on rows padded to
    the largest shard and one `index_select` with the precomputed `gather_index(plan)`."""
    if index is None:
        idx, n_max = gather_index(plan)
        index = (torch.as_tensor(idx, device=x_own.device), n_max)
    idx_t, n_max = index
    pad = x_own if plan.n_own == n_max else torch.cat((x_own, x_own.new_zeros((n_max - plan.n_own, x_own.shape[1]))))
    buf = x_own.new_empty((plan.world, n_max, x_own.shape[1]))
    Transport(group, plan.world).all_gather_rows(buf, pad.contiguous())
    return buf.view(-1, x_own.shape[1]).index_select(0, idx_t)


class ShardedPath(object):
    """Sharded DataAggregation + Bipartite on this rank's GPU, replicated SpatialAggregation / read-out.

    sta_csr: (rowptr, col) of the station graph; A_src_src: global [2, E]; pos_global: [G, 3]; pos_sta: optional [S, 3]
    station positions -> station processing order (the same on every rank: the halo rows of `wv` travel in that order).
    `overlap=False` runs the window as four sequential steps (stage 1, exchange, stage 2, all-gather) on one stream: the
    A/B reference for the overlapped schedule (bit-identical results)."""

    def __init__(self, n_sta, n_grid, sta_csr, A_src_src, pos_global, world, rank, device, group=None, scale_rel=30000.0,
                 pos_sta=None, overlap=True, halo="a2a", emulate=False):
        from . import engine
        self.group = group
        self.n_sta, self.n_grid = int(n_sta), int(n_grid)
        order = engine.sfc_order(np.asarray(pos_global))
        self.plan = ShardPlan(A_src_src, n_grid, world, rank, order)
        p = self.plan
        self.overlap = bool(overlap)
        if halo not in ("a2a", "p2p"):
            raise ValueError("halo must be 'a2a' (one all_to_all_single) or 'p2p' (one send / receive pair per peer)")
        self.halo = halo
        self.local = engine.HipPath(n_sta, p.n_own, sta_csr, (torch.from_numpy(p.src_rowptr), torch.from_numpy(p.src_col)),
                                    n_grid_ext=p.n_ext, grid_order=p.proc_order, scale_rel=scale_rel, device=device,
                                    sta_order=engine.sfc_order(np.asarray(pos_sta)) if pos_sta is not None else None)
        self.full = engine.HipPath(1, n_grid, (torch.zeros(2, dtype=torch.int32), torch.zeros(0, dtype=torch.int32)),
                                   engine.csr_from_edges(torch.as_tensor(A_src_src), n_grid), grid_order=None,
                                   scale_rel=scale_rel, device=device)
        self.device = dev = self.local.device
        self.transport = Transport(group, world, emulate=emulate)
        S = self.n_sta
        pitch = int(self.local.lib.genie_ws_v_pitch(self.local.ctx))
        self._pitch = pitch
        # static exchange state: send-row index, send buffer, the halo part of `wv` inside the workspace (received in place)
        self._send_idx = torch.as_tensor(np.concatenate(p.send_local).astype(np.int64) if world > 1 else np.zeros(0, np.int64), device=dev)
        self._send_buf = torch.empty((int(self._send_idx.numel()), S * pitch), dtype=torch.float32, device=dev)
        idx, n_max = gather_index(p)
        self._gather = (torch.as_tensor(idx, device=dev), n_max)
        self._bip_pad = torch.zeros((n_max, 15), dtype=torch.float32, device=dev)
        self.comm_stream = torch.cuda.Stream(device=dev) if self.local.device.type == "cuda" else None
        self.tail_stream = torch.cuda.Stream(device=dev) if self.local.device.type == "cuda" else None
        self._tail_done = None

    def set_weights(self, named):
        self.local.set_weights(named)
        self.full.set_weights(named)

    def sync_weights(self, params, view=None):
        """`HipPath.sync_weights` for both contexts (the P-sized shard and the replicated G-sized tail)."""
        self.local.sync_weights(params, view)
        self.full.sync_weights(params, view)

    def row_index(self, own_only=False):
        """int64 device tensor: the rows of a full `[G * S, C]` per-product-node tensor (p = g * S + s, process_utils.py:720-722)
        that make up this rank's local rows (owned + halo blocks, or the owned blocks only). Built once."""
        key = "_rows_own" if own_only else "_rows_ext"
        idx = getattr(self, key, None)
        if idx is None:
            g = torch.as_tensor(self.plan.own_global if own_only else self.plan.ext_global, dtype=torch.int64, device=self.device)
            idx = (g.view(-1, 1) * self.n_sta + torch.arange(self.n_sta, dtype=torch.int64, device=self.device).view(1, -1)).reshape(-1)
            setattr(self, key, idx)
        return idx

    def local_rows(self, t, name, cols, own_only=False):
        """This rank's rows of a per-product-node input: a `ShardRows` is taken as it is (its row count is checked), a full
        `[G * S, cols]` tensor (the reference's argument) is cut down by ONE `index_select` on the device it lives on."""
        n_loc = (self.plan.n_own if own_only else self.plan.n_ext) * self.n_sta
        if isinstance(t, ShardRows):
            if t.own_only != own_only and self.plan.n_halo:
                if own_only and not t.own_only:           # owned blocks come first in the local order
                    return lp_f32(t.t, name)[:n_loc]
                raise ValueError("%s: rows of the owned source nodes only, the halo rows are needed too" % name)
            return lp_f32(t.t, name, (n_loc, cols))
        t = torch.as_tensor(t)
        if tuple(t.shape) != (self.n_grid * self.n_sta, cols):
            raise ValueError("%s: expected shape (%d, %d) (all product nodes) or a ShardRows of %d local rows, got %s"
                             % (name, self.n_grid * self.n_sta, cols, n_loc, tuple(t.shape)))
        idx = self.row_index(own_only)
        if t.device != idx.device:
            return t.index_select(0, idx.to(t.device)).to(self.device, torch.float32)
        return t.index_select(0, idx).float()

    def set_edge_features(self, pos_sta, pos_src_global):
        """`use_updated_model_definition` on the shard (genie_set_edge_features): the mean edge features of the owned source nodes
        come from the local CSR over the EXTENDED node list, so the positions of the halo nodes travel with the plan, not per window."""
        ext = torch.as_tensor(self.plan.ext_global, dtype=torch.long)
        pos = torch.as_tensor(pos_src_global).float()
        self.local.set_edge_features(torch.as_tensor(pos_sta).float().to(self.device), pos[ext].contiguous().to(self.device))

    def set_absolute_pos(self, pos_sta, pos_src_global):
        """`use_absolute_pos` on the shard (genie_set_absolute_pos): scaled positions of the stations and of the extended node list."""
        ext = torch.as_tensor(self.plan.ext_global, dtype=torch.long)
        pos = torch.as_tensor(pos_src_global).float()
        self.local.set_absolute_pos(torch.as_tensor(pos_sta).float().to(self.device), pos[ext].contiguous().to(self.device))

    def wv_view(self):
        """Float view [n_ext*S, 16] of the projected operand `wv` inside the local workspace."""
        lp = self.local
        ptr = lp.lib.genie_ws_v_ptr(lp.ctx, lp._ws_ptr)
        off = int(ptr) - lp.ws.data_ptr()
        n = self.plan.n_ext * self.n_sta * self._pitch
        return lp.ws[off: off + 4 * n].view(torch.float32).view(self.plan.n_ext * self.n_sta, self._pitch)

    def _exchange(self, wv):
        """Pack the SEND rows, all-to-all, halo rows received in place (the halo part of `wv` is contiguous, grouped by owner)."""
        p, S = self.plan, self.n_sta
        if p.world == 1 and not self.transport.on and not self.transport.emulate:
            return                      # no process group at all. (With one, EVERY rank enters the collective, also a rank with
                                        # nothing to exchange and the single rank of a world-size-1 group: the RCCL code path of the
                                        # 1-GPU tests and of `bench.py --gpus 1 --mode sharded`)
        blocks = wv[: p.n_own * S].view(p.n_own, S * self._pitch)
        recv = wv[p.n_own * S:].view(p.n_halo, S * self._pitch)
        if self.halo == "p2p":       # bucketed by destination: pack of peer q + 1 under the transfer of peer q, largest pair first
            if getattr(self, "_send_idx_q", None) is None:
                self._send_idx_q = [torch.as_tensor(x.astype(np.int64), device=self.device) for x in p.send_local]
            self.transport.halo_p2p_rows(recv, self._send_buf, p.recv_counts, p.send_counts, p.rank,
                                         pack=lambda q, view: torch.index_select(blocks, 0, self._send_idx_q[q], out=view))
            return
        if self._send_idx.numel():
            torch.index_select(blocks, 0, self._send_idx, out=self._send_buf)
        self.transport.all_to_all_rows(recv, self._send_buf, p.recv_counts, p.send_counts)

    def front(self, Slice_ext, Mask_ext, edge_attr_own):
        """Sharded DataAggregation + Bipartite of one window on the current stream (+ the communication stream): returns the
        all-gathered Bipartite output `[G, 15]` in global order, produced on the current stream."""
        p, S = self.plan, self.n_sta
        lp = self.local
        lp.check_input_range()         # (verdicts of the windows that have completed: engine.HipPath.check_input_range)
        P_ext = p.n_ext * S
        Slice_ext = lp_f32(Slice_ext, "Slice", (P_ext, 4))
        Mask_ext = lp_f32(Mask_ext, "Mask", (P_ext, 4))
        edge_attr_own = lp_f32(edge_attr_own, "edge_attr", (p.n_own * S, 3))
        ea = getattr(lp, "_static_ea", None)
        if ea is None or ea.data_ptr() != edge_attr_own.data_ptr() or edge_attr_own._version != lp._static_ea_version:
            # the static edge_attr of the owned rows: registered once, so that stage 2 reads its processing-order copy and runs
            # the straight-line row-layout kernel (k_stage2_ord) as the unsharded path does
            lp.set_static_edge_attr(edge_attr_own)
        Mask_own = Mask_ext[: p.n_own * S]
        wv = self.wv_view()
        main = torch.cuda.current_stream(self.device)
        (s0, s1), (n0, n1), n = p.r_send, p.r_need, p.n_own
        if not self.overlap or (p.world == 1 and not self.transport.on and not self.transport.emulate):
            lp.da_stage1_range(Slice_ext, Mask_ext, 0, n, True)
            self._exchange(wv)
            lp.da_stage2_partials_range(Mask_own, edge_attr_own, 0, n)
        else:
            comm = self.comm_stream
            lp.da_stage1_range(Slice_ext, Mask_ext, s0, s1, True)                 # rows other ranks wait for
            ev = torch.cuda.Event()
            ev.record(main)
            comm.wait_event(ev)
            with torch.cuda.stream(comm):
                self._exchange(wv)
                halo = torch.cuda.Event()
                halo.record(comm)
            lp.da_stage1_range(Slice_ext, Mask_ext, s1, n, False)                 # under the exchange
            lp.da_stage2_partials_range(Mask_own, edge_attr_own, 0, n0)           # nodes without halo neighbours
            lp.da_stage2_partials_range(Mask_own, edge_attr_own, n1, n)
            main.wait_event(halo)
            lp.da_stage2_partials_range(Mask_own, edge_attr_own, n0, n1)
        bip_own = lp.bipartite_readout()
        return bip_own

    def gather_and_tail(self, bip_own, pos_global, tail=None):
        """All-gather of the Bipartite output, SpatialAggregation x3 (replicated) and `tail(x_spatial)` (the read-outs)."""
        p = self.plan
        idx_t, n_max = self._gather
        pad = self._bip_pad if p.n_own != n_max else None
        if pad is not None:
            pad[: p.n_own].copy_(bip_own)
        buf = bip_own.new_empty((p.world, n_max, 15))
        self.transport.all_gather_rows(buf, pad if pad is not None else bip_own)
        o = self.full.spatial_agg3(buf.view(-1, 15).index_select(0, idx_t), pos_global)
        return o if tail is None else tail(o)

    def path_fwd(self, Slice_ext, Mask_ext, edge_attr_own, pos_global, tail=None, pipelined=False):
        """Slice_ext / Mask_ext: [n_ext*S, 4] rows of owned then halo source nodes (local order). Returns x_spatial [G,30]
        (or `tail(x_spatial)`). `pipelined`: the all-gather and the G-sized kernels run on the tail stream, where they overlap
        the NEXT window's P-sized kernels (windows are independent in the apply loop); the result is then produced on
        `self.tail_stream` -- consume it there or after `wait_tail()`."""
        bip_own = self.front(Slice_ext, Mask_ext, edge_attr_own)
        if not pipelined:
            return self.gather_and_tail(bip_own, pos_global, tail)
        main = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        ts = self.tail_stream
        ts.wait_event(ev)
        bip_own.record_stream(ts)
        with torch.cuda.stream(ts):
            out = self.gather_and_tail(bip_own, pos_global, tail)
            self._tail_done = torch.cuda.Event()
            self._tail_done.record(ts)
        return out

    def wait_tail(self):
        if self._tail_done is not None:
            torch.cuda.current_stream(self.device).wait_event(self._tail_done)


def lp_f32(t, name, shape):
    from . import engine
    return engine._f32(t, name, shape)
