"""CPU (gloo, world_size 2): the source-node sharding plan, the halo all-to-all and the owned-row all-gather of
genie_amd/dist.py, with the oracle standing in for the per-rank HIP stages (test infrastructure only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genie_amd import dist as gdist
from genie_amd import engine, graph, synthetic
from oracle import genie_oracle as O
from tests.util import Case, max_abs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _csr_mean(x_blocks, rowptr, col, n_own):
    """mean over in-neighbours of [n_ext, S, C] blocks -> [n_own, S, C]; empty neighbourhood -> 0."""
    out = torch.zeros((n_own,) + tuple(x_blocks.shape[1:]), dtype=x_blocks.dtype)
    for i in range(n_own):
        c = col[rowptr[i]:rowptr[i + 1]].long()
        if c.numel():
            out[i] = x_blocks[c].sum(0) / c.numel()
    return out


def sharded_oracle_rank(plan, geom, w, Slice, Mask, edge_attr, group):
    """What one rank does, with oracle arithmetic: returns the all-gathered Bipartite output [G,15]."""
    S = geom.n_sta
    ext = torch.from_numpy(plan.ext_global)
    rows = (ext.view(-1, 1) * S + torch.arange(S).view(1, -1)).reshape(-1)
    Se, Me = Slice[rows], Mask[rows]
    n_own, n_ext = plan.n_own, plan.n_ext
    sta_nbr = graph.neighbour_table(geom.A_sta_sta, S)
    rp, col = torch.from_numpy(plan.src_rowptr), torch.from_numpy(plan.src_col)
    pre = "DataAggregation"
    h0 = O.act(O.linear(torch.cat((Se, Me), -1), w, pre + ".init_trns"), w, pre + ".activate")          # all ext rows
    h0o, Mo = h0[: n_own * S], Me[: n_own * S]
    n1 = O._gather_mean_sta(O.act(h0o, w, pre + ".activate11").view(n_own, S, -1), sta_nbr).reshape(n_own * S, -1)
    n2 = _csr_mean(O.act(h0, w, pre + ".activate12").view(n_ext, S, -1), rp, col, n_own).reshape(n_own * S, -1)
    tr1 = O.linear(torch.cat((h0o, n1, Mo), 1), w, pre + ".l1_t1_2")
    tr2 = O.linear(torch.cat((h0o, n2, Mo), 1), w, pre + ".l1_t2_2")
    h1 = O.act(torch.cat((tr1, tr2), 1), w, pre + ".activate1")
    u = O.act(O.linear(h1, w, pre + ".l2_t1_1"), w, pre + ".activate21")
    v = O.act(O.linear(h1, w, pre + ".l2_t2_1"), w, pre + ".activate22")
    W1, W2 = w[pre + ".l2_t1_2.weight"], w[pre + ".l2_t2_2.weight"]
    wu = u @ W1[:, 60:90].T
    wv_own = (v @ W2[:, 60:90].T).contiguous()
    wv_halo = gdist.exchange_halo_rows(wv_own, plan, S, group)                                           # <- all-to-all
    wv = torch.cat((wv_own, wv_halo), 0)
    c1 = h1 @ W1[:, :60].T + Mo @ W1[:, 90:94].T + w[pre + ".l2_t1_2.bias"]
    c2 = h1 @ W2[:, :60].T + Mo @ W2[:, 90:94].T + w[pre + ".l2_t2_2.bias"]
    m1 = O._gather_mean_sta(wu.view(n_own, S, -1), sta_nbr).reshape(n_own * S, -1)
    m2 = _csr_mean(wv.view(n_ext, S, -1), rp, col, n_own).reshape(n_own * S, -1)
    x_latent = O.act(torch.cat((c1 + m1, c2 + m2), 1), w, pre + ".activate2")
    own_rows = rows[: n_own * S]
    bip_own = O.bipartite_read_in_structured(w, x_latent, edge_attr[own_rows], Mo, S, n_own)
    return gdist.allgather_owned(bip_own, plan, group)                                                    # <- all-gather


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        geom = synthetic.Geometry(9, 120, L=120e3, n_query=10, seed=31)
        win = synthetic.make_window(geom, 200, seed=32)
        w = Case("tiny_6x40").weights
        Slice, Mask = torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"])
        ea = torch.from_numpy(geom.edge_attr())
        order = engine.morton_order(geom.x_grid)
        plan = gdist.ShardPlan(geom.A_src_src, geom.n_grid, world, rank, order)
        bip = sharded_oracle_rank(plan, geom, w, Slice, Mask, ea, None)
        sta_nbr = graph.neighbour_table(geom.A_sta_sta, geom.n_sta)
        src_nbr = graph.neighbour_table(geom.A_src_src, geom.n_grid)
        xl = O.data_aggregation_structured(w, Slice, Mask, sta_nbr, src_nbr, geom.n_sta, geom.n_grid)
        ref = O.bipartite_read_in_structured(w, xl, ea, Mask, geom.n_sta, geom.n_grid)
        ret[rank] = (max_abs(bip, ref), float(ref.abs().max()), plan.n_own, plan.n_halo)
    finally:
        dist.destroy_process_group()


def test_sharded_path_world2_gloo_matches_unsharded():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        err, scale, n_own, n_halo = ret[rank]
        assert n_own == 60 and 0 < n_halo <= 60
        assert err <= 1e-5 * max(1.0, scale), (rank, err)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_plan_invariants(world):
    geom = synthetic.Geometry(5, 203, L=100e3, n_query=5, seed=5)
    order = engine.morton_order(geom.x_grid)
    plans = [gdist.ShardPlan(geom.A_src_src, geom.n_grid, world, r, order) for r in range(world)]
    owned = np.concatenate([p.own_global for p in plans])
    assert sorted(owned.tolist()) == list(range(geom.n_grid))                 # a partition of the source nodes
    for p in plans:
        # every neighbour of an owned node is owned or in the halo, and local columns point at the right global node
        tab = graph.neighbour_table(geom.A_src_src, geom.n_grid).numpy()
        for k, g in enumerate(p.own_global):
            loc = p.src_col[p.src_rowptr[k]:p.src_rowptr[k + 1]]
            assert np.array_equal(p.ext_global[loc], tab[g])
        assert not np.intersect1d(p.own_global, p.halo_global).size
        # send/recv symmetry: what q expects from p is exactly what p sends to q
        for q in plans:
            if q.rank != p.rank:
                assert np.array_equal(p.own_global[p.send_local[q.rank]], q.need[q.rank][p.rank])
                assert p.send_counts[q.rank] == q.recv_counts[p.rank]
        # processing order = [SEND only | SEND and NEED | NEED only | interior]: the SEND / NEED nodes are exactly the
        # sub-ranges r_send / r_need of it (what genie_da_stage1_range / genie_da_stage2_partials_range are launched on)
        assert sorted(p.proc_order.tolist()) == list(range(p.n_own))
        sent = set(np.concatenate(p.send_local).astype(int).tolist()) if world > 1 else set()
        needy = {k for k in range(p.n_own) if (p.src_col[p.src_rowptr[k]:p.src_rowptr[k + 1]] >= p.n_own).any()}
        assert set(p.proc_order[p.r_send[0]:p.r_send[1]].tolist()) == sent
        assert set(p.proc_order[p.r_need[0]:p.r_need[1]].tolist()) == needy
        assert p.r_send[0] == 0 and p.r_send[0] <= p.r_need[0] <= p.r_send[1] <= p.r_need[1] <= p.n_own
        idx, n_max = gdist.gather_index(p)
        assert n_max == max(q.n_own for q in plans) and len(set(idx.tolist())) == geom.n_grid
        for q in plans:
            assert np.array_equal(idx[q.own_global], q.rank * n_max + np.arange(q.n_own))
    if world == 1:
        assert plans[0].n_halo == 0 and plans[0].r_send == (0, 0) and plans[0].r_need == (0, 0)


def test_shard_rows_and_shard_resolution():
    """The plumbing of the sharded drop-in class that needs no GPU: how (rank, world, group) is resolved, that a plan of several ranks
    without a process group refuses its collectives instead of silently missing the peers' rows, and the emulation transport (one
    virtual rank alone: copies of the same size)."""
    assert gdist.resolve_shard(None, None) is None
    assert gdist.resolve_shard(None, (3, 8)) == (3, 8, None)
    with pytest.raises(ValueError):
        gdist.resolve_shard(None, (8, 8))
    if not dist.is_initialized():
        with pytest.raises(RuntimeError, match="not initialised"):
            gdist.resolve_shard(True, None)
    r = gdist.ShardRows(torch.zeros(6, 4), own_only=True)
    assert r.own_only and tuple(r.shape) == (6, 4)
    t = gdist.Transport(None, world=4)
    if not t.on:
        with pytest.raises(RuntimeError, match="process group"):
            t.all_to_all_rows(torch.zeros(2, 3), torch.zeros(2, 3), [2, 0, 0, 0], [2, 0, 0, 0])
        with pytest.raises(RuntimeError, match="process group"):
            t.all_gather_rows(torch.zeros(4, 2, 3), torch.zeros(2, 3))
    one = gdist.Transport(None, world=1)
    if not one.on:
        out = torch.zeros(1, 2, 3)
        one.all_gather_rows(out, torch.ones(2, 3))            # a one-rank plan needs no group
        assert float(out.sum()) == 6.0
    e = gdist.Transport(None, world=4, emulate=True)
    send, recv = torch.arange(6.0).view(3, 2), torch.zeros(5, 2)
    e.all_to_all_rows(recv, send, [5, 0, 0, 0], [3, 0, 0, 0])
    assert torch.equal(recv, send[torch.arange(5) % 3])
    packed = []
    e.halo_p2p_rows(recv, send, [0, 5, 0, 0], [0, 3, 0, 0], 0, pack=lambda q, view: packed.append(q))
    assert packed == [1]
    out = torch.zeros(4, 3, 2)
    e.all_gather_rows(out, send)
    assert all(torch.equal(out[k], send) for k in range(4))
