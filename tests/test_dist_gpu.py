"""GPU: the source-node-sharded path (genie_amd/dist.py) through its real schedule -- sub-range stage launches, the halo
all-to-all on the communication stream, all-gather, replicated tail -- with one process per rank.

* two processes sharing ONE GPU over gloo (host-staged transport): every kernel launch, stream dependency and buffer of the
  multi-GPU schedule runs for real; only the collective itself is not RCCL. Runs on the 1-GPU test box.
* two processes on two GPUs over RCCL ("nccl"): skipped unless the box has >= 2 GPUs.
Both compare every rank's result with the unsharded HIP path bit for bit, with the overlap schedule on and off.
"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, backend, same_gpu, ret):
    import torch.distributed as dist
    from genie_amd import dist as gdist
    from genie_amd import engine, synthetic
    from tests.util import Case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = "cuda:0" if same_gpu else "cuda:%d" % rank
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, G = 40, 900
        geom = synthetic.Geometry(S, G, L=200e3, n_query=20, seed=41)
        wins = [synthetic.make_window(geom, 400, seed=42, window=k) for k in range(3)]
        wd = {k: v.to(dev) for k, v in Case("odd_33x257").weights.items()}
        ea = torch.from_numpy(geom.edge_attr())
        pos = torch.from_numpy(geom.x_grid).float().to(dev)
        sta_csr = engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S)
        hp = engine.HipPath(S, G, sta_csr, engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G),
                            grid_order=engine.sfc_order(geom.x_grid), device=dev, sta_order=engine.sfc_order(geom.locs))
        hp.set_weights(wd)
        refs = [hp.path_fwd(torch.from_numpy(w["Slice"]).to(dev), torch.from_numpy(w["Mask"]).to(dev), ea.to(dev), pos)[0].clone()
                for w in wins]
        res = {}
        for overlap in (True, False):
            sp = gdist.ShardedPath(S, G, sta_csr, geom.A_src_src, geom.x_grid, world, rank, dev, pos_sta=geom.locs, overlap=overlap)
            sp.set_weights(wd)
            p = sp.plan
            ext = torch.from_numpy(p.ext_global)
            rows = (ext.view(-1, 1) * S + torch.arange(S).view(1, -1)).reshape(-1)
            own_rows = rows[: p.n_own * S]
            sp.local.ws.fill_(255)                                       # NaN-poisoned workspace
            outs = []
            for w in wins:                                               # plain: everything on the current stream
                outs.append(sp.path_fwd(torch.from_numpy(w["Slice"])[rows].to(dev), torch.from_numpy(w["Mask"])[rows].to(dev),
                                        ea[own_rows].to(dev), pos))
            torch.cuda.synchronize()
            ok_plain = all(torch.equal(o, r) for o, r in zip(outs, refs))
            outs = []
            for w in wins:                                               # pipelined: all-gather + tail on the tail stream
                outs.append(sp.path_fwd(torch.from_numpy(w["Slice"])[rows].to(dev), torch.from_numpy(w["Mask"])[rows].to(dev),
                                        ea[own_rows].to(dev), pos, pipelined=True))
            sp.wait_tail()
            torch.cuda.synchronize()
            ok_piped = all(torch.equal(o, r) for o, r in zip(outs, refs))
            res[overlap] = (ok_plain, ok_piped)
            if overlap:
                res["plan"] = (p.n_own, p.n_halo, p.r_send, p.r_need)
        ret[rank] = res
    finally:
        dist.destroy_process_group()


def _run(world, backend, same_gpu):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), backend, same_gpu, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        r = ret[rank]
        n_own, n_halo, r_send, r_need = r["plan"]
        assert n_halo > 0 and r_send[1] > 0 and r_need[1] > r_need[0]
        assert r_need[1] < n_own, "the test geometry must leave interior nodes for the overlapped stage-2 launch"
        assert r[True] == (True, True), (rank, "overlapped schedule", r[True])
        assert r[False] == (True, True), (rank, "sequential schedule", r[False])


def test_sharded_path_two_processes_on_one_gpu_match_unsharded():
    _run(2, "gloo", True)


def test_sharded_path_three_processes_on_one_gpu_match_unsharded():
    _run(3, "gloo", True)


def test_sharded_path_world2_rccl_matches_unsharded():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI); the 1-GPU box runs the gloo form above")
    _run(2, "nccl", False)
