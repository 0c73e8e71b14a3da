#!/usr/bin/env python
"""bench.py — picks/sec through GCN_Detection_Network_extended.forward_fixed_source on MI355X.

One "step" = one forward window (`forward_fixed_source`, reference module.py:999) on fixed shapes with the
graphs already set (`set_adjacencies` runs once per day in the reference, process_continuous_days.py:622-649),
inputs (`Slice`, `Mask`) resident in HBM. picks/sec = N_picks x windows/sec.

N = 1 workload: BASELINE.json configs[1] = 200 stations / 10 000 grid nodes / 50 000 picks (cfg2).
N > 1 (default `--mode sharded --config cfg4_2000x50k`): BASELINE.json configs[3] = 2000 stations / 50 000 grid nodes /
500 000 picks, ONE window sharded over source nodes across the N ranks (genie_amd/dist.py: halo all-to-all of the
projected `wv` rows + all-gather of the [G,15] Bipartite output over RCCL/xGMI) => strong scaling.
`--mode replicas`: window-parallel replicas of cfg2 (no collective, weak scaling); `--mode stream`: config 5.

`python bench.py --gpus N` launches its own N ranks (torch.distributed.run on 127.0.0.1) when it is not already
running under one (WORLD_SIZE unset); under `torchrun` it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.

Prints ONE JSON line (rank 0). Extra objects: `roofline` (path-level algorithmic bytes B_alg x windows/s
against the 8 TB/s HBM peak; per-kernel HIP-event times and counter traffic as extras) and `cpu_baseline` (the oracle's
reference-formulation forward on the host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from genie_amd import graph, module, synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32 matrix/vector peak

# Algorithmic bytes: whole path B_alg = 1532*P + 816*G bytes per window (reference dataflow at layer
# granularity). The HIP path moves fewer real bytes than that because h0/h1/u/v never leave the registers; per
# product node and per kernel group (gathers counted once per row = perfect cache):
#   stage 1 = k_split_rows + k_stage1_h2: Slice+Mask 32 R, split rows 32 W + 32 R, message mask 4 W, c 120 W, wu+wv 120 W = 340 B
#   stage 2 = k_stage2_h2u              : c 120 R, wu+wv 120 R, message mask 4 R, edge_attr fragments 32 R             = 276 B
B_NODE = {"k_stage1": 340.0, "k_stage2": 276.0}
# ALGORITHMIC FLOPs per product node (SURVEY.md 8d split by kernel; 2 per MAC): stage 1 = init_trns 240 + layer-1 3840
# + l2_t*_1 3600 + l2_t*_2 2820 MACs + 690 layer-1 gather adds; stage 2 = Bipartite fc1 990 MACs + 690 gather adds.
F_NODE = {"k_stage1": 2.0 * (240 + 3840 + 3600 + 2820) + 690.0, "k_stage2": 2.0 * 990 + 690.0}
# What k_stage1_h2 EXECUTES on the 16-bit matrix pipe: 120 v_mfma_f32_32x32x16_f16 (32768 FLOP each) per 32 nodes = three
# fp16 partial products per fp32 product, init_trns recomputed for the 23 neighbours, 30-wide blocks padded to 32.
F16_EXEC_FLOP_NODE = 120 * 32768.0 / 32.0
F16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16
FLOP_NODE = 22980.0 + 1380.0   # reference dense + gather adds per product node


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--settle", type=int, default=1500,
                    help="untimed windows run during set-up, before the W warm-up steps: the GPU clocks need ~0.5 s of sustained "
                         "load to settle (a cold start reads 5-10 %% slow for the first few hundred windows; DESIGN.md section 5)")
    ap.add_argument("--config", default=None, choices=sorted(synthetic.CONFIGS),
                    help="default: cfg2_200x10k (N = 1, replicas, stream), cfg4_2000x50k (sharded with N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--windows", type=int, default=4, help="distinct synthetic pick windows cycled through")
    ap.add_argument("--no-pipeline", action="store_true", help="single-stream forward_fixed_source per window")
    ap.add_argument("--push-flush", action="store_true",
                    help="with --tail-batch 1: push_window / flush_windows (genie_tail_batched over one window) instead of "
                         "forward_fixed_source_pipelined (the same tail launched call by call)")
    ap.add_argument("--tail-batch", type=int, default=None,
                    help="windows per G-sized tail (push_window / flush_windows), 1..16; 1 = one tail per window. Default: 16 (the tail "
                         "kernels are latency-bound: same box 0.582 ms per window at 8, 0.576 at 16; DESIGN.md section 5)")
    ap.add_argument("--mode", default=None, choices=["replicas", "sharded", "stream", "train"],
                    help="default: the cfg2 window pipeline at N = 1; at N > 1 ONE cfg4 window sharded over source nodes with an "
                         "RCCL halo all-to-all + all-gather per window (strong scaling). replicas = window-parallel copies of "
                         "cfg2 (weak scaling, no collective); stream = config 5")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="no GPU: launch the ranks, build the sharding plan and run the two collectives of the sharded path on CPU "
                         "tensors over gloo (checks the launcher / rank plumbing and the exchange; prints a JSON line, no timing)")
    ap.add_argument("--no-overlap", action="store_true", help="sharded: sequential schedule (stage 1, exchange, stage 2, all-gather)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N > 1: nccl = RCCL over xGMI, one GPU per rank (the measurement). gloo = a FUNCTIONAL check of the N > 1 code path on "
                         "a box with fewer GPUs than ranks: the ranks share the visible GPUs round robin and the collectives are staged through "
                         "host memory (dist.Transport); its timings mean nothing and the line says so")
    ap.add_argument("--cpu-windows", type=int, default=3, help="cpu_baseline: timed windows after one warm-up (median)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1 default run: skip the two rocprofv3 --pmc passes that measure roofline.traffic in this run (the line "
                         "then carries the constants of the committed profile)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the live config-3 training-step figure of the default line")
    ap.add_argument("--no-stream", action="store_true", help="skip the live config-5 streaming figure of the default line")
    ap.add_argument("--no-day-loops", action="store_true", help="skip the refine / association loop figures of the default line")
    ap.add_argument("--emulate-world", type=int, default=None,
                    help="N = 1: time every rank of the W-rank shard plan of the sharded workload alone on this GPU (collectives replaced by "
                         "device copies of the same size) and report per-rank ms, imbalance, halo MB and the projected speed-up over the "
                         "one-GPU time -- labelled projected, no xGMI. With --mode sharded: only that. The default N = 1 line carries it "
                         "for W = 8 as `projected_scaling_8` (skip with --no-cfg4-one-gpu)")
    ap.add_argument("--no-cfg4-one-gpu", action="store_true",
                    help="N = 1 default run: skip the short measurement of the N > 1 workload (config 4, one window sharded over source "
                         "nodes) on this one GPU that the line carries as `sharded_workload_on_one_gpu`")
    return ap.parse_args()


def respawn_under_torchrun(a):
    """`python bench.py --gpus N` with N > 1 outside a launcher: start N ranks of this script on this node."""
    import socket
    import subprocess
    n_vis = torch.cuda.device_count()
    if a.gpus > n_vis and not a.dry_run_cpu and a.backend != "gloo":
        print("bench.py: --gpus %d needs %d visible GPUs, this node shows %d; nothing was launched" % (a.gpus, a.gpus, n_vis), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_model(geom, dev, seed=0):
    torch.manual_seed(seed)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev).eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                             torch.from_numpy(geom.edge_attr()).to(dev),
                             torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev))
    return net


def physical_cores():
    try:
        import psutil
        return int(psutil.cpu_count(logical=False) or os.cpu_count() or 1)
    except Exception:
        return int(os.cpu_count() or 1)


def measure_traffic(timeout_s=150):
    """HBM-side bytes per launch of the P-sized kernels, measured now: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE:
    the two do not fit one pass) over tools/stage_profile.py (the same kernels on the same config-2 workload, 5 windows), averaged
    per kernel; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at
    64 B). Returns {"k_stage1": bytes, "k_stage2": bytes} or None when rocprofv3 is unavailable / fails (the line then carries the
    traffic fields of the line are then null: no constant stands in for a measurement)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) or k == "HSA_TOOLS_LIB" for k in os.environ):
        return None          # this process is being profiled itself: no nested profiler
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="genie_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([rp, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(REPO, "tools", "stage_profile.py"), "cfg2_200x10k", "5"],
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                disp = {}
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    key = (row["Kernel_Name"], row["Dispatch_Id"])
                    disp[key] = disp.get(key, 0.0) + float(row["Counter_Value"])
                for (kname, _), v in disp.items():
                    for tag in ("k_split_rows", "k_stage1_h2", "k_stage2_h2"):
                        if tag in kname:
                            acc.setdefault(tag, []).append(v)
            if not all(t in acc for t in ("k_split_rows", "k_stage1_h2", "k_stage2_h2")):
                return None
            per[counter] = {t: float(np.mean(v)) for t, v in acc.items()}
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    byts = lambda t: (2.0 * per["FETCH_SIZE"][t] + per["WRITE_SIZE"][t]) * 1024.0
    return {"k_stage1": byts("k_split_rows") + byts("k_stage1_h2"), "k_stage2": byts("k_stage2_h2")}


def cpu_baseline(net, geom, win, n_timed=3):
    """Oracle ('port' of the reference formulation: explicit product edge lists, gather + scatter-mean) timed on this box's
    host cores on the same workload: one warm-up window, then the median of `n_timed` windows with torch's
    default thread count; and a single-thread figure on a bounded sample (the first G/10 source nodes of the same window with
    their own kNN graph, one warm-up + one timed run, scaled linearly in the number of product nodes)."""
    from oracle import genie_oracle as O
    w = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    S, G = geom.n_sta, geom.n_grid
    A_in_sta, A_in_src, A_src_in_prod, _ = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    args = (w, torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"]), A_in_sta, A_in_src,
            torch.from_numpy(geom.edge_attr()), A_src_in_prod, torch.from_numpy(geom.A_src_src),
            torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(),
            torch.from_numpy(geom.t_query).float())
    times = []
    with torch.no_grad():
        for k in range(1 + max(1, n_timed)):
            t0 = time.perf_counter()
            y, x = O.forward_fixed_source(*args)
            if k:
                times.append(time.perf_counter() - t0)
    del args, A_in_sta, A_in_src, A_src_in_prod
    # single thread, bounded sample
    Gs = max(64, G // 10)
    A_sub = graph.knn_graph(geom.x_grid[:Gs] / 1000.0, min(synthetic.K_SPC, Gs - 1))
    B_in_sta, B_in_src, B_src_in_prod, _ = graph.cartesian_product_edges(geom.A_sta_sta, A_sub, S, Gs)
    sub = (w, torch.from_numpy(win["Slice"][: Gs * S]), torch.from_numpy(win["Mask"][: Gs * S]), B_in_sta, B_in_src,
           torch.from_numpy(geom.edge_attr(slice(0, Gs))), B_src_in_prod, torch.from_numpy(A_sub),
           torch.from_numpy(geom.x_grid[:Gs]).float(), torch.from_numpy(geom.x_query[: max(10, geom.x_query.shape[0] // 10)]).float(),
           torch.from_numpy(geom.t_query).float())
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with torch.no_grad():
            O.forward_fixed_source(*sub)
            t0 = time.perf_counter()
            O.forward_fixed_source(*sub)
            t1 = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthreads)
    return y, x, float(np.median(times)), times, t1 * (G / float(Gs)), Gs


SPARSE_PICKS = 400      # picks of the parity window whose masks gate (the headline window of 50 000 picks saturates Mask: mean 0.999997;
                        # 5 000 picks = 25 per station in a 140-s window under the 3-s kernel still leave only 0.04 % all-zero rows)


def sparse_window(geom, n_picks=SPARSE_PICKS, seed=9):
    """A window of the same shape with FEW picks (default 400 on 200 stations), so that a third of the product nodes has an all-zero
    `Mask` row and the `mask.max(1)` gate of Bipartite_ReadIn (module.py:226-229) and the mask inputs of DataAggregation really
    select: the parity check VERDICT round 4 asked for next to the saturated headline window."""
    return synthetic.make_window(geom, n_picks, seed=seed, window=7)


def sparse_window_parity(net, geom, locs, xg, xq, tq, dev, n_picks=SPARSE_PICKS):
    """max |HIP - oracle| on `sparse_window` (one oracle run on the host cores, reference formulation)."""
    from oracle import genie_oracle as O
    win = sparse_window(geom, n_picks)
    w = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    S, G = geom.n_sta, geom.n_grid
    A_in_sta, A_in_src, A_src_in_prod, _ = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    with torch.no_grad():
        yc, xc = O.forward_fixed_source(w, torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"]), A_in_sta, A_in_src,
                                        torch.from_numpy(geom.edge_attr()), A_src_in_prod, torch.from_numpy(geom.A_src_src),
                                        torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(),
                                        torch.from_numpy(geom.t_query).float())
        yg, xg_ = net.forward_fixed_source(torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev), None, None, None,
                                           locs, xg, xq, tq)
    M = win["Mask"]
    return {"n_picks": int(win["n_picks"]), "mask_mean": round(float(M.mean()), 4), "all_zero_mask_rows": round(float((M.max(1) == 0).mean()), 4),
            "max_abs_y_vs_cpu": float((yg.cpu() - yc).abs().max()), "max_abs_x_vs_cpu": float((xg_.cpu() - xc).abs().max()),
            "max_abs_y": float(yc.abs().max())}


def day_loops_leg(net, geom, dev, n_sources=8, n_rand_query=112000):
    """The two other per-day loops of the caller on the config-2 shape (genie_amd/apply.py): the refine pass (process_continuous_days.py:926-980:
    per candidate source one window embedded on the device, forward_fixed_source read out at a cloud of 112 000 random queries,
    process_config.yaml:25, argmax) and the association pass (:1020-1065: per refined source a 4-output forward_fixed with the window's
    pick lists). A synthetic hour of picks (250 picks / station / day + a few events), 8 candidate sources. Wall clock, synchronised."""
    from genie_amd import apply
    S, G = geom.n_sta, geom.n_grid
    rng = np.random.default_rng(11)
    n_bg = int(250 * S / 24)
    P = np.stack([rng.uniform(0.0, 3600.0, n_bg), rng.integers(0, S, n_bg).astype(np.float64), np.ones(n_bg), np.ones(n_bg),
                  rng.integers(0, 2, n_bg).astype(np.float64)], axis=1)
    trv = geom.travel_times().astype(np.float32)
    max_t = float(np.ceil(trv.max() + 1.0))
    nodes = rng.choice(G, n_sources, replace=False)
    t_org = np.sort(rng.uniform(300.0, 3300.0, n_sources))
    ev = []
    for g, t0 in zip(nodes, t_org):
        for ph in (0, 1):
            keep = rng.random(S) < 0.8
            tt = t0 + trv[g, keep, ph] + rng.normal(0.0, 0.1, int(keep.sum()))
            ev.append(np.stack([tt, np.nonzero(keep)[0].astype(np.float64), np.ones_like(tt), np.ones_like(tt), np.full_like(tt, ph)], axis=1))
    P = np.concatenate([P] + ev, axis=0)
    P = P[rng.permutation(P.shape[0])]
    sig = synthetic.KERNEL_SIG_T
    t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).float().to(dev)
    A_edges_p, A_edges_s, dt_partition = graph.time_pointers(trv, max_t=max_t, dt=sig / 5.0, k=10, win=2.0 * sig)
    net.A_edges_p, net.A_edges_s = torch.from_numpy(A_edges_p).to(dev), torch.from_numpy(A_edges_s).to(dev)
    net.dt_partition, net.tlatent = t(dt_partition), t(trv.reshape(-1, 2))
    picks = apply.ResidentPicks(P, np.arange(S), S, dev)
    leg = apply.GridLeg(net, geom.x_grid, trv)
    srcs = np.concatenate((geom.x_grid[nodes] + rng.normal(0.0, 2000.0, (n_sources, 3)), (t_org + rng.normal(0.0, 0.5, n_sources)).reshape(-1, 1),
                           np.full((n_sources, 1), 0.5)), axis=1)
    ident = lambda x: x
    L = geom.L
    kw = dict(kernel_sig_t=sig, dt_embed=0.3)
    refine = lambda: apply.refine_sources([leg], picks, srcs, geom.locs, geom.t_query, max_t, np.array([[-15e3, -15e3, -7.5e3]]),
                                          np.array([[30e3, 30e3, 15e3]]), n_rand_query, ident, ident, (0.0, L), (0.0, L), (-40e3, 2e3),
                                          rand=np.random.RandomState(3).rand, ftrns2_device=ident, **kw)
    refine()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    refined, _ = refine()
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    d = np.linalg.norm(refined[:, None, 0:3] - geom.locs[None, :, :], axis=2)
    trv_out_srcs = np.stack((d / synthetic.VP, d / synthetic.VS), axis=2).astype(np.float32)
    assoc = lambda: apply.associate_sources([leg], picks, refined, geom.locs, geom.t_query, max_t, trv_out_srcs, ident,
                                            np.array([0.0, 0.0, 0.0]), **kw)
    assoc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Out_p, Out_s, Save_picks, _ = assoc()
    torch.cuda.synchronize()
    t_as = time.perf_counter() - t0
    n_assoc = int(sum(int(((p > 0.1) | (s_ > 0.1)).sum()) for p, s_ in zip(Out_p, Out_s)))
    return {"config": "synthetic hour on the config-2 shape: %d picks, %d candidate sources" % (P.shape[0], n_sources),
            "refine_ms_per_source": round(t_ref / n_sources * 1e3, 3), "refine_queries_per_source": n_rand_query,
            "refine_max_value": round(float(refined[:, 4].max()), 4),
            "association_ms_per_source": round(t_as / n_sources * 1e3, 3), "picks_per_window": round(float(np.mean([len(x) for x in Save_picks])), 1),
            "picks_above_0.1": n_assoc,
            "note": "refine: query cloud on the device (host draw; `ftrns2_device`), device embedding + kNN of the cloud (genie_knn) + "
                    "forward_fixed_source + argmax per source, one host copy at the end; association: device embedding + pick lists (ResidentPicks.pick_inputs) + forward_fixed per source; Out_p_save / "
                    "Out_s_save stay on the device"}


def main_stream(a, geom, nq, rank, world, dev, dist, emit=True):
    """BASELINE config 5: continuous-day sliding-window inference (86 400 s at 1 s stride), every rank holds the model
    and takes every world-th window; picks + travel-time table resident on the GPU, Slice/Mask embedded on device
    (genie_embed_window), Out_2 stacked on device. No collective in the loop (replicas)."""
    S, G = geom.n_sta, geom.n_grid
    net = build_model(geom, dev)
    hp = net._hip
    rng = np.random.default_rng(5)
    n_day = 250 * S                                                      # ~250 picks / station / day (BSSA NC data)
    P = np.stack([np.sort(rng.uniform(0.0, 86400.0, n_day)), rng.integers(0, S, n_day).astype(np.float64), np.ones(n_day),
                  np.ones(n_day), rng.integers(0, 2, n_day).astype(np.float64)], axis=1)
    trv = geom.travel_times().astype(np.float32)
    max_t, sig, dt = float(np.ceil(trv.max() + 1.0)), synthetic.KERNEL_SIG_T, 0.3
    d_t = torch.from_numpy(P[:, 0].copy()).to(dev)
    d_sta = torch.from_numpy(P[:, 1].astype(np.int32)).to(dev)
    d_ph = torch.from_numpy(P[:, 4].astype(np.int32)).to(dev)
    d_trv = torch.from_numpy(trv.reshape(-1, 2)).to(dev)
    locs = torch.from_numpy(geom.locs).float().to(dev)
    xg = torch.from_numpy(geom.x_grid).float().to(dev)
    xq = torch.from_numpy(geom.x_query).float().to(dev)
    tq = torch.from_numpy(geom.t_query).float().to(dev)
    stride = 1.0
    n_cols = 86400 * 2
    Out_2 = torch.zeros((xq.shape[0], n_cols), dtype=torch.float32, device=dev)
    base = torch.arange(9, device=dev)
    t_all = 1000.0 + stride * (rank + world * np.arange(a.warmup + a.steps))
    lo = np.searchsorted(P[:, 0], t_all - 2.0 * sig, side="right")
    hi = np.searchsorted(P[:, 0], t_all + max_t + 2.0 * sig, side="left")

    acc = [None]
    first = [0]
    net.window_batch = max(1, min(a.tail_batch if a.tail_batch is not None else 16, 16))

    def flush(upto):
        y, x, _ = net.flush_windows(xg, xq, tq)
        with torch.cuda.stream(hp.side_stream):
            if acc[0] is not None:
                hp.side_stream.wait_event(acc[0])             # batches accumulate in order
            for k in range(x.shape[0]):
                Out_2.index_add_(1, base + int(t_all[first[0] + k]), x[k, :, :, 0])
            acc[0] = torch.cuda.Event()
            acc[0].record(hp.side_stream)
        first[0] = upto

    def step(i):
        Slice, Mask = hp.embed_window(d_t[lo[i]:hi[i]], d_sta[lo[i]:hi[i]], d_ph[lo[i]:hi[i]], float(t_all[i]), max_t, sig, dt, d_trv,
                                      presplit=True)
        if net.window_batch == 1:         # one tail per window, launched call by call on alternating side streams
            y, x, _ = net.forward_fixed_source_pipelined(Slice, Mask, None, None, None, locs, xg, xq, tq)
            with torch.cuda.stream(hp.side_stream):
                if acc[0] is not None:
                    hp.side_stream.wait_event(acc[0])         # windows accumulate in order
                Out_2.index_add_(1, base + int(t_all[i]), x[:, :, 0])
                acc[0] = torch.cuda.Event()
                acc[0].record(hp.side_stream)
        elif net.push_window(Slice, Mask) >= net.window_batch:
            flush(i + 1)

    def drain(upto):
        if net.pending_windows:
            flush(upto)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(a.warmup):
            step(i)
        drain(a.warmup)
        barrier()
        t0 = time.perf_counter()
        for i in range(a.warmup, a.warmup + a.steps):
            step(i)
        drain(a.warmup + a.steps)
        barrier()
        dtm = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dtm], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtm = float(t.item())
    wps = world * a.steps / dtm
    ppw = float(np.mean(hi - lo))
    out = {
        "metric": "picks/sec through GCN_Detection_Network_extended.forward_fixed_source (GCS_Network.forward)",
        "value": round(wps * ppw, 1), "unit": "picks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dtm / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 5: continuous-day sliding-window inference, %d stations / %d grid nodes, 1 s stride, "
                               "%.0f picks in a window's range, device embedding + forward + Out_2 stacking" % (S, G, ppw),
                   "n_stations": S, "n_grid": G, "n_picks": ppw, "n_query": nq,
                   "parallelism": "window-parallel replicas x%d" % world},
        "windows_per_s": round(wps, 2), "seconds_of_data_per_wall_second": round(wps * stride, 2),
        "day_86400_windows_wall_s": round(86400.0 / wps, 1),
    }
    if rank == 0 and emit:
        emit_line(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return out


def single_rank_rccl(dev):
    """A world-size-1 RCCL ("nccl") process group on its own TCP store: `--gpus 1 --mode sharded` and the one-GPU legs of the other
    lines then drive the sharded path through the device collectives RCCL executes at N > 1 (all_to_all_single with split sizes
    into workspace views, all_gather_into_tensor, communication / tail streams) instead of skipping them. Returns
    (torch.distributed or None, error text or None)."""
    import socket
    import torch.distributed as dist
    if not dist.is_available():
        return None, "torch.distributed is not available"
    if dist.is_initialized():
        return dist, None
    try:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        # an explicit store hosted HERE: under torchrun `init_method="tcp://..."` makes even rank 0 a client of the launcher's agent
        # store (TORCHELASTIC_USE_AGENT_STORE), i.e. of a store nobody hosts on this port -- a ten-minute connect timeout
        store = dist.TCPStore("127.0.0.1", port, 1, is_master=True, use_libuv=False)
        dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=torch.device(dev))
        return dist, None
    except Exception as e:          # (the line is still produced, with the reason)
        return None, repr(e)[:200]


def sharded_setup(a, geom, n_picks, rank, world, dev, group_on, emulate=False):
    """The drop-in model of rank `rank` of a `world`-rank job on `dev`, with its adjacencies set, and ONE resident window embedded on the
    device for the rank's own rows: `GCN_Detection_Network_extended(..., process_group=True)` -> `set_adjacencies_base` (config 4's product
    edge lists, 2.3 G edges, cannot be materialised: the base graphs stand for them) -> `node_rows(travel times)` -> `embed_window`
    (genie_embed_window over the owned + halo source nodes only; nothing of the window is made on the host)."""
    torch.manual_seed(0)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev, process_group=True if group_on else None,
                                                shard=(rank, world), shard_overlap=not a.no_overlap, shard_emulate=emulate).eval()
    locs = torch.from_numpy(geom.locs).float().to(dev)
    xg = torch.from_numpy(geom.x_grid).float().to(dev)
    xq = torch.from_numpy(geom.x_query).float().to(dev)
    tq = torch.from_numpy(geom.t_query).float().to(dev)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), geom.edge_attr, locs, xg)
    P = synthetic.make_picks(geom, n_picks, seed=2, window=0)
    d_trv = net.node_rows(geom.travel_times)
    sig = synthetic.KERNEL_SIG_T
    Slice, Mask = net.embed_window(torch.from_numpy(P[:, 0].copy()).to(dev), torch.from_numpy(P[:, 1].astype(np.int32)).to(dev),
                                   torch.from_numpy(P[:, 4].astype(np.int32)).to(dev), 0.0, float(np.ceil(geom.max_t + 1.0)), sig,
                                   float(np.round(sig / 10.0, 2)), d_trv)
    del d_trv
    return net, Slice, Mask, locs, xg, xq, tq


def sharded_phases(net, Slice, Mask, xg, xq, tq, barrier, reps):
    """Per-phase HIP-event times of the SEQUENTIAL schedule on this rank: stage 1, exchange, stage 2 (+ Bipartite read-out),
    all-gather + replicated tail + read-outs."""
    sp = net._shard
    p, S, lp = sp.plan, sp.n_sta, sp.local
    dS, dM = sp.local_rows(Slice, "Slice", 4), sp.local_rows(Mask, "Mask", 4)
    knn = net.SpatialAttention.query_table(xq, xg, 10)
    full = net._hip

    def readouts(x_spatial):
        return full.readout_grid(x_spatial, tq), full.readout_query(x_spatial, xg, xq, knn, tq)

    ph = {k: [] for k in ("stage1", "exchange", "stage2", "gather_tail")}
    with torch.no_grad():
        wv = sp.wv_view()
        for _ in range(reps):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            barrier()
            e[0].record()
            lp.da_stage1_range(dS, dM, 0, p.n_own, True)
            e[1].record()
            sp._exchange(wv)
            e[2].record()
            lp.da_stage2_partials_range(dM[: p.n_own * S], net._edge_attr, 0, p.n_own)
            bip_own = lp.bipartite_readout()
            e[3].record()
            sp.gather_and_tail(bip_own, xg, readouts)
            e[4].record()
            torch.cuda.synchronize()
            for k, name in enumerate(("stage1", "exchange", "stage2", "gather_tail")):
                ph[name].append(e[k].elapsed_time(e[k + 1]))
    return {k: round(float(np.median(v)), 4) for k, v in ph.items()}


def main_sharded(a, geom, n_picks, nq, rank, world, dev, dist, emit=True, one_rank_group=True):
    """ONE window per step, product graph sharded over source nodes across the ranks, THROUGH THE DROP-IN CLASS: the model is built
    with `process_group=`, `set_adjacencies_base` makes this rank's shard plan and contexts, every step is one
    `forward_fixed_source_pipelined(Slice, Mask, ...)` (the reference's arguments; Slice / Mask = the rank's own rows from the device
    embedding). Per window one halo all-to-all (64 B per halo product node; issued as soon as stage 1 has produced the rows other ranks
    need, under the rest of stage 1 and the halo-free part of stage 2) and one all-gather of the [G,15] Bipartite output over RCCL/xGMI;
    the replicated G-sized tail + read-outs of window i run on a tail stream under the P-sized kernels of window i+1 (the windows of the
    apply loop are independent, as in the N = 1 pipeline). --no-pipeline: `forward_fixed_source` (everything on one stream)."""
    S, G = geom.n_sta, geom.n_grid
    rccl_err = None
    if dist is None and world == 1 and one_rank_group:
        dist, rccl_err = single_rank_rccl(dev)
    elif dist is None and world == 1:
        rccl_err = "not requested (the one-GPU leg after an N > 1 run: no second rendezvous under the launcher)"
    net, Slice, Mask, locs, xg, xq, tq = sharded_setup(a, geom, n_picks, rank, world, dev, dist is not None)
    sp = net._shard
    p = sp.plan

    def step():
        if a.no_pipeline:
            return net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
        return net.forward_fixed_source_pipelined(Slice, Mask, None, None, None, locs, xg, xq, tq)[:2]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(a.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            y, x = step()
        barrier()
        dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    phases = sharded_phases(net, Slice, Mask, xg, xq, tq, barrier, min(a.steps, 5))
    ranks = world if dist is None else int(dist.get_world_size())
    wps = a.steps / dt
    b_alg = 1532.0 * S * G + 816.0 * G
    halo_bytes = float(p.n_halo) * S * 64.0
    out = {
        "metric": "picks/sec through GCN_Detection_Network_extended.forward_fixed_source (GCS_Network.forward)",
        "value": round(wps * n_picks, 1), "unit": "picks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d stations / %d grid nodes / %d picks per window, forward_fixed_source of ONE window sharded "
                               "over source nodes behind the drop-in class (process_group=), graphs preset, inputs resident in HBM "
                               "(each rank's own rows, embedded on the device)" % (a.config, S, G, n_picks),
                   "n_stations": S, "n_grid": G, "n_picks": n_picks, "n_query": nq,
                   "parallelism": "source-node sharding x%d (halo all-to-all + all-gather per window, %s)"
                                  % (world, "sequential" if a.no_overlap else "exchange overlapped with compute"),
                   "entry_point": "GCN_Detection_Network_extended(process_group=...).set_adjacencies_base / .embed_window / "
                                  + (".forward_fixed_source" if a.no_pipeline else ".forward_fixed_source_pipelined"),
                   "backend": ("nccl (RCCL)" if dist.get_backend() == "nccl" else "gloo: FUNCTIONAL CHECK ONLY, ranks share GPUs, collectives "
                               "staged through the host -- the timings of this line mean nothing") if dist is not None else "none",
                   "rccl_ranks": ranks if dist is not None and dist.get_backend() == "nccl" else 0,
                   "device_collectives": bool(sp.transport.on and sp.transport.device_collectives),
                   "rccl_single_rank_error": rccl_err,
                   "tail_pipelined": not a.no_pipeline,
                   "rank0_plan": {"n_own": p.n_own, "n_halo": p.n_halo, "send_nodes": p.n_send_nodes, "need_nodes": p.n_need_nodes,
                                  "halo_MB_in_per_window": round(halo_bytes / 1e6, 1)}},
        "windows_per_s": round(wps, 2),
        "rank0_phase_ms_sequential": phases,
        "one_gpu_same_config": None,       # N > 1: filled by main() with a live one-rank run of the same workload
        "roofline": {"bound": "hbm", "kernel": "path (B_alg = 1532 P + 816 G bytes per window, SURVEY.md 8d)",
                     "achieved": round(b_alg * wps / 1e9, 1), "peak": HBM_PEAK_GBS * world,
                     "unit": "GB/s", "frac": round(b_alg * wps / 1e9 / (HBM_PEAK_GBS * world), 4), "traffic": None},
    }
    if rank == 0 and emit:
        emit_line(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return out


XGMI_LINK_GBS = 153.0      # MI355X_MICROARCH.md: one xGMI link, per direction


def emulate_world(a, geom, n_picks, dev, W, one_gpu_ms=None, steps=8, warmup=3):
    """What ONE GPU can say about N = W: every rank of the W-rank shard plan of this workload, one after the other, alone on this GPU
    through the drop-in class with `shard_emulate=True` -- its own plan, contexts, device embedding of its owned + halo rows, sub-range
    stage launches, communication / tail streams; the halo all-to-all and the all-gather replaced by device-to-device copies of the
    same number of bytes (no peer exists: the results are not the model's output, only the times mean something). Per rank: the
    overlapped pipelined window (front on the main stream + tail on the tail stream), the front alone (the tail hidden under the next
    window), the sequential phases, halo MB in / out and what those bytes cost on xGMI at %.0f GB/s per peer link if NOT hidden.
    Projection = one-GPU time of the same workload / slowest rank; labelled projected, no xGMI."""
    import copy
    S, G = geom.n_sta, geom.n_grid
    a = copy.copy(a)
    a.no_overlap, a.no_pipeline = False, False
    ranks = []
    for r in range(W):
        net, Slice, Mask, locs, xg, xq, tq = sharded_setup(a, geom, n_picks, r, W, dev, False, emulate=True)
        sp = net._shard
        p = sp.plan
        with torch.no_grad():
            for _ in range(warmup):
                net.forward_fixed_source_pipelined(Slice, Mask, None, None, None, locs, xg, xq, tq)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                net.forward_fixed_source_pipelined(Slice, Mask, None, None, None, locs, xg, xq, tq)
            torch.cuda.synchronize()
            piped = (time.perf_counter() - t0) / steps * 1e3
            # the front alone (stage 1 | copy standing in for the exchange | stage 2 | Bipartite read-out), overlapped schedule
            dS, dM = sp.local_rows(Slice, "Slice", 4), sp.local_rows(Mask, "Mask", 4)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(steps):
                sp.front(dS, dM, net._edge_attr)
            ev[1].record()
            torch.cuda.synchronize()
            front = ev[0].elapsed_time(ev[1]) / steps
        phases = sharded_phases(net, Slice, Mask, xg, xq, tq, torch.cuda.synchronize, 3)
        per_peer_in = [c * S * 64.0 for c in p.recv_counts]
        per_peer_out = [c * S * 64.0 for c in p.send_counts]
        ranks.append({"rank": r, "n_own": p.n_own, "n_halo": p.n_halo, "send_nodes": p.n_send_nodes, "need_nodes": p.n_need_nodes,
                      "pipelined_window_ms": round(piped, 3), "front_ms": round(front, 3), "phase_ms_sequential": phases,
                      "halo_MB_in": round(sum(per_peer_in) / 1e6, 1), "halo_MB_out": round(sum(per_peer_out) / 1e6, 1),
                      "xgmi_ms_if_not_hidden": round(max(max(per_peer_in), max(per_peer_out)) / (XGMI_LINK_GBS * 1e9) * 1e3, 3),
                      "allgather_KB_in": round((W - 1) * max(len(o) for o in p.owned) * 15 * 4 / 1e3, 1)})
        del net, sp, Slice, Mask, dS, dM
        torch.cuda.empty_cache()
    piped = [x["pipelined_window_ms"] for x in ranks]
    seq = [sum(x["phase_ms_sequential"].values()) for x in ranks]
    out = {"world": W, "config": "%d stations / %d grid nodes / %d picks" % (S, G, n_picks), "per_rank": ranks,
           "max_rank_ms": round(max(piped), 3), "mean_rank_ms": round(float(np.mean(piped)), 3),
           "imbalance_max_over_mean": round(max(piped) / float(np.mean(piped)), 3),
           "max_rank_ms_sequential_tail_not_hidden": round(max(seq), 3),
           "max_xgmi_ms_if_not_hidden": max(x["xgmi_ms_if_not_hidden"] for x in ranks),
           "label": "projected, no xGMI: each rank timed alone on one GPU, collectives replaced by device copies of the same size",
           "how": "drop-in class with shard=(r, %d), shard_emulate=True; %d pipelined windows per rank after %d warm-ups" % (W, steps, warmup)}
    if one_gpu_ms is not None:
        worst = max(x["pipelined_window_ms"] + x["xgmi_ms_if_not_hidden"] for x in ranks)
        out["one_gpu_ms"] = one_gpu_ms
        out["projected_speedup_tail_hidden"] = round(one_gpu_ms / max(piped), 2)
        out["projected_speedup_tail_not_hidden"] = round(one_gpu_ms / max(seq), 2)
        out["projected_speedup_tail_hidden_exchange_not_hidden"] = round(one_gpu_ms / worst, 2)
    return out


TRAIN_FLOP_FACTOR = 3.0     # a training step = forward + two backward products per Linear = 3 x the forward FLOPs


def cpu_train_baseline(geom, win, lbl, lbl_q, frac=10):
    """Oracle training step (reference formulation with explicit product edge lists, autograd, Adam) timed on this box's host
    cores on a BOUNDED sample: the first G / frac source nodes of the same window with their own kNN graph, 1 warm-up + 1 timed
    step, scaled linearly in the number of product nodes."""
    from oracle import genie_oracle as O
    S, G = geom.n_sta, geom.n_grid
    Gs = max(64, G // frac)
    nq = max(10, geom.x_query.shape[0] // frac)
    torch.manual_seed(0)
    ref = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu")
    w = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref.state_dict().items()}
    opt = torch.optim.Adam([v for v in w.values() if v.requires_grad], lr=1e-3)
    A_sub = graph.knn_graph(geom.x_grid[:Gs] / 1000.0, min(synthetic.K_SPC, Gs - 1))
    B_in_sta, B_in_src, B_src_in_prod, _ = graph.cartesian_product_edges(geom.A_sta_sta, A_sub, S, Gs)
    sub = (torch.from_numpy(win["Slice"][: Gs * S]), torch.from_numpy(win["Mask"][: Gs * S]), B_in_sta, B_in_src,
           torch.from_numpy(geom.edge_attr(slice(0, Gs))), B_src_in_prod, torch.from_numpy(A_sub),
           torch.from_numpy(geom.x_grid[:Gs]).float(), torch.from_numpy(geom.x_query[:nq]).float(), torch.from_numpy(geom.t_query).float())
    mse = torch.nn.functional.mse_loss
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        opt.zero_grad()
        y, x = O.forward_fixed_source(w, *sub)
        loss = 0.1 * mse(y[:, :, 0], lbl[:Gs].cpu()) + 0.4 * mse(x[:, :, 0], lbl_q[:nq].cpu())
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    return times[-1] * (G / float(Gs)), Gs, times[-1]


def rebuild_leg(net, opt, geom, n_picks, dev, n_samples=3, reps=3):
    """The 4-output training step in the reference's call convention, `mz(*input_tensors)` with the graphs passed per call
    (train_GENIE_model.py:1786), over `n_samples` samples that each bring ANOTHER station subset, source grid and set of product edge
    lists [2, P ks] / [2, P kp] (int64, resident on the GPU as `torch.Tensor(...).to(device)` leaves them, :1722-1783): every step
    verifies the lists (genie_product_check), orders the new grid, rebuilds the HIP context and its tables. Timed against the same
    samples called twice in a row (second call: same tensors, the context is reused). Wall clock around synchronised steps."""
    from genie_amd import train as gtrain
    S, G = geom.n_sta, geom.n_grid
    t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).float().to(dev)
    samples = []
    for i in range(n_samples):
        g = synthetic.Geometry(S - i, G, L=geom.L, n_query=geom.x_query.shape[0], seed=40 + i)
        smp = synthetic.training_sample(g, min(n_picks, 4000), n_src=4, seed=50 + i, window=0)
        A1, A2, A3, A4 = graph.cartesian_product_edges(g.A_sta_sta, g.A_src_src, g.n_sta, G, device=dev)
        ea = graph.GraphEdges(x=t(g.edge_attr()), edge_index=A3)
        eaf = graph.GraphEdges(x=ea.x, edge_index=A3.flip(0).contiguous())
        args = (t(smp["Slice"]), t(smp["Mask"]), A1, A2, ea, eaf, A4, torch.from_numpy(g.A_src_src).to(dev), t(smp["A_edges_p"]).long(),
                t(smp["A_edges_s"]).long(), t(smp["dt_partition"]), t(smp["tlatent"]), t(smp["tpick"]), t(smp["ipick"]).long(),
                t(smp["phase_label"]), t(g.locs), t(g.x_grid), t(g.x_query), t(smp["x_query_src"]), t(g.t_query), t(smp["tq_sample"]),
                t(smp["trv_out_q"]))
        samples.append((args, (t(smp["Lbls"]), t(smp["Lbls_query"]), t(smp["pick_lbls"]))))

    def step(k):
        args, lab = samples[k]
        opt.zero_grad(set_to_none=True)
        loss = gtrain.reference_loss(net(*args), lab, 1)
        loss.backward()
        opt.step()

    for k in range(n_samples):          # warm-up: allocator size classes, the library's memory pool, kernels
        step(k)
        step(k)
    fresh, again = [], []
    for _ in range(reps):
        for k in range(n_samples):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(k)                      # another sample than the previous step: new graph
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(k)                      # the same tensors again: context reused
            torch.cuda.synchronize()
            fresh.append((t1 - t0) * 1e3)
            again.append((time.perf_counter() - t1) * 1e3)
    f, g2 = float(np.median(fresh)), float(np.median(again))
    # the same two loops WITHOUT a synchronisation between the steps (a training loop that does not read the loss back every step):
    # the host prepares a sample's context while the GPU still works on the previous step's backward
    def loop(ks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in ks:
            step(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / len(ks)
    ks = [k for _ in range(reps + 1) for k in range(n_samples)]
    loop(ks[:n_samples])
    fu = loop(ks)
    step(0)
    gu = loop([0] * len(ks))
    net.invalidate_graph_cache()
    return {"step_with_rebuild_ms": round(f, 3), "step_same_graph_ms": round(g2, 3), "rebuild_ms": round(f - g2, 3),
            "unsynchronised_loop": {"step_with_rebuild_ms": round(fu, 3), "step_same_graph_ms": round(gu, 3), "rebuild_ms": round(fu - gu, 3),
                                    "steps": len(ks), "note": "no synchronisation between the steps: the structure checks of a new graph are "
                                    "read after the forward's kernels are issued, so the rebuild runs under the previous step's backward"},
            "samples": n_samples, "stations": [S - i for i in range(n_samples)], "n_grid": G,
            "product_edges_per_sample": int(samples[0][0][2].shape[1] + samples[0][0][3].shape[1]),
            "note": "median over %d steps each; rebuild = Cartesian check of the int64 product edge lists on the device (genie_product_check), base "
                    "tables, space-filling-curve orders on the device, context + tables from the library's memory pool" % (reps * n_samples)}


def main_train(a, geom, n_picks, nq, rank, world, dev, dist, emit=True):
    """BASELINE config 3: the training step on the config-2 shape. One step = forward_fixed_source in train() mode (the whole
    path in HIP in both directions, module._PathTrain) + the y / x terms of the reference's weighted MSE (train_GENIE_model.py:1789)
    + backward + one Adam(1e-3) step (:1861), on a synthetic window resident in HBM. N > 1: independent replicas (the reference has
    no multi-GPU training). The reference's own 4-output step `mz(*input_tensors)` (association heads included, every module in HIP in
    both directions) is timed next to it."""
    S, G = geom.n_sta, geom.n_grid
    P = S * G
    torch.manual_seed(0)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
    t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).float().to(dev)
    locs, xg, xq, tq = t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), locs, xg)
    wins = [synthetic.make_window(geom, n_picks, seed=2, window=rank * 1000 + i) for i in range(max(1, min(a.windows, 2)))]
    dS, dM = [t(w["Slice"]) for w in wins], [t(w["Mask"]) for w in wins]
    rng = np.random.default_rng(7)
    lbl = t(rng.random((G, 9)) * (rng.random((G, 1)) < 0.1))
    lbl_q = t(rng.random((nq, 9)) * (rng.random((nq, 1)) < 0.1))
    mse = torch.nn.functional.mse_loss
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []

    def step(i):
        k = i % len(dS)
        opt.zero_grad(set_to_none=True)
        y, x = net.forward_fixed_source(dS[k], dM[k], None, None, None, locs, xg, xq, tq)
        loss = 0.1 * mse(y[:, :, 0], lbl) + 0.4 * mse(x[:, :, 0], lbl_q)
        loss.backward()
        opt.step()
        losses.append(loss.detach())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    barrier()
    losses.clear()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / a.steps * 1e3
    loss_vals = [float(v) for v in losses]
    # ---- phases with HIP events on the launch stream (the same entry points, called directly)
    hp = net._hip
    knn = net.SpatialAttention.query_table(xq, xg, 10)
    ph = {k: [] for k in ("fwd_front", "fwd_total", "bwd_tail", "bwd_front")}
    for i in range(min(a.steps, 10)):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        e[0].record()
        r_, xl_, save = hp.train_fwd(dS[0], dM[0], net._edge_attr, want_x_latent=False)
        e[1].record()
        del r_, xl_, save
        e[2].record()
        y, x, xs, _, _, save, tsave = hp.path_train_fwd(dS[0], dM[0], net._edge_attr, xg, xq, knn, tq)
        e[3].record()
        d_r, blob = hp.tail_train_bwd(xg, xq, knn, tq, tsave, y[:, :, 0].contiguous(), x[:, :, 0].contiguous())
        e[4].record()
        hp.train_bwd(dS[0], dM[0], net._edge_attr, save, d_r[:, :30])
        e[5].record()
        torch.cuda.synchronize()
        ph["fwd_front"].append(e[0].elapsed_time(e[1])); ph["fwd_total"].append(e[2].elapsed_time(e[3]))
        ph["bwd_tail"].append(e[3].elapsed_time(e[4])); ph["bwd_front"].append(e[4].elapsed_time(e[5]))
        del y, x, xs, save, tsave, d_r, blob
    pms = {k: round(float(np.median(v)), 4) for k, v in ph.items()}
    pms["fwd_tail"] = round(pms["fwd_total"] - pms["fwd_front"], 4)
    pms["adam_loss_and_host"] = round(ms - pms["fwd_total"] - pms["bwd_tail"] - pms["bwd_front"], 4)
    # ---- the reference's 4-output step (association heads included), a few steps
    four = None
    try:
        smp = synthetic.training_sample(geom, min(n_picks, 4000), n_src=4, seed=3, window=0)
        # (the time-pointer tables of this sample; the graphs were set by set_adjacencies_base above)
        net.A_edges_p, net.A_edges_s = t(smp["A_edges_p"]).long(), t(smp["A_edges_s"]).long()
        net.dt_partition, net.tlatent = t(smp["dt_partition"]), t(smp["tlatent"])
        from genie_amd import train as gtrain
        args4 = (t(smp["Slice"]), t(smp["Mask"]), t(smp["tpick"]), t(smp["ipick"]).long(), t(smp["phase_label"]), locs, xg, xq,
                 t(smp["x_query_src"]), tq, t(smp["tq_sample"]), t(smp["trv_out_q"]))
        lab4 = (t(smp["Lbls"]), t(smp["Lbls_query"]), t(smp["pick_lbls"]))

        def step4():
            opt.zero_grad(set_to_none=True)
            loss = gtrain.reference_loss(net.forward_fixed(*args4), lab4, 1)
            loss.backward()
            opt.step()
        for _ in range(2):
            step4()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step4()
        torch.cuda.synchronize()
        four = {"ms_per_step": round((time.perf_counter() - t1) / 5 * 1e3, 3), "n_picks": int(len(smp["tpick"])), "n_src": 4,
                "note": "mz(*input_tensors) + 4-term loss + backward + Adam; every module in HIP in both directions: shared path (source "
                        "queries riding along), BipartiteGraphReadOutOperator + DataAggregationAssociationPhase, LocalSliceLgCollapse P / S, "
                        "StationSourceAttentionMergedPhases; PyTorch: the loss, Adam, index plumbing"}
    except Exception as e:
        four = {"error": repr(e)[:200]}
    # ---- the reference's call convention: `mz(*input_tensors)` with NEW graphs per sample (train_GENIE_model.py:1722-1786: another
    # station subset, grid and product edge lists every sample), so `forward` verifies the lists and rebuilds the HIP context per step
    rebuild = None
    if four is not None and "error" not in four:
        try:
            rebuild = rebuild_leg(net, opt, geom, n_picks, dev)
        except Exception as e:
            rebuild = {"error": repr(e)[:200]}
    flops = TRAIN_FLOP_FACTOR * (FLOP_NODE * P)
    tf = flops / (ms * 1e-3) / 1e12 * 1.0
    out = {
        "metric": "picks/sec through the training step of GCN_Detection_Network_extended (forward + loss + backward + Adam)",
        "value": round(world * n_picks / (ms * 1e-3), 1), "unit": "picks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 3 = %s training step: %d stations / %d grid nodes / %d picks per window, forward_fixed_source in "
                               "train() mode (whole path in HIP in both directions) + 0.1 MSE(y) + 0.4 MSE(x) + backward + Adam(1e-3), "
                               "graphs preset, window resident in HBM" % (a.config, S, G, n_picks),
                   "n_stations": S, "n_grid": G, "n_picks": n_picks, "n_query": nq,
                   "parallelism": "independent replicas x%d" % world if world > 1 else "single GPU"},
        "steps_per_s": round(world * 1e3 / ms, 2), "loss_first_last": [loss_vals[0], loss_vals[-1]],
        "roofline": {"bound": "mfma", "kernel": "training step (3 x the path's forward FLOPs, SURVEY.md 8d; fp32 MFMA kernels)",
                     "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
                     "traffic": None, "alg_flops_per_step": flops, "phase_ms": pms,
                     "hbm_view": {"alg_bytes_per_step": 3.0 * (1532.0 * P + 816.0 * G),
                                  "frac_of_hbm_peak": round(3.0 * (1532.0 * P + 816.0 * G) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
        "four_output_step": four,
        "four_output_step_new_graph_per_sample": rebuild,
        "loss_curve_parity": "asserted against the oracle's autograd + Adam (1e-7 relative over 20-24 steps) at 7 x 45, 20 x 500 and 200 x 300, "
                             "every parameter gradient at 33 x 257 and 200 x 1500 (tests/test_train_gpu.py, tests/test_hip_parity.py); at this "
                             "size (200 x 10 000: the oracle's autograd needs minutes per step) the tests assert finite, decreasing, "
                             "bitwise-reproducible steps only",
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline and emit:
        c_full, gs, c_s = cpu_train_baseline(geom, wins[0], lbl, lbl_q)
        out["cpu_baseline"] = {"value": round(n_picks / c_full, 1), "unit": "picks/s", "cores": int(torch.get_num_threads()), "kind": "port",
                               "physical_cores": physical_cores(),
                               "sample": "oracle training step (reference formulation, autograd, Adam) on the first %d of %d source nodes of the "
                                         "same window with their own kNN graph, 1 warm-up + 1 timed step (%.1f s), scaled x%.1f (linear in "
                                         "product nodes): %.1f s per full step" % (gs, G, c_s, G / float(gs), c_full)}
    if rank == 0 and emit:
        emit_line(json.dumps(out))
    if dist is not None and emit:
        dist.destroy_process_group()
    return out


def main_dry_run_cpu(a, rank, world):
    """Launcher / plan / collective check without a GPU (tests/test_bench_cpu.py): every rank builds its ShardPlan, sends the
    row blocks its peers list and all-gathers a per-node tensor, over gloo on CPU tensors; values encode (node, station). The line also
    carries what the first real multi-GPU run should see on the wire: per rank the halo bytes IN per window (64 B of `wv` per halo
    product node), the largest single (sender -> receiver) pair and its time on one 153-GB/s xGMI link -- the halo is a point-to-point
    pattern, every pair on its own link, so that pair bounds the exchange."""
    from genie_amd import dist as gdist, engine
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    S, G, n_picks, L, nq = synthetic.CONFIGS[a.config or "cfg1_20x500"]
    geom = synthetic.Geometry(S, G, L=L, n_query=8, seed=1)
    plan = gdist.ShardPlan(geom.A_src_src, G, world, rank, engine.sfc_order(geom.x_grid))
    Sp = min(S, 4)               # stations of the test payload (the byte counts below use the real S)
    code = (torch.arange(G).view(-1, 1) * 4096 + torch.arange(Sp).view(1, -1)).float()                # [G, Sp]
    rows_global = torch.stack((code, -code), dim=2)                                                    # [G, Sp, 2]
    own = rows_global[torch.from_numpy(plan.own_global)].reshape(-1, 2).contiguous()
    ok = True
    for mode in ("a2a", "p2p"):
        halo = gdist.exchange_halo_rows(own, plan, Sp, mode=mode)
        ok = ok and torch.equal(halo.view(plan.n_halo, Sp, 2), rows_global[torch.from_numpy(plan.halo_global)])
    per_node = torch.stack([torch.from_numpy(plan.own_global).float() * k for k in (1.0, 2.0, 3.0)], dim=1)
    gathered = gdist.allgather_owned(per_node, plan)
    ok = ok and torch.equal(gathered[:, 0], torch.arange(G).float()) and torch.equal(gathered[:, 2], 3.0 * torch.arange(G).float())
    flag = torch.tensor([1.0 if ok else 0.0])
    row_bytes = float(S) * 64.0                                  # one halo source node = S rows of wv, 64 B each
    mine = {"rank": rank, "n_own": plan.n_own, "n_halo": plan.n_halo, "recv_nodes": list(plan.recv_counts), "send_nodes": list(plan.send_counts)}
    everyone = [mine]
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    if rank == 0:
        everyone = sorted(everyone, key=lambda d: d["rank"])
        consistent = all(everyone[r]["recv_nodes"][q] == everyone[q]["send_nodes"][r] for r in range(world) for q in range(world))
        mb_in = [round(sum(d["recv_nodes"]) * row_bytes / 1e6, 1) for d in everyone]
        pair = max((d["recv_nodes"][q] * row_bytes / 1e6, q, d["rank"]) for d in everyone for q in range(world)) if world > 1 else (0.0, 0, 0)
        emit_line(json.dumps({"dry_run_cpu": True, "n_gpus": world, "ranks": world if dist is None else dist.get_world_size(),
                              "backend": "gloo" if dist is not None else "none", "ok": bool(flag.item() == 1.0) and consistent,
                              "config": {"workload": "%s sharding plan + collectives only" % (a.config or "cfg1_20x500")},
                              "rank0_plan": {"n_own": plan.n_own, "n_halo": plan.n_halo},
                              "halo": {"bytes_per_halo_source_node": row_bytes, "MB_in_per_rank": mb_in, "MB_in_max": max(mb_in),
                                       "MB_in_mean": round(float(np.mean(mb_in)), 1),
                                       "largest_pair_MB": round(pair[0], 1), "largest_pair": "rank %d -> rank %d" % (pair[1], pair[2]),
                                       "largest_pair_ms_at_153_GBs": round(pair[0] / 153.0, 3),
                                       "peers_per_rank": [sum(1 for v in d["recv_nodes"] if v) for d in everyone],
                                       "send_recv_consistent": consistent, "exchange_modes_checked": ["a2a", "p2p"]}}))
    if dist is not None:
        dist.destroy_process_group()
    return 0 if flag.item() == 1.0 else 1


_REAL_STDOUT = None


def guard_stdout():
    """The driver reads ONE JSON line from stdout. Native libraries write there too (RCCL prints a version banner when a
    communicator is created): from here on file descriptor 1 IS stderr, and the JSON line goes to a saved duplicate of the real
    stdout (emit_line)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(text):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(text, flush=True)
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(respawn_under_torchrun(a))
    guard_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.dry_run_cpu:
        sys.exit(main_dry_run_cpu(a, rank, world))
    if a.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; running %d rank(s)" % (a.gpus, world, world), file=sys.stderr)
    if a.mode is None:
        a.mode = "sharded" if world > 1 else "replicas"
    if a.config is None:
        a.config = "cfg4_2000x50k" if (a.mode == "sharded" and world > 1) else "cfg2_200x10k"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "gloo":         # functional check only: ranks share the visible GPUs, collectives staged through the host
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group("gloo")
        else:
            if local_rank >= torch.cuda.device_count():      # (under an external launcher: fail before any rendezvous can hang)
                print("bench.py: rank %d has no GPU (WORLD_SIZE %d, %d visible GPUs): one GPU per rank is required"
                      % (rank, world, torch.cuda.device_count()), file=sys.stderr)
                sys.exit(2)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    S, G, n_picks, L, nq = synthetic.CONFIGS[a.config]
    geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
    if a.mode == "sharded":
        if world == 1 and a.emulate_world:
            o1 = main_sharded(a, geom, n_picks, nq, rank, world, dev, dist, emit=False)
            torch.cuda.empty_cache()
            o1["projected_scaling_%d" % a.emulate_world] = emulate_world(a, geom, n_picks, dev, a.emulate_world, one_gpu_ms=o1["ms_per_step"])
            emit_line(json.dumps(o1))
            return o1
        if world == 1:
            return main_sharded(a, geom, n_picks, nq, rank, world, dev, dist)
        out = main_sharded(a, geom, n_picks, nq, rank, world, dev, dist, emit=False)       # (destroys the process group)
        if rank == 0:
            # the one-GPU figure of the SAME workload, measured live on this rank after the sharded run (not a constant)
            import copy
            a1 = copy.copy(a)
            a1.steps, a1.warmup, a1.no_overlap = 5, 2, False
            torch.cuda.empty_cache()
            try:
                o1 = main_sharded(a1, geom, n_picks, nq, 0, 1, dev, None, emit=False, one_rank_group=False)
                out["one_gpu_same_config"] = {"ms_per_step": o1["ms_per_step"], "value": o1["value"], "unit": "picks/s", "steps": a1.steps,
                                              "speedup": round(o1["ms_per_step"] / out["ms_per_step"], 2),
                                              "scaling_efficiency": round(o1["ms_per_step"] / out["ms_per_step"] / world, 3),
                                              "note": "the N = 1 line of `bench.py --gpus 1` is ANOTHER workload (config 2, the literal call): the strong-"
                                                      "scaling curve of this line's workload starts at this figure",
                                              "rank0_phase_ms_sequential": o1["rank0_phase_ms_sequential"],
                                              "source": "measured live on rank 0 after the sharded run, same code path with one rank"}
            except Exception as e:
                out["one_gpu_same_config"] = {"error": repr(e)[:200]}
            emit_line(json.dumps(out))
        return out
    if a.mode == "stream":
        return main_stream(a, geom, nq, rank, world, dev, dist)
    if a.mode == "train":
        if a.steps == 300 and a.warmup == 30:       # the window defaults are too many for a training step
            a.steps, a.warmup = 50, 5
        return main_train(a, geom, n_picks, nq, rank, world, dev, dist)
    net = build_model(geom, dev)
    # synthetic pick windows of the fixed shape, resident in HBM (each rank its own windows)
    wins = [synthetic.make_window(geom, n_picks, seed=2, window=rank * 1000 + i) for i in range(a.windows)]
    dS = [torch.from_numpy(w["Slice"]).to(dev) for w in wins]
    dM = [torch.from_numpy(w["Mask"]).to(dev) for w in wins]
    locs = torch.from_numpy(geom.locs).float().to(dev)
    xg = torch.from_numpy(geom.x_grid).float().to(dev)
    xq = torch.from_numpy(geom.x_query).float().to(dev)
    tq = torch.from_numpy(geom.t_query).float().to(dev)

    tail_batch = max(1, min(a.tail_batch if a.tail_batch is not None else 16, 16))
    net.window_batch = tail_batch

    def step(i):
        # windows are independent (the apply loop): the P-sized kernels of every window run on the main stream, the G-sized
        # tail + read-outs of every window (or of `tail_batch` windows in one set of launches) on alternating side streams,
        # under the P-sized kernels of the following windows (bitwise-identical results, tests/test_hip_parity.py).
        # --no-pipeline = single stream
        k = i % a.windows
        if a.no_pipeline:
            return net.forward_fixed_source(dS[k], dM[k], None, None, None, locs, xg, xq, tq)
        if tail_batch == 1 and not a.push_flush:
            return net.forward_fixed_source_pipelined(dS[k], dM[k], None, None, None, locs, xg, xq, tq)[:2]
        if net.push_window(dS[k], dM[k]) >= tail_batch:
            return net.flush_windows(xg, xq, tq)[:2]
        return None

    def drain():        # the tails of the windows pushed so far belong to the steps that pushed them
        if net.pending_windows:
            return net.flush_windows(xg, xq, tq)[:2]
        return None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def literal(i):
        # THE call of the metric: forward_fixed_source(Slice, Mask, tpick, ipick, phase_label, locs, x_grid, x_query, t_query) of the
        # reference's signature (module.py:999), one call per window on the current stream, both read-outs included
        k = i % a.windows
        return net.forward_fixed_source(dS[k], dM[k], None, None, None, locs, xg, xq, tq)

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        for i in range(n):
            literal(i)
        barrier()
        return time.perf_counter() - t0

    with torch.no_grad():
        # (1) the command as given: W warm-up calls, K timed calls on a GPU that has done nothing else yet -> `cold_ms_per_step`
        for i in range(a.warmup):
            literal(i)
        dt_cold = timed(a.steps)
        # (2) clock settle: untimed windows (their own key `settle_windows`; the GPU clocks need ~0.5 s of sustained load, a cold
        # start reads 5-10 % slow), then W warm-up calls and K timed calls again -> `value` / `ms_per_step` (the steady state an apply
        # loop over a day runs in), and a longer timed region next to it when K calls last under 0.1 s
        for i in range(a.settle):
            literal(i)
        for i in range(a.warmup):
            literal(i)
        dt = timed(a.steps)
        n_long = max(a.steps, 1000) if dt < 0.1 else None
        dt_long = timed(n_long) if n_long else None
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tc = torch.tensor([dt_cold], device=dev, dtype=torch.float64)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        dt_cold = float(tc.item())
    ms_per_step = dt / a.steps * 1e3
    windows_per_s = world * a.steps / dt
    value = windows_per_s * n_picks

    # ---- the apply loop's window pipeline over independent windows (push_window / flush_windows: P-sized kernels per window on the
    # main stream, G-sized tails of `tail_batch` windows per set of launches on side streams): an extra, not the headline
    pipe_ms = None
    if not a.no_pipeline:
        npipe = max(a.steps, 160)
        with torch.no_grad():
            for i in range(64):
                step(i)
            drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(npipe):
                step(i)
            drain()
            torch.cuda.synchronize()
            pipe_ms = (time.perf_counter() - t0) / npipe * 1e3
    drop_in_ms = ms_per_step

    # ---- dominant-kernel timing with HIP events on the launch stream (staged API = same kernels) ----
    hp = net._hip
    info = hp.stage_precision()
    prec_detail = ("fp32 in / out, fp32 accumulation; P-sized stages %s (fp16 range guard of the committed weights: largest hidden-state bound "
                   "%.4g, largest weight %.4g, limit 60000; mode %s)"
                   % ("multiply on the 16-bit matrix pipe with every fp32 operand as two fp16 pieces (three partial products per product)"
                      if info["f16x2_active"] else "on fp32 MFMA", info["act_bound"], info["weight_bound"], info["mode"]))
    P = S * G
    ev = {k: [] for k in ("k_stage1", "k_stage2", "path")}
    torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(min(a.steps, 20)):
            k = i % a.windows
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            hp.da_stage1(dS[k], dM[k])
            e[1].record()
            _, bip = hp.da_stage2_bipartite(dM[k], net._edge_attr)
            e[2].record()
            o = bip
            for l in (1, 2, 3):
                o = hp.spatial_agg(l, o, xg)
            e[3].record()
            torch.cuda.synchronize()
            ev["k_stage1"].append(e[0].elapsed_time(e[1]))
            ev["k_stage2"].append(e[1].elapsed_time(e[2]))
            ev["path"].append(e[0].elapsed_time(e[3]))
    kms = {k: float(np.median(v)) for k, v in ev.items()}
    b_alg = 1532.0 * P + 816.0 * G
    path_gbs = b_alg * (windows_per_s / world) / 1e9
    # the path is priced against the HBM roofline on its ALGORITHMIC bytes (reference dataflow at layer
    # granularity, B_alg per window); the fused kernels move fewer real bytes, so the per-kernel figures (algorithmic bytes of
    # what each kernel has to touch, HIP-event time on the launch stream, counter traffic from the committed PMC profile) are
    # reported next to it. Stage 1 additionally reports the fp16 FLOPs it executes against the 16-bit matrix peak: it is the
    # one compute-bound kernel of the path.
    kern = {}
    for k in ("k_stage1", "k_stage2"):
        gbs = B_NODE[k] * P / (kms[k] * 1e-3) / 1e9
        kern[k] = {"ms": round(kms[k], 4), "alg_bytes_per_launch": B_NODE[k] * P, "achieved_GBs": round(gbs, 1),
                   "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                   "traffic": None}
    exec_tf = F16_EXEC_FLOP_NODE * P / (kms["k_stage1"] * 1e-3) / 1e12
    kern["k_stage1"]["kernels"] = "k_split_rows_g + k_stage1_h2"
    kern["k_stage1"]["executed_f16"] = {"tflops": round(exec_tf, 1), "peak": F16_MFMA_PEAK_TF, "frac": round(exec_tf / F16_MFMA_PEAK_TF, 4),
                                         "note": "fp32 operands as two fp16 pieces, three partial products per product, fp32 accumulation"}
    kern["k_stage2"]["kernels"] = "k_stage2_h2u"
    fused_bytes = (B_NODE["k_stage1"] + B_NODE["k_stage2"]) * P + 816.0 * G
    roofline = {"bound": "hbm", "kernel": "path (B_alg = 1532 P + 816 G bytes per window, SURVEY.md 8d)",
                "achieved": round(path_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(path_gbs / HBM_PEAK_GBS, 4),
                "traffic": None, "hbm_real_frac": None,
                # the same window time against the bytes the FUSED design itself has to move (616 B per product node: h0 / h1 / u / v never
                # leave the registers) -- the design's own HBM floor, far below `frac`, which prices the reference dataflow's bytes
                "fused_bytes_per_window": fused_bytes,
                "fused_bytes_frac": round(fused_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "pipelined_frac": round(b_alg / (pipe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pipe_ms else None,
                "traffic_source": "not measured in this run (rocprofv3 unavailable, --no-live-traffic, or the bench is itself being profiled)",
                "alg_bytes_per_window": b_alg, "kernels": kern, "single_stream_path_ms": round(kms["path"], 4),
                "fp32_tflops": round(FLOP_NODE * P * (windows_per_s / world) / 1e12, 2)}

    out = {
        "metric": "picks/sec through GCN_Detection_Network_extended.forward_fixed_source (GCS_Network.forward)",
        "value": round(value, 1), "unit": "picks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "dtype_detail": prec_detail,
        "settle_windows": a.settle,
        "cold_ms_per_step": round(dt_cold / a.steps * 1e3, 4),
        "settled_ms_per_step": round(ms_per_step, 4),
        "settled_long_run": {"steps": n_long, "ms_per_step": round(dt_long / n_long * 1e3, 4)} if n_long else None,
        "config": {"workload": "%s: %d stations / %d grid nodes / %d picks per window; one step = ONE literal call forward_fixed_source(Slice, "
                               "Mask, tpick, ipick, phase_label, locs, x_grid, x_query, t_query) of the reference's signature on one stream, both "
                               "read-outs included, graphs preset, inputs resident in HBM. `cold_ms_per_step` = the command as given (%d warm-up "
                               "calls, then %d timed calls on an idle GPU); `value` / `ms_per_step` / `settled_ms_per_step` = %d warm-up + %d timed "
                               "calls again after %d untimed clock-settle windows (`settle_windows`; --settle 0 makes the two the same "
                               "measurement)%s"
                               % (a.config, S, G, n_picks, a.warmup, a.steps, a.warmup, a.steps, a.settle,
                                  "; `settled_long_run` = %d more timed calls (the %d requested ones last under 0.1 s)" % (n_long, a.steps) if n_long else ""),
                   "n_stations": S, "n_grid": G, "n_picks": n_picks, "n_query": nq,
                   "parallelism": "window-parallel replicas x%d" % world if world > 1 else "single GPU"},
        "windows_per_s": round(windows_per_s, 2),
        "pipelined_windows_ms": round(pipe_ms, 4) if pipe_ms else None, "tail_batch": 1 if a.no_pipeline else tail_batch,
        "pipelined_windows_note": "the apply loop's form over independent windows (push_window / flush_windows: P-sized kernels per window, G-sized "
                                  "tails and read-outs of %d windows per set of launches on side streams; bit-identical results): an extra, `value` "
                                  "is the literal call" % tail_batch,
        "drop_in_call_ms": round(drop_in_ms, 4),
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and a.config == "cfg2_200x10k" and not a.no_pipeline and not a.no_live_traffic:
        tr = measure_traffic()
        if tr is not None:
            roofline["traffic"] = tr["k_stage1"] + tr["k_stage2"]
            # the same window time against the bytes the kernels REALLY move: the fused kernels keep h0 / h1 / u / v in registers,
            # so this is far below `frac` (which prices the reference dataflow's bytes); neither P-sized kernel is HBM-bound
            roofline["hbm_real_frac"] = round(roofline["traffic"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roofline["traffic_over_fused_bytes"] = round(roofline["traffic"] / fused_bytes, 3)
            roofline["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over tools/stage_profile.py "
                                          "(same kernels, same workload), (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch of the P-sized kernels")
            for k in ("k_stage1", "k_stage2"):
                kern[k]["traffic"] = tr[k]
    if rank == 0 and world == 1 and a.config == "cfg2_200x10k" and not a.no_pipeline and not a.no_cfg4_one_gpu \
            and torch.cuda.get_device_properties(dev).total_memory > 160e9:
        # The N > 1 lines of this bench run ANOTHER workload (config 4: 2000 stations / 50 000 grid nodes / 500 000 picks, ONE
        # window sharded over source nodes, strong scaling). Its one-GPU figure belongs next to them: measured here through the
        # same code path (a few windows), so that a scaling curve over N = 1, 2, 4, 8 has its own N = 1 point.
        import copy
        a4 = copy.copy(a)
        a4.config, a4.steps, a4.warmup, a4.no_overlap = "cfg4_2000x50k", 5, 2, False
        S4, G4, np4, L4, nq4 = synthetic.CONFIGS[a4.config]
        try:
            o4 = main_sharded(a4, synthetic.Geometry(S4, G4, L=L4, n_query=nq4, seed=1), np4, nq4, 0, 1, dev, None, emit=False)
            out["sharded_workload_on_one_gpu"] = {"config": o4["config"]["workload"], "steps": a4.steps, "ms_per_step": o4["ms_per_step"],
                                                  "value": o4["value"], "unit": "picks/s", "rank0_phase_ms_sequential": o4["rank0_phase_ms_sequential"],
                                                  "roofline_frac": o4["roofline"]["frac"], "backend": o4["config"]["backend"],
                                                  "rccl_ranks": o4["config"]["rccl_ranks"], "device_collectives": o4["config"]["device_collectives"],
                                                  "rccl_single_rank_error": o4["config"]["rccl_single_rank_error"]}
            torch.cuda.empty_cache()
            # ... and what this one GPU can say about N = 8 of it: every rank of the 8-rank plan timed alone (projected, no xGMI)
            W = a.emulate_world or 8
            g4 = synthetic.Geometry(S4, G4, L=L4, n_query=nq4, seed=1)
            out["projected_scaling_%d" % W] = emulate_world(a4, g4, np4, dev, W, one_gpu_ms=o4["ms_per_step"])
            if not a.emulate_world:      # the rest of the curve the driver measures (N = 2, 4), per-rank detail dropped
                curve = {1: 1.0}
                for w in (2, 4):
                    e = emulate_world(a4, g4, np4, dev, w, one_gpu_ms=o4["ms_per_step"], steps=4, warmup=2)
                    curve[w] = e["projected_speedup_tail_hidden"]
                    out["projected_scaling_%d" % w] = {k: v for k, v in e.items() if k != "per_rank"}
                curve[W] = out["projected_scaling_%d" % W]["projected_speedup_tail_hidden"]
                out["projected_strong_scaling_curve_config4"] = {"speedup_by_n_gpus": curve, "label": "projected, no xGMI (tail hidden under the next window)"}
        except Exception as e:       # (the headline line must not depend on it)
            out.setdefault("sharded_workload_on_one_gpu", {"error": repr(e)[:200]})
            out.setdefault("projected_scaling_%d" % (a.emulate_world or 8), {"error": repr(e)[:200]})
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and a.config == "cfg2_200x10k" and not a.no_pipeline and not a.no_train_step:
        # BASELINE config 3 (the training step on this same shape), measured live through `--mode train`'s code path
        import copy
        a3 = copy.copy(a)
        a3.steps, a3.warmup, a3.no_cpu_baseline = 10, 3, True
        try:
            o3 = main_train(a3, geom, n_picks, nq, 0, 1, dev, None, emit=False)
            out["training_step_config3"] = {"config": o3["config"]["workload"], "steps": a3.steps, "ms_per_step": o3["ms_per_step"],
                                            "value": o3["value"], "unit": "picks/s", "roofline_frac_fp32_mfma": o3["roofline"]["frac"],
                                            "phase_ms": o3["roofline"]["phase_ms"], "four_output_step": o3["four_output_step"],
                                            "four_output_step_new_graph_per_sample": o3["four_output_step_new_graph_per_sample"],
                                            "loss_curve_parity": o3["loss_curve_parity"]}
        except Exception as e:
            out["training_step_config3"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and a.config == "cfg2_200x10k" and not a.no_pipeline and not a.no_stream:
        # BASELINE config 5 (continuous-day sliding windows on this shape: device embedding + forward + Out_2 stacking), measured
        # live through `--mode stream`'s code path
        import copy
        a5 = copy.copy(a)
        a5.steps, a5.warmup = 320, 64
        try:
            o5 = main_stream(a5, geom, nq, 0, 1, dev, None, emit=False)
            out["streaming_config5"] = {"config": o5["config"]["workload"], "steps": a5.steps, "ms_per_step": o5["ms_per_step"],
                                        "windows_per_s": o5["windows_per_s"], "day_86400_windows_wall_s": o5["day_86400_windows_wall_s"]}
        except Exception as e:
            out["streaming_config5"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and a.config == "cfg2_200x10k" and not a.no_pipeline and not a.no_day_loops:
        try:
            out["day_loops_config2"] = day_loops_leg(net, geom, dev)
        except Exception as e:
            out["day_loops_config2"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        yc, xc, cdt, ctimes, c1, gs = cpu_baseline(net, geom, wins[0], a.cpu_windows)
        with torch.no_grad():
            yg, xgq = net.forward_fixed_source(dS[0], dM[0], None, None, None, locs, xg, xq, tq)
        out["cpu_baseline"] = {
            "value": round(n_picks / cdt, 1), "unit": "picks/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "physical_cores": physical_cores(),
            "sample": "median of %d windows of the same workload after 1 warm-up (oracle = reference formulation with explicit "
                      "product edge lists, torch CPU fp32, %d threads): %s s" % (len(ctimes), int(torch.get_num_threads()),
                                                                                 ", ".join("%.1f" % t for t in ctimes)),
            "single_thread": {"value": round(n_picks / c1, 1), "unit": "picks/s", "cores": 1,
                              "sample": "first %d of %d source nodes of the same window with their own kNN graph, 1 warm-up + 1 "
                                        "timed run, scaled x%.1f (linear in product nodes): %.1f s per full window"
                                        % (gs, G, G / float(gs), c1)},
            "max_abs_y_vs_cpu": float((yg.cpu() - yc).abs().max()), "max_abs_x_vs_cpu": float((xgq.cpu() - xc).abs().max()),
            "mask_mean_of_that_window": round(float(wins[0]["Mask"].mean()), 6),
            "sparse_window": sparse_window_parity(net, geom, locs, xg, xq, tq, dev),
            "window_of_5000_picks": sparse_window_parity(net, geom, locs, xg, xq, tq, dev, 5000),
            "note": "`cores` is the thread count torch was given, not a scaling claim: the oracle's scatter (index_add_) is serial, so the "
                    "all-thread and the single-thread time per window are about equal on every box seen",
        }
    if rank == 0:
        emit_line(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
