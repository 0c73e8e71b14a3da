"""Golden-vector generator — runs ONLY in the build container (needs /root/reference).

Imports the reference's own `Code/module.py` (with the third-party PyG stack replaced by the
documented-semantics shim in `oracle/ref_shim`, see its README), instantiates the live model
`GCN_Detection_Network_extended` (`module.py:882`), runs `forward_fixed_source` (`module.py:999`) on
seeded synthetic inputs from `genie_amd.synthetic`, and writes inputs + reference outputs +
intermediates (captured with forward hooks on the reference's sub-modules) to `tests/golden/*.npz`.

    python oracle/make_golden.py            # rewrites tests/golden/

The fixtures are data (inputs, weights, expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CODE = "/root/reference/Code"
OUT = os.path.join(REPO, "tests", "golden")


def _import_reference(updated_definition=False, absolute_pos=False, phase_types=True):
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REPO, "oracle", "ref_shim"))
    sys.path.insert(0, REF_CODE)
    sys.path.insert(0, REPO)
    cwd = REF_CODE                # module.py:27-31 reads config.yaml / train_config.yaml from the CWD
    if updated_definition or absolute_pos or not phase_types:
        # `use_updated_model_definition` / `use_absolute_pos` are read from config.yaml at import time (module.py:32-33): import
        # the reference from a scratch directory holding its YAML files with that one flag flipped
        import tempfile
        cwd = tempfile.mkdtemp(prefix="genie_cfg_")
        for f in os.listdir(REF_CODE):
            if f.endswith(".yaml"):
                txt = open(os.path.join(REF_CODE, f)).read()
                if f == "config.yaml" and updated_definition:
                    assert "use_updated_model_definition: False" in txt
                    txt = txt.replace("use_updated_model_definition: False", "use_updated_model_definition: True")
                if f == "config.yaml" and absolute_pos:
                    assert "use_absolute_pos: False" in txt
                    txt = txt.replace("use_absolute_pos: False", "use_absolute_pos: True")
                if f == "config.yaml" and not phase_types:
                    assert "use_phase_types: True" in txt
                    txt = txt.replace("use_phase_types: True", "use_phase_types: False")
                open(os.path.join(cwd, f), "w").write(txt)
    os.chdir(cwd)
    import module as ref_module   # noqa: E402
    assert bool(ref_module.use_updated_model_definition) == bool(updated_definition)
    assert bool(ref_module.use_absolute_pos) == bool(absolute_pos)
    assert bool(ref_module.use_phase_types) == bool(phase_types)
    return ref_module


def _hook_intermediates(mz):
    store = {}

    def mk(name, multi=False):
        def hook(_m, _inp, out):
            if multi:
                store.setdefault(name, []).append(out.detach().clone())
            else:
                store[name] = out.detach().clone()
        return hook

    da = mz.DataAggregation
    handles = [
        da.activate.register_forward_hook(mk("h0")),          # module.py:88
        da.activate1.register_forward_hook(mk("h1")),         # :92
        da.activate21.register_forward_hook(mk("u")),         # :94
        da.activate22.register_forward_hook(mk("v")),         # :95
        da.register_forward_hook(mk("x_latent")),             # :916
        mz.Bipartite_ReadIn.register_forward_hook(mk("bip")),
        mz.SpatialAggregation1.register_forward_hook(mk("sa1")),
        mz.SpatialAggregation2.register_forward_hook(mk("sa2")),
        mz.SpatialAggregation3.register_forward_hook(mk("sa3")),
        mz.SpatialDirect.register_forward_hook(mk("y_latent")),
        mz.SpatialAttention.register_forward_hook(mk("xq")),
    ]
    return store, handles


def run_case(ref, name, geom, Slice, Mask, weights_seed=0, perturb_prelu=False, window=None,
             keep=("h0", "h1", "u", "v", "x_latent", "bip", "sa1", "sa2", "sa3", "y_latent", "xq"),
             row_stride=1, keep64=None, pairs=None, gain=1.0, write=True):
    """`pairs` [2, N] (station, source), sorted by (source, station): run the reference on the IRREGULAR product graph of
    `use_subgraph: True` (process_utils.py:744-849) whose nodes are those pairs; Slice / Mask then have N rows."""
    import torch
    from genie_amd import graph as G

    S, Gn = geom.n_sta, geom.n_grid
    if pairs is None:
        A_prod_sta_sta, A_prod_src_src, A_src_in_prod, A_src_in_sta = G.cartesian_product_edges(
            geom.A_sta_sta, geom.A_src_src, S, Gn)
        spatial_vals = torch.from_numpy(geom.edge_attr())
    else:
        A_src_in_sta = torch.from_numpy(np.asarray(pairs)).long()
        A_prod_sta_sta, A_prod_src_src, A_src_in_prod = G.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
        spatial_vals = torch.from_numpy(geom.edge_attr().reshape(Gn, S, 3)[pairs[1], pairs[0]].copy())
    A_src_src = torch.from_numpy(geom.A_src_src).long()
    results = {}
    for dtype, tag in ((torch.float32, ""), (torch.float64, "64")):
        torch.manual_seed(weights_seed)
        np.random.seed(weights_seed)
        mz = ref.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu")
        if perturb_prelu:
            g = torch.Generator().manual_seed(1234)
            for n_, p_ in mz.named_parameters():
                if p_.numel() == 1:
                    p_.data.fill_(float(0.05 + 0.45 * torch.rand(1, generator=g)))
        if gain != 1.0:            # every Linear weight (not the biases) scaled: activations grow layer by layer, outputs reach O(1)
            for n_, p_ in mz.named_parameters():
                if p_.dim() == 2:
                    p_.data.mul_(gain)
        sd32 = {k: v.detach().clone().numpy() for k, v in mz.state_dict().items()}
        mz = mz.to(dtype).eval()
        store, handles = _hook_intermediates(mz)
        Data = sys.modules["torch_geometric"].data.Data
        A_src_in_edges = Data(x=spatial_vals.to(dtype), edge_index=A_src_in_prod)
        A_Lg_in_src = Data(x=spatial_vals.to(dtype), edge_index=A_src_in_prod.flip(0).contiguous())
        tlatent = torch.from_numpy(geom.travel_times().reshape(-1, 2)).to(dtype)
        # set_adjacencies as process_continuous_days.py:634 does (association tables unused on this path)
        mz.set_adjacencies(A_prod_sta_sta, A_prod_src_src, A_src_in_edges, A_Lg_in_src, A_src_in_sta, A_src_src,
                           torch.zeros(0).long(), torch.zeros(0).long(), torch.zeros(2).to(dtype), tlatent,
                           torch.from_numpy(geom.locs).to(dtype), torch.from_numpy(geom.x_grid).to(dtype))
        with torch.no_grad():
            tp = torch.from_numpy(window["tpick"]).to(dtype) if window else torch.zeros(1).to(dtype)
            ip = torch.from_numpy(window["ipick"]).long() if window else torch.zeros(1).long()
            ph = torch.from_numpy(window["phase_label"]).to(dtype) if window else torch.zeros(1, 1).to(dtype)
            y, x = mz.forward_fixed_source(torch.from_numpy(Slice).to(dtype), torch.from_numpy(Mask).to(dtype),
                                           tp, ip, ph, torch.from_numpy(geom.locs).to(dtype),
                                           torch.from_numpy(geom.x_grid).to(dtype),
                                           torch.from_numpy(geom.x_query).to(dtype),
                                           torch.from_numpy(geom.t_query).to(dtype))
        for h in handles:
            h.remove()
        results["y" + tag] = y.numpy()
        results["x" + tag] = x.numpy()
        for k in (keep if (tag == "" or keep64 is None) else keep64):
            v = store[k].numpy()
            if v.shape[0] == Slice.shape[0] and row_stride > 1:
                v = v[::row_stride]
            results[k + tag] = v
        if tag == "":
            for k, v in sd32.items():
                results["w/" + k] = v.astype(np.float32)
    results.update({
        "n_sta": np.int64(S), "n_grid": np.int64(Gn), "row_stride": np.int64(row_stride),
        "locs": geom.locs, "x_grid": geom.x_grid, "x_query": geom.x_query, "t_query": geom.t_query,
        "A_sta_sta": geom.A_sta_sta, "A_src_src": geom.A_src_src,
        "edge_attr": spatial_vals.numpy().astype(np.float32), "Slice": Slice.astype(np.float32), "Mask": Mask.astype(np.uint8),
    })
    if pairs is not None:
        results["pairs"] = np.asarray(pairs, dtype=np.int64)
    results["weight_gain"] = np.float64(gain)
    if not write:
        return results
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **results)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024.0),
          "| y max %.3e x max %.3e" % (np.abs(results["y"]).max(), np.abs(results["x"]).max()))


def run_embed_case(name, geom, P, t0, kernel_sig_t=3.0, use_sign_input=False):
    """Golden vector for the pick -> Slice/Mask embedding: the reference's own `extract_input_from_data`
    (process_utils.py:460-642) on synthetic picks."""
    import torch
    import process_utils as ref_pu   # noqa: E402  (importable once _import_reference() set sys.path / CWD)
    S, Gn = geom.n_sta, geom.n_grid
    ind_use = np.arange(S)
    trv_times = geom.travel_times().astype(np.float32)                     # [G, S, 2] (x_grids_trv, utils.py:669)
    A_src_in_sta = np.stack([np.tile(np.arange(S), Gn), np.repeat(np.arange(Gn), S)], axis=0)   # process_continuous_days.py:629
    dt = np.round(kernel_sig_t / 10.0, 2)                                  # process_continuous_days.py:608
    max_t = float(np.ceil(trv_times.max() + 1.0))
    [Inpts, Masks], _ = ref_pu.extract_input_from_data(None, P, np.array([t0]), ind_use, geom.locs, geom.x_grid, A_src_in_sta,
                                                        trv_times=trv_times, max_t=max_t, kernel_sig_t=kernel_sig_t, dt=dt,
                                                        use_sign_input=use_sign_input, device="cpu")
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, P=P, t0=np.float64(t0), n_sta=np.int64(S), n_grid=np.int64(Gn), trv_times=trv_times,
                        max_t=np.float64(max_t), kernel_sig_t=np.float64(kernel_sig_t), dt=np.float64(dt),
                        Slice=Inpts[0].numpy().astype(np.float32), Mask=Masks[0].numpy().astype(np.uint8))
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024.0), "nonzero rows", int((Inpts[0].abs().sum(1) > 0).sum()))


def run_pick_inputs_case(name, geom, P, t0, ind_use, kernel_sig_t=3.0, t_win_direct=None):
    """Golden vector for the per-window pick lists: the `[lp_times, lp_stations, lp_phases, lp_meta]` the reference's
    `extract_input_from_data` returns next to `[Inpts, Masks]` (process_utils.py:637, = `extract_pick_inputs_from_data` :644-699 on the
    window's `P_slice`), with `ind_use` a SUBSET of the station file (`P[:, 1]` holds absolute station indices, `locs` the absolute
    set); `t_win_direct`: additionally the reference function called directly on that `P_slice` with another `t_win`, so that the ball
    query trims the list."""
    import process_utils as ref_pu   # noqa: E402
    ind_use = np.asarray(ind_use).astype("int")
    S_all, Gn = geom.n_sta, geom.n_grid
    trv_times = geom.travel_times().astype(np.float32)                     # [G, S_all, 2]
    n_use = len(ind_use)
    A_src_in_sta = np.stack([np.tile(np.arange(n_use), Gn), np.repeat(np.arange(Gn), n_use)], axis=0)
    dt = np.round(kernel_sig_t / 10.0, 2)
    max_t = float(np.ceil(trv_times.max() + 1.0))
    [Inpts, Masks], lp = ref_pu.extract_input_from_data(None, P, np.array([t0]), ind_use, geom.locs, geom.x_grid, A_src_in_sta,
                                                         trv_times=trv_times, max_t=max_t, kernel_sig_t=kernel_sig_t, dt=dt, device="cpu")
    out = dict(P=P, t0=np.float64(t0), n_sta_all=np.int64(S_all), n_grid=np.int64(Gn), ind_use=ind_use.astype(np.int64),
               trv_times=trv_times, max_t=np.float64(max_t), kernel_sig_t=np.float64(kernel_sig_t), dt=np.float64(dt),
               Slice=Inpts[0].numpy().astype(np.float32), Mask=Masks[0].numpy().astype(np.uint8),
               lp_times=np.asarray(lp[0][0], dtype=np.float64), lp_stations=np.asarray(lp[1][0], dtype=np.int64),
               lp_phases=np.asarray(lp[2][0], dtype=np.float64), lp_meta=np.asarray(lp[3][0], dtype=np.float64))
    if t_win_direct is not None:
        from oracle import embed_oracle as E
        P_slice = E.window_pick_slice(P, t0, ind_use, max_t, kernel_sig_t)
        lp2 = ref_pu.extract_pick_inputs_from_data(P_slice, geom.locs, ind_use, np.array([t0]), max_t, t_win=t_win_direct)
        out.update(t_win_direct=np.float64(t_win_direct), lp2_times=np.asarray(lp2[0][0], dtype=np.float64),
                   lp2_stations=np.asarray(lp2[1][0], dtype=np.int64), lp2_phases=np.asarray(lp2[2][0], dtype=np.float64),
                   lp2_meta=np.asarray(lp2[3][0], dtype=np.float64))
        assert len(out["lp2_times"]) < len(out["lp_times"])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024.0), "picks in window", len(out["lp_times"]), "of", P.shape[0])


def main_cfg1_events():
    """`python oracle/make_golden.py --cfg1-events`: SURVEY.md 8c (ii) as specified -- the config-1 shape (20 stations / 500 grid nodes)
    with EVENT-STRUCTURED, non-saturated picks: one synthetic event + 6 noise picks (`cfg1_20x500` holds BASELINE config 1's literal
    2 000 random picks, which saturate Slice / Mask: Mask.mean() = 1.000; under the 3-s kernel even 3 events = 120 picks on 20
    stations still give Mask.mean() 0.95). Here 39 % of the Mask entries are zero and 20 % of the product nodes have an all-zero
    row, so the fixture exercises the mask inputs and the `mask.max(1)` gate at the config-1 shape."""
    ref = _import_reference()
    from genie_amd import synthetic as syn
    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(20, 500, L=100e3, n_query=300, seed=1)
    win = syn.make_window(geom, 30, seed=5)
    M = win["Mask"]
    print("cfg1e window: %d picks, Mask.mean() %.4f, all-zero rows %.4f, Slice > 0.5: %.4f" % (win["n_picks"], M.mean(), (M.max(1) == 0).mean(),
                                                                                              (win["Slice"] > 0.5).mean()))
    run_case(ref, "cfg1e_20x500", geom, win["Slice"], win["Mask"], window=win, row_stride=7, perturb_prelu=True,
             keep=("h0", "h1", "x_latent", "bip", "sa1", "sa2", "sa3", "y_latent", "xq"), keep64=("bip", "sa3"))


def main_picks():
    """`python oracle/make_golden.py --picks`: fixtures `picks_*.npz` of the reference's per-window pick selection."""
    _import_reference()
    from genie_amd import synthetic as syn
    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(14, 60, L=90e3, n_query=5, seed=61)
    rng = np.random.default_rng(63)
    P = syn.make_picks(geom, 420, seed=62)
    P[:, 0] = P[:, 0] * 1.5 + 1000.0
    P[:, 2] = rng.random(P.shape[0])                          # distinct amplitudes / probabilities: lp_meta rows are identifiable
    P[:, 3] = rng.random(P.shape[0])
    # duplicates in (station, time) with the other phase label: their order in the output is the caller's order (stable sorts)
    dup = P[rng.choice(P.shape[0], 25, replace=False)].copy()
    dup[:, 4] = 1.0 - dup[:, 4]
    dup[:, 2] += 1.0
    P = np.concatenate((P, dup), 0)
    P = P[rng.permutation(P.shape[0])]                        # load_picks returns the pick file's order, not a time order (utils.py:983)
    ind_use = np.array([0, 2, 3, 5, 6, 7, 9, 10, 12, 13])     # 10 of the 14 stations of the station file
    run_pick_inputs_case("picks_14x60_a", geom, P, 1004.0, ind_use, kernel_sig_t=3.0, t_win_direct=2.0)
    run_pick_inputs_case("picks_14x60_b", geom, P, 1010.5, np.arange(14), kernel_sig_t=7.0)      # 2 sigma > t_win = 10: the ball trims
    P2 = P.copy()
    P2[:, 0] += 80000.3
    run_pick_inputs_case("picks_14x60_c", geom, P2, 81001.7, ind_use[::2], kernel_sig_t=3.0)


def run_assoc_case(ref, name, geom, win, n_src=4, weights_seed=0, stime=None, pairs=None, zero_phase_columns=False):
    """Golden vector for the 4-output `forward_fixed` (module.py:963-997): source branch + association heads
    (BipartiteGraphReadOutOperator, DataAggregationAssociationPhase, LocalSliceLgCollapse P/S,
    StationSourceAttentionMergedPhases), with the time-pointer tables built by the reference's own
    `utils.assemble_time_pointers_for_stations` (utils.py:602-622, called as at train_GENIE_model.py:1364)."""
    import torch
    import utils as ref_utils   # noqa: E402
    from genie_amd import graph as G
    S, Gn = geom.n_sta, geom.n_grid
    trv = geom.travel_times().astype(np.float32)                                   # [G,S,2]
    max_t = float(np.ceil(trv.max()))
    Slice_in, Mask_in = win["Slice"], win["Mask"]
    if pairs is None:
        A_prod_sta_sta, A_prod_src_src, A_src_in_prod, A_src_in_sta = G.cartesian_product_edges(
            geom.A_sta_sta, geom.A_src_src, S, Gn)
        A_edges_p, A_edges_s, dt_partition = ref_utils.assemble_time_pointers_for_stations(trv, k=10, max_t=max_t, dt=3.0 / 5.0, win=6.0)
        spatial_np, tlat_np = geom.edge_attr(), trv.reshape(-1, 2)
    else:
        # `use_subgraph: True`: the product nodes are the listed (station, source) pairs; the time-pointer tables come from the
        # reference's own builder for such graphs (process_utils.py:851-877, called as at train_GENIE_model.py:1456)
        import process_utils as ref_pu
        A_src_in_sta = torch.from_numpy(np.asarray(pairs)).long()
        A_prod_sta_sta, A_prod_src_src, A_src_in_prod = G.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)

        def trv_pairwise(loc, src):
            d = (src - loc).norm(dim=1)
            return torch.stack((d / syn_VP, d / syn_VS), dim=1)
        A_edges_p, A_edges_s, dt_partition = ref_pu.compute_time_embedding_vectors(
            trv_pairwise, geom.locs.astype(np.float32), geom.x_grid.astype(np.float32), A_src_in_sta, max_t, dt_res=3.0 / 5.0, t_win=6.0)
        rows = pairs[1] * S + pairs[0]
        Slice_in, Mask_in = Slice_in[rows], Mask_in[rows]
        spatial_np, tlat_np = geom.edge_attr()[rows], trv.reshape(-1, 2)[rows]
    if zero_phase_columns:          # what the callers do under use_phase_types: False (process_continuous_days.py:783-786)
        Slice_in, Mask_in = Slice_in.copy(), Mask_in.copy()
        Slice_in[:, 2:] = 0.0
        Mask_in[:, 2:] = 0.0
    torch.manual_seed(weights_seed)
    np.random.seed(weights_seed)
    mz = ref.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu")
    g = torch.Generator().manual_seed(4321)
    for n_, p_ in mz.named_parameters():
        if p_.numel() == 1:
            p_.data.fill_(float(0.05 + 0.45 * torch.rand(1, generator=g)))
    sd32 = {k: v.detach().clone().numpy() for k, v in mz.state_dict().items()}
    mz.eval()
    Data = sys.modules["torch_geometric"].data.Data
    spatial_vals = torch.from_numpy(np.ascontiguousarray(spatial_np))
    A_src_in_edges = Data(x=spatial_vals, edge_index=A_src_in_prod)
    A_Lg_in_src = Data(x=spatial_vals, edge_index=A_src_in_prod.flip(0).contiguous())
    tlatent = torch.from_numpy(np.ascontiguousarray(tlat_np))
    mz.set_adjacencies(A_prod_sta_sta, A_prod_src_src, A_src_in_edges, A_Lg_in_src, A_src_in_sta, torch.from_numpy(geom.A_src_src).long(),
                       torch.from_numpy(A_edges_p).long(), torch.from_numpy(A_edges_s).long(), torch.from_numpy(dt_partition).float(),
                       tlatent, torch.from_numpy(geom.locs).float(), torch.from_numpy(geom.x_grid).float())
    rng = np.random.default_rng(77)
    src_nodes = rng.choice(Gn, n_src, replace=False)
    x_query_src = geom.x_grid[src_nodes] + rng.normal(0, 500.0, (n_src, 3))
    tq_sample = rng.uniform(-2.0, 2.0, n_src).astype(np.float32) if stime is None else np.asarray(stime, dtype=np.float32)
    d = np.linalg.norm(x_query_src[:, None, :] - geom.locs[None, :, :], axis=2)
    trv_out_q = np.stack([d / syn_VP, d / syn_VS], axis=2).astype(np.float32)    # [n_src, S, 2]
    # picks inside the embedding range of the time-pointer table (tpick - dt_partition[0] >= 0)
    keep = (win["tpick"] > dt_partition[0] + 0.5) & (win["tpick"] < dt_partition[-1] - 0.5)
    tpick, ipick, phase = win["tpick"][keep], win["ipick"][keep], win["phase_label"][keep]
    with torch.no_grad():
        out = mz.forward_fixed(torch.from_numpy(np.ascontiguousarray(Slice_in)), torch.from_numpy(np.ascontiguousarray(Mask_in)), torch.from_numpy(tpick),
                               torch.from_numpy(ipick).long(), torch.from_numpy(phase), torch.from_numpy(geom.locs).float(),
                               torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(),
                               torch.from_numpy(x_query_src).float(), torch.from_numpy(geom.t_query).float(),
                               torch.from_numpy(tq_sample), torch.from_numpy(trv_out_q))
    res = {"w/" + k: v.astype(np.float32) for k, v in sd32.items()}
    res.update({"y": out[0].numpy(), "x": out[1].numpy(), "arv_p": out[2].numpy(), "arv_s": out[3].numpy(),
                "n_sta": np.int64(S), "n_grid": np.int64(Gn), "locs": geom.locs, "x_grid": geom.x_grid, "x_query": geom.x_query,
                "t_query": geom.t_query, "A_sta_sta": geom.A_sta_sta, "A_src_src": geom.A_src_src, "edge_attr": np.ascontiguousarray(spatial_np),
                "Slice": Slice_in, "Mask": Mask_in.astype(np.uint8), "tpick": tpick, "ipick": ipick, "phase_label": phase,
                "x_query_src": x_query_src, "tq_sample": tq_sample, "trv_out_q": trv_out_q, "tlatent": np.ascontiguousarray(tlat_np),
                "A_edges_p": np.asarray(A_edges_p), "A_edges_s": np.asarray(A_edges_s), "dt_partition": np.asarray(dt_partition),
                "max_t": np.float64(max_t)})
    if pairs is not None:
        res["pairs"] = np.asarray(pairs, dtype=np.int64)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **res)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024.0), "| picks", len(tpick),
          "arv_p max %.3e arv_s max %.3e" % (np.abs(res["arv_p"]).max(), np.abs(res["arv_s"]).max()))


syn_VP, syn_VS = 6000.0, 3400.0


def main_assoc():
    """Round 3: the pick-sized association heads pinned where the HIP kernels branch: >= 17 stations with uniform degree (pipelined
    kernels, station processing order), one station with > 250 picks (two 192-pick chunks of k_arrivals' streaming softmax), a
    station without any pick, and a call where NO candidate source has |stime| < 2 eps, so that the reference's
    `edge_index[0].max()` (module.py:762-763) is not the null pick."""
    ref = _import_reference()
    from genie_amd import synthetic as syn
    geom = syn.Geometry(20, 60, L=80e3, n_query=30, seed=91)
    P = syn.make_picks(geom, 420, seed=92)
    rng = np.random.default_rng(93)
    extra = np.stack([rng.uniform(-5.0, geom.max_t + 5.0, 270), np.full(270, 3.0), np.ones(270), np.ones(270),
                      rng.integers(0, 2, 270).astype(np.float64)], axis=1)                  # 270 more picks on station 3
    P = np.concatenate([P[P[:, 1] != 7], extra], axis=0)                                      # station 7 has no pick at all
    P = P[np.argsort(P[:, 0], kind="stable")]
    Slice, Mask = syn.make_slice_mask(geom, P, 0.0)
    order = np.lexsort((P[:, 0], P[:, 1]))
    win = {"Slice": Slice, "Mask": Mask, "P": P, "tpick": P[order, 0].astype(np.float32), "ipick": P[order, 1].astype(np.int64),
           "phase_label": P[order, 4].astype(np.float32).reshape(-1, 1), "n_picks": int(P.shape[0])}
    assert int((win["ipick"] == 3).sum()) > 250 and int((win["ipick"] == 7).sum()) == 0
    run_assoc_case(ref, "assoc_20x60", geom, win, n_src=5)
    run_assoc_case(ref, "assoc_20x60_nonull", geom, win, n_src=3, stime=[33.0, -31.5, 36.0])


def main_edges():
    """`python oracle/make_golden.py --edges`: fixtures of the use_updated_model_definition class (DataAggregationEdges,
    module.py:102-174 / :1022-1185); a separate process because the flag is fixed when the reference is imported."""
    ref = _import_reference(updated_definition=True)
    from genie_amd import synthetic as syn

    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(12, 60, L=80e3, n_query=30, seed=91)          # uniform 8 / 15 degrees: the bf16x3 kernels apply
    win = syn.make_window(geom, 150, seed=92)
    run_case(ref, "edges_12x60", geom, win["Slice"], win["Mask"], perturb_prelu=True, window=win, keep64=("bip", "sa3"))
    geom = syn.Geometry(7, 13, L=50e3, n_query=9, seed=93)            # ks = 6, kp = 12: generic CSR kernels
    win = syn.make_window(geom, 40, seed=94)
    run_case(ref, "edges_7x13", geom, win["Slice"], win["Mask"], perturb_prelu=True, window=win,
             keep=("h0", "h1", "x_latent", "bip", "sa3"), keep64=("bip", "sa3"))


def main_abspos():
    """`python oracle/make_golden.py --abspos`: fixtures of the live model with `use_absolute_pos: True` (config.yaml:92): station
    and source positions / (3 scale_rel) appended to every product node's input (module.py:1007), in_channels 4 -> 10."""
    ref = _import_reference(absolute_pos=True)
    from genie_amd import synthetic as syn

    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(12, 60, L=80e3, n_query=30, seed=101)          # uniform 8 / 15 degrees
    win = syn.make_window(geom, 150, seed=102)
    run_case(ref, "abspos_12x60", geom, win["Slice"], win["Mask"], perturb_prelu=True, window=win,
             keep=("h0", "h1", "x_latent", "bip", "sa3"), keep64=("bip", "sa3"))
    geom = syn.Geometry(7, 13, L=50e3, n_query=9, seed=103)            # ks = 6, kp = 12
    win = syn.make_window(geom, 40, seed=104)
    run_case(ref, "abspos_7x13", geom, win["Slice"], win["Mask"], perturb_prelu=True, window=win,
             keep=("h0", "h1", "x_latent", "bip", "sa3"), keep64=("bip", "sa3"))


def main_assoc_variant(flag):
    """`python oracle/make_golden.py --assoc-edges` / `--assoc-abspos`: the 4-output `forward_fixed` of the live model with
    `use_updated_model_definition: True` (DataAggregationAssociationPhaseEdges, module.py:407-480: the mean edge feature enters
    l?_t?_2) resp. `use_absolute_pos: True` (positions appended to the association embedding, module.py:987-988); a separate process
    each because the flags are fixed when the reference is imported."""
    ref = _import_reference(updated_definition=(flag == "edges"), absolute_pos=(flag == "abspos"))
    from genie_amd import synthetic as syn
    geom = syn.Geometry(18, 50, L=70e3, n_query=20, seed=111 if flag == "edges" else 121)     # uniform 8 / 15 degrees, partial last tile
    win = syn.make_window(geom, 220, seed=112 if flag == "edges" else 122)
    run_assoc_case(ref, "assoc_%s_18x50" % flag, geom, win)


def main_edges_abspos():
    """`python oracle/make_golden.py --edges-abspos`: BOTH flags set (`use_updated_model_definition: True` and `use_absolute_pos: True`,
    which the reference's classes accept together, module.py:103-109, :408-412, :1056, :1153): the 2-output fixture
    `edges_abspos_12x60` and the 4-output one `assoc_edges_abspos_18x50`."""
    ref = _import_reference(updated_definition=True, absolute_pos=True)
    from genie_amd import synthetic as syn
    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(12, 60, L=80e3, n_query=30, seed=131)          # uniform 8 / 15 degrees
    win = syn.make_window(geom, 150, seed=132)
    run_case(ref, "edges_abspos_12x60", geom, win["Slice"], win["Mask"], perturb_prelu=True, window=win,
             keep=("h0", "h1", "x_latent", "bip", "sa3"), keep64=("bip", "sa3"))
    geom = syn.Geometry(18, 50, L=70e3, n_query=20, seed=133)          # partial last tile
    win = syn.make_window(geom, 220, seed=134)
    run_assoc_case(ref, "assoc_edges_abspos_18x50", geom, win)


def main_assoc_subgraph():
    """`python oracle/make_golden.py --assoc-subgraph`: the 4-output `forward_fixed` of the live model on an irregular product graph
    (`use_subgraph: True`; the geometry and node list of `--subgraph`), time-pointer tables from the reference's own
    `compute_time_embedding_vectors`."""
    ref = _import_reference()
    from genie_amd import synthetic as syn
    geom = syn.Geometry(14, 50, L=80e3, n_query=21, seed=71)
    rng = np.random.default_rng(72)
    d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
    keep = np.zeros(d.shape, dtype=bool)
    keep[np.arange(d.shape[0])[:, None], np.argsort(d, axis=1)[:, :6]] = True
    keep |= rng.random(d.shape) < 0.12
    src_i, sta_i = np.nonzero(keep)
    pairs = np.stack((sta_i, src_i))
    win = syn.make_window(geom, 180, seed=73)
    run_assoc_case(ref, "assoc_subgraph_14x50", geom, win, pairs=pairs)


def main_assoc_nophase():
    """`python oracle/make_golden.py --assoc-nophase`: the 4-output `forward_fixed` with `use_phase_types: False` (config.yaml:91): the
    reference zeroes `phase_label` inside LocalSliceLgCollapse / StationSourceAttentionMergedPhases (module.py:632-633, :706-707), its
    callers zero the phase-informed columns of Slice / Mask (process_continuous_days.py:783-786)."""
    ref = _import_reference(phase_types=False)
    from genie_amd import synthetic as syn
    geom = syn.Geometry(18, 50, L=70e3, n_query=20, seed=131)
    win = syn.make_window(geom, 220, seed=132)
    run_assoc_case(ref, "assoc_nophase_18x50", geom, win, zero_phase_columns=True)


def main_subgraph():
    """`python oracle/make_golden.py --subgraph`: the live model on an irregular product graph (`use_subgraph: True`): every
    source node keeps its 6 nearest stations plus a few random ones, as the reference's builder keeps the k nearest pairs plus
    those within a distance threshold (process_utils.py:777-794)."""
    ref = _import_reference()
    from genie_amd import synthetic as syn

    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(14, 50, L=80e3, n_query=21, seed=71)
    rng = np.random.default_rng(72)
    d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)          # [G, S]
    keep = np.zeros(d.shape, dtype=bool)
    keep[np.arange(d.shape[0])[:, None], np.argsort(d, axis=1)[:, :6]] = True
    keep |= rng.random(d.shape) < 0.12
    src_i, sta_i = np.nonzero(keep)                                                         # row-major: sorted by (source, station)
    pairs = np.stack((sta_i, src_i))
    full = syn.make_window(geom, 180, seed=73)
    rows = src_i * geom.n_sta + sta_i
    run_case(ref, "subgraph_14x50", geom, full["Slice"][rows], full["Mask"][rows], perturb_prelu=True, window=full,
             keep=("h0", "h1", "x_latent", "bip", "sa1", "sa3", "y_latent"), keep64=("bip", "sa3"), pairs=pairs)
    # the reference's own builder (process_utils.py:744) on the same geometry: fixture for genie_amd.graph.subgraph_product_edges
    import process_utils as pu
    out = pu.extract_inputs_adjacencies_subgraph(geom.locs, geom.x_grid, lambda x: x, lambda x: x, max_deg_offset=0.15,
                                                 k_nearest_pairs=6, k_sta_edges=8, k_spc_edges=15, scale_deg=110e3, device="cpu")
    names = ("A_sta_sta", "A_src_src", "A_prod_sta_sta", "A_prod_src_src", "A_src_in_prod", "A_src_in_sta")
    np.savez_compressed(os.path.join(OUT, "subgraph_builder_14x50.npz"), **{k: np.asarray(v).astype(np.int64) for k, v in zip(names, out)})
    print("wrote subgraph_builder_14x50.npz: %d product nodes" % out[5].shape[1])


def main_subgraph_abspos():
    """`python oracle/make_golden.py --subgraph-abspos`: `use_absolute_pos: True` on an irregular product graph (`use_subgraph: True`):
    fixtures `subgraph_abspos_14x50` (2 outputs) and `assoc_subgraph_abspos_14x50` (4 outputs); geometry and node list of `--subgraph`."""
    ref = _import_reference(absolute_pos=True)
    from genie_amd import synthetic as syn
    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(14, 50, L=80e3, n_query=21, seed=71)
    rng = np.random.default_rng(72)
    d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
    keep = np.zeros(d.shape, dtype=bool)
    keep[np.arange(d.shape[0])[:, None], np.argsort(d, axis=1)[:, :6]] = True
    keep |= rng.random(d.shape) < 0.12
    src_i, sta_i = np.nonzero(keep)
    pairs = np.stack((sta_i, src_i))
    full = syn.make_window(geom, 180, seed=73)
    rows = src_i * geom.n_sta + sta_i
    run_case(ref, "subgraph_abspos_14x50", geom, full["Slice"][rows], full["Mask"][rows], perturb_prelu=True, window=full,
             keep=("h0", "h1", "x_latent", "bip", "sa3"), keep64=("bip", "sa3"), pairs=pairs)
    run_assoc_case(ref, "assoc_subgraph_abspos_14x50", geom, full, pairs=pairs)


def main_embed_sign():
    """`python oracle/make_golden.py --embed-sign`: the pick -> Slice/Mask embedding with `use_sign_input: True` (config.yaml:93): the two
    windows of `embed_14x60_a / _b` through the reference's extract_input_from_data with the flag set."""
    _import_reference()
    from genie_amd import synthetic as syn
    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(14, 60, L=90e3, n_query=5, seed=61)
    P = syn.make_picks(geom, 260, seed=62)
    P[:, 0] += 1000.0
    run_embed_case("embed_sign_14x60_a", geom, P, 1000.0, use_sign_input=True)
    P2 = P.copy()
    P2[:, 0] += 80000.3
    P2 = P2[P2[:, 1] != 5]
    run_embed_case("embed_sign_14x60_b", geom, P2, 81003.1, use_sign_input=True)


def main_subgraph_edges_abspos():
    """`python oracle/make_golden.py --subgraph-edges-abspos`: both model options on an irregular product graph: `assoc_subgraph_edges_abspos_14x50`
    (4 outputs; its (y, x) are the 2-output path)."""
    ref = _import_reference(updated_definition=True, absolute_pos=True)
    from genie_amd import synthetic as syn
    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(14, 50, L=80e3, n_query=21, seed=71)
    rng = np.random.default_rng(72)
    d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
    keep = np.zeros(d.shape, dtype=bool)
    keep[np.arange(d.shape[0])[:, None], np.argsort(d, axis=1)[:, :6]] = True
    keep |= rng.random(d.shape) < 0.12
    src_i, sta_i = np.nonzero(keep)
    pairs = np.stack((sta_i, src_i))
    full = syn.make_window(geom, 180, seed=73)
    run_assoc_case(ref, "assoc_subgraph_edges_abspos_14x50", geom, full, pairs=pairs)


def main_subgraph_edges():
    """`python oracle/make_golden.py --subgraph-edges`: `use_updated_model_definition: True` on an irregular product graph
    (`use_subgraph: True`): the mean edge feature of a product node runs over its PRESENT neighbours only, so the static term is per
    product node there. Fixtures `subgraph_edges_14x50` (2 outputs) and `assoc_subgraph_edges_14x50` (4 outputs); geometry and node list
    of `--subgraph`."""
    ref = _import_reference(updated_definition=True)
    from genie_amd import synthetic as syn
    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(14, 50, L=80e3, n_query=21, seed=71)
    rng = np.random.default_rng(72)
    d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
    keep = np.zeros(d.shape, dtype=bool)
    keep[np.arange(d.shape[0])[:, None], np.argsort(d, axis=1)[:, :6]] = True
    keep |= rng.random(d.shape) < 0.12
    src_i, sta_i = np.nonzero(keep)
    pairs = np.stack((sta_i, src_i))
    full = syn.make_window(geom, 180, seed=73)
    rows = src_i * geom.n_sta + sta_i
    run_case(ref, "subgraph_edges_14x50", geom, full["Slice"][rows], full["Mask"][rows], perturb_prelu=True, window=full,
             keep=("h0", "h1", "x_latent", "bip", "sa3"), keep64=("bip", "sa3"), pairs=pairs)
    run_assoc_case(ref, "assoc_subgraph_edges_14x50", geom, full, pairs=pairs)


def main_scaled():
    """`python oracle/make_golden.py --scaled`: (vi) config-1 shape with every Linear weight multiplied by a common gain chosen
    so that max|y|, max|x| are O(1) (with default-initialised weights the outputs are ~0.03 and the 1e-5 absolute tolerance of
    BASELINE.json is a loose relative one); (vii) 2000 stations x 24 source nodes: the station sum of the Bipartite read-in
    over 2000 terms (the reference's own fp32-vs-fp64 drift there is 2.7e-4 absolute)."""
    ref = _import_reference()
    from genie_amd import synthetic as syn

    os.makedirs(OUT, exist_ok=True)
    geom = syn.Geometry(20, 500, L=100e3, n_query=300, seed=1)
    win = syn.make_window(geom, 2000, seed=2)
    kw = dict(window=win, row_stride=7, perturb_prelu=True, keep=("x_latent", "bip", "sa1", "sa2", "sa3", "y_latent", "xq"),
              keep64=("bip", "sa3"))
    chosen = None
    for gain in (2.0, 2.1, 2.2, 2.3, 2.4, 2.5, 2.6, 2.8, 3.0):
        r = run_case(ref, "o1_20x500", geom, win["Slice"], win["Mask"], gain=gain, write=False, **kw)
        my, mx = float(np.abs(r["y"]).max()), float(np.abs(r["x"]).max())
        print("gain %.2f: max|y| %.3f max|x| %.3f" % (gain, my, mx))
        if min(my, mx) >= 0.5:
            chosen = gain
            break
    assert chosen is not None
    run_case(ref, "o1_20x500", geom, win["Slice"], win["Mask"], gain=chosen, **kw)

    geom = syn.Geometry(2000, 24, L=1000e3, n_query=40, seed=111)
    win = syn.make_window(geom, 24000, seed=112)
    run_case(ref, "s2000_2000x24", geom, win["Slice"], win["Mask"], window=win, perturb_prelu=True, row_stride=97,
             keep=("x_latent", "bip", "sa3"), keep64=("bip", "sa3"))


def main_postproc():
    """`python oracle/make_golden.py --postproc`: the reference's own `LocalMarching` (process_utils.py:40-100) on synthetic
    source candidates (clusters in space-time + isolated points, some exactly tied values), in the call form of the apply script
    (process_continuous_days.py:879: n_steps_max = 2, use_directed = False, scale_depth = 0.2) and with its defaults (directed
    edges, up to 100 steps). Rows are stored sorted (the reference returns them grouped by connected component)."""
    _import_reference()
    import process_utils as pu
    rng = np.random.default_rng(301)
    centres = np.c_[rng.uniform(0, 300e3, (12, 2)), rng.uniform(-30e3, 0, 12), rng.uniform(0, 600.0, 12)]
    pts = []
    for c in centres:
        m = int(rng.integers(2, 14))
        pts.append(np.c_[c[None, :3] + rng.normal(0, 12e3, (m, 3)) * np.array([1, 1, 0.3]), c[3] + rng.normal(0, 2.5, m)])
    pts.append(np.c_[rng.uniform(0, 300e3, (25, 2)), rng.uniform(-30e3, 0, 25), rng.uniform(0, 600.0, 25)])
    X = np.vstack(pts)
    vals = rng.uniform(0.1, 1.0, X.shape[0])
    vals[3] = vals[1]                                   # an exact tie inside a cluster
    srcs = np.c_[X, vals]
    srcs = srcs[np.argsort(srcs[:, 3])]
    ident = lambda x: x
    out = {"srcs": srcs}
    for tag, kw in (("apply", dict(tc_win=4.05, sp_win=27e3, scale_depth=0.2, n_steps_max=2, use_directed=False)),
                    ("default", dict(tc_win=5, sp_win=35e3)),
                    ("wide", dict(tc_win=12.0, sp_win=60e3, scale_depth=0.2, n_steps_max=2, use_directed=False))):
        mp = pu.LocalMarching(device="cpu")
        keep = np.asarray(mp(srcs, ident, **kw))
        keep = keep[np.lexsort(keep.T[::-1])] if len(keep) else np.zeros((0, 5))
        out["keep_" + tag] = keep
        for k, v in kw.items():
            out["kw_%s_%s" % (tag, k)] = np.float64(v)
        print(tag, "kept", len(keep), "of", len(srcs))
    path = os.path.join(OUT, "localmarching.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def main():
    if "--postproc" in sys.argv:
        return main_postproc()
    if "--picks" in sys.argv:
        return main_picks()
    if "--cfg1-events" in sys.argv:
        return main_cfg1_events()
    if "--scaled" in sys.argv:
        return main_scaled()
    if "--embed-sign" in sys.argv:
        return main_embed_sign()
    if "--subgraph-edges-abspos" in sys.argv:
        return main_subgraph_edges_abspos()
    if "--subgraph-abspos" in sys.argv:
        return main_subgraph_abspos()
    if "--subgraph-edges" in sys.argv:
        return main_subgraph_edges()
    if "--edges-abspos" in sys.argv:
        return main_edges_abspos()
    if "--edges" in sys.argv:
        return main_edges()
    if "--subgraph" in sys.argv:
        return main_subgraph()
    if "--abspos" in sys.argv:
        return main_abspos()
    if "--assoc-subgraph" in sys.argv:
        return main_assoc_subgraph()
    if "--assoc-nophase" in sys.argv:
        return main_assoc_nophase()
    if "--assoc-edges" in sys.argv:
        return main_assoc_variant("edges")
    if "--assoc-abspos" in sys.argv:
        return main_assoc_variant("abspos")
    if "--assoc" in sys.argv:
        return main_assoc()
    ref = _import_reference()
    from genie_amd import synthetic as syn

    os.makedirs(OUT, exist_ok=True)

    # (v) 4-output forward_fixed with the association heads
    geom = syn.Geometry(7, 45, L=60e3, n_query=20, seed=81)
    win = syn.make_window(geom, 90, seed=82)
    run_assoc_case(ref, "assoc_7x45", geom, win)

    # (iv) pick -> Slice/Mask embedding (row f-1): two windows, one late in the day (large absolute times)
    geom = syn.Geometry(14, 60, L=90e3, n_query=5, seed=61)
    P = syn.make_picks(geom, 260, seed=62)
    P[:, 0] += 1000.0
    run_embed_case("embed_14x60_a", geom, P, 1000.0)
    P2 = P.copy()
    P2[:, 0] += 80000.3
    P2 = P2[P2[:, 1] != 5]                                                  # one station without any pick
    run_embed_case("embed_14x60_b", geom, P2, 81003.1)
    run_embed_case("embed_sign_14x60_a", geom, P, 1000.0, use_sign_input=True)          # config.yaml:93 (process_utils.py:610-614)
    run_embed_case("embed_sign_14x60_b", geom, P2, 81003.1, use_sign_input=True)

    # (i) tiny, exhaustive intermediates, fp32 + fp64, distinct PReLU slopes, event-structured picks
    geom = syn.Geometry(6, 40, L=60e3, n_query=25, seed=11)
    win = syn.make_window(geom, 60, seed=12)
    run_case(ref, "tiny_6x40", geom, win["Slice"], win["Mask"], perturb_prelu=True, window=win)

    # (ii) config-1 shape: 20 stations / 500 grid nodes / 2k picks, default init under seed 0
    geom = syn.Geometry(20, 500, L=100e3, n_query=300, seed=1)
    win = syn.make_window(geom, 2000, seed=2)
    run_case(ref, "cfg1_20x500", geom, win["Slice"], win["Mask"], window=win, row_stride=7,
             keep=("x_latent", "bip", "sa1", "sa2", "sa3", "y_latent", "xq"), keep64=("bip", "sa3"))

    # (iii) odd sizes; Slice random in [0,1], Mask an INDEPENDENT random {0,1} pattern with all-zero rows
    geom = syn.Geometry(33, 257, L=150e3, n_query=77, seed=21)
    rng = np.random.default_rng(22)
    P = geom.n_prod
    Slice = rng.random((P, 4)).astype(np.float32)
    Mask = (rng.random((P, 4)) < 0.4).astype(np.float32)
    Mask[rng.random(P) < 0.3] = 0.0
    Slice[rng.random(P) < 0.1] = 0.0
    run_case(ref, "odd_33x257", geom, Slice, Mask, perturb_prelu=True, row_stride=11,
             keep=("h0", "h1", "u", "v", "x_latent", "bip", "sa1", "sa2", "sa3", "y_latent", "xq"), keep64=("bip", "sa3"))


if __name__ == "__main__":
    main()
