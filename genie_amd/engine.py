"""Host-side owner of one libgenie_hip context: graphs in CSR form, weight mirror, workspace.

PyTorch is plumbing here (device memory, current stream); every compute call goes through the C ABI
of `include/genie_hip.h` with raw device pointers.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name, shape=None):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s); the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s: expected shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    return t


def knn_device(x_context, x_query, k, exclude_self=False):
    """int32 GPU table [n_query, k]: indices of the k nearest context points of every query, nearest first (genie_knn: exact
    brute-force fp64 search on the device; the reference's `knn(x_context / 1000, x_query / 1000, k)`, module.py:282,
    process_utils.py:718-719). `exclude_self`: query i never lists context i (query set = context set)."""
    lib = _lib.load()
    xc, xq = _f32(x_context, "x_context"), _f32(x_query, "x_query")
    if xc.dim() != 2 or xc.shape[1] != 3 or xq.dim() != 2 or xq.shape[1] != 3:
        raise ValueError("knn_device: positions must be [n, 3]")
    k = int(min(k, xc.shape[0] - (1 if exclude_self else 0)))
    if not 1 <= k <= 16:
        raise ValueError("knn_device: 1 <= k <= 16")
    out = torch.empty((xq.shape[0], k), dtype=torch.int32, device=xq.device)
    with torch.cuda.device(xq.device):
        _lib.check(lib.genie_knn(_ptr(xc), int(xc.shape[0]), _ptr(xq), int(xq.shape[0]), k, 1 if exclude_self else 0, _ptr(out),
                                 _stream()), "genie_knn")
    return out


def knn_graph_device(points, k):
    """Base kNN graph of a point set on the device: the layout of `remove_self_loops(knn(x, x, k + 1).flip(0))`
    (process_utils.py:718-719) as (table int32 [n, k], edge list int64 [2, n * k] with row 0 = neighbour, row 1 = centre)."""
    tab = knn_device(points, points, k, exclude_self=True)
    n, kk = tab.shape
    centre = torch.arange(n, device=tab.device).repeat_interleave(kk)
    return tab, torch.stack((tab.reshape(-1).long(), centre), dim=0)


def subgraph_pairs_device(pos_loc, pos_src, max_deg_offset=5.0, k_nearest_pairs=30, scale_deg=110e3):
    """Product nodes of `use_subgraph: True` on the device (process_utils.py:774-794): every source node is paired with the
    stations within `scale_deg * max_deg_offset` metres (:775-778, float64 as the reference's numpy) and with its
    `k_nearest_pairs` nearest stations (:781, float32 coordinates in km as the reference's `knn`); the union, sorted by
    (source, station) (:784-794). pos_loc [S, 3] / pos_src [G, 3] = `ftrns1(locs)`, `ftrns1(x_grid)` in metres.
    Returns int64 [2, N] (row 0 = station, row 1 = source) on the device of `pos_src`."""
    pl, ps = torch.as_tensor(pos_loc), torch.as_tensor(pos_src)
    dev = ps.device
    pl = pl.to(dev)
    d = (ps.double()[:, None, :] - pl.double()[None, :, :]).pow(2).sum(-1).sqrt()            # [G, S]
    member = d < float(scale_deg) * float(max_deg_offset)
    k = int(min(k_nearest_pairs, pl.shape[0]))
    if k > 0:
        pk, sk = (pl.float() / 1000.0).double(), (ps.float() / 1000.0).double()
        d2 = (sk[:, None, :] - pk[None, :, :]).pow(2).sum(-1)
        near = torch.topk(d2, k, dim=1, largest=False).indices
        member.scatter_(1, near, True)
    src, sta = torch.nonzero(member, as_tuple=True)             # row-major: sorted by (source, station)
    return torch.stack((sta, src), dim=0)


def subgraph_csr_device(pairs, n_grid, sta_csr, src_csr):
    """Product-level CSRs of the irregular product graph (the `subgraph(...)` loops of process_utils.py:824-839) built by
    libgenie_hip on the device: pairs int64 [2, N] sorted by (source, station) on the GPU, (rowptr, col) int32 in-edge CSRs of
    the two base graphs. Returns dict(n_prod, sta_csr, src_csr, seg_rowptr) as `HipPath(subgraph=...)` takes it."""
    lib = _lib.load()
    if not pairs.is_cuda:
        raise ValueError("subgraph_csr_device: pairs must be a GPU tensor")
    dev = pairs.device
    sta = pairs[0].to(torch.int32).contiguous()
    src = pairs[1].to(torch.int32).contiguous()
    n = int(sta.numel())
    if n == 0:
        raise ValueError("subgraph_csr_device: no product nodes")
    key = pairs[1] * (int(pairs[0].max()) + 1) + pairs[0]
    if bool((key[1:] <= key[:-1]).any()):
        raise ValueError("pairs must be sorted by (source, station) without duplicates (process_utils.py:790-794)")
    seg = torch.zeros(n_grid + 1, dtype=torch.int64, device=dev)
    seg[1:] = torch.cumsum(torch.bincount(pairs[1], minlength=n_grid), 0)
    seg = seg.to(torch.int32)
    csr = [t.to(dev).to(torch.int32).contiguous() for t in (sta_csr[0], sta_csr[1], src_csr[0], src_csr[1])]
    c1 = torch.empty(n, dtype=torch.int32, device=dev)
    c2 = torch.empty(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.genie_subgraph_csr_count(_ptr(sta), _ptr(src), n, _ptr(seg), _ptr(csr[0]), _ptr(csr[1]), _ptr(csr[2]),
                                                _ptr(csr[3]), _ptr(c1), _ptr(c2), _stream()), "genie_subgraph_csr_count")
        rp = []
        for c in (c1, c2):
            r = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            r[1:] = torch.cumsum(c, 0)
            if int(r[-1]) >= 2 ** 31:
                raise ValueError("subgraph_csr_device: more than 2^31 product edges")
            rp.append(r.to(torch.int32))
        col1 = torch.empty(max(int(rp[0][-1]), 1), dtype=torch.int32, device=dev)
        col2 = torch.empty(max(int(rp[1][-1]), 1), dtype=torch.int32, device=dev)
        _lib.check(lib.genie_subgraph_csr_fill(_ptr(sta), _ptr(src), n, _ptr(seg), _ptr(csr[0]), _ptr(csr[1]), _ptr(csr[2]),
                                               _ptr(csr[3]), _ptr(rp[0]), _ptr(rp[1]), _ptr(col1), _ptr(col2), _stream()),
                   "genie_subgraph_csr_fill")
    return {"n_prod": n, "sta_csr": (rp[0], col1[:int(rp[0][-1])]), "src_csr": (rp[1], col2[:int(rp[1][-1])]), "seg_rowptr": seg}


def csr_from_edges(edge_index, n_target):
    """[2,E] edge list (row0 = j source, row1 = i target) -> (rowptr int32 [n+1], col int32 [E]);
    in-edges grouped by target in stable edge order."""
    ei = torch.as_tensor(edge_index).long()          # stays on its device: a GPU edge list is sorted / counted on the GPU
    dev = ei.device
    if ei.numel() == 0:
        return torch.zeros(n_target + 1, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
    j, i = ei[0], ei[1]
    lo_hi = torch.stack((i.min(), i.max())).tolist()           # (one read-back for both bounds)
    if lo_hi[1] >= n_target or lo_hi[0] < 0:
        raise ValueError("edge target out of range")
    order = torch.sort(i, stable=True)[1]
    deg = torch.bincount(i, minlength=n_target)
    rowptr = torch.zeros(n_target + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0)
    return rowptr.to(torch.int32), j[order].to(torch.int32).contiguous()


def csr_from_table(nbr):
    """Uniform-degree neighbour table [n, k] -> CSR."""
    nbr = torch.as_tensor(nbr).to(torch.int32)
    n, k = nbr.shape
    rowptr = (torch.arange(n + 1, dtype=torch.int64, device=nbr.device) * k).to(torch.int32)
    return rowptr, nbr.reshape(-1).contiguous()


def _quantise_isotropic(points, bits):
    """All axes on the SAME scale: stations (elevations of +-1 km over a 300-km network) and source grids (40 km of depth) are
    nearly flat, and a per-axis scale lets the noise of the short axis decide the order (median index distance of a station to
    its neighbours 65 instead of 5 of 200; distinct neighbour rows per 16-station tile 61 instead of 39)."""
    x = np.asarray(points, dtype=np.float64)
    if x.ndim != 2 or x.shape[1] != 3:
        raise ValueError("positions must be [n, 3]")
    lo, hi = x.min(0), x.max(0)
    top = float((1 << bits) - 1)
    scale = top / max(float((hi - lo).max()), 1e-9)
    return np.clip(((x - lo) * scale).astype(np.int64), 0, (1 << bits) - 1)


def morton_order(points):
    """Morton (Z-curve, 10 bits on the longest axis) order of a point set: int32 permutation [n]."""
    q = _quantise_isotropic(points, 10)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return np.argsort(code, kind="stable").astype(np.int32)


def hilbert_order(points, bits=10):
    """Hilbert-curve order of a point set (Skilling's axes-to-transpose transform, vectorised over the points): int32 permutation
    [n]. No long jumps like the Z-curve has at its quadrant boundaries: 9 % fewer distinct neighbour rows per block of 64 source
    nodes than `morton_order` on the synthetic 10 000-node grid, but no faster end to end (see `sfc_order`)."""
    X = _quantise_isotropic(points, bits).T.copy()          # [3, n]
    n = X.shape[0]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(n):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, n):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[n - 1] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(n):
        X[i] ^= t
    code = np.zeros(X.shape[1], dtype=np.int64)
    for b in range(bits - 1, -1, -1):
        for i in range(n):
            code = (code << 1) | ((X[i] >> b) & 1)
    return np.argsort(code, kind="stable").astype(np.int32)


def sfc_order(points):
    """The processing order used for source nodes and stations: the Z-curve (measured at config 2: 0.781 ms/window against
    0.785 with the Hilbert curve `hilbert_order`; config 4: 42.1 against 41.7 ms). A GPU tensor is ordered on the GPU
    (`morton_order_device`: the same permutation, no host round trip of the positions) and returned as an int32 GPU tensor."""
    if torch.is_tensor(points) and points.is_cuda:
        return morton_order_device(points)
    return morton_order(points.detach().cpu().numpy() if torch.is_tensor(points) else points)


def morton_order_device(points):
    """`morton_order` of positions resident on the GPU, computed there: the float64 quantisation of `_quantise_isotropic` (IEEE
    arithmetic, truncation toward zero: the same integers), the same bit interleave, a stable sort of the codes -- the same
    permutation as the numpy form (tests/test_hip_parity.py), as an int32 GPU tensor. A context rebuild per training sample
    (train_GENIE_model.py:1722-1786) then costs no copy of the grid to the host and back."""
    x = points.detach().to(torch.float64)
    if x.dim() != 2 or x.shape[1] != 3:
        raise ValueError("positions must be [n, 3]")
    lo, hi = x.min(0)[0], x.max(0)[0]
    top = float((1 << 10) - 1)
    scale = top / torch.clamp((hi - lo).max(), min=1e-9)
    q = ((x - lo) * scale).long().clamp_(0, (1 << 10) - 1)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.sort(code, stable=True)[1].to(torch.int32)


def size_class(n):
    """Allocation sizes of the big per-context / per-step buffers (workspace, kept pre-activations, scratch) rounded up to 1/8-octave
    steps: the reference's training loop changes the station subset per sample, so these sizes change by a fraction of a percent from
    step to step, and PyTorch's caching allocator only reuses a cached block for a request it fits -- exact sizes made every other
    rebuild pay a fresh 1-2 GB hipMalloc (60-80 ms). With size classes a handful of blocks serve every sample."""
    n = int(n)
    if n < 4096:
        return max(n, 1)
    sh = n.bit_length() - 4
    step = 1 << sh
    return (n + step - 1) // step * step


def empty_f32(n, device):
    """Flat float32 buffer of `n` elements cut from a size-class allocation (`size_class`)."""
    return torch.empty(size_class(n), dtype=torch.float32, device=device)[:n]


ASSOC_PREFIXES = ("BipartiteGraphReadOutOperator.", "DataAggregationAssociationPhase.", "LocalSliceLgCollapseP.",
                  "LocalSliceLgCollapseS.", "Arrivals.")


# default arithmetic of the P-sized stages of new contexts (HipPath(stage_precision=...)); the A/B tests patch it
STAGE_PRECISION = "auto"
_PRECISION_MODES = {"auto": 0, "f16x2": 1, "f32": 2}


def time_partition(dt_partition):
    """The uniform time partition behind the time-pointer tables (`dt_partition`, module.py:635) as genie_lslc_fwd / _bwd take it:
    (t0, dt, device tensor or None, length). A GPU tensor stays on the device -- the kernels read t0 = p[0], dt = p[1] - p[0]
    themselves, nothing is read back --; host data (numpy / CPU tensor / list) gives the two numbers. A tuple is passed through."""
    if isinstance(dt_partition, tuple):
        return dt_partition
    n = int(len(dt_partition))
    if torch.is_tensor(dt_partition) and dt_partition.is_cuda:
        if n < 2:
            raise ValueError("dt_partition needs at least two entries")
        return (0.0, 0.0, dt_partition.detach().float().contiguous(), n)
    if torch.is_tensor(dt_partition):
        dt_partition = dt_partition.detach().float().numpy()           # fp32 difference, as the reference's tensor arithmetic
        return (float(dt_partition[0]), float(dt_partition[1] - dt_partition[0]), None, n)
    return (float(dt_partition[0]), float(dt_partition[1] - dt_partition[0]), None, n)


class HipPath(object):
    """One libgenie_hip context bound to the current CUDA(HIP) device.

    sta_csr / src_csr: (rowptr, col) int32 CPU or GPU tensors for the base station graph and the base
    source graph (in-edges grouped by target). `n_grid_ext > n_grid` marks halo source nodes (sharded use).
    """

    def __init__(self, n_sta, n_grid, sta_csr, src_csr, n_grid_ext=None, grid_order=None, scale_rel=30000.0,
                 device=None, subgraph=None, sta_order=None, stage_precision=None):
        """`stage_precision`: "auto" (default: two-piece fp16 operands on the 16-bit matrix pipe while the library's fp16 range guard
        holds for the committed weights, fp32 MFMA otherwise), "f16x2" or "f32" (A/B runs); None = `engine.STAGE_PRECISION`.
        `subgraph` = dict(n_prod, sta_csr, src_csr, seg_rowptr): an irregular product graph (`use_subgraph`) given as
        product-level CSRs + the row range of every source node (genie_ctx_create_subgraph); `sta_csr` is then ignored.
        `grid_order` / `sta_order`: processing orders of the source nodes / stations (e.g. `sfc_order(positions)`);
        internal only, every input and output keeps the caller's order."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GenieHipError("no GPU visible: the HIP path cannot run (there is no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.n_sta, self.n_grid = int(n_sta), int(n_grid)
        self.n_grid_ext = int(n_grid_ext) if n_grid_ext is not None else self.n_grid
        self.scale_rel = float(scale_rel)
        self._n_prod = None
        dev = self.device
        order = None
        if grid_order is not None:
            order = (grid_order if torch.is_tensor(grid_order) else torch.as_tensor(np.asarray(grid_order))).to(dev, torch.int32).contiguous()
            if order.numel() != self.n_grid:
                raise ValueError("grid_order must have n_grid entries")
        self.ctx = ctypes.c_void_p(0)
        if subgraph is not None:
            if self.n_grid_ext != self.n_grid:
                raise ValueError("an irregular product graph cannot be sharded")
            self._n_prod = int(subgraph["n_prod"])
            self._keep = [t.to(dev, torch.int32).contiguous() for t in (
                subgraph["sta_csr"][0], subgraph["sta_csr"][1], subgraph["src_csr"][0], subgraph["src_csr"][1],
                subgraph["seg_rowptr"], src_csr[0], src_csr[1])]
            k = self._keep
            if k[0].numel() != self._n_prod + 1 or k[2].numel() != self._n_prod + 1 or k[4].numel() != self.n_grid + 1:
                raise ValueError("subgraph CSR arrays do not match n_prod / n_grid")
            with torch.cuda.device(dev):
                torch.cuda.synchronize()
                rc = self.lib.genie_ctx_create_subgraph(ctypes.byref(self.ctx), self.n_sta, self.n_grid, self._n_prod,
                                                        _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(k[3]), _ptr(k[4]), _ptr(k[5]),
                                                        _ptr(k[6]), _ptr(order), ctypes.c_float(self.scale_rel))
            _lib.check(rc, "genie_ctx_create_subgraph")
        else:
            self._keep = [t.to(dev, torch.int32).contiguous() for t in (sta_csr[0], sta_csr[1], src_csr[0], src_csr[1])]
            with torch.cuda.device(dev):
                torch.cuda.synchronize()
                rc = self.lib.genie_ctx_create(ctypes.byref(self.ctx), self.n_sta, self.n_grid, self.n_grid_ext,
                                               _ptr(self._keep[0]), _ptr(self._keep[1]), _ptr(self._keep[2]),
                                               _ptr(self._keep[3]), _ptr(order), ctypes.c_float(self.scale_rel))
            _lib.check(rc, "genie_ctx_create")
            # without a station order the identity: stage 2's production kernel reads its rows in processing order
            if torch.is_tensor(sta_order):
                sta_order = sta_order.cpu().numpy()
            so = np.ascontiguousarray(np.asarray(sta_order if sta_order is not None else np.arange(self.n_sta)), dtype=np.int32)
            if so.shape != (self.n_sta,):
                raise ValueError("sta_order must have n_sta entries")
            _lib.check(self.lib.genie_set_station_order(self.ctx, ctypes.c_void_p(so.ctypes.data)), "genie_set_station_order")
        self.set_stage_precision(stage_precision if stage_precision is not None else STAGE_PRECISION)
        self.ws = torch.empty(size_class(int(self.lib.genie_workspace_bytes(self.ctx)) + 256), dtype=torch.uint8, device=dev)
        off = (-self.ws.data_ptr()) % 256
        self._ws_ptr = ctypes.c_void_p(self.ws.data_ptr() + off)
        _lib.check(self.lib.genie_set_slot(self.ctx, self.PLAIN_SLOT), "genie_set_slot")
        # weight mirror (the registry is a property of the library, read once per process: 3 x 161 ctypes calls otherwise)
        if HipPath._registry is None:
            n = self.lib.genie_weights_count()
            HipPath._registry = ([self.lib.genie_weights_name(i).decode() for i in range(n)],
                                 [int(self.lib.genie_weights_numel(i)) for i in range(n)],
                                 [int(self.lib.genie_weights_offset(i)) for i in range(n)], int(self.lib.genie_weights_blob_floats()))
        self.w_names, self.w_numel, self.w_off, n_blob = HipPath._registry
        self._blob = torch.zeros(n_blob, dtype=torch.float32, device=dev)
        self._w_key = None

    _registry = None

    def set_subgraph_stations(self, sta_of_prod):
        """Station index of every product node of an irregular product graph (genie_set_subgraph_stations): enables `embed_window`."""
        if self._n_prod is None:
            raise ValueError("set_subgraph_stations: not an irregular product graph")
        t = torch.as_tensor(sta_of_prod).to(self.device, torch.int32).contiguous()
        if t.numel() != self._n_prod:
            raise ValueError("sta_of_prod must have n_prod entries")
        _lib.check(self.lib.genie_set_subgraph_stations(self.ctx, _ptr(t), _stream()), "genie_set_subgraph_stations")
        torch.cuda.current_stream().synchronize()      # (the library keeps its own copy)

    def set_stage2_workmap(self, blocks_of_four):
        """Work map of the row-layout stage 2 (genie_set_stage2_workmap): True / False, None = the default for the station count."""
        v = -1 if blocks_of_four is None else (1 if blocks_of_four else 0)
        _lib.check(self.lib.genie_set_stage2_workmap(self.ctx, v), "genie_set_stage2_workmap")

    def set_sign_input(self, use_sign_input):
        """`use_sign_input` of config.yaml:93 for the device embedding (genie_set_sign_input): features signed by the series' negative slope."""
        _lib.check(self.lib.genie_set_sign_input(self.ctx, 1 if use_sign_input else 0), "genie_set_sign_input")

    def set_phase_types(self, use_phase_types):
        """`use_phase_types` of config.yaml:91 for the device embedding (genie_set_phase_types)."""
        _lib.check(self.lib.genie_set_phase_types(self.ctx, 1 if use_phase_types else 0), "genie_set_phase_types")

    def set_tail_precision(self, fp64_chains):
        """Arithmetic of the G-sized tail of inference calls: fp64 MFMA chains (default) or fp32 ones (genie_set_tail_precision)."""
        _lib.check(self.lib.genie_set_tail_precision(self.ctx, 1 if fp64_chains else 0), "genie_set_tail_precision")

    def set_stage_precision(self, mode):
        """"auto" | "f16x2" | "f32": arithmetic of the P-sized stages (genie_set_stage_precision)."""
        if mode not in _PRECISION_MODES:
            raise ValueError("stage_precision must be one of %s" % (sorted(_PRECISION_MODES),))
        _lib.check(self.lib.genie_set_stage_precision(self.ctx, _PRECISION_MODES[mode]), "genie_set_stage_precision")

    def stage_precision(self):
        """What runs for the weights set so far: dict(mode, f16x2_active, act_bound, weight_bound) -- the last two are the numbers
        the library's fp16 range guard compares with 60000 (genie_stage_precision). Synchronises when weights were pending."""
        mode, act = ctypes.c_int(0), ctypes.c_int(0)
        ab, wb = ctypes.c_float(0), ctypes.c_float(0)
        _lib.check(self.lib.genie_stage_precision(self.ctx, ctypes.byref(mode), ctypes.byref(act), ctypes.byref(ab), ctypes.byref(wb),
                                                  _stream()), "genie_stage_precision")
        names = {v: k for k, v in _PRECISION_MODES.items()}
        return {"mode": names[mode.value], "f16x2_active": bool(act.value), "act_bound": float(ab.value), "weight_bound": float(wb.value)}

    def __del__(self):
        try:
            if getattr(self, "ctx", None) and self.ctx.value:
                fl, mx = ctypes.c_uint(0), ctypes.c_float(0.0)
                self.lib.genie_index_flags(self.ctx, ctypes.byref(fl), 0)
                self.lib.genie_input_range(self.ctx, ctypes.byref(mx), None, 0)
                if fl.value or mx.value != 0.0:
                    import warnings
                    warnings.warn("genie_amd: a HIP context is destroyed with unread device-side verdicts (index flags %#x, input magnitude "
                                  "%.6g): results of its last calls were computed from clamped indices / outside the verified fp16 range"
                                  % (fl.value, mx.value), RuntimeWarning)
                self.lib.genie_ctx_destroy(self.ctx)
                self.ctx = ctypes.c_void_p(0)
        except Exception:
            pass

    @property
    def n_prod(self):
        return self._n_prod if self._n_prod is not None else self.n_sta * self.n_grid

    @property
    def n_prod_ext(self):
        return self._n_prod if self._n_prod is not None else self.n_sta * self.n_grid_ext

    # ---- weights -------------------------------------------------------------------------------
    def set_weights(self, named_tensors):
        """Upload the path's parameters (dict name -> tensor, the reference's state_dict names): one fused multi-tensor copy into the
        flat mirror (a training step re-uploads all ~160 tensors after every optimizer step) + genie_weights_set_blob."""
        self.assoc_ready = True
        if getattr(self, "_w_views", None) is None:
            self._w_views = [self._blob[off:off + n] for n, off in zip(self.w_numel, self.w_off)]
        dst, src, zero = [], [], []
        with torch.no_grad():
            for name, n, view in zip(self.w_names, self.w_numel, self._w_views):
                assoc = name.startswith(ASSOC_PREFIXES)
                if name not in named_tensors:
                    if name.endswith(".weight_pos") or name.endswith(".weight_abs"):   # optional columns: absent = plain DataAggregation
                        zero.append(view)
                        continue
                    if assoc:                       # a context used for forward_fixed_source only needs no association heads
                        zero.append(view)
                        self.assoc_ready = False
                        continue
                    raise KeyError("missing parameter %s" % name)
                t = named_tensors[name]
                if t.numel() != n:
                    if assoc:                       # unexpected head shapes
                        zero.append(view)
                        self.assoc_ready = False
                        continue
                    raise ValueError("parameter %s: expected %d elements, got %d" % (name, n, t.numel()))
                t = t.detach().reshape(-1)
                if t.device != view.device or t.dtype != view.dtype:
                    t = t.to(device=view.device, dtype=view.dtype)
                dst.append(view)
                src.append(t)
            if zero:
                torch._foreach_zero_(zero)
            if dst:
                torch._foreach_copy_(dst, src)
        _lib.check(self.lib.genie_weights_set_blob(self.ctx, _ptr(self._blob), self._blob.numel(), _stream()),
                   "genie_weights_set_blob")

    def sync_weights(self, params, view=None):
        """Re-upload only when any of the parameter tensors changed (data_ptr / in-place version). `view` maps the
        state_dict-named tensors to the registry's names / layouts when they differ (DataAggregationEdges)."""
        key = tuple((p.data_ptr(), p._version) for p in params.values())
        if key != self._w_key:
            # tails of earlier windows may still read the weight mirror / packed images on their side streams
            self.wait_tails()
            self.set_weights(view(params) if view is not None else params)
            self._w_key = key

    def input_limit(self):
        """Largest |Slice| / |Mask| entry the f16x2 kernels are verified for with the weights committed so far (genie_input_range)."""
        lim = ctypes.c_float(0.0)
        _lib.check(self.lib.genie_input_range(self.ctx, None, ctypes.byref(lim), 0), "genie_input_range")
        return float(lim.value)

    def check_input_range(self, synchronize=False):
        """Fail loudly when a split pass of an earlier call met inputs beyond what the fp16 range guard verified (genie_input_range: a
        word of host-mapped memory, read without synchronising -- it reflects the calls that have completed; `synchronize=True` waits for
        the device's current stream first, so that every call issued so far is covered). The context is switched to the fp32 kernels,
        which take any input the reference's fp32 arithmetic takes, before the error is raised: the results of the calls issued since the
        last check are invalid and must be recomputed. Called at the top of every entry point that runs stage 1, by `wait_tails`, and
        wherever the host has just waited for the device anyway (module.forward's deferred verdicts, the per-day loops' final copies)."""
        self.check_index_flags(synchronize)
        mx, lim = ctypes.c_float(0.0), ctypes.c_float(0.0)
        _lib.check(self.lib.genie_input_range(self.ctx, ctypes.byref(mx), ctypes.byref(lim), 0), "genie_input_range")
        if mx.value != 0.0:
            _lib.check(self.lib.genie_input_range(self.ctx, ctypes.byref(mx), None, 1), "genie_input_range")   # atomic fetch-and-clear
            self.set_stage_precision("f32")
            raise _lib.GenieHipError(
                "an earlier call handed the f16x2 stage kernels Slice / Mask entries of magnitude %.6g; the committed weights keep their "
                "hidden states inside the fp16 range only up to |input| <= %.6g (the reference's embedding produces [-1, 1]). The results "
                "of the calls issued since the last check are invalid; this context now runs the fp32 kernels (stage_precision 'f32'): "
                "repeat those calls." % (mx.value, lim.value))

    def check_index_flags(self, synchronize=False):
        """Raise IndexError when a pick of an earlier `lslc_fwd` / arrivals call indexed outside its table (genie_index_flags: host-mapped
        memory, no synchronisation -- it reflects the calls that have completed; the kernel clamps such an index). The reference's own
        indexing (module.py:635-640) fails the same deferred way on a GPU: a device-side assertion reported at the next synchronisation.
        Called by every entry point that checks the input range, by `lslc_fwd` itself and by `wait_tails`; `synchronize=True` waits
        for the device's current stream first, so that the calls issued so far are covered."""
        if synchronize:
            torch.cuda.current_stream(self.device).synchronize()
        fl = ctypes.c_uint(0)
        _lib.check(self.lib.genie_index_flags(self.ctx, ctypes.byref(fl), 0), "genie_index_flags")
        if fl.value:
            _lib.check(self.lib.genie_index_flags(self.ctx, ctypes.byref(fl), 1), "genie_index_flags")          # atomic fetch-and-clear
            what = []
            if fl.value & 1:
                what.append("lslc_fwd: a pick lies outside the time-pointer table (tpick outside dt_partition, or ipick outside the stations "
                            "of A_edges)")
            if fl.value & 2:
                what.append("arrivals: a station index `ipick` outside [0, n_sta)")
            raise IndexError("an earlier call on this context met " + "; ".join(what) + ". Its results were computed from clamped indices "
                             "and are invalid")

    def retire(self, synchronize=True):
        """Last look at the device-side verdicts of a context that is being replaced or dropped (`module.forward` builds a new context
        per training sample; nothing reads the flag word once the context is gone): raises what `check_input_range` would.
        `synchronize=False` when the caller has just waited for the device (the constructor of the replacing context does).
        `__del__` cannot raise; it warns instead."""
        if getattr(self, "ctx", None) and self.ctx.value:
            if synchronize:
                with torch.cuda.device(self.device):
                    torch.cuda.synchronize()
            self.check_input_range()

    def discard_flags(self):
        """Forget the device-side verdicts (calls whose results the caller throws away anyway, e.g. a context built on graphs that the
        deferred structure checks then rejected)."""
        _lib.check(self.lib.genie_index_flags(self.ctx, None, 1), "genie_index_flags")
        _lib.check(self.lib.genie_input_range(self.ctx, None, None, 1), "genie_input_range")

    # ---- stages --------------------------------------------------------------------------------
    def da_stage1(self, Slice, Mask, debug=False):
        """Stage 1 (genie_da_stage1). Returns the validated (Slice, Mask) [, h0, h1 when debug]."""
        self.check_input_range()
        Slice = _f32(Slice, "Slice", (self.n_prod_ext, 4))
        Mask = _f32(Mask, "Mask", (self.n_prod_ext, 4))
        if not debug:
            _lib.check(self.lib.genie_da_stage1(self.ctx, _ptr(Slice), _ptr(Mask), self._ws_ptr, _stream()), "genie_da_stage1")
            return Slice, Mask
        h0 = torch.empty((self.n_prod, 30), dtype=torch.float32, device=self.device)
        h1 = torch.empty((self.n_prod, 60), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_da_stage1_debug(self.ctx, _ptr(Slice), _ptr(Mask), _ptr(h0), _ptr(h1), self._ws_ptr,
                                                  _stream()), "genie_da_stage1_debug")
        return Slice, Mask, h0, h1

    def da_stage1_range(self, Slice, Mask, gi_begin, gi_end, first):
        """Stage 1 of the owned source nodes at positions [gi_begin, gi_end) of the processing order (genie_da_stage1_range);
        `first`: the first range call of this window (runs the split pass over all rows)."""
        _lib.check(self.lib.genie_da_stage1_range(self.ctx, _ptr(Slice), _ptr(Mask), int(gi_begin), int(gi_end), 1 if first else 0,
                                                  self._ws_ptr, _stream()), "genie_da_stage1_range")

    def da_stage2_partials_range(self, Mask, edge_attr, gi_begin, gi_end):
        """Stage-2 partials of the owned source nodes at positions [gi_begin, gi_end) (genie_da_stage2_partials_range)."""
        _lib.check(self.lib.genie_da_stage2_partials_range(self.ctx, _ptr(Mask), _ptr(edge_attr), None, int(gi_begin), int(gi_end),
                                                           self._ws_ptr, _stream()), "genie_da_stage2_partials_range")

    def bipartite_readout(self):
        bip = torch.empty((self.n_grid, 15), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_bipartite_readout(self.ctx, _ptr(bip), self._ws_ptr, _stream()), "genie_bipartite_readout")
        return bip

    def da_stage2_bipartite(self, Mask, edge_attr, want_x_latent=False):
        edge_attr = _f32(edge_attr, "edge_attr", (self.n_prod, 3))
        self._refresh_static_edge_attr(edge_attr)
        x_latent = torch.empty((self.n_prod, 30), dtype=torch.float32, device=self.device) if want_x_latent else None
        bip = torch.empty((self.n_grid, 15), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_da_stage2_bipartite(self.ctx, _ptr(Mask), _ptr(edge_attr), _ptr(x_latent), _ptr(bip),
                                                      self._ws_ptr, _stream()), "genie_da_stage2_bipartite")
        return x_latent, bip

    def spatial_agg(self, layer, x_in, pos):
        c_in = 15 if layer == 1 else 30
        x_in = _f32(x_in, "x_in", (self.n_grid, c_in))
        pos = _f32(pos, "pos", (self.n_grid, 3))
        out = torch.empty((self.n_grid, 30), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_spatial_agg_fwd(self.ctx, layer, _ptr(x_in), _ptr(pos), _ptr(out), self._ws_ptr,
                                                  _stream()), "genie_spatial_agg_fwd")
        return out

    def spatial_agg3(self, x_in15, pos):
        """SpatialAggregation1 -> 2 -> 3 chained (genie_spatial_agg3_fwd, module.py:1012-1014): [G, 15] -> [G, 30]."""
        x_in15 = _f32(x_in15, "x_in15", (self.n_grid, 15))
        pos = _f32(pos, "pos", (self.n_grid, 3))
        out = torch.empty((self.n_grid, 30), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_spatial_agg3_fwd(self.ctx, _ptr(x_in15), _ptr(pos), _ptr(out), self._ws_ptr, _stream()),
                   "genie_spatial_agg3_fwd")
        return out

    def path_fwd(self, Slice, Mask, edge_attr, pos, want_x_latent=False, want_bip=False):
        """Fused DataAggregation -> Bipartite_ReadIn -> SpatialAggregation1..3 (module.py:1010-1014)."""
        self.check_input_range()
        P = self.n_prod
        Slice = _f32(Slice, "Slice", (P, 4))
        Mask = _f32(Mask, "Mask", (P, 4))
        edge_attr = _f32(edge_attr, "edge_attr", (P, 3))
        self._refresh_static_edge_attr(edge_attr)
        pos = _f32(pos, "pos", (self.n_grid, 3))
        out = torch.empty((self.n_grid, 30), dtype=torch.float32, device=self.device)
        x_latent = torch.empty((P, 30), dtype=torch.float32, device=self.device) if want_x_latent else None
        bip = torch.empty((self.n_grid, 15), dtype=torch.float32, device=self.device) if want_bip else None
        _lib.check(self.lib.genie_path_fwd(self.ctx, _ptr(Slice), _ptr(Mask), _ptr(edge_attr), _ptr(pos), _ptr(out),
                                           _ptr(x_latent), _ptr(bip), self._ws_ptr, _stream()), "genie_path_fwd")
        return out, x_latent, bip

    def _new_side_stream(self, prio=0):
        return torch.cuda.Stream(device=self.device, priority=prio)

    def _next_window_slot(self):
        """Workspace slot the NEXT window (forward_pipelined / window_push) will run under; 0 for the single-stream calls."""
        bt = getattr(self, "_bt", None)
        if bt is not None or self.window_batch > 1:
            return (bt["group"] * self.window_batch + bt["n"]) if bt is not None else 0
        if getattr(self, "_ev_tail", None) is not None:
            return self._win % len(self._ev_tail)
        return 0

    def _crosses_to(self, stream, *tensors):
        """Tensors allocated on the current stream and consumed on `stream`: tell the caching allocator (record_stream), so
        that a temporary freed by the caller is not handed out again before the side stream has read it."""
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(stream)

    # ---- two-stream window pipeline ----------------------------------------------------------------
    def forward_pipelined(self, Slice, Mask, edge_attr, pos, x_query, knn_idx, t_query):
        """One forward_fixed_source window as a stream pipeline over independent windows: the P-sized kernels (stage 1,
        stage 2) on the current stream, the G-sized tail (Bipartite read-out, SpatialAggregation x3, read-out heads: short
        latency-bound kernels) on a side stream, where it overlaps the NEXT windows' P-sized kernels. The persistent P-sized
        kernels fill every CU, so a tail kernel only advances when their workgroups retire and one tail takes about as long as
        a whole window; consecutive windows therefore alternate between two side streams, and every
        buffer that crosses the stream boundary exists once per window in flight (`genie_set_slot`). Results are
        bit-identical to `path_fwd` + read-outs. Returns (y, x, done_event); y / x are produced on `self.side_stream` (the
        side stream of THIS window) — consume them there or wait for `done_event`; `wait_tails()` joins all of them. (Measured alternative, rejected: also moving stage 2 to its own stream so that it overlaps the
        next stage 1 — the two P-sized kernels slow each other down more than the overlap gains)"""
        self.check_input_range()
        P = self.n_prod
        Slice = _f32(Slice, "Slice", (P, 4))
        Mask = _f32(Mask, "Mask", (P, 4))
        edge_attr = _f32(edge_attr, "edge_attr", (P, 3))
        self._refresh_static_edge_attr(edge_attr)
        pos = _f32(pos, "pos", (self.n_grid, 3))
        if getattr(self, "_ev_tail", None) is None:
            n_tail = 2
            self._pl_streams = [self._new_side_stream() for _ in range(n_tail)]
            self.side_streams = list(getattr(self, "side_streams", None) or []) + self._pl_streams
            self._win = 0
            self._ev_tail = [None] * (n_tail + 1)
        main = torch.cuda.current_stream(self.device)
        bt = getattr(self, "_bt", None)
        if bt is not None:          # batched tails (window_push / windows_flush) use the same workspace slots: join them first
            for k, ev in enumerate(bt["ev"]):
                if ev is not None:
                    main.wait_event(ev)
                    bt["ev"][k] = None
        slot = self._win % len(self._ev_tail)
        side = self.side_stream = self._pl_streams[self._win % len(self._pl_streams)]   # where this window's y / x are produced
        self._win += 1
        _lib.check(self.lib.genie_set_slot(self.ctx, slot), "genie_set_slot")
        if self._ev_tail[slot] is not None:
            main.wait_event(self._ev_tail[slot])          # the tail that last used this slot's scratch has finished
        st = ctypes.c_void_p(main.cuda_stream)
        _lib.check(self.lib.genie_da_stage1(self.ctx, _ptr(Slice), _ptr(Mask), self._ws_ptr, st), "genie_da_stage1")
        _lib.check(self.lib.genie_da_stage2_partials(self.ctx, _ptr(Mask), _ptr(edge_attr), None, self._ws_ptr, st),
                   "genie_da_stage2_partials")
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        x_query, t_query = _f32(x_query, "x_query"), _f32(t_query, "t_query")
        self._crosses_to(side, pos, x_query, knn_idx, t_query)
        # per-window tails next to the persistent P-sized kernels: fewer tail workgroups (each has a CU to itself while it lives)
        # cost the main stream less, as long as the chain of eight kernels still ends within two windows. 3/8 and 1/4 of the CUs
        # measured best at config 2 (0.852 -> 0.824 ms per window on the same box; 128 / 64: 0.870, 64 / 32: 0.931, 96 / 96: 0.890;
        # worse without the pipeline and with batched tails, hence set here only).
        cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        _lib.check(self.lib.genie_set_tail_grid(self.ctx, (3 * cus) // 8, cus // 4), "genie_set_tail_grid")
        with torch.cuda.stream(side):
            ss = ctypes.c_void_p(side.cuda_stream)
            bip = torch.empty((self.n_grid, 15), dtype=torch.float32, device=self.device)
            x_spatial = torch.empty((self.n_grid, 30), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.genie_bipartite_readout(self.ctx, _ptr(bip), self._ws_ptr, ss), "genie_bipartite_readout")
            _lib.check(self.lib.genie_spatial_agg3_fwd(self.ctx, _ptr(bip), _ptr(pos), _ptr(x_spatial), self._ws_ptr, ss),
                       "genie_spatial_agg3_fwd")
            y = self.readout_grid(x_spatial, t_query)
            x = self.readout_query(x_spatial, pos, x_query, knn_idx, t_query)
            done = torch.cuda.Event()
            done.record(side)
        _lib.check(self.lib.genie_set_tail_grid(self.ctx, 0, 0), "genie_set_tail_grid")
        self._crosses_to(main, y, x)       # produced on the side stream, usually consumed by the caller on the main one
        self._ev_tail[slot] = done
        _lib.check(self.lib.genie_set_slot(self.ctx, self.PLAIN_SLOT), "genie_set_slot")
        return y, x, done

    # ---- window pipeline with batched tails: stage 1 / 2 per window, one G-sized tail per `window_batch` windows -------
    MAX_BATCH = 16
    PLAIN_SLOT = 32       # workspace slot of the single-stream calls (path_fwd, read-outs): never one of a pending window's
    window_batch = 1      # windows per tail (set_window_batch); 1 = every window gets its own tail

    def set_window_batch(self, n):
        """Windows per batched tail, 1..16. Measured at config 2: the batched tail needs 104 us of GPU time per window against 186
        us for per-window tails, but its long persistent read-out workgroups hold CUs that the next stage-1 workgroups wait for:
        for resident windows one tail per window is ~1 % faster end to end, with the device embedding in the loop batches of 8
        are 3.6 % faster. Default 1 (results arrive per window); `apply_windows_device` and the bench use 16."""
        n = int(n)
        if not 1 <= n <= self.MAX_BATCH:
            raise ValueError("window batch must be in [1, %d]" % self.MAX_BATCH)
        if getattr(self, "_bt", None) is not None and self._bt["n"]:
            raise RuntimeError("set_window_batch: windows pending, call windows_flush first")
        if getattr(self, "_bt", None) is not None:
            self.wait_tails()
            self._bt = None
        self.window_batch = n

    def window_push(self, Slice, Mask, edge_attr):
        """Stage 1 + stage 2 of one window on the current stream, into the next workspace slot of the open batch; returns the
        number of windows now pending (call `windows_flush` when it reaches `window_batch`, or earlier). The plain
        single-stream calls (`path_fwd`, read-outs) have a slot of their own for the G-sized buffers (PLAIN_SLOT) and may be mixed
        with pending windows ON THE SAME STREAM only: the P-sized c / wu / wv rows exist GENIE_NBIG = 4 times (slot % 4), so a plain
        call shares its copy with the window slots 0, 4, 8, ...; stream order is what keeps them apart."""
        self.check_input_range()
        P = self.n_prod
        Slice = _f32(Slice, "Slice", (P, 4))
        Mask = _f32(Mask, "Mask", (P, 4))
        edge_attr = _f32(edge_attr, "edge_attr", (P, 3))
        self._refresh_static_edge_attr(edge_attr)
        if getattr(self, "_bt", None) is None:
            groups = 3 if self.window_batch == 1 else 2        # batches in flight (at most 32 workspace slots)
            self._bt = {"group": 0, "n": 0, "ev": [None] * groups, "turn": 0,
                        "streams": [self._new_side_stream() for _ in range(2)]}
            self.side_streams = list(getattr(self, "side_streams", None) or []) + self._bt["streams"]
        bt = self._bt
        if bt["n"] >= self.window_batch:
            raise RuntimeError("window_push: %d windows pending, call windows_flush first" % bt["n"])
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_ev_tail", None) is not None:      # tails of forward_pipelined windows use the same workspace slots: join them first
            for k, ev in enumerate(self._ev_tail):
                if ev is not None:
                    main.wait_event(ev)
                    self._ev_tail[k] = None
        if bt["n"] == 0 and bt["ev"][bt["group"]] is not None:
            main.wait_event(bt["ev"][bt["group"]])          # the tail that last read these slots has finished
        _lib.check(self.lib.genie_set_slot(self.ctx, bt["group"] * self.window_batch + bt["n"]), "genie_set_slot")
        st = ctypes.c_void_p(main.cuda_stream)
        _lib.check(self.lib.genie_da_stage1(self.ctx, _ptr(Slice), _ptr(Mask), self._ws_ptr, st), "genie_da_stage1")
        _lib.check(self.lib.genie_da_stage2_partials(self.ctx, _ptr(Mask), _ptr(edge_attr), None, self._ws_ptr, st),
                   "genie_da_stage2_partials")
        _lib.check(self.lib.genie_set_slot(self.ctx, self.PLAIN_SLOT), "genie_set_slot")
        bt["n"] += 1
        return bt["n"]

    def windows_flush(self, pos, x_query, knn_idx, t_query):
        """The G-sized tail of all pending windows (genie_tail_batched) on a side stream, where it overlaps the P-sized kernels
        of the following windows. Returns (y [n, G, T, 1], x [n, Q, T, 1], done_event), produced on `self.side_stream`: consume
        them there or after the event. Consecutive tails alternate between two side streams."""
        bt = getattr(self, "_bt", None)
        if bt is None or bt["n"] == 0:
            raise RuntimeError("windows_flush: no pending window")
        n = bt["n"]
        pos = _f32(pos, "pos", (self.n_grid, 3))
        x_query = _f32(x_query, "x_query")
        nq = x_query.shape[0]
        if tuple(knn_idx.shape) != (nq, 10) or knn_idx.dtype != torch.int32 or not knn_idx.is_cuda:
            raise ValueError("knn_idx must be an int32 GPU tensor of shape [n_query, 10]")
        knn_idx = knn_idx.contiguous()
        tq = _f32(t_query, "t_query").reshape(-1)
        main = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        side = self.side_stream = bt["streams"][bt["turn"]]
        side.wait_event(ev)
        self._crosses_to(side, pos, x_query, knn_idx, tq)
        with torch.cuda.stream(side):
            x_spatial = torch.empty((n, self.n_grid, 30), dtype=torch.float32, device=self.device)
            y = torch.empty((n, self.n_grid, tq.numel(), 1), dtype=torch.float32, device=self.device)
            x = torch.empty((n, nq, tq.numel(), 1), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.genie_tail_batched(self.ctx, bt["group"] * self.window_batch, n, _ptr(pos), _ptr(x_query), _ptr(knn_idx),
                                                   nq, 10, _ptr(tq), tq.numel(), _ptr(x_spatial), _ptr(y), _ptr(x), self._ws_ptr,
                                                   ctypes.c_void_p(side.cuda_stream)), "genie_tail_batched")
            done = torch.cuda.Event()
            done.record(side)
        self._crosses_to(main, y, x)
        bt["ev"][bt["group"]] = done
        bt["group"] = (bt["group"] + 1) % len(bt["ev"])
        bt["turn"] ^= 1
        bt["n"] = 0
        return y, x, done

    def set_static_edge_attr(self, edge_attr):
        """Register the static edge_attr [P, 3] tensor (genie_set_static_edge_attr): with a station processing order stage 2 then
        reads a processing-order copy whenever it is passed this same tensor. The tensor must stay alive and unchanged."""
        if self._n_prod is not None:
            return
        edge_attr = _f32(edge_attr, "edge_attr", (self.n_prod, 3))
        self._static_ea = edge_attr               # kept alive: the library recognises it by address
        self._static_ea_version = edge_attr._version
        _lib.check(self.lib.genie_set_static_edge_attr(self.ctx, _ptr(edge_attr), _stream()), "genie_set_static_edge_attr")

    def _refresh_static_edge_attr(self, edge_attr):
        """An in-place edit of the registered edge_attr tensor (same address, new `_version`) refreshes the library's copy."""
        ea = getattr(self, "_static_ea", None)
        if ea is not None and edge_attr is not None and edge_attr.data_ptr() == ea.data_ptr() and (
                edge_attr is not ea or edge_attr._version != self._static_ea_version):
            self.set_static_edge_attr(edge_attr)

    def wait_tails(self, stream=None):
        """Make `stream` (default: the current one) wait for every window tail issued so far by `forward_pipelined`."""
        self.check_input_range()
        stream = stream or torch.cuda.current_stream(self.device)
        for s in getattr(self, "side_streams", None) or ():
            stream.wait_stream(s)

    def readout_grid(self, x_spatial, t_query):
        """y[n_grid, T, 1] = TemporalAttention(SpatialDirect(x_spatial), t_query) (module.py:1015-1016)."""
        x_spatial = _f32(x_spatial, "x_spatial", (self.n_grid, 30))
        tq = _f32(t_query, "t_query").reshape(-1)
        out = torch.empty((self.n_grid, tq.numel(), 1), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_readout_grid(self.ctx, _ptr(x_spatial), _ptr(tq), tq.numel(), _ptr(out), _stream()),
                   "genie_readout_grid")
        return out

    def readout_query(self, x_spatial, x_grid, x_query, knn_idx, t_query):
        """x[Q, T, 1] = TemporalAttention(SpatialAttention(x_spatial, x_query, x_grid), t_query) (module.py:1017-1018);
        knn_idx int32 [Q, 10] = the 10 nearest grid nodes of every query."""
        x_spatial = _f32(x_spatial, "x_spatial", (self.n_grid, 30))
        x_grid = _f32(x_grid, "x_grid", (self.n_grid, 3))
        x_query = _f32(x_query, "x_query")
        nq = x_query.shape[0]
        if tuple(knn_idx.shape) != (nq, 10) or knn_idx.dtype != torch.int32 or not knn_idx.is_cuda:
            raise ValueError("knn_idx must be an int32 GPU tensor of shape [n_query, 10]")
        knn_idx = knn_idx.contiguous()
        tq = _f32(t_query, "t_query").reshape(-1)
        out = torch.empty((nq, tq.numel(), 1), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_readout_query(self.ctx, _ptr(x_spatial), _ptr(x_grid), _ptr(x_query), _ptr(knn_idx), nq, 10,
                                                _ptr(tq), tq.numel(), _ptr(out), self._ws_ptr, _stream()), "genie_readout_query")
        return out

    def readouts_forked(self, x_spatial, x_grid, x_query, knn_idx, t_query):
        """Both read-outs of one window, (y [n_grid, T, 1], x [Q, T, 1]), with the grid read-out forked onto a side stream: the two are
        independent given x_spatial (module.py:1015-1016 / :1017-1018) and each is a short launch that fills a fraction of the GPU (10 000
        nodes = 625 wave tiles on 1 024 SIMDs), so one stream runs them back to back for 26 + 37 us where two streams overlap them.
        The current stream joins the side stream before returning: both results are valid on the current stream, as with the plain
        calls, and bit-identical to them."""
        main = torch.cuda.current_stream(self.device)
        side = getattr(self, "_ro_stream", None)
        if side is None:
            side = self._ro_stream = self._new_side_stream()
        x_spatial = _f32(x_spatial, "x_spatial", (self.n_grid, 30))
        tq = _f32(t_query, "t_query").reshape(-1)
        y = torch.empty((self.n_grid, tq.numel(), 1), dtype=torch.float32, device=self.device)      # (allocated on the current stream)
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        _lib.check(self.lib.genie_readout_grid(self.ctx, _ptr(x_spatial), _ptr(tq), tq.numel(), _ptr(y), ctypes.c_void_p(side.cuda_stream)),
                   "genie_readout_grid")
        join = torch.cuda.Event()
        join.record(side)
        x = self.readout_query(x_spatial, x_grid, x_query, knn_idx, tq)
        main.wait_event(join)
        return y, x

    def readout_grid_latent(self, x_spatial, t_query):
        """(y [n_grid, T, 1], y_latent [n_grid, 30] = SpatialDirect(x_spatial)) (module.py:978-979; genie_readout_grid_latent)."""
        x_spatial = _f32(x_spatial, "x_spatial", (self.n_grid, 30))
        tq = _f32(t_query, "t_query").reshape(-1)
        out = torch.empty((self.n_grid, tq.numel(), 1), dtype=torch.float32, device=self.device)
        lat = torch.empty((self.n_grid, 30), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_readout_grid_latent(self.ctx, _ptr(x_spatial), _ptr(tq), tq.numel(), _ptr(out), _ptr(lat), _stream()),
                   "genie_readout_grid_latent")
        return out, lat

    def spatial_attention(self, x_spatial, x_grid, x_query, knn_idx, t_query):
        """SpatialAttention(x_spatial, x_query, x_grid) [Q, 30] (module.py:981, `x_src` of the candidate sources) by the read-out
        kernel (genie_readout_query_latent; its TemporalAttention output is discarded: Q is a handful of sources)."""
        x_spatial = _f32(x_spatial, "x_spatial", (self.n_grid, 30))
        x_grid = _f32(x_grid, "x_grid", (self.n_grid, 3))
        x_query = _f32(x_query, "x_query")
        nq = x_query.shape[0]
        if tuple(knn_idx.shape) != (nq, 10) or knn_idx.dtype != torch.int32 or not knn_idx.is_cuda:
            raise ValueError("knn_idx must be an int32 GPU tensor of shape [n_query, 10]")
        tq = _f32(t_query, "t_query").reshape(-1)
        out = torch.empty((nq, tq.numel(), 1), dtype=torch.float32, device=self.device)
        lat = torch.empty((nq, 30), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_readout_query_latent(self.ctx, _ptr(x_spatial), _ptr(x_grid), _ptr(x_query), _ptr(knn_idx.contiguous()), nq, 10,
                                                       _ptr(tq), tq.numel(), _ptr(out), _ptr(lat), self._ws_ptr, _stream()),
                   "genie_readout_query_latent")
        return lat

    def assoc_fwd(self, y_latent, mask_src, x_latent, Mask, edge_attr):
        """BipartiteGraphReadOutOperator + DataAggregationAssociationPhase (module.py:986-990) in HIP (genie_assoc_fwd):
        y_latent [G,30], mask_src [G] or [G,1], x_latent [P,30], Mask [P,4], edge_attr [P,3] -> [P,30]. Call after the
        DataAggregation stage 2 of the same window (it reuses the c / wu / wv rows of the current workspace slot)."""
        if not getattr(self, "assoc_ready", False):
            raise _lib.GenieHipError("association-head parameters were not uploaded (or have another model definition's shapes)")
        P = self.n_prod
        y_latent = _f32(y_latent, "y_latent", (self.n_grid, 30))
        mask_src = _f32(mask_src, "mask_src").reshape(-1)
        if mask_src.numel() != self.n_grid:
            raise ValueError("mask_src must have n_grid entries")
        x_latent = _f32(x_latent, "x_latent", (P, 30))
        Mask = _f32(Mask, "Mask", (P, 4))
        edge_attr = _f32(edge_attr, "edge_attr", (P, 3))
        self._refresh_static_edge_attr(edge_attr)
        need = int(self.lib.genie_assoc_workspace_bytes(self.ctx))
        if getattr(self, "_assoc_ws", None) is None or self._assoc_ws.numel() * 4 < need:
            self._assoc_ws = empty_f32((need + 3) // 4, self.device)
        out = torch.empty((P, 30), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_assoc_fwd(self.ctx, _ptr(y_latent), _ptr(mask_src), _ptr(x_latent), _ptr(Mask), _ptr(edge_attr),
                                            _ptr(out), _ptr(self._assoc_ws), self._ws_ptr, _stream()), "genie_assoc_fwd")
        return out

    def assoc_train_fwd(self, y_latent, mask_src, x_latent, Mask, edge_attr):
        """Training forward of the P-sized association heads (genie_assoc_train_fwd): as `assoc_fwd`, returns (s [P, 30], asave)."""
        if not getattr(self, "assoc_ready", False):
            raise _lib.GenieHipError("association-head parameters were not uploaded (or have another model definition's shapes)")
        P = self.n_prod
        y_latent = _f32(y_latent, "y_latent", (self.n_grid, 30))
        mask_src = _f32(mask_src, "mask_src").reshape(-1)
        x_latent, Mask, edge_attr = _f32(x_latent, "x_latent", (P, 30)), _f32(Mask, "Mask", (P, 4)), _f32(edge_attr, "edge_attr", (P, 3))
        need = int(self.lib.genie_assoc_workspace_bytes(self.ctx))
        if getattr(self, "_assoc_ws", None) is None or self._assoc_ws.numel() * 4 < need:
            self._assoc_ws = empty_f32((need + 3) // 4, self.device)
        asave = empty_f32(int(self.lib.genie_assoc_train_save_floats(self.ctx)), self.device)
        out = empty_f32(P * 30, self.device).view(P, 30)
        _lib.check(self.lib.genie_assoc_train_fwd(self.ctx, _ptr(y_latent), _ptr(mask_src), _ptr(x_latent), _ptr(Mask), _ptr(edge_attr),
                                                  _ptr(out), _ptr(asave), _ptr(self._assoc_ws), self._ws_ptr, _stream()), "genie_assoc_train_fwd")
        return out, asave

    def assoc_train_bwd(self, y_latent, mask_src, x_latent, Mask, edge_attr, asave, d_s):
        """Backward of `assoc_train_fwd` (genie_assoc_train_bwd): d_s [P, 30] -> (d_y_latent [G, 30], dict parameter name -> gradient)."""
        P, G = self.n_prod, self.n_grid
        d_s = _f32(d_s, "d_s", (P, 30))
        mask_src = _f32(mask_src, "mask_src").reshape(-1)
        need = int(self.lib.genie_assoc_train_scratch_floats(self.ctx))
        if getattr(self, "_train_scratch", None) is None or self._train_scratch.numel() < need:
            self._train_scratch = empty_f32(need, self.device)
        d_ylat = torch.empty((G, 30), dtype=torch.float32, device=self.device)
        blob = torch.empty(self._blob.numel(), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_assoc_train_bwd(self.ctx, _ptr(y_latent), _ptr(mask_src), _ptr(x_latent), _ptr(Mask), _ptr(edge_attr),
                                                  _ptr(asave), _ptr(d_s), _ptr(self._train_scratch), _ptr(d_ylat), _ptr(blob), _stream()),
                   "genie_assoc_train_bwd")
        return d_ylat, {name: blob[off:off + n] for name, n, off in zip(self.w_names, self.w_numel, self.w_off)}

    def lslc_fwd(self, head, s_rows, a_edges, dt_partition, tpick, ipick, phase_label, tlatent, col, eps):
        """LocalSliceLgCollapse P (head 0) / S (head 1), module.py:610-659, in HIP (genie_lslc_fwd): s_rows [P, 30], a_edges int32
        [n_sta * l_dt * 10] time-pointer table, dt_partition [l_dt], tpick / phase_label fp32 [n], ipick int32 [n], tlatent
        [P, C] fp32 with the phase's theoretical arrival in column `col` -> [n, 15]."""
        if not getattr(self, "assoc_ready", False):
            raise _lib.GenieHipError("association-head parameters were not uploaded (or have another model definition's shapes)")
        s_rows = _f32(s_rows, "s_rows", (self.n_prod, 30))
        tpick = _f32(tpick, "tpick").reshape(-1)
        n = int(tpick.numel())
        phase_label = _f32(phase_label, "phase_label").reshape(-1)
        tlatent = _f32(tlatent, "tlatent")
        if (a_edges.dtype != torch.int32 or not a_edges.is_cuda or ipick.dtype != torch.int32 or ipick.numel() != n
                or phase_label.numel() != n or tlatent.dim() != 2 or tlatent.shape[0] != self.n_prod):
            raise ValueError("lslc_fwd: a_edges / ipick must be int32 GPU tensors, one ipick / phase_label per pick, tlatent [P, C]")
        a_edges, ipick = a_edges.contiguous(), ipick.contiguous()
        # the reference indexes the time-pointer table with floor((tpick - t0) / dt) and ipick and fails on an index outside it
        # (module.py:635-640); the kernel clamps such an index and reports it (check_index_flags: raised at the next call, no host
        # synchronisation here -- until round 5 the bounds were read back per call)
        self.check_index_flags()
        t0, dt, dtp, l_dt = time_partition(dt_partition)
        out = torch.empty((n, 15), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_lslc_fwd(self.ctx, int(head), _ptr(s_rows), _ptr(a_edges), int(a_edges.numel()), l_dt,
                                           t0, dt, _ptr(dtp), float(eps), _ptr(tlatent), int(tlatent.shape[1]), int(col), _ptr(tpick), _ptr(ipick),
                                           _ptr(phase_label), n, _ptr(out), _stream()), "genie_lslc_fwd")
        return out

    def lslc_bwd(self, s_rows, a_edges_ps, dt_partition, tpick, ipick, phase_label, tlatent, eps, d_p, d_s):
        """Backward of the two `lslc_fwd` heads of a training step (genie_lslc_bwd + genie_seg_rows): d_p / d_s [n, 15] (None = zero) ->
        (d_s_rows [P, 30], dict parameter name -> gradient). The edge sort by product node is index plumbing (torch.sort, stable)."""
        s_rows = _f32(s_rows, "s_rows", (self.n_prod, 30))
        tpick = _f32(tpick, "tpick").reshape(-1)
        n = int(tpick.numel())
        phase_label = _f32(phase_label, "phase_label").reshape(-1)
        tlatent = _f32(tlatent, "tlatent")
        t0, dt, dtp, l_dt = time_partition(dt_partition)
        dev = self.device
        ds = torch.zeros((self.n_prod, 30), dtype=torch.float32, device=dev)
        blob = torch.zeros(self._blob.numel(), dtype=torch.float32, device=dev)
        part = torch.empty(int(self.lib.genie_lslc_bwd_part_floats(n)), dtype=torch.float32, device=dev)
        for head, d_out in ((0, d_p), (1, d_s)):
            if d_out is None or n == 0:
                continue
            d_out = _f32(d_out, "d_out", (n, 15))
            a_edges = a_edges_ps[head]
            erow = torch.empty((n * 10, 32), dtype=torch.float32, device=dev)
            etgt = torch.empty(n * 10, dtype=torch.int32, device=dev)
            _lib.check(self.lib.genie_lslc_bwd(self.ctx, head, _ptr(s_rows), _ptr(a_edges), int(a_edges.numel()), l_dt, t0, dt,
                                               _ptr(dtp), float(eps), _ptr(tlatent), int(tlatent.shape[1]), head, _ptr(tpick), _ptr(ipick), _ptr(phase_label),
                                               n, _ptr(d_out), _ptr(erow), _ptr(etgt), _ptr(part), _ptr(blob), _stream()), "genie_lslc_bwd")
            order = torch.sort(etgt, stable=True)[1].to(torch.int32)
            _lib.check(self.lib.genie_seg_rows(_ptr(erow), _ptr(etgt), _ptr(order), n * 10, _ptr(ds), _stream()), "genie_seg_rows")
        return ds, {name: blob[off:off + k] for name, k, off in zip(self.w_names, self.w_numel, self.w_off)}

    def _arrivals_args(self, stime, src_embed, trv_src, arrival_p, arrival_s, tpick, ipick, phase_label):
        if not getattr(self, "assoc_ready", False):
            raise _lib.GenieHipError("association-head parameters were not uploaded (or have another model definition's shapes)")
        stime = _f32(stime, "stime").reshape(-1)
        n_src = int(stime.numel())
        src_embed = _f32(src_embed, "src_embed", (n_src, 30))
        trv_src = _f32(trv_src, "trv_src")
        if trv_src.dim() != 3 or trv_src.shape[0] != n_src or trv_src.shape[2] != 2:
            raise ValueError("arrivals: trv_src must be [n_src, n_sta, 2]")
        n_sta = int(trv_src.shape[1])
        tpick = _f32(tpick, "tpick").reshape(-1)
        n = int(tpick.numel())
        arrival_p, arrival_s = _f32(arrival_p, "arrival_p", (n, 15)), _f32(arrival_s, "arrival_s", (n, 15))
        phase_label = _f32(phase_label, "phase_label").reshape(-1)
        if n == 0 or n_src == 0:
            raise ValueError("arrivals: needs at least one pick and one source")
        ip = ipick.reshape(-1).long().contiguous()
        if ip.numel() != n or phase_label.numel() != n:
            raise ValueError("arrivals: one station index and one phase label per pick")
        # one segment per station, empty ones included (their launches find no target and return): every size is known on the host, so
        # nothing here waits for the device (until round 5: two read-backs for the index range and torch.unique_consecutive's own). A
        # station index outside [0, n_sta) is reported by the device (check_index_flags: IndexError at the next call) and clamped here.
        self.check_index_flags()
        _lib.check(self.lib.genie_index_check(self.ctx, _ptr(ip), n, 0, n_sta, 2, _stream()), "genie_index_check")
        ipc = ip.clamp(0, n_sta - 1)
        order = torch.sort(ipc, stable=True)[1]
        counts = torch.zeros(n_sta, dtype=torch.int64, device=ip.device).scatter_add_(0, ipc, torch.ones_like(ipc))
        seg_start = torch.cumsum(counts, 0) - counts
        i32 = lambda t: t.to(torch.int32).contiguous()
        segs = (i32(order), torch.arange(n_sta, dtype=torch.int32, device=ip.device), i32(seg_start), i32(counts))
        return n_src, n_sta, n, (stime, src_embed, trv_src, arrival_p, arrival_s, tpick, phase_label), segs

    def arrivals_fwd(self, stime, src_embed, trv_src, arrival_p, arrival_s, tpick, ipick, phase_label, eps, train=False):
        """StationSourceAttentionMergedPhases (`Arrivals`, module.py:662-775) in HIP (genie_arrivals_fwd): stime [n_src], src_embed
        [n_src, 30], trv_src [n_src, n_sta, 2], arrival_p / arrival_s [n, 15], tpick / phase_label [n], ipick integer [n] ->
        [n_src, n, 2]. `edge_index[0].max()` (module.py:762-763: the pick the reference treats as the null pick; the real null pick
        whenever some source has |stime| < 2 eps) is found on the device (k_arr_e0max). train=True (genie_arrivals_train_fwd): also
        returns the state `arrivals_bwd` needs."""
        n_src, n_sta, n, data, segs = self._arrivals_args(stime, src_embed, trv_src, arrival_p, arrival_s, tpick, ipick, phase_label)
        ctx = torch.empty(n_src * 192, dtype=torch.float32, device=self.device)
        flag = torch.empty(1, dtype=torch.int32, device=self.device)
        out = torch.empty((n_src, n, 2), dtype=torch.float32, device=self.device)
        head = (self.ctx, n_src, _ptr(data[0]), _ptr(data[1]), _ptr(data[2]), n_sta, _ptr(data[3]), _ptr(data[4]), _ptr(data[5]), _ptr(data[6]),
                n, _ptr(segs[0]), _ptr(segs[1]), _ptr(segs[2]), _ptr(segs[3]), int(segs[1].numel()), float(eps), _ptr(ctx), _ptr(flag))
        if not train:
            _lib.check(self.lib.genie_arrivals_fwd(*head, _ptr(out), _stream()), "genie_arrivals_fwd")
            return out
        save = torch.empty(int(self.lib.genie_arrivals_train_save_floats(n_src, n)), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_arrivals_train_fwd(*head, _ptr(out), _ptr(save), _stream()), "genie_arrivals_train_fwd")
        return out, (data, segs, ctx, flag, save, float(eps))

    def arrivals_bwd(self, state, d_out):
        """Backward of `arrivals_fwd(..., train=True)` (genie_arrivals_bwd): d_out [n_src, n, 2] -> (d_src_embed [n_src, 30],
        d_arrival_p [n, 15], d_arrival_s [n, 15], dict parameter name -> gradient)."""
        data, segs, ctx, flag, save, eps = state
        n_src, n, n_sta, n_useg = int(data[0].numel()), int(data[5].numel()), int(data[2].shape[1]), int(segs[1].numel())
        d_out = _f32(d_out, "d_out", (n_src, n, 2))
        dev = self.device
        scratch = torch.empty(int(self.lib.genie_arrivals_bwd_scratch_floats(n_src, n, n_useg)), dtype=torch.float32, device=dev)
        blob = torch.zeros(self._blob.numel(), dtype=torch.float32, device=dev)
        d_src = torch.empty((n_src, 30), dtype=torch.float32, device=dev)
        d_p = torch.empty((n, 15), dtype=torch.float32, device=dev)
        d_s = torch.empty((n, 15), dtype=torch.float32, device=dev)
        _lib.check(self.lib.genie_arrivals_bwd(self.ctx, n_src, _ptr(data[0]), _ptr(data[1]), _ptr(data[2]), n_sta, _ptr(data[3]), _ptr(data[4]),
                                               _ptr(data[5]), _ptr(data[6]), n, _ptr(segs[0]), _ptr(segs[1]), _ptr(segs[2]), _ptr(segs[3]), n_useg,
                                               eps, _ptr(ctx), _ptr(flag), _ptr(save), _ptr(d_out), _ptr(scratch), _ptr(d_src), _ptr(d_p),
                                               _ptr(d_s), _ptr(blob), _stream()), "genie_arrivals_bwd")
        return d_src, d_p, d_s, {name: blob[off:off + k] for name, k, off in zip(self.w_names, self.w_numel, self.w_off)}

    def train_fwd(self, Slice, Mask, edge_attr, want_x_latent=True):
        """Training forward of DataAggregation + the P-sized half of Bipartite_ReadIn (genie_da_train_fwd): returns
        (r [G, 30] = station sums of the gated messages, x_latent [P, 30] or None, save = the pre-activations the backward needs)."""
        self.check_input_range()
        P = self.n_prod
        Slice, Mask = _f32(Slice, "Slice", (P, 4)), _f32(Mask, "Mask", (P, 4))
        edge_attr = _f32(edge_attr, "edge_attr", (P, 3))
        save = empty_f32(int(self.lib.genie_train_save_floats(self.ctx)), self.device)
        r = torch.empty((self.n_grid, 30), dtype=torch.float32, device=self.device)
        x_latent = empty_f32(P * 30, self.device).view(P, 30) if want_x_latent else None
        _lib.check(self.lib.genie_da_train_fwd(self.ctx, _ptr(Slice), _ptr(Mask), _ptr(edge_attr), _ptr(save), _ptr(x_latent), _ptr(r),
                                               self._ws_ptr, _stream()), "genie_da_train_fwd")
        return r, x_latent, save

    def train_bwd(self, Slice, Mask, edge_attr, save, d_r):
        """Backward of `train_fwd` (genie_da_train_bwd): d_r [G, 30] -> dict parameter name -> gradient (views into one blob laid
        out like the weight mirror) for every DataAggregation parameter and Bipartite_ReadIn.fc1 / activate1."""
        d = torch.zeros((self.n_grid, 32), dtype=torch.float32, device=self.device)
        d[:, :30] = d_r
        need = int(self.lib.genie_train_scratch_floats(self.ctx))
        if getattr(self, "_train_scratch", None) is None or self._train_scratch.numel() < need:
            self._train_scratch = empty_f32(need, self.device)
        blob = torch.empty(self._blob.numel(), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_da_train_bwd(self.ctx, _ptr(Slice), _ptr(Mask), _ptr(edge_attr), _ptr(save), _ptr(d),
                                               _ptr(self._train_scratch), _ptr(blob), _stream()), "genie_da_train_bwd")
        return {name: blob[off:off + n] for name, n, off in zip(self.w_names, self.w_numel, self.w_off)}

    # ---- training step of forward_fixed_source: the whole path in HIP, both directions ----------------------------------
    def reverse_query_table(self, knn_idx):
        """The query kNN table [Q, 10] reversed (index plumbing for genie_tail_train_bwd): (rowptr int32 [n_grid + 1], edge ids
        int32 [Q * 10]) = for every grid node the attention edges `i * 10 + k` that end in it, ascending. Cached per table."""
        key = (knn_idx.data_ptr(), knn_idx._version, tuple(knn_idx.shape))
        if getattr(self, "_rknn_key", None) != key:
            flat = knn_idx.reshape(-1).long()
            order = torch.sort(flat, stable=True)[1]
            rowptr = torch.zeros(self.n_grid + 1, dtype=torch.int64, device=knn_idx.device)
            counts = torch.zeros(self.n_grid, dtype=torch.int64, device=knn_idx.device).scatter_add_(0, flat, torch.ones_like(flat))
            rowptr[1:] = torch.cumsum(counts, 0)          # (torch.bincount would read the largest index back: a host wait mid-backward)
            self._rknn = (rowptr.to(torch.int32), order.to(torch.int32).contiguous())
            self._rknn_key, self._rknn_ref = key, knn_idx
        return self._rknn

    def path_train_fwd(self, Slice, Mask, edge_attr, pos, x_query, knn_idx, t_query, want_x_latent=False, want_y_latent=False):
        """Training forward of `forward_fixed_source` (genie_da_train_fwd + genie_tail_train_fwd): returns (y [G, T, 1], x [Q, T, 1],
        x_spatial [G, 30] (a view into tsave), y_latent [G, 30] or None, x_latent [P, 30] or None, save, tsave)."""
        self.check_input_range()
        P = self.n_prod
        Slice, Mask = _f32(Slice, "Slice", (P, 4)), _f32(Mask, "Mask", (P, 4))
        edge_attr = _f32(edge_attr, "edge_attr", (P, 3))
        pos = _f32(pos, "pos", (self.n_grid, 3))
        x_query = _f32(x_query, "x_query")
        nq = int(x_query.shape[0])
        if tuple(knn_idx.shape) != (nq, 10) or knn_idx.dtype != torch.int32 or not knn_idx.is_cuda:
            raise ValueError("knn_idx must be an int32 GPU tensor of shape [n_query, 10]")
        knn_idx = knn_idx.contiguous()
        tq = _f32(t_query, "t_query").reshape(-1)
        dev, G = self.device, self.n_grid
        save = empty_f32(int(self.lib.genie_train_save_floats(self.ctx)), dev)
        tsave = empty_f32(int(self.lib.genie_tail_train_save_floats(self.ctx)), dev)
        r = torch.empty((G, 30), dtype=torch.float32, device=dev)
        x_latent = empty_f32(P * 30, dev).view(P, 30) if want_x_latent else None
        y_latent = torch.empty((G, 30), dtype=torch.float32, device=dev) if want_y_latent else None
        y = torch.empty((G, tq.numel(), 1), dtype=torch.float32, device=dev)
        x = torch.empty((nq, tq.numel(), 1), dtype=torch.float32, device=dev)
        _lib.check(self.lib.genie_da_train_fwd(self.ctx, _ptr(Slice), _ptr(Mask), _ptr(edge_attr), _ptr(save), _ptr(x_latent), _ptr(r),
                                               self._ws_ptr, _stream()), "genie_da_train_fwd")
        _lib.check(self.lib.genie_tail_train_fwd(self.ctx, _ptr(pos), _ptr(x_query), _ptr(knn_idx), nq, 10, _ptr(tq), tq.numel(),
                                                 _ptr(tsave), _ptr(y_latent), _ptr(y), _ptr(x), self._ws_ptr, _stream()),
                   "genie_tail_train_fwd")
        x_spatial = tsave[112 * G:142 * G].view(G, 30)
        return y, x, x_spatial, y_latent, x_latent, save, tsave

    def path_train_bwd(self, Slice, Mask, edge_attr, pos, x_query, knn_idx, t_query, save, tsave, d_y, d_x, d_xs=None, d_ylat=None,
                       d_qlat=None):
        """Backward of `path_train_fwd` (genie_train_bwd): upstream gradients d_y [G, T], d_x [Q, T] (+ optional d_xs [G, 30] on
        x_spatial, d_ylat [G, 30] on y_latent and d_qlat [Q, 30] on the SpatialAttention output of the query rows, from consumers
        outside the path) -> dict parameter name -> gradient (views into one blob laid out like the weight mirror), every
        parameter of the path."""
        dev, G, P = self.device, self.n_grid, self.n_prod
        Slice, Mask = _f32(Slice, "Slice", (P, 4)), _f32(Mask, "Mask", (P, 4))      # raw pointers below: same normalisation as the forward
        edge_attr = _f32(edge_attr, "edge_attr", (P, 3))
        pos = _f32(pos, "pos", (G, 3))
        x_query = _f32(x_query, "x_query")
        nq = int(x_query.shape[0])
        if tuple(knn_idx.shape) != (nq, 10) or knn_idx.dtype != torch.int32 or not knn_idx.is_cuda:
            raise ValueError("knn_idx must be an int32 GPU tensor of shape [n_query, 10]")
        knn_idx = knn_idx.contiguous()
        tq = _f32(t_query, "t_query").reshape(-1)
        T = tq.numel()
        d_y = _f32(d_y, "d_y").reshape(G, T)
        d_x = _f32(d_x, "d_x").reshape(nq, T)
        d_xs = _f32(d_xs, "d_xs", (G, 30)) if d_xs is not None else None
        d_ylat = _f32(d_ylat, "d_ylat", (G, 30)) if d_ylat is not None else None
        d_qlat = _f32(d_qlat, "d_qlat", (nq, 30)) if d_qlat is not None else None
        rp, re = self.reverse_query_table(knn_idx)
        need_t = int(self.lib.genie_tail_train_scratch_floats(self.ctx, nq))
        need_f = int(self.lib.genie_train_scratch_floats(self.ctx))
        if getattr(self, "_tail_scratch", None) is None or self._tail_scratch.numel() < need_t:
            self._tail_scratch = empty_f32(need_t, dev)
        if getattr(self, "_train_scratch", None) is None or self._train_scratch.numel() < need_f:
            self._train_scratch = empty_f32(need_f, dev)
        d_r = torch.empty((G, 32), dtype=torch.float32, device=dev)
        blob = torch.empty(int(self.lib.genie_train_grad_floats()), dtype=torch.float32, device=dev)
        _lib.check(self.lib.genie_train_bwd(self.ctx, _ptr(Slice), _ptr(Mask), _ptr(edge_attr), _ptr(save), _ptr(pos), _ptr(x_query),
                                            _ptr(knn_idx), _ptr(rp), _ptr(re), nq, 10, _ptr(tq), T, _ptr(tsave), _ptr(d_y), _ptr(d_x),
                                            _ptr(d_xs), _ptr(d_ylat), _ptr(d_qlat), _ptr(self._tail_scratch), _ptr(self._train_scratch), _ptr(d_r),
                                            _ptr(blob), _stream()), "genie_train_bwd")
        return {name: blob[off:off + n] for name, n, off in zip(self.w_names, self.w_numel, self.w_off)}

    def tail_train_bwd(self, pos, x_query, knn_idx, t_query, tsave, d_y, d_x, d_xs=None, d_ylat=None):
        """The tail half of `path_train_bwd` alone (genie_tail_train_bwd): returns (d_r [G, 32], gradient blob). Phase timing
        (bench.py --mode train) and tests; `train_bwd(Slice, Mask, edge_attr, save, d_r[:, :30])` is the other half."""
        dev, G = self.device, self.n_grid
        x_query = _f32(x_query, "x_query")
        nq = int(x_query.shape[0])
        tq = _f32(t_query, "t_query").reshape(-1)
        T = tq.numel()
        d_y, d_x = _f32(d_y, "d_y").reshape(G, T), _f32(d_x, "d_x").reshape(nq, T)
        rp, re = self.reverse_query_table(knn_idx)
        need_t = int(self.lib.genie_tail_train_scratch_floats(self.ctx, nq))
        if getattr(self, "_tail_scratch", None) is None or self._tail_scratch.numel() < need_t:
            self._tail_scratch = empty_f32(need_t, dev)
        d_r = torch.empty((G, 32), dtype=torch.float32, device=dev)
        blob = torch.empty(int(self.lib.genie_train_grad_floats()), dtype=torch.float32, device=dev)
        _lib.check(self.lib.genie_tail_train_bwd(self.ctx, _ptr(_f32(pos, "pos", (G, 3))), _ptr(x_query), _ptr(knn_idx), _ptr(rp), _ptr(re), nq, 10,
                                                 _ptr(tq), T, _ptr(tsave), _ptr(d_y), _ptr(d_x), _ptr(d_xs), _ptr(d_ylat), None,
                                                 _ptr(self._tail_scratch), _ptr(d_r), _ptr(blob), _stream()), "genie_tail_train_bwd")
        return d_r, blob

    def nbr_mean(self, x_sta=None, x_src=None):
        """Neighbour means over the product graph of [P, C] rows (C <= 32): (mean over station neighbours of x_sta, mean over
        source neighbours of x_src); genie_nbr_mean on rows padded to 16 / 32 floats ([P, 30] rows as they are)."""
        outs = []
        args = []
        width = None
        for x in (x_sta, x_src):
            if x is None:
                args += [None, None]
                outs.append(None)
                continue
            x = _f32(x, "x")
            C = x.shape[1]
            w = 30 if C == 30 else (16 if C <= 16 else 32)        # [P, 30] rows go through unpadded
            if C > 32 or x.shape[0] != self.n_prod or (width is not None and w != width):
                raise ValueError("nbr_mean: rows must be [n_prod, C <= 32] of one padded width")
            width = w
            xp = x if C == w else torch.nn.functional.pad(x, (0, w - C))
            o = torch.empty_like(xp)
            args += [xp, o]
            outs.append((o, C))
        _lib.check(self.lib.genie_nbr_mean(self.ctx, _ptr(args[0]), _ptr(args[2]), _ptr(args[1]), _ptr(args[3]), width, _stream()),
                   "genie_nbr_mean")
        return tuple(None if o is None else o[0][:, :o[1]] for o in outs)

    def nbr_mean_bwd(self, g_sta=None, g_src=None):
        """Adjoint of `nbr_mean` (genie_nbr_mean_bwd): gradients w.r.t. the rows that were averaged."""
        outs, args, width = [], [], None
        for g in (g_sta, g_src):
            if g is None:
                args += [None, None]
                outs.append(None)
                continue
            g = _f32(g, "grad")
            C = g.shape[1]
            w = 30 if C == 30 else (16 if C <= 16 else 32)
            if C > 32 or g.shape[0] != self.n_prod or (width is not None and w != width):
                raise ValueError("nbr_mean_bwd: rows must be [n_prod, C <= 32] of one padded width")
            width = w
            gp = g if C == w else torch.nn.functional.pad(g, (0, w - C))
            o = torch.empty_like(gp)
            args += [gp, o]
            outs.append((o, C))
        _lib.check(self.lib.genie_nbr_mean_bwd(self.ctx, _ptr(args[0]), _ptr(args[2]), _ptr(args[1]), _ptr(args[3]), width, _stream()),
                   "genie_nbr_mean_bwd")
        return tuple(None if o is None else o[0][:, :o[1]] for o in outs)

    def prelu_bwd(self, x, dy, slope):
        """Backward of a single-slope PReLU over a contiguous fp32 tensor (genie_prelu_bwd): returns (dx, dslope[1])."""
        x, dy = _f32(x, "x"), _f32(dy, "dy", tuple(x.shape))
        slope = _f32(slope, "slope", (1,))
        dx, ds = torch.empty_like(x), torch.empty_like(slope)
        scratch = torch.empty(2048, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.genie_prelu_bwd(_ptr(x), _ptr(dy), _ptr(slope), x.numel(), _ptr(dx), _ptr(ds), _ptr(scratch), _stream()),
                   "genie_prelu_bwd")
        return dx, ds

    def linear_bwd_wb(self, x, dy, bias=True):
        """Weight / bias gradients of y = x W^T + b over contiguous rows x [N, K], dy [N, M] (genie_linear_bwd_wb)."""
        x = _f32(x, "x")
        N, K = x.shape
        dy = _f32(dy, "dy")
        M = dy.shape[1]
        if dy.shape[0] != N:
            raise ValueError("linear_bwd_wb: x and dy must have the same number of rows")
        dW = torch.empty((M, K), dtype=torch.float32, device=x.device)
        db = torch.empty(M, dtype=torch.float32, device=x.device) if bias else None
        if N == 0:
            dW.zero_()
            if bias:
                db.zero_()
            return dW, db
        scratch = torch.empty(int(self.lib.genie_linear_bwd_scratch_floats(K)), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.genie_linear_bwd_wb(_ptr(x), _ptr(dy), N, K, M, _ptr(dW), _ptr(db), _ptr(scratch), _stream()),
                   "genie_linear_bwd_wb")
        return dW, db

    def set_absolute_pos(self, pos_sta, pos_src):
        """`use_absolute_pos` (config.yaml:92): station [n_sta,3] / source [n_grid_ext,3] positions appended (scaled by
        1 / (3 scale_rel)) to every product node's input; `None, None` = off (genie_set_absolute_pos)."""
        if pos_sta is None or pos_src is None:
            _lib.check(self.lib.genie_set_absolute_pos(self.ctx, None, None, _stream()), "genie_set_absolute_pos")
            return
        if self._n_prod is not None:      # irregular product graph: positions per PRODUCT node (the station's, the source node's)
            pos_sta = _f32(pos_sta, "pos_sta", (self._n_prod, 3))
            pos_src = _f32(pos_src, "pos_src", (self._n_prod, 3))
        else:
            pos_sta = _f32(pos_sta, "pos_sta", (self.n_sta, 3))
            pos_src = _f32(pos_src, "pos_src", (self.n_grid_ext, 3))
        _lib.check(self.lib.genie_set_absolute_pos(self.ctx, _ptr(pos_sta), _ptr(pos_src), _stream()), "genie_set_absolute_pos")
        torch.cuda.current_stream(self.device).synchronize()

    def set_edge_features(self, pos_sta, pos_src):
        """DataAggregationEdges (module.py:102-174): station / source-node positions [n,3] from which the library derives
        the mean edge features; `None, None` switches back to plain DataAggregation (genie_set_edge_features). On an irregular product
        graph both arguments are [n_prod, 3]: the station's and the source node's position of every product node (the mean edge
        feature of a node runs over its present neighbours, so the static terms are per product node there)."""
        if pos_sta is None or pos_src is None:
            _lib.check(self.lib.genie_set_edge_features(self.ctx, None, None, _stream()), "genie_set_edge_features")
            return
        if self._n_prod is not None:      # irregular product graph: positions per PRODUCT node (the station's, the source node's)
            pos_sta = _f32(pos_sta, "pos_sta", (self._n_prod, 3))
            pos_src = _f32(pos_src, "pos_src", (self._n_prod, 3))
        else:
            pos_sta = _f32(pos_sta, "pos_sta", (self.n_sta, 3))
            pos_src = _f32(pos_src, "pos_src", (self.n_grid_ext, 3))
        _lib.check(self.lib.genie_set_edge_features(self.ctx, _ptr(pos_sta), _ptr(pos_src), _stream()), "genie_set_edge_features")
        torch.cuda.current_stream(self.device).synchronize()     # the position tensors may be temporaries

    def set_scale_t(self, scale_t):
        _lib.check(self.lib.genie_set_scale_t(self.ctx, ctypes.c_float(float(scale_t))), "genie_set_scale_t")

    def embed_window(self, pick_t, pick_sta, pick_phase, t0, max_t, kernel_sig_t, dt, trv, presplit=False):
        """Slice, Mask [n_grid_ext*n_sta, 4] for the window starting at t0, from picks resident on the GPU
        (process_utils.py:460-642). pick_t float64, pick_sta / pick_phase int32 GPU tensors; trv [rows, 2] fp32.
        `presplit`: also leave the split rows of the f16x2 stage-1 kernel in the workspace, so that the next stage-1 call on
        exactly these (Slice, Mask) skips its split pass (genie_embed_window_split; do not modify them in between)."""
        rows = self.n_prod_ext
        trv = _f32(trv, "trv", (rows, 2))
        n = int(pick_t.numel())
        if n:
            if pick_t.dtype != torch.float64 or pick_sta.dtype != torch.int32 or pick_phase.dtype != torch.int32:
                raise ValueError("pick_t must be float64, pick_sta / pick_phase int32")
            if not (pick_t.is_cuda and pick_sta.is_cuda and pick_phase.is_cuda):
                raise ValueError("pick arrays must live on the GPU")
        n_time = int(self.lib.genie_embed_ntime(float(t0), float(max_t), float(kernel_sig_t), float(dt)))
        need = 2 * self.n_sta * n_time
        if getattr(self, "_emb", None) is None or self._emb.numel() < need:
            self._emb = torch.empty(need, dtype=torch.float32, device=self.device)
        Slice = torch.empty((rows, 4), dtype=torch.float32, device=self.device)
        Mask = torch.empty((rows, 4), dtype=torch.float32, device=self.device)
        common = (self.ctx, _ptr(pick_t) if n else None, _ptr(pick_sta) if n else None, _ptr(pick_phase) if n else None, n,
                  float(t0), float(max_t), float(kernel_sig_t), float(dt), _ptr(trv), _ptr(self._emb), _ptr(Slice), _ptr(Mask))
        if presplit:
            # the message-mask row goes to the workspace copy of the slot this window will run under (forward_pipelined /
            # window_push pick the same one next); a mismatch is repaired by a device copy inside genie_da_stage1
            _lib.check(self.lib.genie_set_slot(self.ctx, self._next_window_slot()), "genie_set_slot")
            try:
                _lib.check(self.lib.genie_embed_window_split(*common, self._ws_ptr, _stream()), "genie_embed_window_split")
            finally:
                _lib.check(self.lib.genie_set_slot(self.ctx, self.PLAIN_SLOT), "genie_set_slot")
        else:
            _lib.check(self.lib.genie_embed_window(*common, _stream()), "genie_embed_window")
        return Slice, Mask

    def export(self, which):
        """Parity/debug: de-padded copy of a workspace intermediate: 0 = c [P,30] (node-local layer-2 terms),
        1 = wu [P,15], 2 = wv [P,15] (u / v projected through the neighbour-mean columns of l2_t1_2 / l2_t2_2)."""
        cols = {0: 30, 1: 15, 2: 15}[which]
        out = torch.empty((self.n_prod, cols), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.genie_ws_export(self.ctx, which, self._ws_ptr, _ptr(out), _stream()), "genie_ws_export")
        return out
