"""Host-side graph construction for the station x source-grid product graph.

Mirrors the *layout contract* of the reference's one-time graph setup
(`/root/reference/Code/process_utils.py:701-742`, same code in `train_GENIE_model.py:1140-1149`):

* base kNN graphs `A_sta_sta [2, S*ks]`, `A_src_src [2, G*kp]` with row 0 = neighbour (message
  source j), row 1 = centre (message target i), grouped by centre, self loops removed
  (`process_utils.py:718-719`: `remove_self_loops(knn(x/1000, x/1000, k+1).flip(0))`);
* product-graph node id `p = g*S + s`;
* `A_prod_sta_sta = A_sta_sta.repeat(1, G) + S*arange(G).repeat_interleave(S*ks)`  (:720)
* `A_prod_src_src = S*A_src_src.repeat(1, S) + arange(S).repeat_interleave(G*kp)`  (:721)
* `A_src_in_prod  = [arange(P); arange(G).repeat_interleave(S)]`                     (:722)
* `A_src_in_sta   = [tile(arange(S), G); arange(G).repeat(S)]`  (`process_continuous_days.py:629`)

The HIP path never consumes the `[2, E]` product edge lists: `base_tables_from_product` recovers the
dense neighbour tables `sta_nbr[S, ks]`, `src_nbr[G, kp]` (int32) and verifies the Cartesian structure.
"""
import numpy as np
import torch
from scipy.spatial import cKDTree


def knn_graph(points, k):
    """Exact kNN graph of a point set with itself, self excluded.

    Returns int64 ndarray [2, N*k]: row 0 = neighbour j, row 1 = centre i, grouped by i
    (the layout of `remove_self_loops(knn(x, x, k+1).flip(0))`, process_utils.py:718).
    `points` are used as given (the reference passes coordinates in km).
    """
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    k = int(min(k, n - 1))
    _, idx = cKDTree(pts).query(pts, k=k + 1)
    idx = np.asarray(idx).reshape(n, k + 1)
    out = np.empty((n, k), dtype=np.int64)
    for i in range(n):  # drop self (robust to the self not being returned first on exact ties)
        row = idx[i]
        row = row[row != i]
        out[i] = row[:k]
    centre = np.repeat(np.arange(n, dtype=np.int64), k)
    return np.stack([out.reshape(-1), centre], axis=0)


def k_sta_effective(k_sta_edges, n_sta):
    """`k_sta_edges = np.minimum(k_sta_edges, len(ind_use) - 2)` (process_utils.py:712)."""
    return int(min(k_sta_edges, n_sta - 2))


def cartesian_product_edges(A_sta_sta, A_src_src, n_sta, n_grid, device="cpu"):
    """Explicit product-graph edge lists exactly as the reference builds them (process_utils.py:720-722).

    Only for small/medium sizes (E = P*(ks+kp) int64 pairs); the HIP path does not need them.
    """
    A_sta_sta = torch.as_tensor(A_sta_sta, dtype=torch.long, device=device)
    A_src_src = torch.as_tensor(A_src_src, dtype=torch.long, device=device)
    e_sta = A_sta_sta.shape[1]
    e_src = A_src_src.shape[1]
    ar = torch.arange
    A_prod_sta_sta = (A_sta_sta.repeat(1, n_grid)
                      + n_sta * ar(n_grid, device=device).repeat_interleave(e_sta).view(1, -1)).contiguous()
    A_prod_src_src = (n_sta * A_src_src.repeat(1, n_sta)
                      + ar(n_sta, device=device).repeat_interleave(e_src).view(1, -1)).contiguous()
    A_src_in_prod = torch.cat((ar(n_sta * n_grid, device=device).view(1, -1),
                               ar(n_grid, device=device).repeat_interleave(n_sta).view(1, -1)), dim=0).contiguous()
    A_src_in_sta = torch.cat((ar(n_sta, device=device).repeat(n_grid).view(1, -1),
                              ar(n_grid, device=device).repeat_interleave(n_sta).view(1, -1)), dim=0).contiguous()
    return A_prod_sta_sta, A_prod_src_src, A_src_in_prod, A_src_in_sta


class GraphEdges(object):
    """Duck-typed stand-in for PyG `Data(x=..., edge_index=...)` (process_continuous_days.py:631-632)."""

    def __init__(self, x=None, edge_index=None):
        self.x = x
        self.edge_index = edge_index

    def to(self, device):
        if self.x is not None:
            self.x = self.x.to(device)
        if self.edge_index is not None:
            self.edge_index = self.edge_index.to(device)
        return self


def neighbour_table(edge_index, n_nodes):
    """[2, n*k] edge list (row0 = j, row1 = i, every node exactly k in-edges) -> int32 table [n, k].

    Order of neighbours within a row follows the edge order (stable), so means are summed in the
    reference's edge order. Raises ValueError if in-degrees are not uniform.
    """
    ei = torch.as_tensor(edge_index).long().cpu()
    if ei.numel() == 0:
        return torch.zeros((n_nodes, 0), dtype=torch.int32)
    j, i = ei[0], ei[1]
    deg = torch.bincount(i, minlength=n_nodes)
    k = int(deg.max().item())
    if int(deg.min().item()) != k:
        raise ValueError("neighbour_table: non-uniform in-degree (min %d, max %d); use the CSR path"
                         % (int(deg.min().item()), k))
    order = torch.sort(i, stable=True)[1]
    return j[order].view(n_nodes, k).to(torch.int32).contiguous()


def base_tables_from_product(A_in_sta, A_in_src, n_sta, n_grid, check=True, defer=False):
    """Recover base tables from the reference's product edge lists and verify the Cartesian structure.

    A_in_sta [2, P*ks], A_in_src [2, P*kp] as built at process_utils.py:720-721. Returns
    (sta_nbr int32 [S, ks], src_nbr int32 [G, kp]). Raises ValueError when the lists are not the
    full Cartesian product of two uniform-degree base graphs (e.g. `use_subgraph: True`).
    `defer` (GPU lists only): nothing is read back; returns (sta_nbr, src_nbr, verdict) with `verdict` a bool GPU tensor [3]
    (not Cartesian, non-uniform station degree, non-uniform source degree) the caller reads when it next synchronises anyway --
    the tables are cut from the lists' first blocks either way, so a caller may build on them and discard the work if the verdict
    turns out bad.
    """
    A_in_sta = torch.as_tensor(A_in_sta)
    A_in_src = torch.as_tensor(A_in_src)
    S, G = int(n_sta), int(n_grid)
    if A_in_sta.shape[1] % G != 0 or A_in_src.shape[1] % S != 0:
        raise ValueError("product edge lists are not a multiple of (n_grid, n_sta): not Cartesian")
    e_sta = A_in_sta.shape[1] // G
    e_src = A_in_src.shape[1] // S
    if A_in_sta.is_cuda and A_in_src.is_cuda and check and e_sta > 0 and e_src > 0:
        return _base_tables_from_product_device(A_in_sta, A_in_src, S, G, e_sta, e_src, defer)
    if defer:
        raise ValueError("base_tables_from_product(defer=True) takes edge lists resident on the GPU")
    if e_sta == 0 or e_src == 0:
        # a base graph without any edge (found by the randomized sweep: 3 stations whose few edges were all dropped): no uniform-degree
        # table to cut -- the caller's general (CSR) path takes it
        raise ValueError("neighbour_table: a base graph has no edges; use the CSR path")
    base_sta = A_in_sta[:, :e_sta]
    if int(base_sta.max().item()) >= S:
        raise ValueError("first block of A_in_sta leaves source node 0: not Cartesian")
    # A_in_src block for station 0: ids are S*g + 0
    blk = A_in_src[:, :e_src]
    if int((blk % S).abs().max().item()) != 0:
        raise ValueError("first block of A_in_src is not station 0: not Cartesian")
    base_src = torch.div(blk, S, rounding_mode="floor")
    if check:
        dev = A_in_sta.device
        off = S * torch.arange(G, device=dev).repeat_interleave(e_sta).view(1, -1)
        if not torch.equal(A_in_sta, base_sta.repeat(1, G) + off):
            raise ValueError("A_in_sta is not A_sta_sta (x) I_G: not Cartesian")
        off = torch.arange(S, device=dev).repeat_interleave(e_src).view(1, -1)
        if not torch.equal(A_in_src, S * base_src.repeat(1, S) + off):
            raise ValueError("A_in_src is not I_S (x) A_src_src: not Cartesian")
    return neighbour_table(base_sta, S), neighbour_table(base_src, G)


def _base_tables_from_product_device(A_in_sta, A_in_src, S, G, e_sta, e_src, defer=False):
    """`base_tables_from_product` for edge lists resident on the GPU (the training call convention hands `forward` new lists per
    sample, train_GENIE_model.py:1722-1786): the Cartesian structure is verified by one pass of `genie_product_check` over the lists
    (no materialised copy), the base tables are cut from their first blocks on the device, and every verdict is read back with ONE
    synchronisation. Returns int32 GPU tables."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    dev = A_in_sta.device
    A1 = A_in_sta if (A_in_sta.dtype == torch.int64 and A_in_sta.is_contiguous()) else A_in_sta.long().contiguous()
    A2 = A_in_src if (A_in_src.dtype == torch.int64 and A_in_src.is_contiguous()) else A_in_src.long().contiguous()
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.genie_product_check(ctypes.c_void_p(A1.data_ptr()), int(A1.shape[1]), ctypes.c_void_p(A2.data_ptr()), int(A2.shape[1]),
                                           S, G, ctypes.c_void_p(flags.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "genie_product_check")
    base_sta = A1[:, :e_sta]
    base_src = torch.div(A2[:, :e_src], S, rounding_mode="floor")

    def table(base, n, e):
        # uniform in-degree k = e / n, in-edges of a node in edge order (neighbour_table's contract); ids clamped so that a list the
        # check above rejects cannot index out of range before the verdict is read
        k = e // n
        i = base[1].clamp(0, n - 1)
        # (scatter_add_ on a size known here: torch.bincount reads the largest index back)
        deg = torch.zeros(n, dtype=torch.int64, device=dev).scatter_add_(0, i, torch.ones_like(i))
        deg_bad = (deg != k).any() if k * n == e else torch.ones((), dtype=torch.bool, device=dev)
        order = torch.sort(i, stable=True)[1]
        # (neighbour ids clamped as well: with `defer` a context is built on this table before the verdict is read, and a list whose first
        # block leaves its node range -- which the check flags -- must not make a kernel read out of bounds meanwhile)
        return base[0][order][: n * k].clamp(0, n - 1).view(n, k).to(torch.int32).contiguous(), deg_bad

    sta_tab, bad1 = table(base_sta, S, e_sta)
    src_tab, bad2 = table(base_src, G, e_src)
    verdict = torch.stack((flags[0] != 0, bad1, bad2))
    if defer:
        return sta_tab, src_tab, verdict
    verdict = verdict.tolist()          # the one synchronisation
    if verdict[0]:
        which = int(flags.item())
        raise ValueError("%s: not Cartesian" % ("A_in_sta is not A_sta_sta (x) I_G" if which & 1 else "A_in_src is not I_S (x) A_src_src"))
    if verdict[1] or verdict[2]:
        raise ValueError("neighbour_table: non-uniform in-degree; use the CSR path")
    return sta_tab, src_tab


def subgraph_product_edges(A_sta_sta, A_src_src, A_src_in_sta):
    """Edge lists of the IRREGULAR product graph of `use_subgraph: True` (process_utils.py:744-849).

    A_src_in_sta [2, N] lists the product nodes as (station, source) pairs, sorted by (source, station)
    (process_utils.py:790-794). Product node n is connected
      * in A_prod_sta_sta to node m when both have the same source node and (sta_m -> sta_n) is an edge of A_sta_sta
        (the station graph induced on the station set of that source node, process_utils.py:824-826);
      * in A_prod_src_src to node m when both have the same station and (src_m -> src_n) is an edge of A_src_src
        (the source graph induced on the source set of that station, :828-839).
    Returns (A_prod_sta_sta [2,E1], A_prod_src_src [2,E2], A_src_in_prod [2,N]) with row 0 = neighbour j, row 1 = centre i,
    in-edges grouped by centre."""
    A_sta = np.asarray(A_sta_sta, dtype=np.int64)
    A_src = np.asarray(A_src_src, dtype=np.int64)
    pairs = np.asarray(A_src_in_sta, dtype=np.int64)
    sta, src = pairs[0], pairs[1]
    N = sta.size
    if np.any(np.diff(src) < 0) or np.any((np.diff(src) == 0) & (np.diff(sta) <= 0)):
        raise ValueError("A_src_in_sta must be sorted by (source, station) without duplicates")
    n_sta_all = int(max(sta.max(initial=0), A_sta.max(initial=0))) + 1
    keys = src * n_sta_all + sta                      # increasing: the node id of a (station, source) pair is its rank

    def induced(A, centre_base, other_is_station):
        """for every product node n and every base in-edge j -> centre_base[n]: the product node (j paired with n's other
        coordinate), if it exists; in-edges of a node keep the base graph's edge order."""
        order = np.argsort(A[1], kind="stable")
        tgt, nb = A[1][order], A[0][order]
        n_base = int(max(centre_base.max(initial=0), tgt.max(initial=-1))) + 1
        ptr = np.zeros(n_base + 1, dtype=np.int64)
        np.add.at(ptr, tgt + 1, 1)
        ptr = np.cumsum(ptr)
        deg = ptr[centre_base + 1] - ptr[centre_base]
        centre = np.repeat(np.arange(N), deg)
        first = np.repeat(ptr[centre_base], deg)
        within = np.arange(centre.size) - np.repeat(np.cumsum(deg) - deg, deg)
        j = nb[first + within]
        key = (src[centre] * n_sta_all + j) if other_is_station else (j * n_sta_all + sta[centre])
        pos = np.searchsorted(keys, key)
        ok = (pos < N) & (keys[np.minimum(pos, N - 1)] == key)
        return torch.from_numpy(np.stack((pos[ok], centre[ok])))

    A1 = induced(A_sta, sta, True)
    A2 = induced(A_src, src, False)
    A_src_in_prod = torch.stack((torch.arange(N), torch.from_numpy(src.copy())), dim=0)
    return A1, A2, A_src_in_prod


def time_pointers(trv_out, max_t=None, dt=1.0, k=10, win=10.0):
    """Per-station time-bin -> k nearest product nodes tables of the association heads, as
    `assemble_time_pointers_for_stations` builds them (`/root/reference/Code/utils.py:602-622`, called at
    `train_GENIE_model.py:1364`, `process_continuous_days.py:620`).

    trv_out [G, S, 2]: theoretical P / S travel time per (source node, station). For every station i and every time step t of
    `dt_partition = arange(-win, win + max_t + dt, dt)` the k source nodes whose travel time to i is nearest t (nearest
    first), as product-node ids `g * S + i`. Returns (edges_p, edges_s, dt_partition): int64 [S * len(dt_partition) * k],
    laid out [station][time step][k] (what `LocalSliceLgCollapse` indexes with `ipick * l_dt * k + t_index * k + arange(k)`,
    module.py:635-637)."""
    trv = np.asarray(trv_out)
    n_src, n_sta = trv.shape[0], trv.shape[1]
    if max_t is None:
        max_t = trv.max()
    dt_partition = np.arange(-win, win + max_t + dt, dt)
    k = int(min(k, n_src))
    # The reference ranks ALL source nodes per (station, time step) with a stable argsort of |trv - t| and keeps k: ties go to
    # the lower node id. Same result from the 2k candidates around t's insertion point in the station's sorted travel times,
    # ranked by (distance, node id), as long as no node outside that window is as near as the k-th candidate (a tie the window
    # cannot see); the rare rows where one is are ranked in full like the reference does.
    n_t = len(dt_partition)
    span = np.arange(-k, k)[None, :]
    out = []
    for ph in (0, 1):
        e = np.empty((n_sta, n_t, k), dtype=np.int64)
        for i in range(n_sta):
            col = trv[:, i, ph].astype(np.float64)
            order = np.argsort(col, kind="stable")
            pos = np.searchsorted(col[order], dt_partition, side="left")[:, None] + span               # [n_t, 2k] sorted positions
            ok = (pos >= 0) & (pos < n_src)
            cand = order[np.clip(pos, 0, n_src - 1)]                                                    # node ids
            d = np.where(ok, np.abs(col[cand] - dt_partition[:, None]), np.inf)
            by_id = np.argsort(np.where(ok, cand, n_src), axis=1, kind="stable")
            d_id, cand_id = np.take_along_axis(d, by_id, 1), np.take_along_axis(cand, by_id, 1)
            by_d = np.argsort(d_id, axis=1, kind="stable")[:, :k]
            e[i] = np.take_along_axis(cand_id, by_d, 1) * n_sta + i
            d_k = np.take_along_axis(d_id, by_d[:, -1:], 1)[:, 0]
            lo, hi = pos[:, 0] - 1, pos[:, -1] + 1
            sv = col[order]
            d_out = np.minimum(np.where(lo >= 0, np.abs(sv[np.clip(lo, 0, n_src - 1)] - dt_partition), np.inf),
                               np.where(hi < n_src, np.abs(sv[np.clip(hi, 0, n_src - 1)] - dt_partition), np.inf))
            for r in np.nonzero(d_out <= d_k)[0]:
                e[i, r] = np.argsort(np.abs(col - dt_partition[r]), kind="stable")[:k] * n_sta + i
        out.append(e.reshape(-1))
    return out[0], out[1], dt_partition
