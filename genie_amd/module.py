"""Host-side mirror of the reference model interface for the hot path.

Class / attribute / parameter names follow `/root/reference/Code/module.py` so that `state_dict()` keys
(158 tensors) and the `forward*` / `set_adjacencies` signatures are drop-in compatible:

    GCN_Detection_Network_extended            module.py:882-1020 (live variant, use_updated_model_definition=False)
      .DataAggregation        DataAggregation          module.py:52-98     -> HIP (libgenie_hip)
      .Bipartite_ReadIn       BipartiteGraphOperator   module.py:214-229   -> HIP
      .SpatialAggregation1..3 SpatialAggregation       module.py:231-249   -> HIP
      .SpatialDirect / .TemporalAttention / .SpatialAttention               module.py:251-331   -> HIP (read-out kernels)
      .BipartiteGraphReadOutOperator / .DataAggregationAssociationPhase / .LocalSliceLgCollapseP/S / .Arrivals   -> HIP

The sub-module classes hold PARAMETERS only (names and shapes of the reference, so checkpoints load strictly): none of them has a
PyTorch `forward` here. Every computation goes through `libgenie_hip.so` from the model's `forward*` methods and raises if the
library or a GPU is missing (no CPU / eager fallback). The plain-PyTorch restatements of the heads that the tests compare the
kernels with live in `tests/restatements.py`.
"""
import math

import numpy as np
import torch
from torch import nn

from . import dist as _dist
from . import engine as _engine
from . import graph as _graph

SCALE_REL = 30000.0         # config.yaml:73
KERNEL_SIG_T = 3.0          # train_config.yaml:17
SCALE_T = 3.0 * KERNEL_SIG_T  # module.py:40
EPS = 5.0 * KERNEL_SIG_T      # module.py:41


def _path_param_dict(net):
    """state_dict-name -> Parameter for the modules that run in HIP."""
    out = {}
    for mod_name in ("DataAggregation", "Bipartite_ReadIn", "SpatialAggregation1", "SpatialAggregation2",
                     "SpatialAggregation3", "SpatialDirect", "TemporalAttention", "SpatialAttention",
                     "BipartiteGraphReadOutOperator", "DataAggregationAssociationPhase", "LocalSliceLgCollapseP",
                     "LocalSliceLgCollapseS", "Arrivals"):
        mod = getattr(net, mod_name)
        for n, p in mod.named_parameters():
            out[mod_name + "." + n] = p
    return out


def _train_path_params():
    """state_dict names of every parameter the HIP training step differentiates (the whole `forward_fixed_source` path): the
    dead layers of the reference (DataAggregation.l1_t1_1 / l1_t2_1, SpatialAttention.param_vector / f_direct) get no gradient
    there either."""
    names = ["DataAggregation.%s.%s" % (l, k) for l in ("init_trns", "l1_t1_2", "l1_t2_2", "l2_t1_1", "l2_t2_1", "l2_t1_2", "l2_t2_2")
             for k in ("weight", "bias")]
    names += ["DataAggregation.%s.weight" % a for a in ("activate", "activate11", "activate12", "activate1", "activate21", "activate22", "activate2")]
    names += ["Bipartite_ReadIn.%s" % n for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "activate1.weight", "activate2.weight")]
    for k in (1, 2, 3):
        names += ["SpatialAggregation%d.%s" % (k, n) for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fglobal.weight",
                                                               "fglobal.bias", "activate1.weight", "activate2.weight", "activate3.weight")]
    names += ["SpatialDirect.f_direct.weight", "SpatialDirect.f_direct.bias", "SpatialDirect.activate.weight"]
    names += ["TemporalAttention.%s.%s" % (l, k) for l in ("temporal_query_1", "temporal_query_2", "f_context_1", "f_context_2",
                                                             "f_values_1", "f_values_2", "proj_1", "proj_2") for k in ("weight", "bias")]
    names += ["TemporalAttention.activate%d.weight" % k for k in (1, 2, 3, 4, 5)]
    names += ["SpatialAttention.%s.%s" % (l, k) for l in ("f_queries", "f_context", "f_values", "proj") for k in ("weight", "bias")]
    names += ["SpatialAttention.activate1.weight", "SpatialAttention.activate2.weight"]
    return tuple(names)


TRAIN_PATH_PARAMS = _train_path_params()


class _PathTrain(torch.autograd.Function):
    """A training step of the whole `forward_fixed_source` path in HIP, both directions: forward =
    genie_da_train_fwd (stage kernels, pre-activations kept) + genie_tail_train_fwd (the inference tail); backward =
    genie_train_bwd (tail passes, then the three P-sized passes, weight gradients reduced in a fixed order). Outputs
    (y, x, x_spatial, y_latent, x_latent): x_spatial / y_latent are differentiable too, for the association heads of the 4-output
    `forward` that consume them. `params` (order TRAIN_PATH_PARAMS) are listed so that autograd routes their gradients; their
    values are read from the library's weight mirror (synchronised by the caller)."""

    @staticmethod
    def forward(ctx, Slice, Mask, edge_attr, pos, x_query, knn, t_query, hip, want_latents, n_src_rows, *params):
        # the last n_src_rows query rows are the source queries of the 4-output forward (module.py:981): only their
        # SpatialAttention output (`x_src`) is used
        y, x, xs, ylat, xl, save, tsave = hip.path_train_fwd(Slice, Mask, edge_attr, pos, x_query, knn, t_query,
                                                              want_x_latent=want_latents, want_y_latent=want_latents)
        nq = x_query.shape[0] - n_src_rows
        ctx.hip, ctx.n_src_rows = hip, n_src_rows
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(Slice, Mask, edge_attr, pos, x_query, knn, t_query, save, tsave)
        ctx.set_materialize_grads(False)
        xs = xs.clone()
        if n_src_rows:
            x_src = hip.spatial_attention(xs, pos, x_query[nq:], knn[nq:], t_query)
            x = x[:nq].contiguous()
        else:
            x_src = xs.new_zeros(0)
        if not want_latents:
            ylat, xl = xs.new_zeros(0), xs.new_zeros(0)
        ctx.mark_non_differentiable(xl)
        return y, x, xs, ylat, xl, x_src

    @staticmethod
    def backward(ctx, d_y, d_x, d_xs, d_ylat, _d_xl, d_xsrc):
        Slice, Mask, edge_attr, pos, x_query, knn, t_query, save, tsave = ctx.saved_tensors
        hp, ns = ctx.hip, ctx.n_src_rows
        T, nq_all, dev = t_query.numel(), x_query.shape[0], Slice.device
        d_y = d_y if d_y is not None else torch.zeros((hp.n_grid, T), dtype=torch.float32, device=dev)
        d_xa = torch.zeros((nq_all, T), dtype=torch.float32, device=dev)
        if d_x is not None:
            d_xa[:nq_all - ns] = d_x.reshape(nq_all - ns, T)
        d_qlat = None
        if ns and d_xsrc is not None:
            d_qlat = torch.zeros((nq_all, 30), dtype=torch.float32, device=dev)
            d_qlat[nq_all - ns:] = d_xsrc
        g = hp.path_train_bwd(Slice, Mask, edge_attr, pos, x_query, knn, t_query, save, tsave, d_y, d_xa, d_xs, d_ylat, d_qlat)
        return (None,) * 10 + tuple(_join_variant_columns(g, n, s) for n, s in zip(TRAIN_PATH_PARAMS, ctx.shapes))


TRAIN_ASSOC_PARAMS = tuple(
    ["BipartiteGraphReadOutOperator.%s" % n for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "activate1.weight", "activate2.weight")]
    + ["DataAggregationAssociationPhase.%s.%s" % (l, k) for l in ("init_trns", "l1_t1_1", "l1_t2_1", "l1_t1_2", "l1_t2_2", "l2_t1_1", "l2_t2_1",
                                                                 "l2_t1_2", "l2_t2_2") for k in ("weight", "bias")]
    + ["DataAggregationAssociationPhase.%s.weight" % a for a in ("activate", "activate11", "activate12", "activate1", "activate21", "activate22",
                                                                "activate2")])


class _AssocTrain(torch.autograd.Function):
    """The P-sized association heads of a training step (BipartiteGraphReadOutOperator + DataAggregationAssociationPhase,
    module.py:986-990) in HIP in both directions: forward = genie_assoc_train_fwd (the inference kernels, pre-activations kept),
    backward = genie_assoc_train_bwd. Differentiable inputs: y_latent and the parameters (order TRAIN_ASSOC_PARAMS; values read from
    the library's weight mirror); x_latent arrives detached (module.py:990), the masks carry no gradient."""

    @staticmethod
    def forward(ctx, y_latent, mask_src, x_latent, Mask, edge_attr, hip, *params):
        s, asave = hip.assoc_train_fwd(y_latent, mask_src, x_latent, Mask, edge_attr)
        ctx.hip = hip
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(y_latent, mask_src, x_latent, Mask, edge_attr, asave)
        return s

    @staticmethod
    def backward(ctx, d_s):
        y_latent, mask_src, x_latent, Mask, edge_attr, asave = ctx.saved_tensors
        d_ylat, g = ctx.hip.assoc_train_bwd(y_latent, mask_src, x_latent, Mask, edge_attr, asave, d_s.contiguous())
        return (d_ylat, None, None, None, None, None) + tuple(_join_variant_columns(g, n, sh) for n, sh in zip(TRAIN_ASSOC_PARAMS, ctx.shapes))


TRAIN_LSLC_PARAMS = tuple("LocalSliceLgCollapse%s.%s" % (h, n) for h in ("P", "S")
                           for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "activate1.weight", "activate2.weight"))


class _LslcTrain(torch.autograd.Function):
    """LocalSliceLgCollapse P and S (module.py:610-659, :991-992) of a training step in HIP in both directions: forward = the
    inference kernel (genie_lslc_fwd), backward = genie_lslc_bwd (forward recomputed per tile) + genie_seg_rows (the gathered rows'
    gradients summed per product node in a fixed order). Differentiable inputs: the association embedding s [P, 30] and the two
    heads' parameters (order TRAIN_LSLC_PARAMS)."""

    @staticmethod
    def forward(ctx, s, a_edges_p, a_edges_s, dt_partition, tpick, ipick32, phase_label, tlatent, eps, hip, *params):
        arv_p = hip.lslc_fwd(0, s, a_edges_p, dt_partition, tpick, ipick32, phase_label, tlatent, 0, eps)
        arv_s = hip.lslc_fwd(1, s, a_edges_s, dt_partition, tpick, ipick32, phase_label, tlatent, 1, eps)
        ctx.hip, ctx.eps, ctx.dt_partition = hip, eps, dt_partition
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(s, a_edges_p, a_edges_s, tpick, ipick32, phase_label, tlatent)
        ctx.set_materialize_grads(False)
        return arv_p, arv_s

    @staticmethod
    def backward(ctx, d_p, d_s):
        s, a_edges_p, a_edges_s, tpick, ipick32, phase_label, tlatent = ctx.saved_tensors
        ds_rows, g = ctx.hip.lslc_bwd(s, (a_edges_p, a_edges_s), ctx.dt_partition, tpick, ipick32, phase_label, tlatent, ctx.eps,
                                      d_p.contiguous() if d_p is not None else None, d_s.contiguous() if d_s is not None else None)
        return (ds_rows,) + (None,) * 9 + tuple(g[n].view(sh) for n, sh in zip(TRAIN_LSLC_PARAMS, ctx.shapes))


TRAIN_ARR_PARAMS = tuple("Arrivals.%s" % n for n in (
    "f_arrival_query_1.weight", "f_arrival_query_1.bias", "f_arrival_query_2.weight", "f_arrival_query_2.bias",
    "f_src_context_1.weight", "f_src_context_1.bias", "f_src_context_2.weight", "f_src_context_2.bias",
    "f_values_1.weight", "f_values_1.bias", "f_values_2.weight", "f_values_2.bias", "proj_1.weight", "proj_1.bias",
    "proj_2.weight", "proj_2.bias", "activate1.weight", "activate2.weight", "activate3.weight", "activate4.weight"))


class _ArrivalsTrain(torch.autograd.Function):
    """StationSourceAttentionMergedPhases (module.py:662-775, :993) of a training step in HIP in both directions: forward = the
    inference kernels keeping the per-target softmax state (genie_arrivals_train_fwd), backward = genie_arrivals_bwd.
    Differentiable inputs: src_embed (`x_src`), arrival_p, arrival_s and the head's parameters (order TRAIN_ARR_PARAMS)."""

    @staticmethod
    def forward(ctx, stime, src_embed, trv_src, arrival_p, arrival_s, tpick, ipick, phase_label, eps, hip, *params):
        out, state = hip.arrivals_fwd(stime, src_embed, trv_src, arrival_p, arrival_s, tpick, ipick, phase_label, eps, train=True)
        ctx.hip, ctx.state = hip, state
        ctx.shapes = [tuple(p.shape) for p in params]
        return out

    @staticmethod
    def backward(ctx, d_out):
        d_src, d_p, d_s, g = ctx.hip.arrivals_bwd(ctx.state, d_out.contiguous())
        ctx.state = None
        return (None, d_src, None, d_p, d_s, None, None, None, None, None) + tuple(g[n].view(sh) for n, sh in zip(TRAIN_ARR_PARAMS, ctx.shapes))


class DataAggregation(nn.Module):
    """Parameters of reference `DataAggregation` (module.py:53-83), incl. the two layers it defines but never
    applies (`l1_t1_1`, `l1_t2_1`) so checkpoints load strictly. Compute: HIP stages 0-2."""

    def __init__(self, in_channels, out_channels, n_hidden=30, n_dim_mask=4, use_absolute_pos=False):
        super().__init__()
        if use_absolute_pos:
            in_channels = in_channels + 3 * 2          # module.py:56-57

        self.in_channels, self.out_channels, self.n_hidden = in_channels, out_channels, n_hidden
        self.activate = nn.PReLU()
        self.init_trns = nn.Linear(in_channels + n_dim_mask, n_hidden)
        self.l1_t1_1 = nn.Linear(n_hidden, n_hidden)
        self.l1_t1_2 = nn.Linear(2 * n_hidden + n_dim_mask, n_hidden)
        self.l1_t2_1 = nn.Linear(in_channels, n_hidden)
        self.l1_t2_2 = nn.Linear(2 * n_hidden + n_dim_mask, n_hidden)
        self.activate11 = nn.PReLU()
        self.activate12 = nn.PReLU()
        self.activate1 = nn.PReLU()
        self.l2_t1_1 = nn.Linear(2 * n_hidden, n_hidden)
        self.l2_t1_2 = nn.Linear(3 * n_hidden + n_dim_mask, out_channels)
        self.l2_t2_1 = nn.Linear(2 * n_hidden, n_hidden)
        self.l2_t2_2 = nn.Linear(3 * n_hidden + n_dim_mask, out_channels)
        self.activate21 = nn.PReLU()
        self.activate22 = nn.PReLU()
        self.activate2 = nn.PReLU()


class DataAggregationEdges(nn.Module):
    """Parameters of the `use_updated_model_definition: True` variant (module.py:102-174): as DataAggregation, but every
    message carries 4 edge features, so l1_t?_2 is [30, 68] and l2_t?_2 is [15, 98] (columns: node, neighbour mean, EDGE
    FEATURES, Mask). Same state_dict keys and shapes as the reference class; computed by libgenie_hip."""

    def __init__(self, in_channels, out_channels, n_hidden=30, n_dim_mask=4, ndim_proj=3, use_absolute_pos=False):
        super().__init__()
        if use_absolute_pos:
            in_channels = in_channels + 3 * 2          # module.py:106-107 (both options together)
        ne = ndim_proj + 1
        self.activate = nn.PReLU()
        self.init_trns = nn.Linear(in_channels + n_dim_mask, n_hidden)
        self.l1_t1_1 = nn.Linear(n_hidden, n_hidden)                                  # unused by forward (module.py:157)
        self.l1_t1_2 = nn.Linear(2 * n_hidden + n_dim_mask + ne, n_hidden)
        self.l1_t2_1 = nn.Linear(in_channels, n_hidden)                               # unused
        self.l1_t2_2 = nn.Linear(2 * n_hidden + n_dim_mask + ne, n_hidden)
        self.activate11 = nn.PReLU()
        self.activate12 = nn.PReLU()
        self.activate1 = nn.PReLU()
        self.l2_t1_1 = nn.Linear(2 * n_hidden, n_hidden)
        self.l2_t1_2 = nn.Linear(3 * n_hidden + n_dim_mask + ne, out_channels)
        self.l2_t2_1 = nn.Linear(2 * n_hidden, n_hidden)
        self.l2_t2_2 = nn.Linear(3 * n_hidden + n_dim_mask + ne, out_channels)
        self.activate21 = nn.PReLU()
        self.activate22 = nn.PReLU()
        self.activate2 = nn.PReLU()


def _split_abs_columns(named):
    """Registry view under use_absolute_pos: init_trns.weight [30, 14] = [Slice 4 | station pos 3 | source pos 3 | Mask 4]
    -> the usual [30, 8] plus `init_trns.weight_abs` [30, 6] (include/genie_hip.h, genie_set_absolute_pos); the association phase's
    init_trns [30, 56] = [s 15 | station pos 3 | source pos 3 | x_latent 30 | mask 5] (module.py:987-988) likewise."""
    out = dict(named)
    W = named["DataAggregation.init_trns.weight"].detach()
    out["DataAggregation.init_trns.weight"] = torch.cat((W[:, :4], W[:, 10:]), dim=1).contiguous()
    out["DataAggregation.init_trns.weight_abs"] = W[:, 4:10].contiguous()
    k = "DataAggregationAssociationPhase.init_trns.weight"
    if k in named and named[k].shape[1] == 56:
        W = named[k].detach()
        out[k] = torch.cat((W[:, :15], W[:, 21:]), dim=1).contiguous()
        out[k + "_abs"] = W[:, 15:21].contiguous()
    return out


def _split_edge_columns(named):
    """Registry view of DataAggregationEdges / DataAggregationAssociationPhaseEdges weights: the edge-feature columns of l?_t?_2 go
    to `<name>_pos`, the rest keeps the DataAggregation column layout (include/genie_hip.h, genie_set_edge_features)."""
    out = dict(named)
    for mod in ("DataAggregation", "DataAggregationAssociationPhase"):
        for lay, n_in in (("l1_t1_2", 60), ("l1_t2_2", 60), ("l2_t1_2", 90), ("l2_t2_2", 90)):
            k = "%s.%s.weight" % (mod, lay)
            if k not in named:
                continue
            W = named[k].detach()
            out[k] = torch.cat((W[:, :n_in], W[:, n_in + 4:]), dim=1).contiguous()
            out[k + "_pos"] = W[:, n_in:n_in + 4].contiguous()
    return out


def _join_variant_columns(g, name, shape):
    """Inverse of `_split_edge_columns` / `_split_abs_columns` on a dict of registry-layout gradients: the gradient of state_dict
    entry `name` in the parameter's own `shape` (the static-term columns, kept under `<name>_pos` / `<name>_abs` in the library's
    weight mirror, go back where those functions cut them out)."""
    w = g[name]
    if name.endswith(".weight") and len(shape) == 2 and w.numel() != shape[0] * shape[1]:
        extra = g.get(name + "_pos") if (name + "_pos") in g else g.get(name + "_abs")
        ne = shape[1] - w.numel() // shape[0]
        if extra is None or extra.numel() != shape[0] * ne:
            raise RuntimeError("gradient of %s: registry columns do not add up to the parameter's shape %r" % (name, tuple(shape)))
        mod, lay = name.split(".")[0], name.split(".")[1]
        at = {"l1_t1_2": 60, "l1_t2_2": 60, "l2_t1_2": 90, "l2_t2_2": 90,
              "init_trns": 15 if mod == "DataAggregationAssociationPhase" else 4}[lay]
        w, extra = w.view(shape[0], -1), extra.view(shape[0], ne)
        return torch.cat((w[:, :at], extra, w[:, at:]), dim=1)
    return w.view(shape)


class BipartiteGraphOperator(nn.Module):
    """Parameters of reference `BipartiteGraphOperator` (module.py:215-222)."""

    def __init__(self, ndim_in, ndim_out, ndim_edges=3):
        super().__init__()
        self.fc1 = nn.Linear(ndim_in + ndim_edges, ndim_in)
        self.fc2 = nn.Linear(ndim_in, ndim_out)
        self.activate1 = nn.PReLU()
        self.activate2 = nn.PReLU()


class SpatialAggregation(nn.Module):
    """Parameters of reference `SpatialAggregation` (module.py:232-241)."""

    def __init__(self, in_channels, out_channels, scale_rel=SCALE_REL, n_dim=3, n_global=5, n_hidden=30):
        super().__init__()
        self.fc1 = nn.Linear(in_channels + n_dim + n_global, n_hidden)
        self.fc2 = nn.Linear(n_hidden + in_channels, out_channels)
        self.fglobal = nn.Linear(in_channels, n_global)
        self.activate1 = nn.PReLU()
        self.activate2 = nn.PReLU()
        self.activate3 = nn.PReLU()
        self.scale_rel = scale_rel


class SpatialDirect(nn.Module):
    """Parameters of module.py:251-260 (computed by k_readout_m<0>)."""

    def __init__(self, inpt_dim, out_channels):
        super().__init__()
        self.f_direct = nn.Linear(inpt_dim, out_channels)
        self.activate = nn.PReLU()



def knn_query_edges(x_context, x_query, k):
    """Exact kNN of each query in the context set on `x/1000` (module.py:282), by one HIP kernel (genie_knn: brute force, fp64
    distances). Returns LongTensor [2, Q*k]: row 0 = context j, row 1 = query i (the `.flip(0)` layout). GPU tensors only: the
    product has no CPU path (the torch form the CPU tests use lives in tests/restatements.py)."""
    if not x_context.is_cuda:
        raise _engine._lib.GenieHipError("knn_query_edges: positions must live on the GPU (no CPU fallback)")
    idx = _engine.knn_device(x_context, x_query, k).long()
    row_q = torch.arange(x_query.shape[0], device=x_query.device).repeat_interleave(idx.shape[1])
    return torch.stack([idx.reshape(-1), row_q], dim=0)


class SpatialAttention(nn.Module):
    """Parameters of module.py:262-297 (kNN k=10 of the queries into the grid, per-edge q/c/v, segment softmax, mean over heads:
    k_ro_pre_m + k_readout_m<1>) and the cache of the query set's kNN table (genie_knn)."""

    def __init__(self, inpt_dim, out_channels, n_dim, n_latent, n_hidden=30, n_heads=5, scale_rel=SCALE_REL):
        super().__init__()
        self.param_vector = nn.Parameter(nn.init.xavier_uniform_(torch.empty(1, n_heads, n_latent)))  # unused (module.py:266,292)
        self.f_queries = nn.Linear(n_dim, n_heads * n_latent)
        self.f_context = nn.Linear(inpt_dim + n_dim, n_heads * n_latent)
        self.f_values = nn.Linear(inpt_dim + n_dim, n_heads * n_latent)
        self.f_direct = nn.Linear(inpt_dim, out_channels)                                            # unused (module.py:270)
        self.proj = nn.Linear(n_latent, out_channels)
        self.scale = math.sqrt(n_latent)
        self.n_heads, self.n_latent, self.scale_rel = n_heads, n_latent, scale_rel
        self.activate1 = nn.PReLU()
        self.activate2 = nn.PReLU()
        self._edge_cache = {}

    def query_edges(self, x_query, x_context, k):
        """kNN edges of a query set, cached for the set last used. The cache entry HOLDS the two position tensors: while it
        lives their storage cannot be freed and handed to another tensor, so (address, in-place version, shape) identifies
        the contents (a freed-and-reallocated query set of the same shape would otherwise hit with stale neighbours)."""
        key = (x_query.data_ptr(), x_query._version, tuple(x_query.shape), x_query.dtype, x_context.data_ptr(),
               x_context._version, tuple(x_context.shape), x_context.dtype, k)
        hit = self._edge_cache.get("key") == key
        if not hit:
            edges = knn_query_edges(x_context, x_query, k)
            self._edge_cache = {"key": key, "edges": edges, "refs": (x_query, x_context),
                                "table": edges[0].view(x_query.shape[0], -1).to(torch.int32).contiguous()}
        return self._edge_cache["edges"]

    def invalidate_query_cache(self):
        self._edge_cache = {}

    def query_table(self, x_query, x_context, k=10):
        """int32 [Q, k] table of the k nearest context nodes of every query (cached per query set)."""
        self.query_edges(x_query, x_context, k)
        return self._edge_cache["table"]


class TemporalAttention(nn.Module):
    """Parameters of module.py:299-331 (dense: score * value, mean over heads; no softmax; inside k_readout_m)."""

    def __init__(self, inpt_dim, out_channels, n_latent, n_hidden=30, n_heads=5, scale_t=SCALE_T):
        super().__init__()
        self.temporal_query_1 = nn.Linear(1, n_hidden)
        self.temporal_query_2 = nn.Linear(n_hidden, n_heads * n_latent)
        self.f_context_1 = nn.Linear(inpt_dim, n_hidden)
        self.f_context_2 = nn.Linear(n_hidden, n_heads * n_latent)
        self.f_values_1 = nn.Linear(inpt_dim, n_hidden)
        self.f_values_2 = nn.Linear(n_hidden, n_heads * n_latent)
        self.proj_1 = nn.Linear(n_latent, n_hidden)
        self.proj_2 = nn.Linear(n_hidden, out_channels)
        self.scale = math.sqrt(n_latent)
        self.n_heads, self.n_latent, self.scale_t = n_heads, n_latent, scale_t
        self.activate1 = nn.PReLU()
        self.activate2 = nn.PReLU()
        self.activate3 = nn.PReLU()
        self.activate4 = nn.PReLU()
        self.activate5 = nn.PReLU()


class BipartiteGraphReadOutOperator(nn.Module):
    """Association head, module.py:333-352: parameters (computed by genie_assoc_fwd / genie_assoc_train_fwd).
    `A_Lg_in_src.edge_index = [g(p); p]`: one edge per product node, so the 'add' aggregation is the identity."""

    def __init__(self, ndim_in, ndim_out, ndim_edges=3):
        super().__init__()
        self.fc1 = nn.Linear(ndim_in + ndim_edges, ndim_in)
        self.fc2 = nn.Linear(ndim_in, ndim_out)
        self.activate1 = nn.PReLU()
        self.activate2 = nn.PReLU()


class DataAggregationAssociationPhase(nn.Module):
    """Association head, module.py:356-403 (and its Edges form, :407-480, with `n_edge = 4`): parameters (computed by genie_assoc_fwd /
    genie_assoc_train_fwd)."""

    def __init__(self, in_channels, out_channels, n_hidden=30, n_dim_latent=30, n_dim_mask=5, n_edge=0):
        super().__init__()
        # n_edge = 4 under use_updated_model_definition: edge-feature columns of l?_t?_2 (module.py:416-423)
        self.activate = nn.PReLU()
        self.init_trns = nn.Linear(in_channels + n_dim_latent + n_dim_mask, n_hidden)
        self.l1_t1_1 = nn.Linear(n_hidden, n_hidden)
        self.l1_t1_2 = nn.Linear(2 * n_hidden + n_dim_mask + n_edge, n_hidden)
        self.l1_t2_1 = nn.Linear(n_hidden, n_hidden)
        self.l1_t2_2 = nn.Linear(2 * n_hidden + n_dim_mask + n_edge, n_hidden)
        self.activate11 = nn.PReLU()
        self.activate12 = nn.PReLU()
        self.activate1 = nn.PReLU()
        self.l2_t1_1 = nn.Linear(2 * n_hidden, n_hidden)
        self.l2_t1_2 = nn.Linear(3 * n_hidden + n_dim_mask + n_edge, out_channels)
        self.l2_t2_1 = nn.Linear(2 * n_hidden, n_hidden)
        self.l2_t2_2 = nn.Linear(3 * n_hidden + n_dim_mask + n_edge, out_channels)
        self.activate21 = nn.PReLU()
        self.activate22 = nn.PReLU()
        self.activate2 = nn.PReLU()


class LocalSliceLgCollapse(nn.Module):
    """Association head, module.py:610-659: per pick, the k = 10 product nodes of its station whose theoretical arrival is
    nearest the pick time (time-pointer table `A_edges`), edge MLP, mean. Parameters (computed by genie_lslc_fwd / genie_lslc_bwd)."""

    def __init__(self, ndim_in, ndim_out, n_edge=2, n_hidden=30, eps=EPS):
        super().__init__()
        self.fc1 = nn.Linear(ndim_in + n_edge, n_hidden)
        self.fc2 = nn.Linear(n_hidden, ndim_out)
        self.activate1 = nn.PReLU()
        self.activate2 = nn.PReLU()
        self.eps = eps


class StationSourceAttentionMergedPhases(nn.Module):
    """Association head, module.py:662-775 (use_sparse = True, use_neighbor_assoc_edges = False): parameters (computed by
    genie_arrivals_fwd / genie_arrivals_bwd, which never materialise the pick x pick edge list of module.py:703-718)."""

    def __init__(self, ndim_src_in, ndim_arv_in, ndim_out, n_latent, ndim_extra=1, n_heads=5, n_hidden=30, eps=EPS):
        super().__init__()
        self.f_arrival_query_1 = nn.Linear(2 * ndim_arv_in + 6, n_hidden)
        self.f_arrival_query_2 = nn.Linear(n_hidden, n_heads * n_latent)
        self.f_src_context_1 = nn.Linear(ndim_src_in + ndim_extra + 2, n_hidden)
        self.f_src_context_2 = nn.Linear(n_hidden, n_heads * n_latent)
        self.f_values_1 = nn.Linear(2 * ndim_arv_in + ndim_extra + 7, n_hidden)
        self.f_values_2 = nn.Linear(n_hidden, n_heads * n_latent)
        self.proj_1 = nn.Linear(n_latent, n_hidden)
        self.proj_2 = nn.Linear(n_hidden, ndim_out)
        self.activate1 = nn.PReLU()
        self.activate2 = nn.PReLU()
        self.activate3 = nn.PReLU()
        self.activate4 = nn.PReLU()
        self.n_heads, self.n_latent, self.eps = n_heads, n_latent, eps


class GCN_Detection_Network_extended(nn.Module):
    """Drop-in for the reference class of the same name (module.py:882-1020).

    `forward_fixed_source` (module.py:999) runs DataAggregation -> Bipartite_ReadIn -> SpatialAggregation1..3
    as ONE fused call into libgenie_hip (`genie_path_fwd`) on the graphs cached by `set_adjacencies`, then the
    read-out heads (`genie_readout_grid` / `genie_readout_query`). `use_absolute_pos=True` (config.yaml:92, +6 input
    channels) and `use_updated_model_definition=True` (config.yaml:95) are served for `forward_fixed_source` and for the
    4-output `forward` / `forward_fixed`, in eval mode and as training steps (train() mode with gradients enabled).

    Multi-GPU (genie_amd/dist.py): `process_group=` (a `torch.distributed` group, True = the default group; one process
    per GPU, RCCL over xGMI) makes the SAME calls run source-node sharded: `set_adjacencies*` build this rank's shard plan and contexts,
    `forward_fixed_source` takes the reference's arguments and returns the replicated `(y, x)` -- bit-equal to the unsharded model --,
    `embed_window` / `node_rows` / `genie_amd.apply` produce and consume only the rank's owned + halo rows. `forward` / `forward_fixed`
    (association heads) and training steps are not sharded and raise.
    """

    def __init__(self, ftrns1, ftrns2, scale_rel=SCALE_REL, use_absolute_pos=False, device="cuda",
                 use_updated_model_definition=False, use_phase_types=True, use_sign_input=False, process_group=None, shard=None,
                 shard_overlap=True, shard_halo="a2a", shard_emulate=False):
        super().__init__()
        # (rank, world, group) of a source-node-sharded model, None = one GPU holds the whole product graph
        self._shard_cfg = _dist.resolve_shard(process_group, shard)
        # shard_emulate: ONE virtual rank alone on its GPU, collectives replaced by copies of the same size -- timing only, results
        # are not the model's output (dist.Transport; bench.py --emulate-world)
        self._shard_opts = {"overlap": bool(shard_overlap), "halo": shard_halo, "emulate": bool(shard_emulate)}
        self._shard = None
        # config.yaml:93: the pick -> Slice / Mask embedding tags every feature with the sign of the negative slope of the series it is read
        # from (process_utils.py:610-614); the model itself is unchanged. Applies to the device embedding (`genie_amd.apply`, `embed_window`)
        self.use_sign_input = bool(use_sign_input)
        # config.yaml:91. False = the pick phase labels are ignored: LocalSliceLgCollapse and StationSourceAttentionMergedPhases
        # zero `phase_label` (module.py:632-633, :706-707); the callers also zero the phase-informed columns 2, 3 of Slice / Mask
        # (process_continuous_days.py:783-786, train_GENIE_model.py:1707-1709), which `genie_amd.apply` does under the same flag
        self.use_phase_types = bool(use_phase_types)
        # config.yaml:92: station / source positions appended to the inputs (module.py:916). It combines with
        # use_updated_model_definition as in the reference (module.py:103-109, :408-412, :1056, :1153)
        # config.yaml:95. True = the class of module.py:1022-1185: DataAggregationEdges on the hot path; its
        # `forward_fixed_source` is served here, its 4-output `forward` / `forward_fixed` (different association heads) are not
        self.use_updated_model_definition = bool(use_updated_model_definition)
        self.DataAggregation = (DataAggregationEdges(4, 15, use_absolute_pos=use_absolute_pos) if self.use_updated_model_definition
                                else DataAggregation(4, 15, use_absolute_pos=use_absolute_pos)).to(device)
        self.Bipartite_ReadIn = BipartiteGraphOperator(30, 15, ndim_edges=3).to(device)
        self.SpatialAggregation1 = SpatialAggregation(15, 30, scale_rel=scale_rel).to(device)
        self.SpatialAggregation2 = SpatialAggregation(30, 30, scale_rel=scale_rel).to(device)
        self.SpatialAggregation3 = SpatialAggregation(30, 30, scale_rel=scale_rel).to(device)
        self.SpatialDirect = SpatialDirect(30, 30).to(device)
        self.SpatialAttention = SpatialAttention(30, 30, 3, 15, scale_rel=scale_rel).to(device)
        self.TemporalAttention = TemporalAttention(30, 1, 15).to(device)
        self.BipartiteGraphReadOutOperator = BipartiteGraphReadOutOperator(30, 15).to(device)
        self.DataAggregationAssociationPhase = DataAggregationAssociationPhase(
            15 + (6 if use_absolute_pos else 0), 15, n_edge=4 if self.use_updated_model_definition else 0).to(device)
        self.LocalSliceLgCollapseP = LocalSliceLgCollapse(30, 15).to(device)
        self.LocalSliceLgCollapseS = LocalSliceLgCollapse(30, 15).to(device)
        self.Arrivals = StationSourceAttentionMergedPhases(30, 15, 2, 15, n_heads=3).to(device)
        self.use_absolute_pos = use_absolute_pos
        self.scale_rel = scale_rel
        self.ftrns1 = ftrns1
        self.ftrns2 = ftrns2
        self._hip = None
        self._path_params = None
        self._edge_attr = None

    def _weight_split(self):
        """state_dict view -> the library's registry view (static-term columns of the two other model definitions under their own names)."""
        if self.use_updated_model_definition and self.use_absolute_pos:
            return lambda named: _split_edge_columns(_split_abs_columns(named))
        return _split_edge_columns if self.use_updated_model_definition else (_split_abs_columns if self.use_absolute_pos else None)

    # ---- graphs --------------------------------------------------------------------------------
    def _configure_engine(self):
        """The model-level options every new HIP context gets, whichever builder made it (Cartesian or `use_subgraph`): the weight
        registry view, TemporalAttention's time scale, and the two flags of the device pick embedding (config.yaml:91, :93)."""
        self._path_params = _path_param_dict(self)
        for hp in ((self._hip,) if self._shard is None else (self._shard.local, self._shard.full)):
            hp.set_scale_t(self.TemporalAttention.scale_t)
            hp.set_phase_types(self.use_phase_types)
            hp.set_sign_input(self.use_sign_input)

    def _build_engine(self, sta_csr, src_csr, n_sta, n_grid, pos_src, pos_loc=None):
        # (positions on the GPU are ordered there: no host round trip per context, which the training call convention builds per sample)
        dev = next(self.parameters()).device
        if self._shard_cfg is not None:
            return self._build_engine_sharded(sta_csr, src_csr, n_sta, n_grid, pos_src, pos_loc, dev)
        order = _engine.sfc_order(pos_src) if pos_src is not None else None
        sta_order = _engine.sfc_order(pos_loc) if pos_loc is not None else None
        new = _engine.HipPath(n_sta, n_grid, sta_csr, src_csr, grid_order=order, scale_rel=self.scale_rel,
                              device=dev, sta_order=sta_order)
        self._retire_engine()
        self._hip = new
        self._configure_engine()
        if self.use_updated_model_definition:
            if pos_loc is None or pos_src is None:
                raise ValueError("use_updated_model_definition=True needs station and source positions")
            self._hip.set_edge_features(pos_loc.to(dev), pos_src.to(dev))                 # module.py:1102-1111
        if self.use_absolute_pos:
            if pos_loc is None or pos_src is None:
                raise ValueError("use_absolute_pos=True needs station and source positions")
            self._hip.set_absolute_pos(pos_loc.to(dev), pos_src.to(dev))                  # module.py:1007

    def _build_engine_sharded(self, sta_csr, src_csr, n_sta, n_grid, pos_src, pos_loc, dev):
        """This rank's part of the product graph (genie_amd/dist.py): the shard plan from the base source graph and the space-filling-
        curve order of the source nodes (the same arithmetic on every rank, no collective), the local context over the owned + halo
        source nodes, the replicated G-sized context. `self._hip` is the replicated one: the read-out heads run on it unchanged."""
        if pos_src is None or pos_loc is None:
            raise ValueError("a sharded model needs station and source positions (the shard plan follows the source nodes' order)")
        rank, world, group = self._shard_cfg
        rp, col = [torch.as_tensor(t).cpu().long().numpy() for t in src_csr]
        A_src = np.stack((col, np.repeat(np.arange(n_grid, dtype=np.int64), np.diff(rp))))
        to_np = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        new = _dist.ShardedPath(n_sta, n_grid, sta_csr, A_src, to_np(pos_src), world, rank, dev, group=group,
                                scale_rel=self.scale_rel, pos_sta=to_np(pos_loc), **self._shard_opts)
        self._retire_engine()
        self._shard = new
        self._hip = self._shard.full
        self._hip.side_stream = self._shard.tail_stream                    # where the pipelined windows' (y, x) are produced
        self._hip.side_streams = [self._shard.tail_stream, self._shard.comm_stream]
        self._configure_engine()
        if self.use_updated_model_definition:
            self._shard.set_edge_features(pos_loc, pos_src)
        if self.use_absolute_pos:
            self._shard.set_absolute_pos(pos_loc, pos_src)

    def _set_edge_attr(self, edge_attr, n_sta, n_grid):
        """`A_src_in_edges.x` [P, 3] registered with the context(s); a sharded model keeps the rows of its owned source nodes only (a full
        tensor is cut down once, a `ShardRows` of the owned rows is taken as it is: config 4's 1.2 GB need not exist on any rank).
        A callable `edge_attr(source_node_ids) -> [len(ids) * S, 3]` is evaluated for the source nodes this model holds, in blocks."""
        if callable(edge_attr):
            edge_attr = self._rows_from_callable(edge_attr, n_grid, own_only=True)
        if self._shard is not None:
            self._edge_attr = self._shard.local_rows(edge_attr, "A_src_in_edges.x", 3, own_only=True).contiguous()
            self._edge_attr_version = self._edge_attr._version
            self._shard.local.set_static_edge_attr(self._edge_attr)
            return
        self._edge_attr = _engine._f32(edge_attr, "A_src_in_edges.x", (n_sta * n_grid, 3))
        self._edge_attr_version = self._edge_attr._version
        self._hip.set_static_edge_attr(self._edge_attr)

    def _rows_from_callable(self, fn, n_grid, own_only=False, block=2048):
        """`fn(ids)` -> rows of the source nodes `ids` ([len, S, C] or [len * S, C]), evaluated block by block for every source node of
        an unsharded model, or for the owned (+ halo) source nodes of this rank; returned on the device (a `ShardRows` when sharded)."""
        dev = self._hip.device
        if self._shard is not None:
            p = self._shard.plan
            ids = p.own_global if own_only else p.ext_global
        else:
            ids = np.arange(n_grid, dtype=np.int64)
        parts = []
        for i in range(0, len(ids), block):
            r = fn(ids[i:i + block])
            r = r if torch.is_tensor(r) else torch.from_numpy(np.ascontiguousarray(r))
            parts.append(r.reshape(-1, r.shape[-1]).to(dev, torch.float32))
        t = torch.cat(parts, 0) if len(parts) != 1 else parts[0]
        return _dist.ShardRows(t, own_only=own_only) if self._shard is not None else t

    @property
    def is_sharded(self):
        return self._shard_cfg is not None

    @property
    def shard_plan(self):
        """The `genie_amd.dist.ShardPlan` of this rank (after `set_adjacencies*`), None for an unsharded model."""
        return self._shard.plan if self._shard is not None else None

    def _not_sharded(self, what):
        if self._shard_cfg is not None:
            raise NotImplementedError("%s is not available on a source-node-sharded model (process_group / shard): the sharded calls are "
                                      "set_adjacencies*, forward_fixed_source[_pipelined], embed_window, node_rows and genie_amd.apply" % what)

    def set_adjacencies(self, A_in_sta, A_in_src, A_src_in_edges, A_Lg_in_src, A_src_in_sta, A_src, A_edges_p,
                        A_edges_s, dt_partition, tlatent, pos_loc, pos_src, _defer_checks=False):
        """Same 12 arguments as module.py:941. The product edge lists are reduced to the base kNN graphs (the
        Cartesian structure is verified) and handed to libgenie_hip once; nothing is rebuilt per window.
        `_defer_checks` (used by `forward`, whose graphs change per training sample): with GPU lists of the full product's size the
        context is built from the lists' first blocks at once and the verdicts of the structure checks stay on the device
        (`self._pending_checks`) until `_resolve_pending_checks` reads them -- `forward` does after issuing its kernels, so the host
        prepares a sample's context while the GPU still works on the previous step instead of waiting for it to drain."""
        self._pending_checks = None
        self.A_in_sta, self.A_in_src = A_in_sta, A_in_src
        self.A_src_in_edges, self.A_Lg_in_src = A_src_in_edges, A_Lg_in_src
        self.A_src_in_sta, self.A_src = A_src_in_sta, A_src
        self.A_edges_p, self.A_edges_s = A_edges_p, A_edges_s
        self.dt_partition, self.tlatent = dt_partition, tlatent
        n_sta, n_grid = int(pos_loc.shape[0]), int(pos_src.shape[0])
        n_prod = int(A_src_in_sta.shape[1])
        cartesian = n_prod == n_sta * n_grid
        verdict = None
        defer = (_defer_checks and cartesian and torch.is_tensor(A_in_sta) and torch.is_tensor(A_in_src) and A_in_sta.is_cuda and A_in_src.is_cuda
                 and A_in_sta.shape[1] > 0 and A_in_src.shape[1] > 0 and A_in_sta.shape[1] % n_grid == 0 and A_in_src.shape[1] % n_sta == 0)
        if cartesian:
            try:
                if defer:
                    sta_nbr, src_nbr, verdict = _graph.base_tables_from_product(A_in_sta, A_in_src, n_sta, n_grid, defer=True)
                else:
                    sta_nbr, src_nbr = _graph.base_tables_from_product(A_in_sta, A_in_src, n_sta, n_grid)
            except ValueError:
                cartesian = False
        if not cartesian:
            self._not_sharded("an irregular product graph (use_subgraph)")
            return self._set_adjacencies_subgraph(A_in_sta, A_in_src, A_src_in_edges, A_src_in_sta, A_src, n_sta, n_grid,
                                                  pos_loc, pos_src)
        src_csr = _engine.csr_from_table(src_nbr)
        A_src_t = torch.as_tensor(A_src)
        kp = int(src_nbr.shape[1])
        # the reference hands over the very edge list the product was built from (process_utils.py:719-721: in-edges grouped by centre,
        # centres ascending): one comparison on the lists' own device; any other ordering of the same graph takes the CSR comparison
        literal = A_src_t.device == src_nbr.device and tuple(A_src_t.shape) == (2, n_grid * kp) and kp > 0
        if verdict is not None and not literal:
            # not even the shape of the literal form (e.g. base graphs of non-uniform degree, whose tables above are meaningless): nothing
            # to build on, take the path with the checks up front
            return self.set_adjacencies(A_in_sta, A_in_src, A_src_in_edges, A_Lg_in_src, A_src_in_sta, A_src, A_edges_p, A_edges_s,
                                        dt_partition, tlatent, pos_loc, pos_src)
        if literal:
            centre = torch.arange(n_grid, device=A_src_t.device, dtype=A_src_t.dtype).view(-1, 1)
            literal = ((A_src_t[0].view(n_grid, kp) == src_nbr) & (A_src_t[1].view(n_grid, kp) == centre)).all()
            if verdict is not None:
                verdict = torch.cat((verdict, (~literal).view(1)))           # read with the others (_resolve_pending_checks)
                literal = True
            else:
                literal = bool(literal)                                       # one read-back
        if not literal:
            src_from_A = _engine.csr_from_edges(A_src, n_grid)
            a, b = [t.cpu() for t in src_from_A], [t.cpu() for t in src_csr]
            if a[0].shape != b[0].shape or a[1].shape != b[1].shape or not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])):
                raise ValueError("A_src is not the base graph of A_in_src")
        self._build_engine(_engine.csr_from_table(sta_nbr), src_csr, n_sta, n_grid, pos_src, pos_loc)
        self._set_edge_attr(A_src_in_edges.x, n_sta, n_grid)
        self._pending_checks = verdict

    def _resolve_pending_checks(self):
        """Read the verdicts a deferred `set_adjacencies` left on the device (ONE read-back). True: the context stands. False: the lists
        are not the Cartesian product the context was built for (or A_src is not written as the product's base graph): the caller
        repeats `set_adjacencies` without deferral -- which takes the general path or raises, as it always did -- and recomputes."""
        v, self._pending_checks = getattr(self, "_pending_checks", None), None
        if v is None:
            return True
        ok = not any(v.tolist())
        # the read-back has waited for every kernel this forward issued on the stream: their device-side verdicts (a pick outside the
        # time-pointer table, a station index outside the model, inputs beyond the verified fp16 range) are in -- raise them now, the
        # next sample brings a new context and nothing would read them again
        if ok:
            self._hip.check_input_range()
        else:
            self._hip.discard_flags()
        return ok

    def _retire_engine(self):
        """The context(s) about to be replaced: called right AFTER the replacing context was constructed (its constructor waits for the
        device), so the verdicts of every call issued on the old one are in; raises what they hold."""
        old = [h for h in ((self._hip,) if self._shard is None else (self._shard.local, self._shard.full)) if h is not None]
        self._hip = self._shard = None
        for h in old:
            h.retire(synchronize=False)

    def _set_adjacencies_subgraph(self, A_in_sta, A_in_src, A_src_in_edges, A_src_in_sta, A_src, n_sta, n_grid, pos_loc, pos_src):
        """`use_subgraph: True` (config.yaml:86, process_utils.py:744-849): the product nodes are the pairs listed in
        A_src_in_sta (grouped by source node) and the edge lists are irregular: product-level CSRs, generic HIP kernels."""
        pairs = torch.as_tensor(A_src_in_sta).long().cpu()
        n_prod = int(pairs.shape[1])
        src_of = pairs[1]
        if n_prod == 0 or bool((src_of[1:] < src_of[:-1]).any()):
            raise ValueError("A_src_in_sta must list the product nodes grouped by source node (process_utils.py:790-794)")
        seg = torch.zeros(n_grid + 1, dtype=torch.int32)
        seg[1:] = torch.cumsum(torch.bincount(src_of, minlength=n_grid), 0).to(torch.int32)
        sub = {"n_prod": n_prod, "sta_csr": _engine.csr_from_edges(A_in_sta, n_prod),
               "src_csr": _engine.csr_from_edges(A_in_src, n_prod), "seg_rowptr": seg}
        order = _engine.sfc_order(pos_src.detach().cpu().numpy())
        dev = next(self.parameters()).device
        new = _engine.HipPath(n_sta, n_grid, None, _engine.csr_from_edges(A_src, n_grid), grid_order=order,
                              scale_rel=self.scale_rel, device=dev, subgraph=sub)
        self._retire_engine()
        self._hip = new
        self._hip.set_subgraph_stations(pairs[0])
        self._configure_engine()
        if self.use_updated_model_definition:       # module.py:1059-1072 on the irregular edge lists: positions per product node
            if pos_loc is None or pos_src is None:
                raise ValueError("use_updated_model_definition=True needs station and source positions")
            pl, ps = _engine._f32(pos_loc.to(dev), "pos_loc"), _engine._f32(pos_src.to(dev), "pos_src")
            self._hip.set_edge_features(pl[pairs[0].to(dev)].contiguous(), ps[pairs[1].to(dev)].contiguous())
        if self.use_absolute_pos:                   # module.py:1007 / :1056: the positions of a product node's station and source node
            if pos_loc is None or pos_src is None:
                raise ValueError("use_absolute_pos=True needs station and source positions")
            pl, ps = _engine._f32(pos_loc.to(dev), "pos_loc"), _engine._f32(pos_src.to(dev), "pos_src")
            self._hip.set_absolute_pos(pl[pairs[0].to(dev)].contiguous(), ps[pairs[1].to(dev)].contiguous())
        self._edge_attr = _engine._f32(A_src_in_edges.x, "A_src_in_edges.x", (n_prod, 3))
        self._edge_attr_version = self._edge_attr._version

    def set_adjacencies_subgraph_from_positions(self, pos_loc, pos_src, edge_attr=None, k_sta_edges=10, k_spc_edges=15,
                                                max_deg_offset=5.0, k_nearest_pairs=30, scale_deg=110e3,
                                                scale_pairwise_sta_in_src_distances=100e3):
        """`use_subgraph: True` set up entirely on the device: `extract_inputs_adjacencies_subgraph` (process_utils.py:744-849,
        its defaults) without the host loops. Base kNN graphs by `genie_knn` (:782-783), the product nodes = every source node
        paired with the stations within `scale_deg * max_deg_offset` and its `k_nearest_pairs` nearest stations (:775-794),
        the product-level CSRs of the two induced graphs (:824-839) by `genie_subgraph_csr_count / _fill`.
        pos_loc [S, 3] / pos_src [G, 3] Cartesian metres; edge_attr [N, 3] for the N product nodes, a callable
        `edge_attr(pairs) -> [N, 3]`, or None = `(pos_src[source] - pos_loc[station]) / scale_pairwise_sta_in_src_distances`
        (:811). Returns (A_sta_sta, A_src_src, A_src_in_sta) with A_src_in_sta int64 [2, N] = the product nodes as
        (station, source) pairs in node order; Slice / Mask / edge_attr rows follow that order."""
        self._not_sharded("an irregular product graph (use_subgraph)")
        dev = next(self.parameters()).device
        pos_loc, pos_src = _engine._f32(pos_loc.to(dev), "pos_loc"), _engine._f32(pos_src.to(dev), "pos_src")
        n_sta, n_grid = int(pos_loc.shape[0]), int(pos_src.shape[0])
        sta_tab, A_sta = _engine.knn_graph_device(pos_loc, _graph.k_sta_effective(k_sta_edges, n_sta))
        src_tab, A_src = _engine.knn_graph_device(pos_src, min(k_spc_edges, n_grid - 1))
        pairs = _engine.subgraph_pairs_device(pos_loc, pos_src, max_deg_offset, k_nearest_pairs, scale_deg)
        src_csr = _engine.csr_from_table(src_tab)
        sub = _engine.subgraph_csr_device(pairs, n_grid, _engine.csr_from_table(sta_tab), src_csr)
        order = _engine.sfc_order(pos_src.detach().cpu().numpy())
        new = _engine.HipPath(n_sta, n_grid, None, src_csr, grid_order=order, scale_rel=self.scale_rel, device=dev, subgraph=sub)
        self._retire_engine()
        self._hip = new
        self._hip.set_subgraph_stations(pairs[0])
        self._configure_engine()
        if self.use_updated_model_definition:
            self._hip.set_edge_features(pos_loc[pairs[0].long()].contiguous(), pos_src[pairs[1].long()].contiguous())
        if self.use_absolute_pos:
            self._hip.set_absolute_pos(pos_loc[pairs[0].long()].contiguous(), pos_src[pairs[1].long()].contiguous())
        if edge_attr is None:
            edge_attr = (pos_src[pairs[1]] - pos_loc[pairs[0]]) / float(scale_pairwise_sta_in_src_distances)
        elif callable(edge_attr):
            edge_attr = edge_attr(pairs)
        self._edge_attr = _engine._f32(edge_attr, "edge_attr", (sub["n_prod"], 3))
        self.A_src = A_src
        return A_sta, A_src, pairs

    def set_adjacencies_from_positions(self, pos_loc, pos_src, edge_attr, k_sta_edges=8, k_spc_edges=15):
        """One-time graph setup entirely on the device (process_utils.py:701-742 without its host detours): the two base kNN
        graphs `remove_self_loops(knn(x / 1000, x / 1000, k + 1).flip(0))` (:718-719, `k_sta_edges = min(k_sta_edges, n_sta - 2)`
        :712) by `genie_knn`, handed to libgenie_hip as device CSR arrays; the product edge lists of :720-721 are never built.
        pos_loc [S, 3] / pos_src [G, 3] Cartesian metres (`ftrns1(locs)`, `ftrns1(x_grid)`), edge_attr [S * G, 3]
        (`A_src_in_edges.x`, process_continuous_days.py:630)."""
        dev = next(self.parameters()).device
        pos_loc, pos_src = _engine._f32(pos_loc.to(dev), "pos_loc"), _engine._f32(pos_src.to(dev), "pos_src")
        n_sta, n_grid = int(pos_loc.shape[0]), int(pos_src.shape[0])
        sta_tab, A_sta = _engine.knn_graph_device(pos_loc, _graph.k_sta_effective(k_sta_edges, n_sta))
        src_tab, A_src = _engine.knn_graph_device(pos_src, min(k_spc_edges, n_grid - 1))
        self.A_src = A_src
        self._build_engine(_engine.csr_from_table(sta_tab), _engine.csr_from_table(src_tab), n_sta, n_grid, pos_src, pos_loc)
        self._set_edge_attr(edge_attr, n_sta, n_grid)
        return A_sta, A_src

    def set_adjacencies_base(self, A_sta_sta, A_src_src, edge_attr, pos_loc, pos_src, A_edges_p=None, A_edges_s=None,
                             dt_partition=None, tlatent=None):
        """Same effect as `set_adjacencies` from the BASE graphs only (process_utils.py:718-719), for sizes
        where the explicit product edge lists cannot be materialised (config 4: 2.3 G edges). The last four arguments are
        `set_adjacencies`' time-pointer tables and travel times (module.py:941): needed by the 4-output `forward_fixed` only."""
        n_sta, n_grid = int(pos_loc.shape[0]), int(pos_src.shape[0])
        self.A_edges_p, self.A_edges_s, self.dt_partition, self.tlatent = A_edges_p, A_edges_s, dt_partition, tlatent
        self.A_src = torch.as_tensor(A_src_src)
        self._build_engine(_engine.csr_from_edges(A_sta_sta, n_sta), _engine.csr_from_edges(A_src_src, n_grid),
                           n_sta, n_grid, pos_src, pos_loc)
        self._set_edge_attr(edge_attr, n_sta, n_grid)

    # ---- hot path ------------------------------------------------------------------------------
    def _path(self, Slice, Mask, x_temp_cuda_cart, want_x_latent=False, want_bip=False):
        if self._hip is None:
            raise RuntimeError("call set_adjacencies(...) before forward_fixed*/forward_fixed_source")
        if self._shard is not None:
            # this rank's owned + halo rows -> stage 1 | halo exchange | stage 2 -> all-gather of [G, 15] -> replicated tail
            if want_x_latent or want_bip:
                self._not_sharded("x_latent / the Bipartite output of the whole grid")
            sp = self._shard
            sp.sync_weights(self._path_params, self._weight_split())
            return sp.path_fwd(sp.local_rows(Slice, "Slice", 4), sp.local_rows(Mask, "Mask", 4), self._edge_attr,
                               x_temp_cuda_cart), None, None
        self._hip.sync_weights(self._path_params, self._weight_split())
        return self._hip.path_fwd(Slice, Mask, self._edge_attr, x_temp_cuda_cart, want_x_latent, want_bip)

    def _differentiable(self):
        return self.training and torch.is_grad_enabled()

    def _path_train(self, Slice, Mask, x_temp_cuda_cart, x_query_cart, t_query, want_latents=False, x_query_src_cart=None):
        """Training-mode `forward_fixed_source`: the whole path in HIP in both directions (`_PathTrain`). Used
        when the module is in train() mode with gradients enabled; eval / no_grad calls take the fused inference kernels.
        Returns (y, x, x_spatial, y_latent, x_latent, x_src); the latents only with `want_latents`, x_src (SpatialAttention at the
        source queries, module.py:981) only with `x_query_src_cart` (the 4-output forward)."""
        if self._hip is None:
            raise RuntimeError("call set_adjacencies(...) first")
        self._not_sharded("a training step (train() mode with gradients enabled; use eval() / no_grad)")
        hp = self._hip
        hp.sync_weights(self._path_params, self._weight_split())
        knn = self.SpatialAttention.query_table(x_query_cart, x_temp_cuda_cart, 10)
        n_src_rows = 0
        if x_query_src_cart is not None:          # the source queries ride along as extra query rows (same kernels, same backward)
            n_src_rows = int(x_query_src_cart.shape[0])
            knn = torch.cat((knn, _engine.knn_device(x_temp_cuda_cart, x_query_src_cart, 10)), 0)
            x_query_cart = torch.cat((_engine._f32(x_query_cart, "x_query"), _engine._f32(x_query_src_cart, "x_query_src")), 0)
        # normalised ONCE, before autograd saves them: the backward hands these very buffers to genie_train_bwd as raw pointers (a
        # float64 or non-contiguous Slice / Mask would otherwise give a correct forward and wrong DataAggregation gradients)
        P = hp.n_prod
        Slice, Mask = _engine._f32(Slice, "Slice", (P, 4)), _engine._f32(Mask, "Mask", (P, 4))
        return _PathTrain.apply(Slice, Mask, self._edge_attr, x_temp_cuda_cart, x_query_cart, knn, t_query, hp, bool(want_latents), n_src_rows,
                                *[self._path_params[n] for n in TRAIN_PATH_PARAMS])

    def forward_fixed_source(self, Slice, Mask, tpick, ipick, phase_label, locs_use_cart, x_temp_cuda_cart,
                             x_query_cart, t_query):
        """module.py:999-1020. `tpick`, `ipick`, `phase_label` are accepted and ignored, as in the reference."""
        if self._differentiable():
            y, x = self._path_train(Slice, Mask, x_temp_cuda_cart, x_query_cart, t_query)[:2]
            return y, x
        x_spatial, _, _ = self._path(Slice, Mask, x_temp_cuda_cart)                       # :1010-1014
        knn = self.SpatialAttention.query_table(x_query_cart, x_temp_cuda_cart, 10)        # :282 (cached per query set)
        # :1015-1018: the grid read-out (y) and the query read-out (x) side by side on two streams, joined before returning
        return self._hip.readouts_forked(x_spatial, x_temp_cuda_cart, x_query_cart, knn, t_query)

    def forward_fixed_source_pipelined(self, Slice, Mask, tpick, ipick, phase_label, locs_use_cart, x_temp_cuda_cart,
                                       x_query_cart, t_query):
        """Throughput variant of `forward_fixed_source` for loops over independent windows (the apply loop,
        process_continuous_days.py:761-810): same arithmetic and results, but the G-sized tail of the window runs on
        `self._hip.side_stream` (one of two alternating side streams) so it overlaps the next windows' P-sized kernels. Returns (y, x, done_event): consume
        y / x on that stream (`with torch.cuda.stream(net._hip.side_stream)`) or after `done_event.wait()`."""
        if self._hip is None:
            raise RuntimeError("call set_adjacencies(...) before forward_fixed*/forward_fixed_source")
        knn = self.SpatialAttention.query_table(x_query_cart, x_temp_cuda_cart, 10)
        if self._shard is not None:
            # the all-gather, the replicated tail and both read-outs of this window on the shard's tail stream (= `_hip.side_stream`),
            # under the P-sized kernels and the halo exchange of the next window
            sp, full = self._shard, self._hip
            sp.sync_weights(self._path_params, self._weight_split())
            xq, tq = _engine._f32(x_query_cart, "x_query"), _engine._f32(t_query, "t_query")
            full._crosses_to(sp.tail_stream, x_temp_cuda_cart, xq, knn, tq)
            y, x = sp.path_fwd(sp.local_rows(Slice, "Slice", 4), sp.local_rows(Mask, "Mask", 4), self._edge_attr, x_temp_cuda_cart,
                               tail=lambda xs: (full.readout_grid(xs, tq), full.readout_query(xs, x_temp_cuda_cart, xq, knn, tq)),
                               pipelined=True)
            full._crosses_to(torch.cuda.current_stream(full.device), y, x)
            return y, x, sp._tail_done
        self._hip.sync_weights(self._path_params, self._weight_split())
        return self._hip.forward_pipelined(Slice, Mask, self._edge_attr, x_temp_cuda_cart, x_query_cart, knn, t_query)

    def node_rows(self, table, cols=None):
        """A per-product-node table of the caller ([G, S, C] or [G * S, C]; numpy or tensor; e.g. the travel times `x_grids_trv[i]` the
        embedding gathers, process_utils.py:599-608) as a float32 tensor on the model's device, in the form `embed_window` and the
        forward calls take: all rows for an unsharded model; for a sharded one a `ShardRows` of this rank's owned + halo rows only, cut out
        where the table lives (a host table never reaches the device in full)."""
        dev = self._hip.device
        if isinstance(table, _dist.ShardRows):
            return table
        if callable(table):          # table(source_node_ids) -> their rows: evaluated for the source nodes this model / rank holds
            n_grid = self._shard.n_grid if self._shard is not None else self._hip.n_grid
            return self._rows_from_callable(table, n_grid)
        t = table if torch.is_tensor(table) else torch.from_numpy(np.ascontiguousarray(table))
        if t.dim() == 3:
            t = t.reshape(t.shape[0] * t.shape[1], t.shape[2])
        if cols is not None and t.shape[1] != cols:
            raise ValueError("node_rows: expected %d columns, got %d" % (cols, t.shape[1]))
        if self._shard is None:
            return t.to(dev, torch.float32).contiguous()
        return _dist.ShardRows(self._shard.local_rows(t, "table", int(t.shape[1])).contiguous())

    def embed_window(self, pick_t, pick_sta, pick_phase, t0, max_t, kernel_sig_t, dt, trv, presplit=False):
        """(Slice, Mask) of the window starting at t0 from picks resident on the GPU (genie_embed_window = extract_input_from_data,
        process_utils.py:460-642; arguments of `engine.HipPath.embed_window`). `trv`: `node_rows(travel times)`. On a sharded model the
        rank embeds its owned + halo source nodes only and gets a pair of `ShardRows`, which the forward calls take as they are."""
        if self._hip is None:
            raise RuntimeError("call set_adjacencies(...) first")
        if self._shard is None:
            return self._hip.embed_window(pick_t, pick_sta, pick_phase, t0, max_t, kernel_sig_t, dt, trv, presplit=presplit)
        trv = self._shard.local_rows(trv, "trv", 2)
        Slice, Mask = self._shard.local.embed_window(pick_t, pick_sta, pick_phase, t0, max_t, kernel_sig_t, dt, trv, presplit=presplit)
        return _dist.ShardRows(Slice), _dist.ShardRows(Mask)

    def push_window(self, Slice, Mask):
        """Batched throughput form of `forward_fixed_source` for loops over independent windows (the apply loop,
        process_continuous_days.py:761-810): `push_window` runs the P-sized part of one window, `flush_windows` the G-sized
        tail and both read-outs of every pushed window in one set of launches (at most `net.window_batch` windows).
        Same arithmetic, bit-identical results. Returns the number of pending windows."""
        if self._hip is None:
            raise RuntimeError("call set_adjacencies(...) before forward_fixed*/forward_fixed_source")
        self._not_sharded("push_window / flush_windows (batched tails; use forward_fixed_source_pipelined)")
        self._hip.sync_weights(self._path_params, self._weight_split())
        return self._hip.window_push(Slice, Mask, self._edge_attr)

    def flush_windows(self, x_temp_cuda_cart, x_query_cart, t_query):
        """(y [n, G, T, 1], x [n, Q, T, 1], done_event) of the n pushed windows, in push order, produced on
        `self._hip.side_stream` (consume them there, or after `done_event.wait()`)."""
        knn = self.SpatialAttention.query_table(x_query_cart, x_temp_cuda_cart, 10)
        return self._hip.windows_flush(x_temp_cuda_cart, x_query_cart, knn, t_query)

    @property
    def window_batch(self):
        return self._hip.window_batch if self._hip is not None else 1

    @window_batch.setter
    def window_batch(self, n):
        if self._hip is None:
            raise RuntimeError("call set_adjacencies(...) first")
        if self._shard is not None:          # a shard's tail follows its own all-gather: one tail per window
            if int(n) != 1:
                self._not_sharded("window_batch > 1")
            return
        self._hip.set_window_batch(n)

    @property
    def pending_windows(self):
        bt = getattr(self._hip, "_bt", None) if self._hip is not None else None
        return bt["n"] if bt else 0

    def forward_fixed(self, Slice, Mask, tpick, ipick, phase_label, locs_use_cart, x_temp_cuda_cart, x_query_cart,
                      x_query_src_cart, t_query, tq_sample, trv_out_q):
        """module.py:963-997 (and :1128-1161 of the use_updated_model_definition class): (y, x, arv_p, arv_s). Every module runs
        in HIP: the shared front and read-outs, the P-sized association heads (genie_assoc_fwd; under use_updated_model_definition
        / use_absolute_pos their static per-station / per-source-node terms are added inside the same kernels), LocalSliceLgCollapse
        P / S (genie_lslc_fwd) and the arrival head (genie_arrivals_fwd). In train() mode with gradients enabled the same modules
        differentiate in HIP (`_PathTrain`, `_AssocTrain`, `_LslcTrain`, `_ArrivalsTrain`; all three model definitions on Cartesian product graphs, the default one on irregular ones too)."""
        if self._hip is None:
            raise RuntimeError("call set_adjacencies(...) before forward_fixed")
        self._not_sharded("forward / forward_fixed (the association heads)")
        if getattr(self, "A_edges_p", None) is None or getattr(self, "tlatent", None) is None:
            raise RuntimeError("forward_fixed needs the time-pointer tables and travel times (A_edges_p, A_edges_s, dt_partition, tlatent) "
                               "of set_adjacencies(...) / set_adjacencies_base(...)")
        hp = self._hip
        if not getattr(hp, "assoc_ready", False) and hp is not None:
            hp.sync_weights(self._path_params, self._weight_split())
        train = self._differentiable()
        if train:       # training step (train_GENIE_model.py:1786): the shared path in HIP in both directions
            y, x, x_spatial, y_latent, x_latent, x_src = self._path_train(Slice, Mask, x_temp_cuda_cart, x_query_cart, t_query, want_latents=True,
                                                                          x_query_src_cart=x_query_src_cart)      # :973-982
        else:
            x_spatial, x_latent, _ = self._path(Slice, Mask, x_temp_cuda_cart, want_x_latent=True)  # :973-977
            y, y_latent = hp.readout_grid_latent(x_spatial, t_query)                                 # :978-979
            knn = self.SpatialAttention.query_table(x_query_cart, x_temp_cuda_cart, 10)
            x = hp.readout_query(x_spatial, x_temp_cuda_cart, x_query_cart, knn, t_query)            # :980,982
            knn_src = _engine.knn_device(x_temp_cuda_cart, x_query_src_cart, 10)
            x_src = hp.spatial_attention(x_spatial, x_temp_cuda_cart, x_query_src_cart, knn_src, t_query)   # :981
        if not getattr(hp, "assoc_ready", False):
            raise _engine._lib.GenieHipError("the association heads' parameters are not in the HIP context (unexpected shapes)")
        mask_out = 1.0 * (y[:, :, 0].detach().max(1, keepdim=True)[0] > 0.01)                        # :985
        Maskf = _engine._f32(Mask, "Mask")
        if train:       # the kernels of genie_assoc_fwd with their pre-activations kept, backward in HIP
            s = _AssocTrain.apply(y_latent, mask_out, x_latent.detach(), Maskf, self._edge_attr, hp,
                                  *[self._path_params[n] for n in TRAIN_ASSOC_PARAMS])
        else:           # :986-990 as three P-sized HIP passes
            s = hp.assoc_fwd(y_latent, mask_out, x_latent, Maskf, self._edge_attr)
        n_src = int(x_query_src_cart.shape[0])
        if not self.use_phase_types:      # module.py:632-633, :706-707
            phase_label = phase_label * 0.0
        if len(tpick) == 0:      # no pick in the window: the reference returns two empty [n_src, 0, 1] tensors
            e = s.new_zeros((n_src, 0, 1))
            return y, x, e, e.clone()
        # :991-992 (genie_lslc_fwd); the int32 copies of the static time-pointer tables are cached with the tables
        key = (self.A_edges_p.data_ptr(), self.A_edges_s.data_ptr(), self.A_edges_p._version, self.A_edges_s._version)
        if getattr(self, "_a_edges_key", None) != key:
            self._a_edges_i32 = (self.A_edges_p.to(s.device).to(torch.int32).contiguous(),
                                 self.A_edges_s.to(s.device).to(torch.int32).contiguous())
            self._a_edges_key, self._a_edges_refs = key, (self.A_edges_p, self.A_edges_s)
        ip32, tl, eps = ipick.to(torch.int32), self.tlatent, self.LocalSliceLgCollapseP.eps
        dtp = self.dt_partition                   # in the form the lslc calls take it (a GPU tensor stays on the device: no read-back)
        dkey = (dtp.data_ptr(), dtp._version) if torch.is_tensor(dtp) else id(dtp)
        if getattr(self, "_dt_host_key", None) != dkey:
            self._dt_host, self._dt_host_key, self._dt_host_ref = _engine.time_partition(dtp), dkey, dtp
        dth = self._dt_host
        if train:
            arv_p, arv_s = _LslcTrain.apply(s, self._a_edges_i32[0], self._a_edges_i32[1], dth, _engine._f32(tpick, "tpick"),
                                            ip32, _engine._f32(phase_label, "phase_label"), _engine._f32(tl, "tlatent"), eps, hp,
                                            *[self._path_params[n] for n in TRAIN_LSLC_PARAMS])
            arv = _ArrivalsTrain.apply(tq_sample, x_src, trv_out_q, arv_p, arv_s, tpick, ipick, phase_label, self.Arrivals.eps, hp,
                                       *[self._path_params[n] for n in TRAIN_ARR_PARAMS])              # :993
        else:
            arv_p = hp.lslc_fwd(0, s, self._a_edges_i32[0], dth, tpick, ip32, phase_label, tl, 0, eps)
            arv_s = hp.lslc_fwd(1, s, self._a_edges_i32[1], dth, tpick, ip32, phase_label, tl, 1, eps)
            arv = hp.arrivals_fwd(tq_sample, x_src, trv_out_q, arv_p, arv_s, tpick, ipick, phase_label, self.Arrivals.eps)   # :993
        return y, x, arv[:, :, 0].unsqueeze(-1), arv[:, :, 1].unsqueeze(-1)                          # :995-997

    def invalidate_graph_cache(self):
        """Forget the graphs `forward` cached (next call rebuilds the HIP context) and the cached query kNN table."""
        self._fwd_key = self._fwd_refs = None
        self.SpatialAttention.invalidate_query_cache()

    def forward(self, Slice, Mask, A_in_sta, A_in_src, A_src_in_edges, A_Lg_in_src, A_src_in_sta, A_src, A_edges_p,
                A_edges_s, dt_partition, tlatent, tpick, ipick, phase_label, locs_use_cart, x_temp_cuda_cart,
                x_query_cart, x_query_src_cart, t_query, tq_sample, trv_out_q):
        """module.py:908-939: same as `forward_fixed` with the graphs passed per call (the training call convention,
        train_GENIE_model.py:1786). The HIP context is rebuilt when any tensor that defines the graphs (edge lists, product
        node list, positions) changes address, shape or in-place version; the cache holds those tensors, so an address cannot
        be recycled while it is the key. The per-call tensors that do not shape the context (edge_attr, time-pointer tables,
        tlatent) are simply taken from this call. `invalidate_graph_cache()` forces a rebuild."""
        self._not_sharded("forward / forward_fixed (the association heads)")

        def tk(t):
            return (t.data_ptr(), t._version, tuple(t.shape), t.dtype) if torch.is_tensor(t) else id(t)
        graph_tensors = (A_in_sta, A_in_src, A_src, A_src_in_sta, locs_use_cart, x_temp_cuda_cart)
        key = tuple(tk(t) for t in graph_tensors)
        adj = (A_in_sta, A_in_src, A_src_in_edges, A_Lg_in_src, A_src_in_sta, A_src, A_edges_p, A_edges_s, dt_partition, tlatent,
               locs_use_cart, x_temp_cuda_cart)
        call = (Slice, Mask, tpick, ipick, phase_label, locs_use_cart, x_temp_cuda_cart, x_query_cart, x_query_src_cart, t_query, tq_sample,
                trv_out_q)
        if getattr(self, "_fwd_key", None) != key:
            self._fwd_key = self._fwd_refs = None
            # deferral pays only while the callers' lists are written in the literal form the device checks recognise: once a sample
            # failed them (valid graphs in another edge order, base graphs of non-uniform degree, ...) every later sample would pay a
            # discarded forward + a second build, so the model remembers and builds with the checks up front from then on
            if getattr(self, "_defer_failed", False):
                self.set_adjacencies(*adj)
            else:
                try:
                    self.set_adjacencies(*adj, _defer_checks=True)
                except Exception:
                    # building on unverified (clamped) tables can fail in ways the checked path reports properly or does not hit at all
                    self._defer_failed = True
                    self.set_adjacencies(*adj)
            self._fwd_key, self._fwd_refs = key, graph_tensors
        else:
            self.A_src_in_edges, self.A_Lg_in_src = A_src_in_edges, A_Lg_in_src
            self.A_edges_p, self.A_edges_s, self.dt_partition, self.tlatent = A_edges_p, A_edges_s, dt_partition, tlatent
            ea = _engine._f32(A_src_in_edges.x, "A_src_in_edges.x", tuple(self._edge_attr.shape))
            if ea.data_ptr() != self._edge_attr.data_ptr() or ea._version != getattr(self, "_edge_attr_version", None):
                self._edge_attr, self._edge_attr_version = ea, ea._version
                self._hip.set_static_edge_attr(ea)
        deferred = getattr(self, "_pending_checks", None) is not None
        try:
            out = self.forward_fixed(*call)
            standing = not deferred or self._resolve_pending_checks()
        except (IndexError, _engine._lib.GenieHipError):
            if not deferred or getattr(self, "_pending_checks", None) is None:
                raise        # a checked context (or verdicts already read as good): a genuine error of this call
            # raised while running on tables that were never verified: decide below, with the checks up front
            self._pending_checks = None
            self._hip.discard_flags()
            standing = False
        if not standing:
            # the graphs were not what the context was built for: build again with the checks up front (general path or ValueError)
            self._defer_failed = True
            self._fwd_key = self._fwd_refs = None
            self.set_adjacencies(*adj)
            self._fwd_key, self._fwd_refs = key, graph_tensors
            out = self.forward_fixed(*call)
        return out
