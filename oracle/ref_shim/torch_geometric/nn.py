"""Test-only stand-in for torch_geometric.nn.MessagePassing (gather -> message -> scatter).

Semantics (PyG docs): j = edge_index[0] (source), i = edge_index[1] (target);
a `message` argument `foo_j` / `foo_i` is kwargs['foo'] indexed by j / i along node_dim (a tuple
kwarg supplies (source side, target side)); `edge_index`, `index` (= i), `size_i`, `size_j`,
`dim_size` are special; anything else is passed through. Output = scatter(message, i, node_dim,
dim_size = number of target nodes, reduce = aggr).
"""
import inspect
import torch
from torch_scatter import scatter


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim
        self._msg_params = None

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs

    def propagate(self, edge_index, size=None, **kwargs):
        if self._msg_params is None:
            self._msg_params = list(inspect.signature(self.message).parameters.keys())
        j, i = edge_index[0], edge_index[1]
        size = [None, None] if size is None else list(size)

        def _set(side, data):
            n = data.size(self.node_dim)
            if size[side] is None:
                size[side] = n

        args = {}
        for name in self._msg_params:
            if name in ("edge_index",):
                args[name] = edge_index
            elif name == "index":
                args[name] = i
            elif name.endswith("_i") or name.endswith("_j"):
                base = name[:-2]
                if name in ("size_i", "size_j", "edge_index_i", "edge_index_j"):
                    continue
                data = kwargs[base]
                side = 0 if name.endswith("_j") else 1
                if isinstance(data, (tuple, list)):
                    _set(0, data[0]); _set(1, data[1])
                    data = data[side]
                else:
                    _set(0, data); _set(1, data)
                idx = j if side == 0 else i
                args[name] = data.index_select(self.node_dim, idx)
            elif name in kwargs:
                args[name] = kwargs[name]
        if "size_i" in self._msg_params:
            args["size_i"] = size[1]
        if "size_j" in self._msg_params:
            args["size_j"] = size[0]
        if "dim_size" in self._msg_params:
            args["dim_size"] = size[1]
        if "edge_index_i" in self._msg_params:
            args["edge_index_i"] = i
        if "edge_index_j" in self._msg_params:
            args["edge_index_j"] = j
        msg = self.message(**args)
        dim_size = size[1]
        if dim_size is None:
            dim_size = int(i.max().item()) + 1
        out = scatter(msg, i, dim=self.node_dim, dim_size=dim_size, reduce=self.aggr)
        return self.update(out)
