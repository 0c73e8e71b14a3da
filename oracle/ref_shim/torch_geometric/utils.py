"""Test-only stand-ins for the torch_geometric.utils functions the reference imports.

Semantics follow the PyG documentation.
"""
import torch


def remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    if edge_attr is None:
        return edge_index[:, keep], None
    return edge_index[:, keep], edge_attr[keep]


def softmax(src, index, ptr=None, num_nodes=None, dim=0):
    """Segment softmax along dim 0: subtract per-segment max, exp, divide by (segment sum + 1e-16)."""
    n = int(index.max().item()) + 1 if num_nodes is None else num_nodes
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    mx = torch.full(shape, float("-inf"), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, idx, src.detach(), reduce="amax", include_self=True)
    out = (src - mx.gather(0, idx)).exp()
    den = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add(0, idx, out)
    return out / (den.gather(0, idx) + 1e-16)


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max().item()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.get_default_dtype(), device=index.device)
    return out.scatter_add_(0, index, torch.ones_like(index, dtype=out.dtype))


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None, return_edge_mask=False):
    n = int(edge_index.max().item()) + 1 if num_nodes is None else num_nodes
    if subset.dtype == torch.bool:
        node_mask = subset
        subset = node_mask.nonzero().view(-1)
    else:
        node_mask = torch.zeros(n, dtype=torch.bool, device=edge_index.device)
        node_mask[subset] = True
    edge_mask = node_mask[edge_index[0]] & node_mask[edge_index[1]]
    ei = edge_index[:, edge_mask]
    ea = edge_attr[edge_mask] if edge_attr is not None else None
    if relabel_nodes:
        relabel = torch.zeros(n, dtype=torch.long, device=edge_index.device)
        relabel[subset] = torch.arange(subset.numel(), device=edge_index.device)
        ei = relabel[ei]
    if return_edge_mask:
        return ei, ea, edge_mask
    return ei, ea


def to_undirected(edge_index, *a, **k):
    ei = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    return torch.unique(ei, dim=1)


def to_networkx(data, *a, **k):
    """Directed networkx graph with nodes 0 .. num_nodes-1 (num_nodes inferred from edge_index when the Data object carries no
    node features, as PyG does) and one edge per column of edge_index. Used by LocalMarching (process_utils.py:59-60)."""
    import networkx as nx
    ei = data.edge_index
    n = int(data.x.shape[0]) if getattr(data, "x", None) is not None else (int(ei.max().item()) + 1 if ei.numel() else 0)
    g = nx.DiGraph()
    g.add_nodes_from(range(n))
    g.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
    return g
