"""CPU: the stand-ins of the reference's un-vendored third-party stack (`oracle/ref_shim/`: torch_scatter.scatter, torch_geometric's
MessagePassing / softmax / remove_self_loops / degree, torch_cluster.knn -- none installable here, none pinned by the reference) against
(1) the KNOWN ANSWERS those packages publish in their own documentation (the examples of the torch_scatter README / `scatter` docstring,
`torch_geometric.utils.softmax`, `remove_self_loops`, `degree`, `torch_cluster.knn`), and (2) the oracle's own restatements of the same
semantics (`oracle/genie_oracle.py`), which until round 5 met the shim only through whole-model outputs.
The fixtures of tests/golden/ were generated through this shim: this file is what pins the shim itself."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(os.path.dirname(HERE), "oracle", "ref_shim")


@pytest.fixture(scope="module")
def shim():
    """The shim packages imported under their third-party names (as oracle/make_golden.py does), removed from sys.modules afterwards."""
    names = ("torch_scatter", "torch_cluster", "torch_geometric", "torch_geometric.utils", "torch_geometric.nn", "torch_geometric.data")
    saved = {n: sys.modules.pop(n) for n in list(sys.modules) if n in names}
    sys.path.insert(0, SHIM)
    try:
        import torch_scatter, torch_cluster                      # noqa: E401
        import torch_geometric.utils as U
        import torch_geometric.nn as N
        yield {"scatter": torch_scatter.scatter, "knn": torch_cluster.knn, "U": U, "N": N}
    finally:
        sys.path.remove(SHIM)
        for n in names:
            sys.modules.pop(n, None)
        sys.modules.update(saved)


def test_scatter_known_answers_of_the_torch_scatter_documentation(shim):
    scatter = shim["scatter"]
    src = torch.tensor([[2.0, 0, 1, 4, 3], [0, 2, 1, 3, 4]])
    index = torch.tensor([[4, 5, 4, 2, 3], [0, 0, 2, 2, 1]])
    # README of torch_scatter ("scatter_max"): out = [[0, 0, 4, 3, 2, 0], [2, 4, 3, 0, 0, 0]]: empty segments of a max are ZERO
    assert torch.equal(scatter(src, index, dim=-1, dim_size=6, reduce="max"), torch.tensor([[0.0, 0, 4, 3, 2, 0], [2, 4, 3, 0, 0, 0]]))
    # `scatter` docstring ("sum"): the same inputs summed
    assert torch.equal(scatter(src, index, dim=-1, dim_size=6, reduce="sum"), torch.tensor([[0.0, 0, 4, 3, 3, 0], [2, 4, 4, 0, 0, 0]]))
    # mean = sum / clamp(count, 1): an empty segment is 0, not NaN
    assert torch.equal(scatter(src, index, dim=-1, dim_size=6, reduce="mean"), torch.tensor([[0.0, 0, 4, 3, 1.5, 0], [1, 4, 2, 0, 0, 0]]))
    # the form the reference uses (process_utils.py:563: 1-D index broadcast over rows, dim 0, 'max' of NON-NEGATIVE values)
    v = torch.tensor([0.3, 0.9, 0.2, 0.0, 0.5])
    assert torch.equal(scatter(v, torch.tensor([1, 1, 3, 3, 0]), dim=0, dim_size=5, reduce="max"), torch.tensor([0.5, 0.9, 0.0, 0.2, 0.0]))
    rows = torch.arange(12.0).view(4, 3)
    assert torch.equal(scatter(rows, torch.tensor([2, 0, 2, 2]), dim=0, dim_size=3, reduce="mean"),
                       torch.tensor([[3.0, 4, 5], [0, 0, 0], [5, 6, 7]]))


def test_pyg_utils_known_answers(shim):
    U = shim["U"]
    # torch_geometric.utils.softmax docstring: src = [1, 1, 1, 1], index = [0, 0, 1, 2] -> [0.5, 0.5, 1, 1]
    assert torch.allclose(U.softmax(torch.ones(4), torch.tensor([0, 0, 1, 2])), torch.tensor([0.5, 0.5, 1.0, 1.0]))
    # remove_self_loops docstring: [[0, 1, 0], [1, 0, 0]] -> [[0, 1], [1, 0]]
    assert torch.equal(U.remove_self_loops(torch.tensor([[0, 1, 0], [1, 0, 0]]))[0], torch.tensor([[0, 1], [1, 0]]))
    # degree docstring: row = [0, 1, 0, 2, 0] -> [3, 1, 1]
    assert torch.equal(U.degree(torch.tensor([0, 1, 0, 2, 0]), dtype=torch.long), torch.tensor([3, 1, 1]))
    # a large-magnitude segment does not overflow (the per-segment maximum is subtracted first)
    out = U.softmax(torch.tensor([[1000.0], [1001.0], [-5.0]]), torch.tensor([0, 0, 1]))
    assert torch.allclose(out[:, 0], torch.tensor([1.0 / (1.0 + np.e), np.e / (1.0 + np.e), 1.0]), atol=1e-6)


def test_knn_known_answer_of_the_torch_cluster_documentation(shim):
    # torch_cluster.knn docstring: x = [[-1,-1],[-1,1],[1,-1],[1,1]], y = [[-1,0],[1,0]], k = 2 -> [[0,0,1,1],[0,1,2,3]]
    x = torch.tensor([[-1.0, -1], [-1, 1], [1, -1], [1, 1]])
    y = torch.tensor([[-1.0, 0], [1, 0]])
    a = shim["knn"](x, y, 2)
    assert torch.equal(a[0], torch.tensor([0, 0, 1, 1]))
    assert sorted(a[1][:2].tolist()) == [0, 1] and sorted(a[1][2:].tolist()) == [2, 3]        # (equidistant pairs: order undefined upstream)


def test_message_passing_gathers_sources_and_scatters_to_targets(shim):
    """`propagate(edge_index, x=..., pos=...)`: `x_j = x[edge_index[0]]`, `pos_i = pos[edge_index[1]]`, the messages reduced over the
    edges that END in a node (flow source_to_target), a node without in-edges gets 0 -- computed by hand on a 4-node graph."""
    MP = shim["N"].MessagePassing

    class Diff(MP):
        def __init__(self, aggr):
            super().__init__(aggr)

        def forward(self, x, pos, ei):
            return self.propagate(ei, x=x, pos=pos)

        def message(self, x_j, pos_i, pos_j):
            return torch.cat((x_j, pos_i - pos_j), dim=-1)

    x = torch.tensor([[1.0], [10.0], [100.0], [1000.0]])
    pos = torch.tensor([[0.0], [1.0], [3.0], [6.0]])
    ei = torch.tensor([[0, 1, 3, 0], [2, 2, 2, 1]])                      # 0->2, 1->2, 3->2, 0->1
    add = Diff("add")(x, pos, ei)
    assert torch.equal(add, torch.tensor([[0.0, 0.0], [1.0, 1.0], [1011.0, (3 - 0) + (3 - 1) + (3 - 6)], [0.0, 0.0]]))
    mean = Diff("mean")(x, pos, ei)
    assert torch.allclose(mean, torch.tensor([[0.0, 0.0], [1.0, 1.0], [337.0, 2.0 / 3.0], [0.0, 0.0]]))


def test_oracle_restatements_equal_the_shim_on_random_segments(shim):
    """oracle/genie_oracle.py restates scatter 'sum' / 'mean', the mean propagate and the segment softmax for itself; on random
    inputs with EMPTY segments both statements give the same numbers (exactly for the sums and means, to rounding for the softmax)."""
    from oracle import genie_oracle as O
    g = torch.Generator().manual_seed(5)
    n, E, C = 37, 400, 7
    index = torch.randint(0, n, (E,), generator=g)
    index[index == 5] = 6                                                 # node 5 receives nothing
    index[index == 20] = 21
    msg = torch.randn(E, C, generator=g)
    scatter, U = shim["scatter"], shim["U"]
    assert torch.equal(O.scatter_sum(msg, index, n), scatter(msg, index, dim=0, dim_size=n, reduce="sum"))
    assert torch.equal(O.scatter_mean(msg, index, n), scatter(msg, index, dim=0, dim_size=n, reduce="mean"))
    assert float(O.scatter_mean(msg, index, n)[5].abs().max()) == 0.0
    a, b = O.segment_softmax(msg, index, n), U.softmax(msg, index, num_nodes=n)
    assert float((a - b).abs().max()) <= 1e-7
    seg = torch.zeros(n, C).index_add_(0, index, a)
    present = torch.bincount(index, minlength=n) > 0
    assert torch.allclose(seg[present], torch.ones(int(present.sum()), C), atol=1e-6) and float(seg[~present].abs().max()) == 0.0
    x = torch.randn(n, C, generator=g)
    ei = torch.stack((torch.randint(0, n, (E,), generator=g), index))
    MP = shim["N"].MessagePassing
    assert torch.equal(O.propagate_mean(x, ei), MP("mean").propagate(ei, x=x))
