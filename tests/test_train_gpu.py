"""GPU: BASELINE config 3 -- the training loop (forward -> 4-term weighted MSE -> backward per sample -> Adam(1e-3) per batch,
train_GENIE_model.py:1383-1392, :1786-1789, :1843-1861) through the drop-in class, loss-curve parity against the oracle's
autograd + torch.optim.Adam on the CPU. Tolerance: relative deviation of the loss per step <= 1e-4."""
import numpy as np
import pytest
import torch

from genie_amd import graph, module, synthetic, train
from tests.util import Case, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(geom, smp, dev):
    """The 22 positional tensors of `mz(*input_tensors)` (train_GENIE_model.py:1770-1786) + the labels, on `dev`."""
    S, G = geom.n_sta, geom.n_grid
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    ea = t(geom.edge_attr())
    d1 = graph.GraphEdges(x=ea, edge_index=A_src_in_prod.to(dev))
    d2 = graph.GraphEdges(x=ea, edge_index=A_src_in_prod.flip(0).contiguous().to(dev))
    inputs = [t(smp["Slice"]), t(smp["Mask"]), A_in_sta.to(dev), A_in_src.to(dev), d1, d2, A_src_in_sta.to(dev),
              t(geom.A_src_src, torch.long), t(smp["A_edges_p"], torch.long), t(smp["A_edges_s"], torch.long), t(smp["dt_partition"]),
              t(smp["tlatent"]), t(smp["tpick"]), t(smp["ipick"], torch.long), t(smp["phase_label"]), t(geom.locs), t(geom.x_grid),
              t(geom.x_query), t(smp["x_query_src"]), t(geom.t_query), t(smp["tq_sample"]), t(smp["trv_out_q"])]
    labels = (t(smp["Lbls"]), t(smp["Lbls_query"]), t(smp["pick_lbls"]))
    return inputs, labels


def _oracle_curve(w0, geom, samples, n_steps):
    """The same loop with the oracle: weights as autograd leaves, torch.optim.Adam(lr 1e-3), one step per batch."""
    from oracle import genie_oracle as O
    S, G = geom.n_sta, geom.n_grid
    w = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in w0.items()}
    opt = torch.optim.Adam([v for v in w.values() if v.requires_grad], lr=0.001)
    A_in_sta, A_in_src, A_src_in_prod, _ = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    c = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    losses = []
    for _ in range(n_steps):
        opt.zero_grad()
        total = 0.0
        for smp in samples:
            out = O.forward_fixed(w, c(smp["Slice"]), c(smp["Mask"]), A_in_sta, A_in_src, c(geom.edge_attr()), A_src_in_prod,
                                  c(geom.A_src_src, torch.long), c(smp["A_edges_p"], torch.long), c(smp["A_edges_s"], torch.long),
                                  c(smp["dt_partition"]), c(smp["tlatent"]), c(smp["tpick"]), c(smp["ipick"], torch.long),
                                  c(smp["phase_label"]), c(geom.x_grid), c(geom.x_query), c(smp["x_query_src"]), c(geom.t_query),
                                  c(smp["tq_sample"]), c(smp["trv_out_q"]), S)
            loss = train.reference_loss(out, (c(smp["Lbls"]), c(smp["Lbls_query"]), c(smp["pick_lbls"])), len(samples))
            loss.backward()
            total += float(loss.item())
        opt.step()
        losses.append(total)
    return losses, w


@pytest.mark.parametrize("shape", ["7x45", "20x500"])
def test_training_loss_curve_matches_oracle_adam(shape):
    """24 Adam steps over a batch of two samples (same station set and grid, different pick windows): the loss of every step
    within 1e-4 relative of the oracle's, the loss goes down, and the trained weights agree."""
    S, G, n_picks, nq = {"7x45": (7, 45, 90, 20), "20x500": (20, 500, 500, 300)}[shape]
    geom = synthetic.Geometry(S, G, L=100e3, n_query=nq, seed=1)
    samples = [synthetic.training_sample(geom, n_picks, seed=3, window=k) for k in range(2)]
    w0 = Case("tiny_6x40").weights               # distinct PReLU slopes
    torch.manual_seed(0)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    net.train()
    opt = train.make_optimizer(net)
    batch = [_inputs(geom, smp, DEV) for smp in samples]
    n_steps = 24
    got = [train.train_step(net, opt, batch) for _ in range(n_steps)]
    want, w = _oracle_curve(w0, geom, samples, n_steps)
    rel = [abs(a - b) / abs(b) for a, b in zip(got, want)]
    print("loss curve %s: first %.6g last %.6g (oracle %.6g -> %.6g), max relative deviation %.3g" %
          (shape, got[0], got[-1], want[0], want[-1], max(rel)))
    assert max(rel) <= 1e-4, rel
    assert got[-1] < 0.9 * got[0]
    # the trained weights themselves: Adam's normalised update amplifies tiny gradient differences where a gradient is ~0,
    # so compare the tensors that moved (24 steps x 1e-3) to 2 % of the distance they moved
    moved = 0
    for k, p in net.named_parameters():
        d = float((w[k].detach() - w0[k]).abs().max())
        if d < 5e-3:
            continue
        moved += 1
        assert max_abs(p.detach().cpu(), w[k].detach()) <= 0.02 * d + 1e-6, (k, max_abs(p.detach().cpu(), w[k].detach()), d)
    assert moved >= 60


def test_training_step_200_stations_matches_structured_oracle():
    """One training step of the 2-output `forward_fixed_source` at the config-2/3 station count (200 stations x 1500 source
    nodes = 300 000 product nodes): loss and every gradient of the path against the structured oracle's autograd."""
    from oracle import genie_oracle as O
    S, G = 200, 1500
    geom = synthetic.Geometry(S, G, L=300e3, n_query=500, seed=1)
    win = synthetic.make_window(geom, 7500, seed=2)
    w0 = Case("cfg1_20x500").weights
    rng = np.random.default_rng(5)
    lbl = torch.from_numpy(rng.random((G, 9)).astype(np.float32))
    lbl_q = torch.from_numpy(rng.random((500, 9)).astype(np.float32))
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    net.train()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), t(geom.locs),
                             t(geom.x_grid))
    y, x = net.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query),
                                    t(geom.t_query))
    mse = torch.nn.functional.mse_loss
    loss = 0.1 * mse(y[:, :, 0], lbl.to(DEV)) + 0.4 * mse(x[:, :, 0], lbl_q.to(DEV))
    loss.backward()
    w = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    yo, xo = O.forward_fixed_source_structured(w, c(win["Slice"]), c(win["Mask"]), graph.neighbour_table(geom.A_sta_sta, S),
                                               graph.neighbour_table(geom.A_src_src, G), c(geom.edge_attr()),
                                               torch.from_numpy(geom.A_src_src), c(geom.x_grid), c(geom.x_query), c(geom.t_query), S, G)
    lo = 0.1 * mse(yo[:, :, 0], lbl) + 0.4 * mse(xo[:, :, 0], lbl_q)
    lo.backward()
    assert abs(float(loss) - float(lo)) <= 1e-5 * abs(float(lo))
    checked = 0
    for k, p in net.named_parameters():
        if w[k].grad is None:
            continue
        assert p.grad is not None, k
        tol = 1e-5 * max(1e-3, float(w[k].grad.abs().max())) + 1e-9
        assert max_abs(p.grad.cpu(), w[k].grad) <= tol, (k, max_abs(p.grad.cpu(), w[k].grad), tol)
        checked += 1
    assert checked >= 85


def test_hip_training_front_is_used_deterministic_and_equals_the_autograd_formulation(monkeypatch):
    """The P-sized front of a training step runs as HIP passes in both directions (genie_da_train_fwd / genie_da_train_bwd):
    it is the path `forward_fixed_source` takes in train() mode, its gradients are bitwise reproducible (fixed-order reduction of
    per-wave partials) and equal those of the per-node autograd formulation kept for A/B (GENIE_TRAIN_AUTOGRAD=1)."""
    S, G = 40, 300
    geom = synthetic.Geometry(S, G, L=200e3, n_query=60, seed=3)
    win = synthetic.make_window(geom, 900, seed=4)
    w0 = Case("tiny_6x40").weights
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
    rng = np.random.default_rng(9)
    cy, cx = t(rng.normal(0, 1, (G, 9, 1))), t(rng.normal(0, 1, (60, 9, 1)))

    def run(env):
        if env:
            monkeypatch.setenv("GENIE_TRAIN_AUTOGRAD", "1")
        else:
            monkeypatch.delenv("GENIE_TRAIN_AUTOGRAD", raising=False)
        net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
        net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
        net.train()
        net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), t(geom.locs),
                                 t(geom.x_grid))
        calls = []
        orig = net._hip.train_bwd
        net._hip.train_bwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        y, x = net.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query),
                                        t(geom.t_query))
        ((y * cy).sum() + (x * cx).sum()).backward()
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, len(calls), y.detach(), x.detach()

    g1, n1, y1, x1 = run(False)
    g2, n2, y2, x2 = run(False)
    g3, n3, y3, x3 = run(True)
    assert n1 == 1 and n2 == 1 and n3 == 0
    assert set(g1) == set(g3) and len(g1) >= 85
    for k in g1:
        tol = max(1e-4 * float(g3[k].abs().max()), 1e-6)     # the G- / Q-sized tail under autograd sums with atomics: two runs of
        assert max_abs(g1[k], g2[k]) <= tol, k                # the SAME path differ by ~1e-6 of a gradient's scale
        assert max_abs(g1[k], g3[k]) <= tol, (k, max_abs(g1[k], g3[k]), tol)
    assert max_abs(y1, y3) <= 1e-6 and max_abs(x1, x3) <= 1e-6
    # the HIP passes themselves are bitwise reproducible for a given upstream gradient
    from genie_amd import engine
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), grid_order=engine.sfc_order(geom.x_grid), device=DEV,
                        sta_order=engine.sfc_order(geom.locs))
    hp.set_weights({k: v.to(DEV) for k, v in w0.items()})
    Sl, Mk, ea = t(win["Slice"]), t(win["Mask"]), t(geom.edge_attr())
    r, xl, save = hp.train_fwd(Sl, Mk, ea)
    d_r = t(rng.normal(0, 1, (G, 30)))
    ga = {k: v.clone() for k, v in hp.train_bwd(Sl, Mk, ea, save, d_r).items()}
    gb = hp.train_bwd(Sl, Mk, ea, save, d_r)
    assert all(torch.equal(ga[k], gb[k]) for k in ga)
    # and the training forward equals the inference kernels' x_latent
    _, xl_inf, _ = hp.path_fwd(Sl, Mk, ea, t(geom.x_grid), True, False)
    assert max_abs(xl, xl_inf) <= 2e-6
