"""Host synchronisations inside the GPU-only apply loop (apply.apply_windows_device) on a small setup: python tools/sync_probe_apply.py"""
import os, sys, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from genie_amd import apply, module, synthetic
from tests.util import Case

DEV = "cuda:0"
S, G = 20, 200
geom = synthetic.Geometry(S, G, L=60e3, n_query=12, seed=71)
P = synthetic.make_picks(geom, 3000, seed=72)
P[:, 0] = P[:, 0] * 0.25 + 5000.0
P = P[np.argsort(P[:, 0], kind="stable")]
trv = geom.travel_times().astype(np.float32)
c = Case("tiny_6x40")
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
net.eval()
net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(DEV),
                         torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV))
max_t = float(np.ceil(trv.max() + 1.0))
run = lambda: apply.apply_windows_device(net, geom, P, trv, step_size="half", min_required_picks=5, max_t=max_t, tail_batch=16)
out, times = run()
torch.cuda.synchronize()
print("windows", len(times))


def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "genie_amd" in f.filename]
    print("SYNC:", str(message)[:50], "<-", " | ".join("%s:%d" % (f.filename.split("/")[-1], f.lineno) for f in st[-4:]))


warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
run()
torch.cuda.set_sync_debug_mode("default")
print("done")
