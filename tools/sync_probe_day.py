"""Host synchronisations inside the per-day loops (apply.refine_sources / associate_sources / apply_windows_device) on a small setup:
python tools/sync_probe_day.py  (torch.cuda.set_sync_debug_mode('warn'); every wait is printed with the genie_amd frames that led to it)."""
import os, sys, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genie_amd import apply
from tests.test_day_loops_gpu import _Setup

s = _Setup()
ga = s.geom_all
rng = np.random.default_rng(5)
nodes = rng.choice(s.G, 4, replace=False)
srcs = np.concatenate((ga.x_grid[nodes], rng.uniform(6995.0, 7010.0, (4, 1)), np.full((4, 1), 0.5)), axis=1)
off_min, off_rng = np.array([[-5e3, -5e3, -3e3]]), np.array([[10e3, 10e3, 6e3]])
ident = lambda x: x
ranges = ((0.0, 60e3), (0.0, 60e3), (-40e3, 2e3))
kw = dict(kernel_sig_t=s.sig, dt_embed=s.dt)


def run():
    ref, _ = apply.refine_sources([s.leg], s.picks, srcs, s.locs, s.tq, s.max_t, off_min, off_rng, 300, ident, ident, *ranges,
                                  rand=np.random.RandomState(77).rand, ftrns2_device=ident, **kw)
    d = np.linalg.norm(s.locs[None, :, :] - ref[:, None, 0:3], axis=2)
    trv = np.stack((d / 6000.0, d / 3500.0), axis=2)
    return apply.associate_sources([s.leg], s.picks, ref, s.locs, s.tq, s.max_t, trv, ident, np.array([0.0, 0.0, 0.0]), **kw)


run()
torch.cuda.synchronize()


def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "genie_amd" in f.filename]
    print("SYNC:", str(message)[:50], "<-", " | ".join("%s:%d" % (f.filename.split("/")[-1], f.lineno) for f in st[-4:]))


warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
run()
torch.cuda.set_sync_debug_mode("default")
print("done")
