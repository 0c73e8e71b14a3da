"""cProfile of the refine / association loops of bench.day_loops_leg on the config-2 shape: python tools/day_loops_prof.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from genie_amd import synthetic  # noqa: E402

S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
net = bench.build_model(geom, "cuda:0")
print(bench.day_loops_leg(net, geom, "cuda:0"))
pr = cProfile.Profile()
pr.enable()
print(bench.day_loops_leg(net, geom, "cuda:0"))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
