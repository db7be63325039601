"""Diagnostic: cProfile of GP hyper-parameter training on the drop-in classes (Model_learning.reinforce_model, N=300, D=6): where the host
time of an epoch goes (the figure behind the bench line's ``fit_model``)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import workloads
dev = torch.device("cuda", 0)
workloads.time_fit_model(dev, 300, 20)
pr = cProfile.Profile()
pr.enable()
s, n = workloads.time_fit_model(dev, 300, 200)
pr.disable()
print("ms per epoch per GP", 1e3 * s)
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
