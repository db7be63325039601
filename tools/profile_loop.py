"""Diagnostic: cProfile of MC_PILCO.reinforce_policy on the drop-in classes (host-side cost per optimizer step at the headline shape)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mcp_boot, torch
from mc_pilco_amd import workloads
dev = torch.device("cuda", 0)
workloads.time_reinforce_policy(dev, 20, False)
pr = cProfile.Profile()
pr.enable()
s, c0, c1 = workloads.time_reinforce_policy(dev, 200, False)
pr.disable()
print("ms per step", 1e3 * s)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
