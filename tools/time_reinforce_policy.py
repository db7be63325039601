"""Diagnostic: optimizer steps per second of MC_PILCO.reinforce_policy on the drop-in classes at the headline shape
(cart-pole, N=300, M=400, T=150, B=200) -- the same work as bench.py's step plus the loop's monitors and NaN check.
(bench.py reports the same figure as ``loop_ms_per_step``.)    python tools/time_reinforce_policy.py [steps] [pms]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import workloads
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pms = len(sys.argv) > 2 and sys.argv[2] == "pms"
s, c0, c1 = workloads.time_reinforce_policy(torch.device("cuda", 0), steps, pms)
print("reinforce_policy%s: %d steps (+1 reference rollout) -> %.2f ms per step, %.3e particle-steps/s; cost %.4f -> %.4f"
      % (" [PMS]" if pms else "", steps, 1e3 * s, 400 * 150 / s, c0, c1))
