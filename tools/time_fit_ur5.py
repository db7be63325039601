"""Diagnostic: GP training epoch at the UR5 shape alone (6 GPs, N = 400, D = 24, SE + poly(1)), for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import workloads
s6, n6 = workloads.time_fit_model_ur5(torch.device("cuda", 0), 400, int(sys.argv[1]) if len(sys.argv) > 1 else 50)
print("fit_model on the device, UR5 shape: N=%d, D=24, SE+poly(1): %.2f ms per epoch for all 6 GPs" % (n6, 1e3 * s6))
