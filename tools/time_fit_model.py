"""Diagnostic: time of one GP hyper-parameter training epoch on the device (GP_prior.fit_model with the analytic
marginal-likelihood gradient kernel) at the cart-pole size (bench.py reports the same figure as ``fit_model``).
    python tools/time_fit_model.py [N] [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import workloads
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ep = int(sys.argv[2]) if len(sys.argv) > 2 else 200
s, n = workloads.time_fit_model(torch.device("cuda", 0), N, ep)
print("fit_model on the device: N=%d, D=6: %.2f ms per epoch per GP (%d epochs x 2 GPs)" % (n, 1e3 * s, ep))
if "ur5" in sys.argv:
    s6, n6 = workloads.time_fit_model_ur5(torch.device("cuda", 0), 400, 50)
    print("fit_model on the device, UR5 shape: N=%d, D=24, SE+poly(1): %.2f ms per epoch for all 6 GPs" % (n6, 1e3 * s6))
