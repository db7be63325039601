"""Diagnostic: device time of mcp_chol_factor / mcp_chol_inverse alone (one matrix), per form of mcp_debug_set_chol_mfma.
    python tools/time_chol.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, numpy as np, torch
from mc_pilco_amd import hipabi, ops

dev = torch.device("cuda", 0)
for N in [int(a) for a in sys.argv[1:]] or [300, 400]:
    rs = np.random.RandomState(0)
    A = rs.randn(N, N + 3)
    K = torch.tensor(A @ A.T / (N + 3) + 0.1 * np.eye(N), dtype=torch.float64, device=dev)
    for form in ((1, 2) if N <= 1152 else (1,)):
        hipabi.lib().mcp_debug_set_chol_mfma(form)
        U, _, _ = ops.chol_factor(K)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            ops.chol_factor(K)
        ev[1].record()
        for _ in range(20):
            ops.chol_inverse(U)
        ev[2].record()
        torch.cuda.synchronize()
        print("N=%d form %d: factor %.1f us (incl. the copy of K), inverse + K^-1 %.1f us" % (N, form, 50 * ev[0].elapsed_time(ev[1]), 50 * ev[1].elapsed_time(ev[2])))
    hipabi.lib().mcp_debug_set_chol_mfma(1)
