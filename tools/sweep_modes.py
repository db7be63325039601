"""Diagnostic: forward-kernel time for forced (particles/workgroup, XLDS) modes at a given swarm size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
name = sys.argv[1]; M = int(sys.argv[2]); Tn = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device("cuda", 0)
w = workloads.build(name, device=dev, M=M, T=Tn)
x0 = w.sample_x0()
lib = hipabi.lib()
for ppw in (2, 4, 16):
    for xl in ((1,) if ppw == 16 else (1, 0)):
        lib.mcp_debug_set_particles_per_wg(ppw); lib.mcp_debug_set_fwd_mode(xl, 0)
        try:
            for i in range(2):
                ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 3
            for i in range(n):
                ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=5 + i), x0, w.T, w.p_drop)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print("%s M=%d ppw=%d xlds=%d : fwd %.2f ms  -> %.3e particle-steps/s (fwd only)" % (name, M, ppw, xl, dt * 1e3, M * w.T / dt), flush=True)
        except Exception as e:
            print(name, M, ppw, xl, "failed:", e)
lib.mcp_debug_set_particles_per_wg(0); lib.mcp_debug_set_fwd_mode(-1, 0)
