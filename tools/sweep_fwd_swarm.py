"""Diagnostic: forward rollout time against swarm size for the launch forms that compete in the middle range: automatic dispatch,
everything unsharded, and the GP-sharded 16-particle kernel forced.   python tools/sweep_fwd_swarm.py [workload] [M ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
dev = torch.device("cuda", 0)
L = hipabi.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
for M in [int(a) for a in sys.argv[2:]] or [512, 640, 800, 1024, 1280, 1536, 2000]:
    w = workloads.build(name, device=dev, M=M)
    x0 = w.sample_x0()
    for ppw, mode, label in ((0, -1, "auto"), (0, 0, "unsharded"), (16, 1, "16-particle kernel, GP-sharded")):
        L.mcp_debug_set_particles_per_wg(ppw)
        L.mcp_debug_set_gp_sharding(mode)
        ts = []
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%s M=%d %s: forward %.3f ms (particles per workgroup %d, sharded launches %d, status %d)"
              % (name, M, label, min(ts), L.mcp_debug_last_particles_per_wg(), L.mcp_debug_last_gp_sharded(), int(status.item())), flush=True)
    L.mcp_debug_set_particles_per_wg(0)
    L.mcp_debug_set_gp_sharding(-1)
