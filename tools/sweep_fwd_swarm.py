import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
dev = torch.device("cuda", 0)
L = hipabi.lib()
for M in (512, 640, 800, 1024, 1280, 1536, 2000):
    w = workloads.build("c1", device=dev, M=M)
    x0 = w.sample_x0()
    for mode, name in ((-1, "auto"), (0, "unsharded")):
        L.mcp_debug_set_gp_sharding(mode)
        ts = []
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("c1 M=%d %s: forward %.3f ms (ppw %d, sharded launches %d, status %d)" % (M, name, min(ts), L.mcp_debug_last_particles_per_wg(), L.mcp_debug_last_gp_sharded(), int(status.item())), flush=True)
    L.mcp_debug_set_gp_sharding(-1)
