import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
name = sys.argv[1]; T = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
w = workloads.build(name, device=dev, T=T)
x0 = w.sample_x0()
for pb in (0, 1, 2, 4):
    hipabi.lib().mcp_debug_set_bwd_particles(pb)
    try:
        ts = []
        for i in range(4):
            for p in w.params: p.grad = None
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop)
            c, s = ops.expected_cost(w.cost, st)
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); c.backward(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(name, "T", w.T, "M", w.M, "bwd PB", pb, "ms", min(ts[1:]))
    except Exception as ex:
        print("PB", pb, "failed:", ex)
hipabi.lib().mcp_debug_set_bwd_particles(0)
