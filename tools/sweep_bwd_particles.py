import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
dev = torch.device("cuda", 0)
for name, M in [("c1", 400), ("c1", 800), ("c1", 1024), ("c1", 2000), ("c1", 4000)]:
    w = workloads.build(name, device=dev, M=M)
    x0 = w.sample_x0()
    for pb in (1, 2, 4):
        hipabi.lib().mcp_debug_set_bwd_particles(pb)
        ts = []
        for i in range(4):
            for p in w.params:
                p.grad = None
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop)
            c, sd = ops.expected_cost(w.cost, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); c.backward(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%s M=%d PB=%d: backward %.3f ms" % (name, M, pb, min(ts)), flush=True)
    hipabi.lib().mcp_debug_set_bwd_particles(0)
