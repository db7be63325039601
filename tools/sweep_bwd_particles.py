import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
dev = torch.device("cuda", 0)
"""Diagnostic: backward sweep time per particles-per-workgroup setting over the swarm size (0 = the automatic choice, which below 513 particles
is the latency-lean sweep).   python tools/sweep_bwd_particles.py [workload]"""
wl = sys.argv[1] if len(sys.argv) > 1 else "c1"
for name, M in [(wl, 400), (wl, 800), (wl, 1024), (wl, 1536), (wl, 2000), (wl, 2560), (wl, 3072), (wl, 4000)]:
    w = workloads.build(name, device=dev, M=M)
    x0 = w.sample_x0()
    for pb in (0, 1, 2, 4):
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
