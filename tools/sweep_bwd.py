"""Diagnostic: backward-kernel time against swarm size (how much do co-resident workgroups slow each other?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
dev = torch.device("cuda", 0)
for M in [int(a) for a in sys.argv[2:]] or [128, 256, 400, 512, 768, 1024]:
    w = workloads.build(name, device=dev, M=M)
    x0 = w.sample_x0()
    ts = []
    for i in range(5):
        for p in w.params:
            p.grad = None
        st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop)
        c, sd = ops.expected_cost(w.cost, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        c.backward()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%s M=%d: backward (cost bwd + rollout bwd + reduce) %.3f ms (min of 5) -> %.0f cycles/particle-step-slot at 2.4 GHz"
          % (name, M, min(ts), min(ts) * 1e-3 * 2.4e9 / w.T), flush=True)
