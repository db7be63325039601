"""Diagnostic: the row-split cluster against the one-workgroup form on the full ur5_script shape (M = 200, T = 200), same on-device noise:
how far the two summation orders drift apart over the horizon, the status word, and the forward time of either form.
    python tools/row_split_soak.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads

dev = torch.device("cuda", 0)
w = workloads.build("ur5_script", device=dev)
L = hipabi.lib()
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
x0 = w.sample_x0()
out, ms = {}, {}
# forms: (row parts, deal): one workgroup per (tile, GP); two parts with a tile's members on one XCD (round 5); two / three parts dealt row part major (round 6)
for split in ((0, 0), (2, 0), (2, 1), (3, 1)):
    L.mcp_debug_set_row_split(split[0])
    L.mcp_debug_set_cluster_map(split[1])
    with torch.no_grad():
        for i in range(3):
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=7, call=1), x0, w.T, w.p_drop)
        assert L.mcp_debug_last_row_split() == split[0]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        bad = 0
        for i in range(rep):
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=7, call=1), x0, w.T, w.p_drop)
            bad |= int(status.item())
        ev[1].record()
        torch.cuda.synchronize()
    out[split] = (st.clone(), inp.clone())
    ms[split] = ev[0].elapsed_time(ev[1]) / rep
    print("row parts %d, deal %d: forward %.3f ms, status over %d rollouts %d" % (split[0], split[1], ms[split], rep, bad))
L.mcp_debug_set_row_split(-1)
L.mcp_debug_set_cluster_map(-1)
for split in ((2, 0), (2, 1), (3, 1)):
    d = (out[split][0] - out[(0, 0)][0]).abs().amax(dim=(1, 2))
    print("row parts %d, deal %d: max |states - states(one workgroup)| at t = 1, 10, 50, 100, 199:" % split, " ".join("%.2e" % float(d[t]) for t in (1, 10, 50, 100, 199)),
          "| inputs %.2e (|u| <= %.1f)" % (float((out[split][1] - out[(0, 0)][1]).abs().max()), float(out[(0, 0)][1].abs().max())))
