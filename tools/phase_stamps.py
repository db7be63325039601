"""Diagnostic: per-phase cycle shares of rollout_fwd_kernel (workgroup 0), via the runtime-gated
s_memtime stamps.  Read the SHARES, not the totals (the stamps serialise nothing but add a branch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
ppw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
Mo = int(sys.argv[3]) if len(sys.argv) > 3 else None
To = int(sys.argv[4]) if len(sys.argv) > 4 else None
pd = float(sys.argv[5]) if len(sys.argv) > 5 else 0.25
w = workloads.build(name, device=dev, M=Mo, T=To, p_drop=pd)
buf = torch.zeros(32, dtype=torch.int64, device=dev)
x0 = w.sample_x0()
hipabi.lib().mcp_debug_set_particles_per_wg(ppw % 100)
hipabi.lib().mcp_debug_set_fwd_lean(0 if 100 <= ppw < 200 else -1)  # 1xx: the general sharded kernel; 2xx / automatic: the lean one where it applies
if ppw >= 100:  # 101 / 102 / 104 (general kernel), 201 / 202 / 204 (lean kernel): force the GP-sharded launch with that cluster size
    hipabi.lib().mcp_debug_set_gp_sharding(1)
elif ppw:
    hipabi.lib().mcp_debug_set_gp_sharding(0)
for i in range(2):
    ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop, meas=w.meas)
hipabi.lib().mcp_debug_set_stamp_block(int(os.environ.get("MCP_STAMP_BLOCK", "0")))  # which workgroup is stamped
hipabi.lib().mcp_debug_set_stamp_buffer(buf.data_ptr())
ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=9), x0, w.T, w.p_drop, meas=w.meas)
torch.cuda.synchronize()
hipabi.lib().mcp_debug_set_stamp_buffer(None)
v = buf.cpu().tolist()
names = ["S(state/feat)", "PHI", "U", "K", "V(matvec)", "vsum", "J", "F+integrate", "(hand-off wait, inside F)"]
if hipabi.lib().mcp_debug_last_fwd_lean():  # the lean kernel's five barrier intervals (the serial section of wave 0 is "F")
    names = ["S (wave 0)", "policy+K(state)", "-", "u+K(exp)", "-", "-", "V + J (to the barrier)", "F+hand-off+integrate", "(hand-off wait, inside F)"]
tot = sum(v[:8])
# backward stamps
w.params[0].grad = None
st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=10), x0, w.T, w.p_drop, meas=w.meas)
c, sd = ops.expected_cost(w.cost, st)
buf2 = torch.zeros(16, dtype=torch.int64, device=dev)
buf2[1] = buf2[2] = 1 << 62
hipabi.lib().mcp_debug_set_bwd_stamp_buffer(buf2.data_ptr())
c.backward()
torch.cuda.synchronize()
hipabi.lib().mcp_debug_set_bwd_stamp_buffer(None)
v2 = buf2.cpu().tolist()
if hipabi.lib().mcp_debug_last_bwd_lean():
    print("bwd (lean sweep) workgroup run times: max %d min %d core cycles; first start -> last start %.1f us, first start -> last end %.1f us" % (v2[0], v2[1], (v2[3] - v2[2]) / 100.0, (v2[4] - v2[2]) / 100.0))
    print("bwd (lean sweep) per step cycles, wave 0: chain %.0f | wait at barrier 1 %.0f | prepare next step %.0f | wait at barrier 2 %.0f" % tuple(x / w.T for x in (v2[8], v2[9], v2[10], v2[11])))
    print("bwd (lean sweep) per step cycles, wave 1: before barrier 1 (exp, Philox, distances) %.0f | wait %.0f | after it (fma, wave sum) %.0f | wait at barrier 2 %.0f" % tuple(x / w.T for x in (v2[12], v2[13], v2[14], v2[15])))
else:
    if hipabi.lib().mcp_debug_last_bwd_pipe():  # (round 6: one particle per workgroup on the wide classes -- wave 0's view of a step)
        print("bwd (pipelined form) per step cycles, wave 0: chain %.0f | wait for the RBF waves' first half + park %.0f | features of the next step %.0f | wait for their adjoint half %.0f"
              % tuple(x / w.T for x in (v2[8], v2[9], v2[10], v2[11])))
    else:
        print("bwd per step cycles: serial(wave0) %.0f | barrier1 %.0f | RBF stage %.0f | park+barrier2 %.0f" % tuple(x / w.T for x in (v2[8], v2[9], v2[10], v2[11])))
print("workload", name, "T", w.T, "M", w.M, "ppw forced", ppw, "launched", hipabi.lib().mcp_debug_last_particles_per_wg(), "gp-sharded", hipabi.lib().mcp_debug_last_gp_sharded(), "lean", hipabi.lib().mcp_debug_last_fwd_lean(), "total cycles", tot, "-> per step", tot / (w.T - 1))
print("tile kernel detail (wave 0, per step, all GPs): K setup %.0f, K tiles %.0f | J setup %.0f, J batches %.0f cyc" % tuple((0 if (i == 14 and hipabi.lib().mcp_debug_last_row_split()) else v[i]) / (w.T - 1) for i in (12, 13, 14, 15)))
if not hipabi.lib().mcp_debug_last_row_split():  # (the row-split cluster's phase F uses these slots: printed at the end)
    print("J finish (wave 0, per step, all GPs): wait for the other waves %.0f, park + barrier %.0f, add + barrier %.0f cyc" % tuple(v[i] / (w.T - 1) for i in (9, 10, 11)))
if hipabi.lib().mcp_debug_last_particles_per_wg() == 16:
    print("tile kernel, phase K per wave (own time, all GPs):", " ".join("%.0f" % (v[16 + i] / (w.T - 1)) for i in range(8)))
    print("tile kernel, phase V per wave (own time to the barrier, all GPs):", " ".join("%.0f" % (v[24 + i] / (w.T - 1)) for i in range(8)))
if hipabi.lib().mcp_debug_last_fwd_lean():
    print("lean kernel, wave 0 per step: S compute (to the barrier) %.0f | F compute (delta, granules, Jacobian stores) %.0f | hand-off poll %.0f | J (own rows) %.0f cyc"
          % tuple(v[i] / (w.T - 1) for i in (15, 9, 10, 12)))
    print("lean kernel, phase V per wave (own time, before the barrier):", " ".join("%.0f" % (v[16 + i] / (w.T - 1)) for i in range(8)))
for n, c in zip(names, v):
    print("%-14s %12d  %5.1f%%  %8.0f cyc/step" % (n, c, 100.0 * c / tot, c / (w.T - 1)))
if hipabi.lib().mcp_debug_last_row_split():
    print("row-split cluster, phase F (thread 0, per step, all GPs): my sums %.0f | partner poll %.0f | finish %.0f | end of J -> end of hand-off %.0f cyc"
          % tuple(v[i] / (w.T - 1) for i in (9, 10, 11, 14)))
