"""Diagnostic: the 16-particle MFMA tile kernel against the small-tile forward kernel on synthetic workloads of varied shape
(same Philox noise): max differences of states, inputs and stored Jacobians, with and without sampling."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import hipabi, ops, workloads
dev = torch.device("cuda", 0)
cases = [("cartpole", 0, 96, 16, 4), ("cartpole", 1, 96, 16, 4), ("cartpole", 2, 96, 16, 4), ("ur5", 0, 40, 16, 4), ("ur5", 1, 40, 16, 4), ("ur5", 0, 96, 16, 4),
         ("ur5", 1, 96, 20, 4), ("cartpole", 0, 300, 40, 5), ("ur5", 1, 400, 24, 4)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) if x.isdigit() else x for x in a.split(",")) for a in sys.argv[1:]]
for cs in cases:
    workloads.CONFIGS["dbg"] = cs
    w = workloads.build("dbg", device=dev)
    torch.manual_seed(3)
    x0 = w.sample_x0()
    out = {}
    for pp in (True, False):
        for ppw in (4, 16):
            hipabi.lib().mcp_debug_set_particles_per_wg(ppw)
            st, inp, jac, status = ops.rollout_forward_raw(w.model, w.policy, ops.NoiseSpec(seed=3, call=1), x0, w.T, w.p_drop, particle_pred=pp)
            torch.cuda.synchronize()
            out[(pp, ppw)] = (st.cpu(), inp.cpu(), jac.cpu(), int(status.item()), hipabi.lib().mcp_debug_last_particles_per_wg())
    hipabi.lib().mcp_debug_set_particles_per_wg(0)
    for pp in (True, False):
        a, b = out[(pp, 4)], out[(pp, 16)]
        print("%-28s pred=%d used=%d/%d status=%d/%d  |dstates| %.2e  |dinputs| %.2e  |djac| %.2e (|jac| %.2e)" % (
            str(cs), pp, a[4], b[4], a[3], b[3], (a[0] - b[0]).abs().max().item(), (a[1] - b[1]).abs().max().item(),
            (a[2] - b[2]).abs().max().item(), a[2].abs().max().item()), flush=True)
