"""Diagnostic (BWX_WSTAMPS experiment build of rollout_bwd.hip): every wave's own intervals of a backward step of the general sweep, workgroup 0.
    python mc-pilco_amd/build.py --variant-bwd bws BWX_WSTAMPS
    MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=mc-pilco_amd/libmcpilco_hip_bws.so python tools/bwd_wave_stamps.py c5"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot  # noqa: F401
import torch

from mc_pilco_amd import hipabi, ops, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
dev = torch.device("cuda", 0)
w = workloads.build(name, device=dev)
x0 = w.sample_x0()
for i in range(2):
    for p in w.params:
        p.grad = None
    st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=i), x0, w.T, w.p_drop, meas=w.meas)
    c, sd = ops.expected_cost(w.cost, st)
    if i == 1:
        buf = torch.zeros(16 + 16 * 5, dtype=torch.int64, device=dev)
        hipabi.lib().mcp_debug_set_bwd_stamp_buffer(buf.data_ptr())
    c.backward()
torch.cuda.synchronize()
hipabi.lib().mcp_debug_set_bwd_stamp_buffer(None)
v = buf.cpu().tolist()
print("workload %s: backward, workgroup 0, cycles per step and wave (lean sweep: %d)" % (name, hipabi.lib().mcp_debug_last_bwd_lean()))
print("wave   chain+prefetch   barrier1   RBF stage   park   barrier2   total")
for wv in range(16):
    r = [x / w.T for x in v[16 + wv * 5:16 + wv * 5 + 5]]
    if sum(r) > 0:
        print("%4d   %12.0f   %8.0f   %9.0f   %5.0f   %8.0f   %6.0f" % ((wv,) + tuple(r) + (sum(r),)))
print("wave 0's own stamps: serial %.0f | barrier1 %.0f | RBF %.0f | park+barrier2 %.0f" % tuple(x / w.T for x in v[8:12]))
