"""Diagnostic: compare the 16-particle tile kernel with the small-tile kernel on a golden rollout fixture."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, mcp_boot, torch
from conftest import load_golden
from gpu_helpers import G, noise_from, packed_model, packed_policy
from mc_pilco_amd import hipabi, ops
name, kind = sys.argv[1], sys.argv[2]
fx = load_golden(name)
model = packed_model(fx, kind); pol = packed_policy(fx, kind, requires_grad=False)
x0 = G(fx["states"][0]); Tn = fx["states"].shape[0]; p = float(fx["p_drop"])
out = {}
for ppw in (4, 16):
    hipabi.lib().mcp_debug_set_particles_per_wg(ppw)
    st, inp, jac, status = ops.rollout_forward_raw(model, pol, noise_from(fx), x0, Tn, p)
    torch.cuda.synchronize()
    print("ppw", ppw, "used", hipabi.lib().mcp_debug_last_particles_per_wg(), "status", int(status.item()))
    out[ppw] = (st.cpu(), inp.cpu(), jac.cpu())
hipabi.lib().mcp_debug_set_particles_per_wg(0)
a, b = out[4], out[16]
for t in range(Tn):
    ds = (a[0][t] - b[0][t]).abs().max().item(); du = (a[1][t] - b[1][t]).abs().max().item()
    dj = (a[2][t] - b[2][t]).abs().max().item() if t < Tn - 1 else 0.0
    print("t=%d  |dstate| %.3e  |dinput| %.3e  |djac| %.3e" % (t, ds, du, dj))
t = 0
dj = (a[2][t] - b[2][t]).abs()  # [M, G, D]
print("jac diff per gp (t=0):", dj.amax(dim=(0, 2)).tolist())
print("jac diff per dim (t=0):", ["%.1e" % v for v in dj.amax(dim=(0, 1)).tolist()])
print("state diff per comp (t=1):", (a[0][1] - b[0][1]).abs().amax(dim=0).tolist())
print("ref jac[0,0,0,:6]", a[2][0, 0, 0, :6].tolist()); print("tile jac[0,0,0,:6]", b[2][0, 0, 0, :6].tolist())
print("---- particle_pred=False (mean path only)")
out = {}
for ppw in (4, 16):
    hipabi.lib().mcp_debug_set_particles_per_wg(ppw)
    st, inp, jac, status = ops.rollout_forward_raw(model, pol, noise_from(fx), x0, Tn, p, particle_pred=False)
    torch.cuda.synchronize()
    print("ppw", ppw, "status", int(status.item()))
    out[ppw] = (st.cpu(), inp.cpu(), jac.cpu())
hipabi.lib().mcp_debug_set_particles_per_wg(0)
a, b = out[4], out[16]
for t in range(min(Tn, 3)):
    ds = (a[0][t] - b[0][t]).abs().max().item(); dj = (a[2][t] - b[2][t]).abs().max().item() if t < Tn - 1 else 0.0
    print("t=%d  |dstate| %.3e  |djac| %.3e" % (t, ds, dj))
print("state diff per comp (t=1):", ["%.1e" % v for v in (a[0][1] - b[0][1]).abs().amax(dim=0).tolist()])
dj = (a[2][0] - b[2][0]).abs()
print("Jmu diff per gp:", ["%.1e" % v for v in dj.amax(dim=(0, 2)).tolist()]); print("Jmu diff per dim:", ["%.1e" % v for v in dj.amax(dim=(0, 1)).tolist()])
print("Jmu diff per particle:", ["%.1e" % v for v in dj.amax(dim=(1, 2)).tolist()])
