"""Round-4 golden vectors, generated like make_golden*.py by IMPORTING THE REFERENCE (read-only at /root/reference) and running
its own classes.  Only arrays are stored -- no reference source text.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tests/golden/make_golden_r4.py

  surface_r4   the pieces of the public surface round 3 lacked (VERDICT r3 "missing" 3-5):
                 Policy.Sum_of_sinusoids (policy_learning/Policy.py:94-150) built under np.random.seed(7): amplitudes / omega / phases
                   (the np.random draw order is the contract) and u(t) at 6 times, squashed and not;
                 Policy.PD_controller (:406-449) on a 3-joint error signal;
                 Cost_function.cart_pole_cost / saturated_distance_from_trajectory (:170-182, :124-147) as plain functions handed to the
                   generic Expected_cost (:25-36): per-particle costs, (cost, std), gradient w.r.t. the states;
                 MPK_GP.get_Sigma_deg / get_Sigma / get_phi (gpr_lib/GP_prior/Sparse_GP.py:426-441, 613-656) and
                   Linear_GP.get_parameters_inv_lemma on a small regression;
                 ode_systems.pend (simulation_class/ode_systems.py:16-31).
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

with contextlib.redirect_stdout(io.StringIO()):
    import gpr_lib.Utils.Parameters_covariance_functions as RPC  # noqa: F401
    import gpr_lib.GP_prior.Sparse_GP as RSP
    import policy_learning.Cost_function as RC
    import policy_learning.Policy as RP
    import simulation_class.ode_systems as RODE

dtype = torch.float64
dev = torch.device("cpu")
torch.set_num_threads(1)


def N(t):
    return t.detach().cpu().numpy().copy()


out = {}

# ---- exploration policies --------------------------------------------------------------------------------------------
np.random.seed(7)
sos = RP.Sum_of_sinusoids(state_dim=4, input_dim=2, num_sin=5, omega_min=0.5, omega_max=6.0, amplitude_min=[0.2, 1.0], amplitude_max=[3.0, 4.0],
                          flg_squash=True, u_max=2.5, dtype=dtype, device=dev)
ts = np.array([0.0, 0.05, 0.3, 1.0, 2.35, 7.5])
out.update(sos_amplitudes=N(sos.amplitudes), sos_omega=N(sos.omega), sos_phases=N(sos.phases), sos_t=ts,
           sos_u=np.stack([N(sos(torch.zeros(1, 4, dtype=dtype), t)) for t in ts]),
           sos_np_u=np.stack([sos.get_np_policy()(np.zeros((1, 4)), t) for t in ts]), sos_after=np.random.rand(3))
np.random.seed(11)
sos2 = RP.Sum_of_sinusoids(state_dim=4, input_dim=1, num_sin=3, omega_min=1.0, omega_max=2.0, amplitude_min=0.5, amplitude_max=1.5,
                           flg_squash=False, u_max=1, dtype=dtype, device=dev)
out.update(sos2_u=np.stack([N(sos2(None, t)) for t in ts]))

rng = np.random.RandomState(3)
traj = torch.tensor(rng.randn(10, 6), dtype=dtype)
pd = RP.PD_controller(state_dim=6, input_dim=3, sqrt_Kp_gains=np.array([2.0, 1.5, 1.0]), sqrt_Kd_gains=np.array([0.5, 0.4, 0.3]),
                      target_traj=traj, flg_squash=True, u_max=np.array([1.0, 2.0, 3.0]), dtype=dtype, device=dev)
xs = torch.tensor(rng.randn(4, 6), dtype=dtype)
out.update(pd_traj=N(traj), pd_states=N(xs), pd_u=np.stack([N(pd(xs, t)) for t in (0, 3, 9)]))

# ---- the two costs as plain functions under the generic Expected_cost ---------------------------------------------------
st = torch.tensor(rng.randn(7, 9, 4) * 1.5, dtype=dtype, requires_grad=True)
inp = torch.zeros(7, 9, 1, dtype=dtype)
tgt, ls = torch.tensor([np.pi, 0.0], dtype=dtype), torch.tensor([3.0, 1.0], dtype=dtype)
ec = RC.Expected_cost(lambda x, u, k: RC.cart_pole_cost(x, u, k, target_state=tgt, lengthscales=ls, angle_index=2, pos_index=0))
cost, std = ec(st, inp, 0)
cost.backward()
out.update(cf_states=N(st), cp_costs=N(RC.cart_pole_cost(st, inp, 0, tgt, ls, 2, 0)), cp_cost=N(cost), cp_std=N(std), cp_grad=N(st.grad))
st2 = torch.tensor(N(st), dtype=dtype, requires_grad=True)
ttraj = torch.tensor(rng.randn(7, 4), dtype=dtype)
ls_var = [torch.tensor([1.0, 2.0], dtype=dtype), torch.tensor([0.5, 3.0], dtype=dtype)]
ec2 = RC.Expected_cost(lambda x, u, k: RC.saturated_distance_from_trajectory(x, u, k, target_traj=ttraj, lengthscales=ls_var,
                                                                               flg_var_lengthscales=True, used_indeces=[0, 3]))
cost2, std2 = ec2(st2, inp, 1)
cost2.backward()
out.update(sd_traj=N(ttraj), sd_ls=np.stack([N(l) for l in ls_var]), sd_cost=N(cost2), sd_std=N(std2), sd_grad=N(st2.grad),
           sd_costs_all=N(RC.saturated_distance_from_trajectory(st2, inp, 0, ttraj, torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=dtype), False, None)))

# ---- MPK_GP regressor-space helpers ---------------------------------------------------------------------------------------
par = np.array([0.3, 0.5, 0.7, 1.1, 1.3, 1.7])
mpk = RSP.MPK_GP(active_dims=np.arange(3), poly_deg=2, Sigma_pos_par_init=par, flg_offset=False, dtype=dtype, device=dev)
X = torch.tensor(rng.randn(5, 3), dtype=dtype)
mpk.current_deg = 1
out.update(mpk_par=par, mpk_X=N(X), mpk_Sigma_deg0=N(mpk.get_Sigma_deg(0)), mpk_Sigma_deg1=N(mpk.get_Sigma_deg(1)), mpk_Sigma_cur1=N(mpk.get_Sigma()),
           mpk_phi=N(mpk.get_phi(X)), mpk_K=N(mpk.get_covariance(X)))
mpk1 = RSP.MPK_GP(active_dims=np.arange(3), poly_deg=1, sigma_n_init=0.1 * np.ones(1), Sigma_pos_par_init=np.array([0.4, 0.6, 0.8, 1.2]), flg_offset=True,
                  dtype=dtype, device=dev)
mpk1.current_deg = 0
Y = torch.tensor(rng.randn(5, 1), dtype=dtype)
out.update(mpk1_phi=N(mpk1.get_phi(X)), mpk1_Sigma=N(mpk1.get_Sigma()), mpk1_Y=N(Y), mpk1_w_lemma=N(mpk1.get_parameters_inv_lemma(X, Y)))

# ---- pendulum ODE -------------------------------------------------------------------------------------------------------
out.update(pend=np.array([RODE.pend([0.3, -0.7], 0.0, 1.2), RODE.pend([2.0, 0.1], 0.0, -0.4)]))

np.savez_compressed(os.path.join(HERE, "surface_r4.npz"), **{k: np.asarray(v) for k, v in out.items()})
print("wrote surface_r4", {k: np.asarray(v).shape for k, v in out.items()})
