"""BUILD CONTAINER ONLY (the reference does not exist on the GPU box): times one policy-gradient step of the reference itself
(imported read-only from /root/reference) and of the oracle port (oracle/mcpilco_oracle.py, bench.py's cpu_baseline) on the SAME
host, same shape (cart-pole SE, N=300, M=400, T=150, dropout 0.25, 1 thread), so that the port's speed can be related to the
reference's (SURVEY 8d asks for "within ~10 %")."""
import contextlib, io, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference"); sys.path.insert(1, ROOT)
os.environ.setdefault("MPLBACKEND", "Agg"); sys.dont_write_bytecode = True
torch.set_num_threads(1)
with contextlib.redirect_stdout(io.StringIO()):
    import gpr_lib.Utils.Parameters_covariance_functions  # noqa
    import model_learning.Model_learning as RML
    import policy_learning.Cost_function as RC
    import policy_learning.Policy as RP
import mcp_boot  # noqa
from mc_pilco_amd import synthetic as sy, workloads
from oracle import mcpilco_oracle as orc
dt = torch.float64
T_ = lambda a: torch.tensor(np.asarray(a), dtype=dt)
pb = workloads.numpy_problem("c1"); c = pb["cfg"]; M, Tn, p = 400, 150, 0.25
# reference objects
rbf = dict(active_dims=np.arange(6), lengthscales_init=np.asarray(c["lengthscales"], float), flg_train_lengthscales=True, lambda_init=np.ones(1),
           flg_train_lambda=False, sigma_n_init=c["sigma_n"] * np.ones(1), sigma_n_num=None, flg_train_sigma_n=True, dtype=dt, device=torch.device("cpu"))
with contextlib.redirect_stdout(io.StringIO()):
    ml = RML.Speed_Model_learning_RBF_angle_state(num_gp=2, init_dict_list=[rbf] * 2, T_sampling=c["Ts"], angle_indeces=c["angle"],
                                                  not_angle_indeces=c["not_angle"], vel_indeces=c["vel"], not_vel_indeces=c["not_vel"], dtype=dt,
                                                  device=torch.device("cpu"))
    roll = sy.cartpole_rollouts(n_roll=5, seed=1)
    x = np.concatenate([r[0] for r in roll], 0)[:301]; u = np.concatenate([r[1] for r in roll], 0)[:301]
    ml.add_data(x, u)
    with torch.no_grad():
        for g in range(2): ml.pretrain_gp(g)
    ml.set_eval_mode()
    pi = pb["policy"]
    pol = RP.Sum_of_gaussians_with_angles(state_dim=4, input_dim=1, num_basis=c["B"], angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                                          lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True,
                                          u_max=c["u_max"], flg_drop=True, dtype=dt, device=torch.device("cpu"))
cf = RC.Cart_pole_cost(target_state=T_(c["cost_target"]), lengthscales=T_(c["cost_ls"]), angle_index=2, pos_index=0)
def ref_step():
    x0 = T_(c["x0_mean"]).reshape(1, -1) + torch.sqrt(T_(c["x0_var"])).reshape(1, -1) * torch.randn(M, 4, dtype=dt)
    xs, us = [x0], [pol(x0, t=0, p_dropout=p)]
    for t in range(1, Tn):
        xn, _, _ = ml.get_next_state(xs[-1], us[-1]); xs.append(xn); us.append(pol(xn, t=t, p_dropout=p))
    cost, _ = cf(torch.stack(xs), torch.stack(us), 0)
    pol.zero_grad(); cost.backward()
ref_step(); t0 = time.perf_counter(); n = 2
for _ in range(n): ref_step()
tr = (time.perf_counter() - t0) / n
sys.path.insert(0, ROOT)
import bench
ob = bench.cpu_baseline(pb, M, Tn, p, 1, budget_s=20)
print("reference (imported): %.3f s/step = %.3e particle-steps/s | oracle port: %.3f s/step = %.3e | port/reference speed %.2f"
      % (tr, M * Tn / tr, ob["s_per_step"], ob["value"], tr / ob["s_per_step"]))
