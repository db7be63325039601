"""Diagnostic: optimizer steps per second of MC_PILCO.reinforce_policy on the drop-in classes at the headline shape
(cart-pole, N=300, M=400, T=150, B=200) -- the same work as bench.py's step plus the loop's monitors and NaN check."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcp_boot, numpy as np, torch
from mc_pilco_amd import synthetic as sy
from mc_pilco_amd.model_learning import Model_learning as ML
from mc_pilco_amd.policy_learning import MC_PILCO, Cost_function, Policy
dev = torch.device("cuda", 0); dt = torch.float64
c = sy.CARTPOLE
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pms = len(sys.argv) > 2 and sys.argv[2] == "pms"
T = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=dev)
rbf = dict(active_dims=np.arange(6), lengthscales_init=np.asarray(c["lengthscales"], dtype=float), flg_train_lengthscales=True, lambda_init=np.ones(1),
           flg_train_lambda=False, sigma_n_init=c["sigma_n"] * np.ones(1), sigma_n_num=None, flg_train_sigma_n=True, dtype=dt, device=dev)
mlp = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"], not_vel_indeces=c["not_vel"],
           dtype=dt, device=dev, init_dict_list=[rbf] * 2)
pi = sy.cartpole_policy_init(B=200, seed=1)
ppar = dict(state_dim=4, input_dim=1, num_basis=200, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
            lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True, u_max=c["u_max"],
            flg_drop=True, dtype=dt, device=dev)
kw = dict(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None, f_model_learning=ML.Speed_Model_learning_RBF_angle_state,
          model_learning_par=mlp, f_rand_exploration_policy=Policy.Random_exploration,
          rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=10.0, dtype=dt), f_control_policy=Policy.Sum_of_gaussians_with_angles,
          control_policy_par=ppar, f_cost_function=Cost_function.Cart_pole_cost,
          cost_function_par=dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0), log_path=None,
          dtype=dt, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    if pms:
        obj = MC_PILCO.MC_PILCO4PMS(pos_indeces=[0, 2], vel_indeces=[1, 3], std_meas_noise=0.01 * np.ones(4), filtering_dict={"fc": 0.5}, **kw)
    else:
        obj = MC_PILCO.MC_PILCO(**kw)
    for xs, us in sy.cartpole_rollouts(n_roll=5):
        obj.model_learning.add_data(np.asarray(xs), np.asarray(us))
    with torch.no_grad():
        for g in range(2):
            obj.model_learning.pretrain_gp(g)
    obj.model_learning.set_eval_mode()
    args = dict(T_control=7.5, num_particles=400, trial_index=0, particles_initial_state_mean=T(c["x0_mean"]), particles_initial_state_var=T(c["x0_var"]),
                flg_particles_init_uniform=False, particles_init_up_bound=None, particles_init_low_bound=None, flg_particles_init_multi_gauss=False,
                lr_list=[0.01], f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)", num_step_print=50, p_dropout_list=[0.25],
                policy_reinit_dict=dict(lenghtscales_par=np.ones(5), centers_par=np.array([np.pi, np.pi, np.pi, 1.0, 1.0]), weight_par=10.0))
    obj.reinforce_policy(opt_steps_list=[10], **args)  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = obj.reinforce_policy(opt_steps_list=[steps], **args)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
print("reinforce_policy%s: %d steps (+1 reference rollout) in %.3f s -> %.2f ms per step, %.3e particle-steps/s; cost %.4f -> %.4f"
      % (" [PMS]" if pms else "", steps, el, 1e3 * el / steps, 400 * 150 * steps / el, out[0][0], out[0][-1]))
