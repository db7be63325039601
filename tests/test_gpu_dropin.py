"""GPU: the drop-in class surface (gpr_lib / Model_learning / Policy / Cost_function / MC_PILCO) reproduces the
reference seed for seed: objects are built exactly like tests/golden/make_golden.py built the reference's,
torch is seeded with the fixture's seed, and apply_policy + cost + backward are compared with the golden
outputs (the reference's own results)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from mc_pilco_amd import synthetic as sy

pytestmark = pytest.mark.gpu
quiet = lambda: contextlib.redirect_stdout(io.StringIO())
dtype = torch.float64


def dev():
    return torch.device("cuda", 0)


def T(a):
    return torch.tensor(np.asarray(a), dtype=dtype, device=dev())


def rbf_dict(D, ls, sigma_n):
    return dict(active_dims=np.arange(D), lengthscales_init=np.asarray(ls, dtype=float), flg_train_lengthscales=True, lambda_init=np.ones(1),
                flg_train_lambda=False, sigma_n_init=sigma_n * np.ones(1), sigma_n_num=None, flg_train_sigma_n=True, dtype=dtype, device=dev())


def mpk_dict(D, deg, weights):
    return dict(active_dims=np.arange(D), poly_deg=deg, Sigma_pos_par_init_list=weights, flg_train_Sigma_pos_par_list=[True] * deg, dtype=dtype,
                device=dev())


def relerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def build_cartpole(fx, deg, sod):
    from mc_pilco_amd.model_learning import Model_learning as ML

    c = sy.CARTPOLE
    par = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
               not_vel_indeces=c["not_vel"], dtype=dtype, device=dev())
    if sod:
        par["approximation_mode"] = "SOD"
        par["approximation_dict"] = {"SOD_threshold_mode": "relative", "SOD_threshold": 0.5, "flg_SOD_permutation": False}
    sig = float(fx["sigma_n"])
    with quiet():
        if deg == 0:
            par["init_dict_list"] = [rbf_dict(6, c["lengthscales"], sig)] * 2
            ml = ML.Speed_Model_learning_RBF_angle_state(**par)
        else:
            pw = [[fx["poly_w%d_gp%d" % (k, g)] for k in range(1, deg + 1)] for g in range(2)]
            par["init_dict_list"] = [[rbf_dict(6, c["lengthscales"], sig), mpk_dict(6, deg, pw[g])] for g in range(2)]
            ml = ML.Speed_Model_learning_RBF_MPK_angle_state(**par)
        ml.add_data(fx["states_tr"], fx["inputs_tr"])
        with torch.no_grad():
            for g in range(2):
                ml.pretrain_gp(g)
        ml.set_eval_mode()
    return ml


def build_mcpilco(fx, ml, B):
    from mc_pilco_amd.policy_learning import MC_PILCO, Cost_function, Policy

    c = sy.CARTPOLE
    ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=fx["pol_ls"].reshape(-1), centers_init=fx["pol_centers"], weight_init=fx["pol_weight"], flg_squash=True,
                u_max=c["u_max"], flg_drop=True, dtype=dtype, device=dev())
    with quiet():
        obj = MC_PILCO.MC_PILCO(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None, f_model_learning=lambda **kw: ml,
                                model_learning_par={}, f_rand_exploration_policy=Policy.Random_exploration,
                                rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=1.0, dtype=dtype),
                                f_control_policy=Policy.Sum_of_gaussians_with_angles, control_policy_par=ppar,
                                f_cost_function=Cost_function.Cart_pole_cost,
                                cost_function_par=dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0),
                                log_path=None, dtype=dtype, device=dev())
    return obj


@pytest.mark.parametrize("name,deg,sod,seed", [("rollout_se", 0, False, 101), ("rollout_se_nodrop", 0, False, 102),
                                                ("rollout_se_sod", 0, True, 103), ("rollout_se_poly2", 2, False, 104)])
def test_seed_for_seed_parity_with_reference(golden, name, deg, sod, seed):
    fx = golden(name)
    ml = build_cartpole(fx, deg, sod)
    # pretrain on the device reproduces the reference's cached operands (and its SOD choice, index for index)
    for g in range(2):
        assert relerr(ml.alpha_list[g], fx["alpha%d" % g]) < 1e-8
        assert relerr(ml.K_X_inv_list[g], fx["Kinv%d" % g]) < 1e-8
        if sod:
            assert [int(i) for i in ml.SOD_indices[g]] == [int(i) for i in fx["sod%d" % g]]
    obj = build_mcpilco(fx, ml, fx["pol_centers"].shape[0])
    obj.noise_mode = "reference"
    M, Tn, p = fx["states"].shape[1], fx["states"].shape[0], float(fx["p_drop"])
    torch.manual_seed(seed)
    st, inp = obj.apply_policy(particles_initial_state_mean=T(fx["x0_mean"]), particles_initial_state_var=T(fx["x0_var"]),
                               flg_particles_init_uniform=False, particles_init_up_bound=None, particles_init_low_bound=None,
                               flg_particles_init_multi_gauss=False, num_particles=M, T_control=Tn, p_dropout=p)
    cost, std = obj.cost_function(st, inp, 0)
    cost.backward()
    assert np.array_equal(st[0].detach().cpu().numpy(), fx["states"][0])  # x0 bit-exact
    assert float((st.detach().cpu() - torch.as_tensor(fx["states"])).abs().max()) < 1e-8
    assert float((inp.detach().cpu() - torch.as_tensor(fx["inputs"])).abs().max()) < 1e-8
    assert abs(float(cost) - float(fx["cost"])) < 1e-10 * abs(float(fx["cost"]))
    assert abs(float(std) - float(fx["std"])) < 1e-9 * max(abs(float(fx["std"])), 1e-3)
    pol = obj.control_policy
    assert relerr(pol.log_lengthscales.grad, fx["g_log_ls"]) < 1e-7
    assert relerr(pol.centers.grad, fx["g_centers"]) < 1e-7
    assert relerr(pol.f_linear.weight.grad, fx["g_weight"]) < 1e-7


def test_state_dict_keys_match_reference_names(golden):
    fx = golden("rollout_se_poly2")
    ml = build_cartpole(fx, 2, False)
    keys = sorted(ml.gp_list[0].state_dict().keys())
    assert keys == sorted(["gp_list.0.sigma_n_log", "gp_list.0.log_lengthscales_par", "gp_list.0.log_lambda_par", "gp_list.0.mean_par",
                           "gp_list.1.gp_list.0.mean_par", "gp_list.1.gp_list.0.Sigma_pos_par", "gp_list.1.gp_list.1.mean_par",
                           "gp_list.1.gp_list.1.Sigma_pos_par"])
    obj = build_mcpilco(fx, ml, fx["pol_centers"].shape[0])
    assert sorted(obj.control_policy.state_dict().keys()) == ["centers", "f_linear.weight", "log_lengthscales"]


def test_single_step_api_matches_fixture(golden):
    """Model_learning.get_next_state (mean prediction) and GP objects' covariance / posterior methods."""
    fx = golden("step_se")
    ml = build_cartpole(fx, 0, False)
    with torch.no_grad():
        nm, mu, var = ml.get_next_state(T(fx["x"]), T(fx["u"]), particle_pred=False)
    assert float((nm.cpu() - torch.as_tensor(fx["next_mean"])).abs().max()) < 1e-9
    assert float((mu.cpu() - torch.as_tensor(fx["mu"])).abs().max()) < 1e-9
    assert float((var.cpu() - torch.as_tensor(fx["var"])).abs().max()) < 1e-9
    # gradient of the single-step posterior w.r.t. the test inputs against finite differences of itself
    gp = ml.gp_list[0]
    z = ml.data_to_gp_input(T(fx["x"]), T(fx["u"]))[:4].clone().requires_grad_(True)
    mu, var = gp.get_estimate_from_alpha(ml.gp_inputs_tr_list[0], z, ml.alpha_list[0], ml.m_X_list[0], K_X_inv=ml.K_X_inv_list[0])
    (mu.sum() + 3.0 * var.sum()).backward()
    h = 1e-6
    for d in range(6):
        zp, zm = z.detach().clone(), z.detach().clone()
        zp[:, d] += h
        zm[:, d] -= h
        fp = gp.get_estimate_from_alpha(ml.gp_inputs_tr_list[0], zp, ml.alpha_list[0], ml.m_X_list[0], K_X_inv=ml.K_X_inv_list[0])
        fm = gp.get_estimate_from_alpha(ml.gp_inputs_tr_list[0], zm, ml.alpha_list[0], ml.m_X_list[0], K_X_inv=ml.K_X_inv_list[0])
        fd = ((fp[0].reshape(-1) + 3.0 * fp[1]) - (fm[0].reshape(-1) + 3.0 * fm[1])) / (2 * h)
        assert float((fd - z.grad[:, d]).abs().max()) < 1e-6 * max(1.0, float(z.grad[:, d].abs().max()))


def test_policy_forward_class_surface(golden):
    from mc_pilco_amd.policy_learning import Policy

    fx = golden("policy")
    with quiet():
        pol = Policy.Sum_of_gaussians_with_target_trajectory(state_dim=24, input_dim=6, num_basis=40, target_traj=fx["traj_target"],
                                                             lengthscales_init=fx["traj_ls"].reshape(-1), centers_init=fx["traj_centers"],
                                                             weight_init=fx["traj_weight"], flg_squash=True, u_max=[1.0] * 6, flg_drop=True,
                                                             dtype=dtype, device=dev())
    with torch.no_grad():
        u = pol(T(fx["traj_x"]), t=int(fx["traj_t"]), p_dropout=0.0)
    assert relerr(u, fx["traj_u0"]) < 1e-12
    pol.noise_mode = "torch_cpu"
    torch.manual_seed(23)
    with torch.no_grad():
        u = pol(T(fx["traj_x"]), t=int(fx["traj_t"]), p_dropout=0.25)
    assert relerr(u, fx["traj_u25"]) < 1e-12


def test_reinforce_policy_runs_and_improves(golden):
    """A short optimisation on the HIP path (philox noise): finite costs, and the cost goes down."""
    fx = golden("rollout_se")
    ml = build_cartpole(fx, 0, False)
    obj = build_mcpilco(fx, ml, fx["pol_centers"].shape[0])
    with quiet():
        costs, stds, st, inp = obj.reinforce_policy(
            T_control=0.05 * 12, num_particles=64, trial_index=0, particles_initial_state_mean=T(fx["x0_mean"]),
            particles_initial_state_var=T(fx["x0_var"]), flg_particles_init_uniform=False, particles_init_up_bound=None,
            particles_init_low_bound=None, flg_particles_init_multi_gauss=False, opt_steps_list=[40], lr_list=[0.05],
            f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)", num_step_print=20, p_dropout_list=[0.25],
            policy_reinit_dict=dict(lenghtscales_par=np.ones(5), centers_par=np.array([np.pi, np.pi, np.pi, 1.0, 1.0]), weight_par=10.0))
    assert costs.shape == (40,) and np.all(np.isfinite(costs)) and np.all(np.isfinite(stds))
    assert st.shape == (12, 64, 4) and inp.shape == (12, 64, 1)
    assert costs[-10:].mean() < costs[:10].mean()


@pytest.mark.parametrize("name,D,deg", [("nll_se", 6, 0), ("nll_se_poly2", 6, 2), ("nll_se_poly1_d24", 24, 1)])
def test_marginal_likelihood_and_gradient_match_reference_autograd(golden, name, D, deg):
    """fit_model's objective: loss and d loss / d every hyper-parameter against the reference's autograd."""
    from mc_pilco_amd.gpr_lib.GP_prior import GP_prior as GP
    from mc_pilco_amd.gpr_lib.GP_prior import Sparse_GP, Stationary_GP
    from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood

    fx = golden(name)
    rbf = dict(rbf_dict(D, fx["lengthscales"], float(fx["sigma_n"])), flg_train_lambda=True)
    with quiet():
        if deg == 0:
            gp = Stationary_GP.RBF(**rbf)
        else:
            pw = [fx["poly_w%d" % k] for k in range(1, deg + 1)]
            gp = GP.Sum_Independent_GP(Stationary_GP.RBF(**rbf), Sparse_GP.get_Volterra_MPK_GP(**mpk_dict(D, deg, pw)))
    loss = Likelihood.Marginal_log_likelihood().loss_and_grad(gp, T(fx["X"]), T(fx["Y"]))
    assert abs(float(loss) - float(fx["loss"])) < 1e-9 * abs(float(fx["loss"]))
    checked = 0
    for n, p in gp.named_parameters():
        key = "grad__" + n
        if key in fx:
            ref = fx[key]
            assert float((p.grad.cpu() - torch.as_tensor(ref)).abs().max()) < 1e-8 * max(1.0, float(np.abs(ref).max())), n
            checked += 1
    assert checked == (3 if deg == 0 else 3 + deg)


def test_reinforce_runs_end_to_end():
    """The whole algorithm on the drop-in: exploration on the simulated cart-pole, GP training (fit_model on the device),
    SOD pretrain, policy optimisation with the fused kernels, policy applied to the system -- two short trials."""
    import tempfile

    from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood
    from mc_pilco_amd.model_learning import Model_learning as ML
    from mc_pilco_amd.policy_learning import MC_PILCO, Cost_function, Policy
    from mc_pilco_amd.simulation_class import ode_systems as f_ode

    np.random.seed(1)
    torch.manual_seed(1)
    c = sy.CARTPOLE
    init = dict(rbf_dict(6, np.ones(6), 1.0))
    mlp = dict(num_gp=2, angle_indeces=[2], not_angle_indeces=[0, 1, 3], T_sampling=0.05, vel_indeces=[1, 3], not_vel_indeces=[0, 2], device=dev(),
               dtype=dtype, approximation_mode="SOD",
               approximation_dict={"SOD_threshold_mode": "relative", "SOD_threshold": 0.5, "flg_SOD_permutation": False},
               init_dict_list=[init] * 2)
    B = 50
    pi = sy.cartpole_policy_init(B=B, seed=2)
    ppar = dict(state_dim=4, input_dim=1, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]), u_max=10.0, num_basis=B,
                dtype=dtype, device=dev(), centers_init=pi["centers"], lengthscales_init=pi["lengthscales"], weight_init=pi["weight"],
                flg_squash=True, flg_drop=True)
    with tempfile.TemporaryDirectory() as tmp, quiet():
        obj = MC_PILCO.MC_PILCO(T_sampling=0.05, state_dim=4, input_dim=1, f_sim=f_ode.cartpole, std_meas_noise=1e-2 * np.ones(4),
                                f_model_learning=ML.Speed_Model_learning_RBF_angle_state, model_learning_par=mlp,
                                f_rand_exploration_policy=Policy.Random_exploration,
                                rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=10.0, dtype=dtype),
                                f_control_policy=Policy.Sum_of_gaussians_with_angles, control_policy_par=ppar,
                                f_cost_function=Cost_function.Cart_pole_cost,
                                cost_function_par=dict(pos_index=0, angle_index=2, target_state=T([np.pi, 0.0]), lengthscales=T([3.0, 1.0])),
                                log_path=tmp, dtype=dtype, device=dev())
        mopt = dict(f_optimizer="lambda p : torch.optim.Adam(p, lr=0.01)", criterion=Likelihood.Marginal_log_likelihood, N_epoch=40, N_epoch_print=20)
        popt = dict(num_particles=32, opt_steps_list=[8, 8], lr_list=[0.01, 0.01], f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)",
                    num_step_print=4, p_dropout_list=[0.25, 0.25], p_drop_reduction=0.125, alpha_diff_cost=0.99, min_diff_cost=0.08,
                    num_min_diff_cost=200, min_step=200, lr_min=0.0025,
                    policy_reinit_dict=dict(lenghtscales_par=np.ones(5), centers_par=np.array([np.pi, np.pi, np.pi, 1.0, 1.0]), weight_par=10.0))
        costs, pstates, pinputs = obj.reinforce(initial_state=np.zeros(4), initial_state_var=1e-4 * np.ones(4), T_exploration=1.5, T_control=1.0,
                                                num_trials=2, model_optimization_opt_list=[mopt] * 2, policy_optimization_dict=popt)
        import pickle

        log = pickle.load(open(tmp + "/log.pkl", "rb"))
    assert len(costs) == 2 and all(np.all(np.isfinite(cl)) for cl in costs)
    assert pstates[0].shape == (20, 32, 4) and pinputs[0].shape == (20, 32, 1)
    for k in ("parameters_gp_0", "gp_inputs_0", "cost_trial_list", "parameters_trial_list", "state_samples_history"):
        assert k in log
    assert len(obj.state_samples_history) == 3  # one exploration + the policy applied after each of the two trials


@pytest.mark.parametrize("fused", [True, False])
def test_pms_seed_for_seed_parity_with_reference(golden, fused):
    """MC_PILCO4PMS.apply_policy (MC_PILCO.py:808-906) on the drop-in: same seed, same draws, same trajectories and gradient --
    through the fused kernels (measurement filter carried per particle, adjoint recursion in the reverse sweep) and through
    the step-wise operator path."""
    from mc_pilco_amd.policy_learning import MC_PILCO, Cost_function, Policy

    fx = golden("rollout_pms")
    ml = build_cartpole(fx, 0, False)
    c = sy.CARTPOLE
    B = fx["pol_centers"].shape[0]
    ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=fx["pol_ls"].reshape(-1), centers_init=fx["pol_centers"], weight_init=fx["pol_weight"], flg_squash=True,
                u_max=c["u_max"], flg_drop=True, dtype=dtype, device=dev())
    with quiet():
        obj = MC_PILCO.MC_PILCO4PMS(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None, f_model_learning=lambda **kw: ml,
                                    model_learning_par={}, f_rand_exploration_policy=Policy.Random_exploration,
                                    rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=1.0, dtype=dtype),
                                    f_control_policy=Policy.Sum_of_gaussians_with_angles, control_policy_par=ppar,
                                    f_cost_function=Cost_function.Cart_pole_cost,
                                    cost_function_par=dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0),
                                    pos_indeces=[int(i) for i in fx["pos_indeces"]], vel_indeces=[int(i) for i in fx["vel_indeces"]],
                                    std_meas_noise=fx["std_meas_noise"], log_path=None, filtering_dict={"fc": float(fx["fc"])}, dtype=dtype,
                                    device=dev())
    obj.noise_mode = "reference"
    obj.fused = fused
    M, Tn, p = fx["states"].shape[1], fx["states"].shape[0], float(fx["p_drop"])
    torch.manual_seed(107)
    st, inp = obj.apply_policy(particles_initial_state_mean=T(fx["x0_mean"]), particles_initial_state_var=T(fx["x0_var"]),
                               flg_particles_init_uniform=False, particles_init_up_bound=None, particles_init_low_bound=None,
                               flg_particles_init_multi_gauss=False, num_particles=M, T_control=Tn, p_dropout=p)
    cost, std = obj.cost_function(st, inp, 0)
    cost.backward()
    assert float((st[0].detach().cpu() - torch.as_tensor(fx["states"][0])).abs().max()) < 1e-15
    assert float((st.detach().cpu() - torch.as_tensor(fx["states"])).abs().max()) < 1e-8
    assert float((inp.detach().cpu() - torch.as_tensor(fx["inputs"])).abs().max()) < 1e-8
    assert abs(float(cost) - float(fx["cost"])) < 1e-10 * abs(float(fx["cost"]))
    pol = obj.control_policy
    assert relerr(pol.log_lengthscales.grad, fx["g_log_ls"]) < 1e-7
    assert relerr(pol.centers.grad, fx["g_centers"]) < 1e-7
    assert relerr(pol.f_linear.weight.grad, fx["g_weight"]) < 1e-7


def test_initial_particle_distributions_match_reference(golden):
    """MC_PILCO.apply_policy's three initial distributions (MC_PILCO.py:634-660) in reference noise mode: the multi-Gaussian
    component indices and the uniform draw reproduce the reference's for the same seed."""
    fx = golden("init_dists")
    fr = golden("rollout_se")
    ml = build_cartpole(fr, 0, False)
    obj = build_mcpilco(fr, ml, fr["pol_centers"].shape[0])
    obj.noise_mode = "reference"
    M = fx["mg_x0"].shape[0]
    torch.manual_seed(int(fx["mg_seed"]))
    x0 = obj.sample_initial_particles(T(fx["means"]), T(fx["vars"]), False, None, None, True, M)
    assert float((x0.cpu() - torch.as_tensor(fx["mg_x0"])).abs().max()) < 1e-15
    torch.manual_seed(int(fx["un_seed"]))
    x0 = obj.sample_initial_particles(T(fx["means"][0]), T(fx["vars"][0]), True, T(fx["ub"]), T(fx["lb"]), False, M)
    assert float((x0.cpu() - torch.as_tensor(fx["un_x0"])).abs().max()) < 1e-15
    # performance mode: same moments (Gaussian), on the device
    obj.noise_mode = "philox"
    torch.manual_seed(0)
    x0 = obj.sample_initial_particles(T(fx["means"][0]), T(fx["vars"][0]), False, None, None, False, 20000)
    assert x0.is_cuda and float((x0.mean(0).cpu() - torch.as_tensor(fx["means"][0])).abs().max()) < 0.01
    assert float((x0.var(0).cpu() / torch.as_tensor(fx["vars"][0]) - 1).abs().max()) < 0.05
