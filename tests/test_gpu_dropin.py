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
    h = 1e-5  # (the quotient's own rounding noise is eps |Kinv| |k|^2 / h ~ 1e-6 at h = 1e-6 with this Kinv: larger than its truncation error by far)
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
    ref_loss = float(np.asarray(fx["loss"]).reshape(-1)[0])
    assert abs(float(loss) - ref_loss) < 1e-9 * abs(ref_loss)
    checked = 0
    for n, p in gp.named_parameters():
        key = "grad__" + n
        if key in fx:
            ref = fx[key]
            assert float((p.grad.cpu() - torch.as_tensor(ref)).abs().max()) < 1e-8 * max(1.0, float(np.abs(ref).max())), n
            checked += 1
    assert checked == (3 if deg == 0 else 3 + deg)


class _OtherCriterion(torch.nn.modules.loss._Loss):
    """A loss that touches all four outputs of GP_prior.forward (the formula of tests/golden/make_golden_r5.py: OtherCriterion)."""

    def forward(self, out, Y):
        m_X, K, Kinv, logdet = out
        r = Y - m_X
        n = Y.shape[0]
        return (0.5 * (r.t() @ Kinv @ r) + 0.3 * logdet + 1e-3 * torch.trace(K) + 0.05 * (Kinv * Kinv).sum() / n).reshape(())


@pytest.mark.parametrize("name,deg", [("fwd_autograd_se", 0), ("fwd_autograd_se_poly2", 2)])
def test_forward_is_an_autograd_graph_under_any_criterion(golden, name, deg):
    """Round 5 (VERDICT r4 missing 4): `GP_prior.forward` carries a graph when its hyper-parameters are trainable, as in the reference
    (GP_prior.py:91-115) -- a criterion other than the marginal likelihood can be differentiated and trained with `fit_model` (:179-230).
    Against the reference's own autograd: loss rel 1e-10, every gradient 1e-8 max(1, |g|); and five Adam epochs of `fit_model` with that criterion,
    every hyper-parameter after every epoch to 1e-8."""
    from mc_pilco_amd.gpr_lib.GP_prior import GP_prior as GP
    from mc_pilco_amd.gpr_lib.GP_prior import Sparse_GP, Stationary_GP

    fx = golden(name)

    def make():
        rbf = dict(rbf_dict(6, fx["lengthscales"], float(fx["sigma_n"])), flg_train_lambda=True)
        with quiet():
            if deg == 0:
                return Stationary_GP.RBF(**rbf)
            pw = [fx["poly_w%d" % k] for k in range(1, deg + 1)]
            return GP.Sum_Independent_GP(Stationary_GP.RBF(**rbf), Sparse_GP.get_Volterra_MPK_GP(**mpk_dict(6, deg, pw)))

    gp = make()
    crit = _OtherCriterion()
    loss = crit(gp(T(fx["X"])), T(fx["Y"]))
    loss.backward()
    ref_loss = float(np.asarray(fx["loss"]).reshape(-1)[0])
    assert abs(float(loss.detach()) - ref_loss) < 1e-10 * abs(ref_loss)
    names = [str(n) for n in fx["names"]]
    pars = dict(gp.named_parameters())
    for n in names:
        ref = fx["grad__" + n].reshape(-1)
        got = pars[n].grad.detach().cpu().numpy().reshape(-1)
        assert float(np.abs(got - ref).max()) < 1e-8 * max(1.0, float(np.abs(ref).max())), n
    # fit_model with this criterion: the reference's loop, five epochs
    gp2 = make()
    traj = {n: [dict(gp2.named_parameters())[n].detach().cpu().numpy().copy()] for n in names}

    def snap():
        for n in names:
            traj[n].append(dict(gp2.named_parameters())[n].detach().cpu().numpy().copy())

    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(T(fx["X"]), T(fx["Y"])), batch_size=fx["X"].shape[0], shuffle=False)
    with quiet():
        gp2.fit_model(trainloader=loader, optimizer=torch.optim.Adam(gp2.parameters(), lr=float(fx["lr"])), criterion=crit,
                      N_epoch=int(fx["n_epoch"]), N_epoch_print=1, f_print=snap)
    for n in names:
        ref = fx["traj__" + n]
        got = np.stack(traj[n]).reshape(ref.shape)
        assert float(np.abs(got - ref).max()) < 1e-8 * max(1.0, float(np.abs(ref).max())), n


@pytest.mark.parametrize("N,D,deg", [(500, 6, 0), (600, 6, 2), (1100, 24, 1)])
def test_training_epoch_kernels_match_the_single_gp_path_at_large_n(N, D, deg):
    """`mcp_nll_epoch` (the batched training epoch: Gram tiles, left-looking Cholesky, four-wave inverse columns, LDS-staged gradient rows or, beyond
    their LDS budget, the row kernel) against `nll_loss_and_grad` (the one-GP path: separate entry points, the row kernel) on the same GP at sizes
    that take the epoch's other instantiations -- N = 500 the LDS-panel Cholesky with 8 tile slots, 600 the global-operand one, 1100 twelve slots and
    the gradient fallback: loss and every gradient to 1e-9 (two independent routes to the same numbers), and the epoch with lr = 0 leaves the
    parameters where they were."""
    from mc_pilco_amd import nll
    from mc_pilco_amd.gpr_lib.GP_prior import GP_prior as GP
    from mc_pilco_amd.gpr_lib.GP_prior import Sparse_GP, Stationary_GP
    from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood

    rs = np.random.RandomState(N)
    X = rs.uniform(-1.0, 1.0, size=(N, D))
    Y = (np.sin(X[:, :2].sum(1)) + 0.05 * rs.randn(N)).reshape(-1, 1)
    rbf = dict(rbf_dict(D, 0.8 + 0.6 * rs.rand(D) + 0.3 * D ** 0.5, 0.3), flg_train_lambda=True)
    with quiet():
        if deg == 0:
            gp = Stationary_GP.RBF(**rbf)
        else:
            pw = [0.3 * rs.rand(D + 1) if k == 1 else 0.3 * rs.rand(2, D) for k in range(1, deg + 1)]
            gp = GP.Sum_Independent_GP(Stationary_GP.RBF(**rbf), Sparse_GP.get_Volterra_MPK_GP(**mpk_dict(D, deg, pw)))
    before = {n: p.detach().clone() for n, p in gp.named_parameters()}
    loss_a = float(Likelihood.Marginal_log_likelihood().loss_and_grad(gp, T(X), T(Y)))
    grads_a = {n: p.grad.detach().clone() for n, p in gp.named_parameters() if p.grad is not None}
    for p in gp.parameters():
        p.grad = None
    fit = nll.BatchedFit([gp], T(X), [T(Y)], [1.0], [torch.optim.Adam(gp.parameters(), lr=0.0)], 1, 10 ** 9)
    assert fit.eligible
    with quiet():
        fit.run()
    assert abs(float(fit.loss[0]) - loss_a) < 1e-9 * abs(loss_a)
    assert len(grads_a) >= 3
    for n, p in gp.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
        if n in grads_a:
            assert float((p.grad - grads_a[n]).abs().max()) < 1e-9 * max(1.0, float(grads_a[n].abs().max())), n


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


# ---- round 2: the optimizer loop, the "next" rows of SURVEY 8f rank 4, cache / pointer lifetime -------------------------------
def _trace_setup(fx):
    ml = build_cartpole(fx, 0, False)
    obj = build_mcpilco(dict(pol_ls=fx["pol_ls"], pol_centers=fx["pol_centers"], pol_weight=fx["pol_weight"]), ml, fx["pol_centers"].shape[0])
    obj.noise_mode = "reference"
    return obj


@pytest.mark.parametrize("tag", ["plain", "forced"])
def test_reinforce_policy_trace_matches_reference(golden, tag):
    """MC_PILCO.reinforce_policy (MC_PILCO.py:375-613) on the drop-in, reference noise mode, same seed: the reference's cost trace over
    6 Adam steps to 1e-8, and the run whose monitors force two learning-rate halvings, two dropout reductions (the second one
    stops the mask draws: the RNG stream shifts) and the early exit after 8 of 12 steps; final policy parameters included."""
    fx = golden("policy_opt_trace")
    obj = _trace_setup(fx)
    kw = dict(opt_steps_list=[6], lr_list=[0.01], p_dropout_list=[0.25]) if tag == "plain" else dict(
        opt_steps_list=[12], lr_list=[0.01], p_dropout_list=[0.25], alpha_diff_cost=0.9, lr_reduction_ratio=0.5, lr_min=0.004,
        p_drop_reduction=0.125, min_diff_cost=1e9, num_min_diff_cost=2, min_step=0)
    torch.manual_seed(int(fx[tag + "_seed"]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        costs, stds, st, inp = obj.reinforce_policy(
            T_control=float(fx["T_control"]), num_particles=int(fx["M"]), trial_index=0, particles_initial_state_mean=T(fx["x0_mean"]),
            particles_initial_state_var=T(fx["x0_var"]), flg_particles_init_uniform=False, particles_init_up_bound=None,
            particles_init_low_bound=None, flg_particles_init_multi_gauss=False, f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)",
            num_step_print=100, policy_reinit_dict=None, **kw)
    assert costs.shape == fx[tag + "_cost_list"].shape
    assert relerr(costs, fx[tag + "_cost_list"]) < 1e-8
    assert relerr(stds, fx[tag + "_std_list"]) < 1e-7
    assert buf.getvalue().count("REDUCING THE LEARNING RATE") == int(fx[tag + "_n_lr_reductions"])
    assert ("EXIT FROM OPTIMIZATION" in buf.getvalue()) == bool(fx[tag + "_early_exit"])
    assert float(np.abs(st - fx[tag + "_last_states"]).max()) < 1e-7 and float(np.abs(inp - fx[tag + "_last_inputs"]).max()) < 1e-7
    pol = obj.control_policy
    assert relerr(pol.centers, fx[tag + "_final_centers"]) < 1e-8
    assert relerr(pol.f_linear.weight, fx[tag + "_final_weight"]) < 1e-8
    assert relerr(pol.log_lengthscales, fx[tag + "_final_log_ls"]) < 1e-8


def test_delta_state_model_matches_reference(golden):
    """Model_learning_RBF_angle_state (Model_learning.py:471-493, 528-580): GP i predicts the change of state i; one step."""
    from mc_pilco_amd.model_learning import Model_learning as ML

    fx = golden("delta_model_step")
    c = sy.CARTPOLE
    with quiet():
        ml = ML.Model_learning_RBF_angle_state(num_gp=4, init_dict_list=[rbf_dict(6, c["lengthscales"], float(fx["sigma_n"]))] * 4,
                                               angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], dtype=dtype, device=dev())
        ml.add_data(fx["states_tr"], fx["inputs_tr"])
        with torch.no_grad():
            for g in range(4):
                ml.pretrain_gp(g)
        ml.set_eval_mode()
    assert relerr(ml.gp_inputs, fx["gp_inputs"]) < 1e-14
    for g in range(4):
        assert relerr(ml.gp_output_list[g], fx["gp_output%d" % g]) < 1e-14
        assert relerr(ml.alpha_list[g], fx["alpha%d" % g]) < 1e-8
    with torch.no_grad():
        nm, mu, var = ml.get_next_state(T(fx["x"]), T(fx["u"]), particle_pred=False)
        torch.manual_seed(12)
        eps = torch.empty(20, 4, dtype=dtype).normal_()
    assert float((mu.cpu() - torch.as_tensor(fx["mu"])).abs().max()) < 1e-9
    assert float((var.cpu() - torch.as_tensor(fx["var"])).abs().max()) < 1e-9
    assert float((nm.cpu() - torch.as_tensor(fx["next_mean"])).abs().max()) < 1e-9
    assert np.array_equal(eps.numpy(), fx["eps"])
    nxt = T(fx["x"]) + mu + torch.sqrt(var) * eps.to(dev())
    assert float((nxt.cpu() - torch.as_tensor(fx["next"])).abs().max()) < 1e-9


def test_simple_costs_match_reference(golden):
    """Expected_distance / Expected_saturated_distance (Cost_function.py:39-101)."""
    from mc_pilco_amd.policy_learning import Cost_function

    fx = golden("simple_costs")
    act = [int(i) for i in fx["active_dims"]]
    for tag, cls in (("dist", Cost_function.Expected_distance), ("sat", Cost_function.Expected_saturated_distance)):
        st = T(fx["states"]).requires_grad_(True)
        cf = cls(target_state=T(fx["target"]), lengthscales=T(fx["lengthscales"]), active_dims=act)
        c, s = cf(st, None, 0)
        c.backward()
        assert abs(float(c) - float(fx[tag + "_cost"])) < 1e-12 * abs(float(fx[tag + "_cost"]))
        assert abs(float(s) - float(fx[tag + "_std"])) < 1e-12 * abs(float(fx[tag + "_std"]))
        assert relerr(st.grad, fx[tag + "_grad"]) < 1e-12


def test_mean_rollout_matches_reference(golden):
    """MC_PILCO.rollout (MC_PILCO.py:347-373): mean-only prediction of a recorded trajectory."""
    fx = golden("mean_rollout")
    ml = build_cartpole(fx, 0, False)
    pi = sy.cartpole_policy_init(B=16, seed=8)
    obj = build_mcpilco(dict(pol_ls=pi["lengthscales"], pol_centers=pi["centers"], pol_weight=pi["weight"]), ml, 16)
    obj.state_samples_history = [fx["x_rec"]]
    obj.input_samples_history = [fx["u_rec"]]
    with torch.no_grad(), quiet():
        traj = obj.rollout(data_collection_index=0)
        traj12 = obj.rollout(data_collection_index=0, T_rollout=12)
    assert traj.shape == fx["traj"].shape and float(np.abs(traj - fx["traj"]).max()) < 1e-8
    assert traj12.shape == (12, 4) and float(np.abs(traj12 - fx["traj12"]).max()) < 1e-9


def test_sod_with_seeded_permutation_matches_reference(golden):
    """GP_prior.get_SOD(flg_permutation=True) (GP_prior.py:244-247): the permutation is torch.randperm on the CPU generator, as
    in the reference, so the same seed visits the samples in the same order; indices exact."""
    from mc_pilco_amd.gpr_lib.GP_prior import Stationary_GP

    fx = golden("sod_permutation")
    with quiet():
        gp = Stationary_GP.RBF(**rbf_dict(6, fx["lengthscales"], float(fx["sigma_n"])))
        torch.manual_seed(int(fx["seed"]))
        idx = gp.get_SOD(T(fx["X"]), T(fx["Y"]), float(fx["thr"]), flg_permutation=True)
    assert [int(i) for i in idx] == [int(i) for i in fx["idx"]]


def test_load_model_from_reference_log(golden, tmp_path):
    """MC_PILCO.load_model_from_log (MC_PILCO.py:711-751) on a log.pkl written from REFERENCE objects (tests/golden/ref_log.pkl: its
    keys, its state_dict names, CPU tensors): data replayed, hyper-parameters restored, pretrain reproduces the reference's alpha.
    Then the drop-in's own log round trip: what it writes, it reads back to the same model."""
    import os
    import pickle
    import shutil

    from conftest import GOLDEN
    from mc_pilco_amd.model_learning import Model_learning as ML

    ex = golden("ref_log_expect")
    c = sy.CARTPOLE

    def fresh():
        par = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
                   not_vel_indeces=c["not_vel"], dtype=dtype, device=dev(), init_dict_list=[rbf_dict(6, np.ones(6), 1.0)] * 2)
        with quiet():
            ml = ML.Speed_Model_learning_RBF_angle_state(**par)
        pi = sy.cartpole_policy_init(B=16, seed=1)
        return build_mcpilco(dict(pol_ls=pi["lengthscales"], pol_centers=pi["centers"], pol_weight=pi["weight"]), ml, 16)

    folder = str(tmp_path) + "/"
    shutil.copy(os.path.join(GOLDEN, "ref_log.pkl"), folder + "log.pkl")
    obj = fresh()
    with quiet():
        obj.load_model_from_log(num_trial=1, folder=folder)
        obj.load_policy_from_log(num_trial=1, folder=folder)
    ml = obj.model_learning
    assert len(obj.state_samples_history) == 2 and ml.gp_inputs.shape == ex["gp_inputs"].shape
    assert relerr(ml.gp_inputs, ex["gp_inputs"]) < 1e-14
    for g in range(2):
        assert relerr(torch.exp(ml.gp_list[g].log_lengthscales_par), ex["ls%d" % g]) < 1e-14
        assert relerr(ml.alpha_list[g], ex["alpha%d" % g]) < 1e-8
    assert relerr(obj.control_policy.centers, ex["pol_centers"]) < 1e-15
    # the drop-in's own log: written with the reference's keys, read back by a fresh object
    obj.log_path = str(tmp_path / "own")
    os.makedirs(obj.log_path)
    with quiet():
        obj._save_log()
    obj2 = fresh()
    with quiet():
        obj2.load_model_from_log(num_trial=1, folder=obj.log_path + "/")
    for g in range(2):
        assert relerr(obj2.model_learning.alpha_list[g], ml.alpha_list[g].detach().cpu().numpy()) < 1e-12
    log = pickle.load(open(obj.log_path + "/log.pkl", "rb"))
    assert sorted(log["parameters_gp_0"][0].keys()) == sorted(["sigma_n_log", "log_lengthscales_par", "log_lambda_par", "mean_par"])


def test_descriptor_pointers_survive_operator_calls_between_rollouts(golden):
    """Round-1 advisor finding: the packed model's descriptor points at device copies of the kernel hyper-parameters; calling the
    single-step operators on the same GP objects between two fused rollouts must not free or move them.  packed() rollout ->
    get_next_state / posterior / get_estimate_from_alpha (with allocator churn) -> second rollout: bit-identical under fixed noise."""
    from mc_pilco_amd import ops

    fx = golden("rollout_se")
    ml = build_cartpole(fx, 0, False)
    obj = build_mcpilco(fx, ml, fx["pol_centers"].shape[0])
    pm, pp = ml.packed(), obj.control_policy.packed()
    x0 = T(fx["states"][0])
    nz = ops.NoiseSpec(eps=T(fx["eps"]), masks=torch.as_tensor(fx["masks"]).to(dev()).contiguous())
    with torch.no_grad():
        a = ops.rollout(pm, pp, nz, x0, fx["states"].shape[0], 0.25)[0].clone()
        for _ in range(3):
            ml.get_next_state(T(fx["states"][1]), T(fx["inputs"][1]), particle_pred=False)
            junk = [torch.zeros(n, dtype=dtype, device=dev()) for n in (6, 7, 64, 1, 13)]  # small blocks the allocator would hand back
            for g in range(2):
                ml.gp_list[g].get_estimate_from_alpha(ml.gp_inputs_tr_list[g], ml.gp_inputs[:5], ml.alpha_list[g], ml.m_X_list[g],
                                                      K_X_inv=ml.K_X_inv_list[g])
            del junk
        assert ml.packed() is pm
        b = ops.rollout(pm, pp, nz, x0, fx["states"].shape[0], 0.25)[0]
    assert torch.equal(a, b)
    assert float((a.cpu() - torch.as_tensor(fx["states"])).abs().max()) < 1e-8


def test_posterior_cache_follows_the_hyper_parameters(golden):
    """Round-1 advisor finding: get_alpha() followed by get_estimate_from_alpha() must see hyper-parameters changed in place in
    between (optimizer step, load_state_dict), even when the new tensors land on the old addresses."""
    from mc_pilco_amd.gpr_lib.GP_prior import Stationary_GP

    fx = golden("kern_se")
    with quiet():
        gp = Stationary_GP.RBF(**rbf_dict(6, fx["lengthscales"], float(fx["sigma_n"])))
    X, Y, Xs = T(fx["X"]), T(fx["Y"]), T(fx["Xs"])
    with torch.no_grad():
        alpha, mX, Kinv = gp.get_alpha(X, Y)
        mu0, var0 = gp.get_estimate_from_alpha(X, Xs, alpha, mX, K_X_inv=Kinv)
        assert relerr(mu0, fx["mu"]) < 1e-9
        gp.log_lengthscales_par.add_(0.3)  # in place, like an optimizer step
        alpha2, mX2, Kinv2 = gp.get_alpha(X, Y)
        mu1, var1 = gp.get_estimate_from_alpha(X, Xs, alpha2, mX2, K_X_inv=Kinv2)
        # same (X, alpha2, Kinv2) tensors, parameters moved back: must repack again, not serve the cached operands
        gp.log_lengthscales_par.sub_(0.3)
        mu2, _ = gp.get_estimate_from_alpha(X, Xs, alpha2, mX2, K_X_inv=Kinv2)
        from mc_pilco_amd import ops

        ref = ops.posterior(ops.PackedGP(gp.kernel_spec(), X, alpha2, Kinv2), Xs)[0]
    assert float((mu1 - mu0).abs().max()) > 1e-4
    assert torch.equal(mu2, ref) and float((mu2 - mu1).abs().max()) > 1e-6


def test_policy_scale_factor_and_per_trial_cost_lengthscales(golden):
    """Sum_of_gaussians(scale_factor=...) (Policy.py:220-222, 252: folded into the operands the kernels see, gradients through the
    fold) and Expected_saturated_distance_from_trajectory(flg_var_lengthscales=True) (Cost_function.py:136-141) against the reference."""
    from mc_pilco_amd.policy_learning import Cost_function, Policy

    fx = golden("options")
    with quiet():
        pol = Policy.Sum_of_gaussians(state_dim=4, input_dim=2, num_basis=fx["sf_centers"].shape[0], lengthscales_init=fx["sf_ls"].reshape(-1),
                                      centers_init=fx["sf_centers"], weight_init=fx["sf_weight"], flg_squash=True, u_max=[3.0, 1.5],
                                      scale_factor=fx["sf_scale"], flg_drop=True, dtype=dtype, device=dev())
    u = pol(T(fx["sf_x"]), t=0, p_dropout=0.0)
    (u * T(fx["sf_wsum"])).sum().backward()
    assert relerr(u, fx["sf_u"]) < 1e-12
    assert relerr(pol.log_lengthscales.grad, fx["sf_g_log_ls"]) < 1e-10
    assert relerr(pol.centers.grad, fx["sf_g_centers"]) < 1e-10
    assert relerr(pol.f_linear.weight.grad, fx["sf_g_weight"]) < 1e-10
    st = T(fx["vl_states"]).requires_grad_(True)
    cf = Cost_function.Expected_saturated_distance_from_trajectory(target_traj=T(fx["vl_target"]), lengthscales=T(fx["vl_ls_all"]),
                                                                   flg_var_lengthscales=True, used_indeces=list(range(12)))
    c, s = cf(st, None, int(fx["vl_trial"]))
    c.backward()
    assert abs(float(c.detach()) - float(fx["vl_cost"])) < 1e-12 * abs(float(fx["vl_cost"]))
    assert abs(float(s) - float(fx["vl_std"])) < 1e-12 * abs(float(fx["vl_std"]))
    assert relerr(st.grad, fx["vl_grad"]) < 1e-12
    c0, _ = cf(T(fx["vl_states"]), None, 0)  # the other trial's lengthscales give another cost
    assert abs(float(c0) - float(c.detach())) > 1e-3


def test_policy_bias_seed_for_seed_parity_with_reference(golden):
    """Sum_of_gaussians_with_angles(flg_bias=True, flg_train_bias=True) through the drop-in MC_PILCO class, reference noise mode."""
    from mc_pilco_amd.policy_learning import MC_PILCO, Cost_function, Policy

    fx = golden("rollout_bias")
    ml = build_cartpole(fx, 0, False)
    c = sy.CARTPOLE
    B = fx["pol_centers"].shape[0]
    ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=fx["pol_ls"].reshape(-1), centers_init=fx["pol_centers"], weight_init=fx["pol_weight"], flg_squash=True,
                u_max=c["u_max"], flg_bias=True, bias_init=fx["pol_bias"], flg_train_bias=True, flg_drop=True, dtype=dtype, device=dev())
    with quiet():
        obj = MC_PILCO.MC_PILCO(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None, f_model_learning=lambda **kw: ml,
                                model_learning_par={}, f_rand_exploration_policy=Policy.Random_exploration,
                                rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=1.0, dtype=dtype),
                                f_control_policy=Policy.Sum_of_gaussians_with_angles, control_policy_par=ppar,
                                f_cost_function=Cost_function.Cart_pole_cost,
                                cost_function_par=dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0),
                                log_path=None, dtype=dtype, device=dev())
    obj.noise_mode = "reference"
    assert sorted(obj.control_policy.state_dict().keys()) == ["centers", "f_linear.bias", "f_linear.weight", "log_lengthscales"]
    M, Tn, p = fx["states"].shape[1], fx["states"].shape[0], float(fx["p_drop"])
    torch.manual_seed(int(fx["seed"]))
    st, inp = obj.apply_policy(particles_initial_state_mean=T(fx["x0_mean"]), particles_initial_state_var=T(fx["x0_var"]),
                               flg_particles_init_uniform=False, particles_init_up_bound=None, particles_init_low_bound=None,
                               flg_particles_init_multi_gauss=False, num_particles=M, T_control=Tn, p_dropout=p)
    cost, std = obj.cost_function(st, inp, 0)
    cost.backward()
    assert float((st.detach().cpu() - torch.as_tensor(fx["states"])).abs().max()) < 1e-8
    assert abs(float(cost.detach()) - float(fx["cost"])) < 1e-10 * abs(float(fx["cost"]))
    pol = obj.control_policy
    assert relerr(pol.f_linear.bias.grad, fx["g_bias"]) < 1e-7
    assert relerr(pol.centers.grad, fx["g_centers"]) < 1e-7 and relerr(pol.f_linear.weight.grad, fx["g_weight"]) < 1e-7


# ---- round 3: the last unpinned branches of the class surface -----------------------------------------------------------------------
def _nan_cost_class():
    from mc_pilco_amd.policy_learning import Cost_function

    class NanOnCalls(Cost_function.Cart_pole_cost):
        """The drop-in's cart-pole cost; the expected cost is NaN on the calls whose index is in ``nan_calls`` (the same wrapper the
        fixture generator puts around the reference's class)."""

        def __init__(self, nan_calls, **kw):
            super().__init__(**kw)
            self.nan_calls, self.calls = set(int(i) for i in nan_calls), 0

        def forward(self, states_sequence, inputs_sequence=None, trial_index=None, group=None, counts=None):
            cost, std = super().forward(states_sequence, inputs_sequence, trial_index, group, counts)
            k = self.calls
            self.calls += 1
            return (cost * float("nan") if k in self.nan_calls else cost), std

    return NanOnCalls


@pytest.mark.parametrize("tag", ["step", "init"])
def test_reinforce_policy_nan_branches_match_reference(golden, tag):
    """MC_PILCO.py:430-456, 479-501, 573-607 on the drop-in (reference noise mode, same seed): ten "try sampling again", the
    re-initialisation with the reference's own torch.rand draws, counters / optimizer / dropout reset and 4 fresh steps whose costs
    are the reference's to 1e-8; and the re-initialisation during the warm-up rollout."""
    from mc_pilco_amd.policy_learning import MC_PILCO, Policy

    fx = golden("policy_nan_trace")
    ml = build_cartpole(fx, 0, False)
    c = sy.CARTPOLE
    B = fx["pol_centers"].shape[0]
    ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=fx["pol_ls"].reshape(-1), centers_init=fx["pol_centers"], weight_init=fx["pol_weight"], flg_squash=True,
                u_max=c["u_max"], flg_drop=True, dtype=dtype, device=dev())
    with quiet():
        obj = MC_PILCO.MC_PILCO(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None, f_model_learning=lambda **kw: ml,
                                model_learning_par={}, f_rand_exploration_policy=Policy.Random_exploration,
                                rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=1.0, dtype=dtype),
                                f_control_policy=Policy.Sum_of_gaussians_with_angles, control_policy_par=ppar, f_cost_function=_nan_cost_class(),
                                cost_function_par=dict(nan_calls=[int(i) for i in fx[tag + "_nan_calls"]], target_state=T(c["cost_target"]),
                                                       lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0),
                                log_path=None, dtype=dtype, device=dev())
    obj.noise_mode = "reference"
    pol = obj.control_policy
    after = {}
    orig = pol.reinit

    def spy(**k):
        orig(**k)
        after.update(log_ls=pol.log_lengthscales.detach().cpu().numpy().copy(), centers=pol.centers.detach().cpu().numpy().copy(),
                     weight=pol.f_linear.weight.detach().cpu().numpy().copy())

    pol.reinit = spy
    torch.manual_seed(int(fx[tag + "_seed"]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        costs, stds, st, inp = obj.reinforce_policy(
            T_control=float(fx["T_control"]), num_particles=int(fx["M"]), trial_index=0, particles_initial_state_mean=T(fx["x0_mean"]),
            particles_initial_state_var=T(fx["x0_var"]), flg_particles_init_uniform=False, particles_init_up_bound=None,
            particles_init_low_bound=None, flg_particles_init_multi_gauss=False, f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)",
            num_step_print=100, opt_steps_list=[int(fx[tag + "_opt_steps"])], lr_list=[0.01], p_dropout_list=[0.25],
            policy_reinit_dict=dict(lenghtscales_par=np.ones(5), centers_par=np.array([np.pi, np.pi, np.pi, 1.0, 1.0]), weight_par=10.0))
    txt = buf.getvalue()
    assert txt.count("Cost is NaN: try sampling again") == int(fx[tag + "_n_retry"])
    assert txt.count("re-initialize control policy") == int(fx[tag + "_n_reinit"])
    assert txt.count("SE filter initialization: Cost is NaN") == int(fx[tag + "_n_init_reinit"])
    assert obj.cost_function.calls == int(fx[tag + "_cost_calls"])
    assert costs.shape == fx[tag + "_cost_list"].shape and np.all(np.isfinite(costs))
    for k in ("log_ls", "centers", "weight"):
        assert np.abs(after[k] - fx[tag + "_after_" + k]).max() < 1e-15, k  # the reference's own draws
    assert relerr(costs, fx[tag + "_cost_list"]) < 1e-8
    assert relerr(stds, fx[tag + "_std_list"]) < 1e-7
    assert float(np.abs(st - fx[tag + "_last_states"]).max()) < 1e-7 and float(np.abs(inp - fx[tag + "_last_inputs"]).max()) < 1e-7
    assert relerr(pol.centers, fx[tag + "_final_centers"]) < 1e-8
    assert relerr(pol.f_linear.weight, fx[tag + "_final_weight"]) < 1e-8
    assert relerr(pol.log_lengthscales, fx[tag + "_final_log_ls"]) < 1e-8


@pytest.mark.parametrize("name,deg", [("fit_trace_se", 0), ("fit_trace_se_poly2", 2)])
def test_fit_model_trajectories_match_reference(golden, name, deg):
    """GP_prior.fit_model on the device (analytic NLL gradient kernels, the caller's Adam) against the reference's own run: 20
    epochs on N=80, every hyper-parameter after every epoch and the loss of every epoch, to 1e-8 (GP_prior.py:179-230,
    Model_learning.py:398-421)."""
    from mc_pilco_amd.gpr_lib.GP_prior import GP_prior as GP
    from mc_pilco_amd.gpr_lib.GP_prior import Sparse_GP, Stationary_GP
    from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood

    fx = golden(name)
    rbf = dict(rbf_dict(6, fx["lengthscales"], float(fx["sigma_n"])), flg_train_lambda=True)
    with quiet():
        if deg == 0:
            gp = Stationary_GP.RBF(**rbf)
        else:
            pw = [fx["poly_w%d" % k] for k in range(1, deg + 1)]
            gp = GP.Sum_Independent_GP(Stationary_GP.RBF(**rbf), Sparse_GP.get_Volterra_MPK_GP(**mpk_dict(6, deg, pw)))
    names = [str(n) for n in fx["names"]]
    assert sorted(n for n, p in gp.named_parameters() if p.requires_grad) == sorted(names)  # the reference's parameter names
    traj = {n: [dict(gp.named_parameters())[n].detach().cpu().numpy().copy()] for n in names}
    losses = []

    class Crit(Likelihood.Marginal_log_likelihood):
        def loss_and_grad(self, gp_, X, Y):
            loss = super().loss_and_grad(gp_, X, Y)
            losses.append(loss)
            return loss

    def snap():
        for n, p in gp.named_parameters():
            if n in traj:
                traj[n].append(p.detach().cpu().numpy().copy())

    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(T(fx["X"]), T(fx["Y"])), batch_size=80, shuffle=False)
    with quiet():
        gp.fit_model(trainloader=loader, optimizer=torch.optim.Adam(gp.parameters(), lr=float(fx["lr"])), criterion=Crit(),
                     N_epoch=int(fx["n_epoch"]), N_epoch_print=1, f_print=snap)
    got = np.array([float(l) for l in losses])
    assert got.shape == fx["losses"].shape and np.abs(got - fx["losses"]).max() < 1e-8 * np.abs(fx["losses"]).max()
    for n in names:
        ref = fx["traj__" + n]
        g = np.stack(traj[n]).reshape(ref.shape)
        assert np.abs(g - ref).max() < 1e-8, n


def test_get_model_learning_performance_and_squashing(golden):
    """MC_PILCO.get_model_learning_performance (MC_PILCO.py:260-306) and Policy.squashing (Policy.py:52-60): the two public methods
    the round-2 signature diff found missing.  One-step predictions on the training trajectory itself: targets = the data's
    velocity increments, MSE near the noise level, variances positive."""
    fx = golden("rollout_se")
    ml = build_cartpole(fx, 0, False)
    obj = build_mcpilco(fx, ml, fx["pol_centers"].shape[0])
    obj.state_samples_history.append(np.asarray(fx["states_tr"]))
    obj.input_samples_history.append(np.asarray(fx["inputs_tr"]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gp_inputs, targets, means, variances = obj.get_model_learning_performance(0)
    assert buf.getvalue().count("MSE gp") == 2
    n = fx["states_tr"].shape[0] - 1
    assert gp_inputs.shape == (n, 6) and len(targets) == len(means) == len(variances) == 2
    for g, v in enumerate((1, 3)):
        assert np.abs(targets[g].reshape(-1) - (fx["states_tr"][1:, v] - fx["states_tr"][:-1, v])).max() < 1e-12
        assert ((targets[g] - means[g]) ** 2).mean() < 4.0 * float(fx["sigma_n"]) ** 2
        assert bool((variances[g] > 0).all())
    pol = obj.control_policy
    u = T([[-30.0], [-1.0], [0.0], [0.5], [40.0]])
    assert float((pol.squashing(u, 10.0) - 10.0 * torch.tanh(u / 10.0)).abs().max()) == 0.0
    u2 = T([[1.0, -3.0], [0.2, 9.0]])
    assert float((pol.squashing(u2, [1.0, 2.0]) - T([1.0, 2.0]) * torch.tanh(u2 / T([1.0, 2.0]))).abs().max()) == 0.0


def test_zero_predictive_variance_raises_like_the_reference_normal():
    """Model_learning.py:704 samples with Normal(mean, sqrt(var)).rsample(): torch's argument validation raises ValueError on a scale
    that is not > 0.  The kernels report it as MCP_STATUS_NONPOS_VAR; the drop-in raises the same exception type (a NaN cost keeps
    taking the retry path, MC_PILCO.py:497).  One training point exactly at the particles' GP input, no noise -> var == 0 exactly."""
    import types

    from gpu_helpers import G, spec_from
    from mc_pilco_amd import hipabi, ops
    from mc_pilco_amd.policy_learning import MC_PILCO

    S, U, B = 4, 1, 16
    X = np.zeros((16, 6))
    X[:, 4] = 1.0            # z = [x0, x1, x3, sin x2, cos x2, u] at x = 0, u = 0
    X[1:, 0] = 50.0 + np.arange(15)  # the other rows far away (k = 0 there)
    sp = spec_from(np.ones(6), 0.0)
    Kinv = np.eye(16)        # k(z, X) = e_0 (lambda = 1)  ->  var = 1 - 1 = 0
    gp = ops.PackedGP(sp, G(X), G(np.zeros(16)), G(Kinv))
    model = ops.PackedModel([gp, gp], S, U, 0.05, [2], [0, 1, 3], [1, 3], [0, 2])
    pol = ops.PackedPolicy("angles", S, torch.log(G(np.ones((1, 5)))), G(np.zeros((B, 5))), G(np.zeros((U, B))), 10.0, True, angle=[2], non_angle=[0, 1, 3])
    x0 = G(np.zeros((8, S)))
    with torch.no_grad():
        st, inp, status = ops.rollout(model, pol, ops.NoiseSpec(seed=1, call=1), x0, 2, 0.0)
    assert int(status.item()) & hipabi.STATUS_NONPOS_VAR and not (int(status.item()) & hipabi.STATUS_NAN)
    assert bool(torch.isfinite(st).all())
    stub = types.SimpleNamespace(gp_sharding=True)
    stub._judge_attempt = lambda *a: MC_PILCO.MC_PILCO._judge_attempt(stub, *a)
    with pytest.raises(ValueError):
        MC_PILCO.MC_PILCO._rollout_failed(stub, torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64))
    with pytest.raises(ValueError):  # a FINITE variance <= 0 raises whatever the cost came out as (sqrt of a negative variance makes it NaN)
        MC_PILCO.MC_PILCO._rollout_failed(stub, torch.tensor([1.0, 0.0, 1.0], dtype=torch.float64))
    assert MC_PILCO.MC_PILCO._rollout_failed(stub, torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)) is True   # NaN alone: the retry path
    assert MC_PILCO.MC_PILCO._rollout_failed(stub, torch.tensor([0.0, 0.0, 0.0], dtype=torch.float64)) is False
    # a NaN state (divergence) reaches the GP as a NaN variance: MCP_STATUS_NAN only -- the retry case, not the ValueError
    x_nan = G(np.zeros((8, S)))
    x_nan[3, 1] = float("nan")
    with torch.no_grad():
        _, _, status = ops.rollout(model, pol, ops.NoiseSpec(seed=1, call=1), x_nan + 1.0, 2, 0.0)
    assert int(status.item()) & hipabi.STATUS_NAN and not (int(status.item()) & hipabi.STATUS_NONPOS_VAR)


# ---- round 4: the optimizer loop without a host sync per step ------------------------------------------------------------------------
def _nan_on_rollout_calls(obj, nan_calls):
    """The rollouts whose number (obj._rollout_calls, the counter that keys the in-kernel noise) is in ``nan_calls`` start from NaN particles, so
    their cost is NaN -- a key that means the same attempt at every pipeline depth.  Round 6: an attempt replayed from a HIP graph keeps that
    counter on the device (mcp_noise.call_dev), and so must this decision: a factor of NaN or exactly 1 on x0, chosen by a device comparison
    while ``obj._call_dev`` is set, by the host's counter otherwise."""
    inner = obj.sample_initial_particles
    nan_calls = set(int(i) for i in nan_calls)
    nan_t = torch.tensor(sorted(nan_calls), dtype=torch.int64, device=dev()).reshape(1, -1)

    def wrapped(*a, **k):
        x0 = inner(*a, **k)
        if obj._call_dev is not None:  # (advanced at the top of the recorded attempt: it holds THIS rollout's number)
            bad = (obj._call_dev.reshape(1, 1) == nan_t).any()
            return x0 * torch.where(bad, torch.full_like(x0[:1, :1], float("nan")), torch.ones_like(x0[:1, :1]))
        return x0 * float("nan") if (obj._rollout_calls + 1) in nan_calls else x0  # (the counter advances behind the draw)

    obj.sample_initial_particles = wrapped


@pytest.mark.parametrize("case", ["plain", "nan_retry", "reinit", "lr_and_exit", "last_step_fails"])
def test_pipelined_optimizer_loop_takes_exactly_the_synchronous_steps(golden, case):
    """reinforce_policy with the in-kernel noise, reading each attempt's outcome one attempt late (pipeline_depth 1, the default) against
    reading it at once (depth 0 -- the mode the reference-noise tests above pin to the reference's own traces): the same cost list,
    the same final parameters, the same returned particles and the same messages, BIT FOR BIT, through NaN retries, the
    re-initialisation after ten failures, learning-rate halvings with dropout reduction, the early exit, and a retry on the very last
    step; afterwards the default generator and the rollout counter stand where the synchronous run leaves them."""
    fx = golden("policy_opt_trace")
    kw = dict(opt_steps_list=[7], lr_list=[0.01], p_dropout_list=[0.25])
    nan_calls = []
    if case == "nan_retry":
        nan_calls = [3, 4, 7]                     # two failures of step 1 (rollouts 3, 4), one of step 3
    elif case == "reinit":
        nan_calls = list(range(4, 14))            # ten failures of step 2: re-initialisation, then 7 fresh steps
    elif case == "lr_and_exit":
        kw = dict(opt_steps_list=[12], lr_list=[0.01], p_dropout_list=[0.25], alpha_diff_cost=0.9, lr_reduction_ratio=0.5, lr_min=0.004,
                  p_drop_reduction=0.125, min_diff_cost=1e9, num_min_diff_cost=2, min_step=0)
    elif case == "last_step_fails":
        nan_calls = [8]                           # (rollout 1 is the warm-up: rollout 8 is the first attempt of step 6, the last)
    out = {}
    for depth in (0, 1):
        obj = _trace_setup(fx)
        obj.noise_mode = "philox"
        obj.pipeline_depth = depth
        obj.capture_attempts = True  # (opt-in since round 6's measurement: a replay is no faster than the eager launches on this runtime; depth 0 never replays)
        _nan_on_rollout_calls(obj, nan_calls)
        torch.manual_seed(1234)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = obj.reinforce_policy(
                T_control=float(fx["T_control"]), num_particles=int(fx["M"]), trial_index=0, particles_initial_state_mean=T(fx["x0_mean"]),
                particles_initial_state_var=T(fx["x0_var"]), flg_particles_init_uniform=False, particles_init_up_bound=None,
                particles_init_low_bound=None, flg_particles_init_multi_gauss=False, f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)",
                num_step_print=3, policy_reinit_dict=dict(lenghtscales_par=np.ones(5), centers_par=np.array([np.pi, np.pi, np.pi, 1.0, 1.0]),
                                                          weight_par=10.0), **kw)
        pol = obj.control_policy
        import re
        # round 6: the pipelined loop replays its attempts from HIP graphs (the first two after every restart run eagerly); the synchronous one never
        assert (obj.attempts_replayed > 0) == (depth == 1), (depth, obj.attempts_replayed)
        assert obj._call_dev is None
        out[depth] = dict(res=res, calls=obj._rollout_calls, rng=torch.cuda.get_rng_state(dev()).clone(), replayed=obj.attempts_replayed,
                          par=[q.detach().cpu().numpy().copy() for q in pol.parameters()],
                          txt=re.sub(r"time elapsed:  [0-9.e+-]+", "time elapsed", buf.getvalue()))
    a, b = out[0], out[1]
    print("%s: %d attempts in all, %d of them graph replays in the pipelined run" % (case, b["calls"] - 1, b["replayed"]))
    for x, y in zip(a["res"], b["res"]):
        assert x.shape == y.shape and np.array_equal(x, y, equal_nan=True)
    for x, y in zip(a["par"], b["par"]):
        assert np.array_equal(x, y)
    assert a["calls"] == b["calls"] and torch.equal(a["rng"], b["rng"])
    assert a["txt"] == b["txt"]
    costs = a["res"][0]
    assert np.all(np.isfinite(costs))
    if case == "plain":
        assert costs.shape == (7,) and "Cost is NaN" not in a["txt"]
    if case == "nan_retry":
        assert costs.shape == (7,) and a["txt"].count("try sampling again") == 3 and a["calls"] == 1 + 7 + 3
    if case == "reinit":
        assert costs.shape == (7,) and a["txt"].count("try sampling again") == 10 and a["txt"].count("re-initialize control policy") == 1
        assert a["calls"] == 1 + 2 + 10 + 7
    if case == "lr_and_exit":
        assert a["txt"].count("REDUCING THE LEARNING RATE") >= 1 and "EXIT FROM OPTIMIZATION" in a["txt"] and costs.shape[0] < 12
    if case == "last_step_fails":
        assert costs.shape == (7,) and a["txt"].count("try sampling again") == 1 and a["calls"] == 1 + 7 + 1


def test_optimizer_loop_with_another_optimizer_updates_through_its_own_step(golden):
    """f_optimizer is the user's: anything but the textbook Adam keeps its own ``step()`` (called once the host has seen that the
    attempt counts) -- SGD here, with a failed attempt in the middle: the update of a failed attempt must not happen."""
    fx = golden("policy_opt_trace")
    obj = _trace_setup(fx)
    obj.noise_mode = "philox"
    _nan_on_rollout_calls(obj, [3])
    w0 = obj.control_policy.f_linear.weight.detach().clone()
    torch.manual_seed(5)
    with quiet():
        costs, _, _, _ = obj.reinforce_policy(
            T_control=float(fx["T_control"]), num_particles=int(fx["M"]), trial_index=0, particles_initial_state_mean=T(fx["x0_mean"]),
            particles_initial_state_var=T(fx["x0_var"]), flg_particles_init_uniform=False, particles_init_up_bound=None,
            particles_init_low_bound=None, flg_particles_init_multi_gauss=False, f_optimizer="lambda p, lr : torch.optim.SGD(p, lr)",
            num_step_print=100, opt_steps_list=[4], lr_list=[0.01], p_dropout_list=[0.25], policy_reinit_dict=None)
    assert costs.shape == (4,) and np.all(np.isfinite(costs)) and obj._rollout_calls == 1 + 4 + 1
    w1 = obj.control_policy.f_linear.weight.detach()
    assert bool(torch.isfinite(w1).all()) and float((w1 - w0).abs().max()) > 0


# ---- round 4: GP hyper-parameter training of all GPs of a model at once ---------------------------------------------------------------
@pytest.mark.parametrize("deg", [0, 2])
def test_batched_gp_training_equals_training_the_gps_one_by_one(golden, deg):
    """Model_learning.reinforce_model trains the G GPs of a model epoch-synchronously (mcp_nll_epoch: every stage one launch whose grid
    carries the GP index, one Adam launch for all parameters) where the reference trains them one after the other
    (Model_learning.py:149-161).  The GPs are independent: the batched run must leave exactly the hyper-parameters, the cached alpha and
    K^-1 of the sequential one (each GP through GP_prior.fit_model on its own) -- bit for bit -- and print the same text."""
    import re

    from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood

    fx = golden("rollout_se_poly2" if deg else "rollout_se")
    opt = dict(f_optimizer="lambda p : torch.optim.Adam(p, lr=0.01)", criterion=Likelihood.Marginal_log_likelihood, N_epoch=12, N_epoch_print=5)
    res = {}
    for mode in ("batched", "one_by_one"):
        ml = build_cartpole(fx, deg, False)
        ml.set_training_mode()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            if mode == "batched":
                ml.reinforce_model([opt, opt])
            else:
                ml.init_gp_models()
                for i in range(ml.num_gp):
                    ml.train_gp(gp_index=i, optimization_opt_dict=opt)  # (GP_prior.fit_model of ONE GP)
                    with torch.no_grad():
                        ml.pretrain_gp(gp_index=i)
        res[mode] = dict(par=[{k: v.detach().cpu().numpy().copy() for k, v in gp.named_parameters()} for gp in ml.gp_list],
                         alpha=[a.detach().cpu().numpy().copy() for a in ml.alpha_list], kinv=[k.detach().cpu().numpy().copy() for k in ml.K_X_inv_list],
                         txt=re.sub(r"Time elapsed: [0-9.e+-]+", "Time elapsed", buf.getvalue()))
    a, b = res["batched"], res["one_by_one"]
    for pa, pb in zip(a["par"], b["par"]):
        assert pa.keys() == pb.keys()
        for k in pa:
            assert np.array_equal(pa[k], pb[k]), k
    for x, y in zip(a["alpha"] + a["kinv"], b["alpha"] + b["kinv"]):
        assert np.array_equal(x, y)
    assert a["txt"] == b["txt"] and a["txt"].count("EPOCH:") == 2 * 3 and a["txt"].count("Final parameters") == 2
    # and training moved the parameters
    ml0 = build_cartpole(fx, deg, False)
    moved = max(float(np.abs(pa[k] - v.detach().cpu().numpy()).max()) for pa, gp in zip(a["par"], ml0.gp_list) for k, v in gp.named_parameters()
                if v.requires_grad or True)
    assert moved > 1e-3


def test_gp_training_with_another_optimizer_keeps_the_callers_step():
    """fit_model with anything but the textbook Adam runs the caller's optimizer on the analytic gradient (the legacy path)."""
    from mc_pilco_amd.gpr_lib.GP_prior import Stationary_GP
    from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood

    rng = np.random.RandomState(0)
    X = T(rng.randn(40, 6))
    Y = T(np.sin(rng.randn(40, 1)))
    with quiet():
        gp = Stationary_GP.RBF(**rbf_dict(6, np.ones(6), 0.3))
        p0 = gp.log_lengthscales_par.detach().clone()
        gp.fit_model(trainloader=[(X, Y)], optimizer=torch.optim.SGD(gp.parameters(), lr=1e-3), criterion=Likelihood.Marginal_log_likelihood(),
                     N_epoch=3, N_epoch_print=10)
    assert float((gp.log_lengthscales_par.detach() - p0).abs().max()) > 0 and bool(torch.isfinite(gp.log_lengthscales_par).all())
