"""CPU: the host-side pieces of the public surface added in round 4, against the reference's own outputs
(tests/golden/surface_r4.npz, written by tests/golden/make_golden_r4.py from the reference): exploration policies (their np.random
draw order is the contract), the two cost functions as plain functions under the generic Expected_cost, MPK_GP's regressor-space
helpers, the pendulum ODE."""
import os

import numpy as np
import torch

import mcp_boot  # noqa: F401
from mc_pilco_amd.gpr_lib.GP_prior import Sparse_GP
from mc_pilco_amd.policy_learning import Cost_function, Policy
from mc_pilco_amd.simulation_class import ode_systems

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surface_r4.npz"))
DT = torch.float64
CPU = torch.device("cpu")


def test_sum_of_sinusoids_draws_and_signal_match_the_reference():
    np.random.seed(7)
    pol = Policy.Sum_of_sinusoids(state_dim=4, input_dim=2, num_sin=5, omega_min=0.5, omega_max=6.0, amplitude_min=[0.2, 1.0],
                                  amplitude_max=[3.0, 4.0], flg_squash=True, u_max=2.5, dtype=DT, device=CPU)
    for k in ("amplitudes", "omega", "phases"):  # bit-exact: the same np.random calls in the same order
        assert np.array_equal(getattr(pol, k).detach().numpy(), FX["sos_" + k]), k
    assert not any(p.requires_grad for p in pol.parameters())
    f = pol.get_np_policy()
    for i, t in enumerate(FX["sos_t"]):
        np.testing.assert_allclose(pol(torch.zeros(1, 4, dtype=DT), t).numpy(), FX["sos_u"][i], rtol=0, atol=1e-15)
        np.testing.assert_allclose(f(np.zeros((1, 4)), t), FX["sos_np_u"][i], rtol=0, atol=1e-15)
    assert np.array_equal(np.random.rand(3), FX["sos_after"])  # the generator is where the reference leaves it
    np.random.seed(11)
    pol2 = Policy.Sum_of_sinusoids(state_dim=4, input_dim=1, num_sin=3, omega_min=1.0, omega_max=2.0, amplitude_min=0.5, amplitude_max=1.5,
                                   flg_squash=False, u_max=1, dtype=DT, device=CPU)
    for i, t in enumerate(FX["sos_t"]):
        np.testing.assert_allclose(pol2(None, t).numpy(), FX["sos2_u"][i], rtol=0, atol=1e-15)


def test_pd_controller_matches_the_reference():
    pd = Policy.PD_controller(state_dim=6, input_dim=3, sqrt_Kp_gains=np.array([2.0, 1.5, 1.0]), sqrt_Kd_gains=np.array([0.5, 0.4, 0.3]),
                              target_traj=torch.tensor(FX["pd_traj"], dtype=DT), flg_squash=True, u_max=np.array([1.0, 2.0, 3.0]), dtype=DT, device=CPU)
    xs = torch.tensor(FX["pd_states"], dtype=DT)
    for i, t in enumerate((0, 3, 9)):
        np.testing.assert_allclose(pd(xs, t).detach().numpy(), FX["pd_u"][i], rtol=1e-14, atol=1e-15)


def test_module_level_cost_functions_under_the_generic_expected_cost():
    st = torch.tensor(FX["cf_states"], dtype=DT, requires_grad=True)
    inp = torch.zeros(7, 9, 1, dtype=DT)
    tgt, ls = torch.tensor([np.pi, 0.0], dtype=DT), torch.tensor([3.0, 1.0], dtype=DT)
    np.testing.assert_allclose(Cost_function.cart_pole_cost(st, inp, 0, tgt, ls, 2, 0).detach().numpy(), FX["cp_costs"], rtol=1e-14, atol=1e-16)
    ec = Cost_function.Expected_cost(lambda x, u, k: Cost_function.cart_pole_cost(x, u, k, target_state=tgt, lengthscales=ls, angle_index=2,
                                                                                   pos_index=0))
    cost, std = ec(st, inp, 0)
    cost.backward()
    assert abs(float(cost) - float(FX["cp_cost"])) < 1e-13 and abs(float(std) - float(FX["cp_std"])) < 1e-13
    np.testing.assert_allclose(st.grad.numpy(), FX["cp_grad"], rtol=1e-12, atol=1e-16)

    st2 = torch.tensor(FX["cf_states"], dtype=DT, requires_grad=True)
    ttraj = torch.tensor(FX["sd_traj"], dtype=DT)
    ls_var = [torch.tensor(l, dtype=DT) for l in FX["sd_ls"]]
    ec2 = Cost_function.Expected_cost(lambda x, u, k: Cost_function.saturated_distance_from_trajectory(
        x, u, k, target_traj=ttraj, lengthscales=ls_var, flg_var_lengthscales=True, used_indeces=[0, 3]))
    cost2, std2 = ec2(st2, inp, 1)
    cost2.backward()
    assert abs(float(cost2) - float(FX["sd_cost"])) < 1e-13 and abs(float(std2) - float(FX["sd_std"])) < 1e-13
    np.testing.assert_allclose(st2.grad.numpy(), FX["sd_grad"], rtol=1e-12, atol=1e-16)
    all_ = Cost_function.saturated_distance_from_trajectory(st2, inp, 0, ttraj, torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=DT), False, None)
    np.testing.assert_allclose(all_.detach().numpy(), FX["sd_costs_all"], rtol=1e-14, atol=1e-16)


def test_mpk_regressor_space_helpers_match_the_reference():
    mpk = Sparse_GP.MPK_GP(active_dims=np.arange(3), poly_deg=2, Sigma_pos_par_init=FX["mpk_par"], flg_offset=False, dtype=DT, device=CPU)
    X = torch.tensor(FX["mpk_X"], dtype=DT)
    np.testing.assert_allclose(mpk.get_Sigma_deg(0).detach().numpy(), FX["mpk_Sigma_deg0"], rtol=1e-15)
    np.testing.assert_allclose(mpk.get_Sigma_deg(1).detach().numpy(), FX["mpk_Sigma_deg1"], rtol=1e-15)
    mpk.current_deg = 1
    np.testing.assert_allclose(mpk.get_Sigma().detach().numpy(), FX["mpk_Sigma_cur1"], rtol=1e-15)
    assert np.array_equal(mpk.get_phi(X).numpy(), FX["mpk_phi"])
    # the kernels' weights are the diagonals of these matrices (what mcp_kernel.w20 / w21 carry)
    w = mpk.factor_weights()
    np.testing.assert_allclose(w[0].numpy(), np.diag(FX["mpk_Sigma_deg0"]), rtol=1e-15)
    np.testing.assert_allclose(w[1].numpy(), np.diag(FX["mpk_Sigma_deg1"]), rtol=1e-15)
    # and phi Sigma_0 phi^T * phi Sigma_1 phi^T is the reference's Gram
    phi = FX["mpk_phi"]
    np.testing.assert_allclose((phi @ FX["mpk_Sigma_deg0"] @ phi.T) * (phi @ FX["mpk_Sigma_deg1"] @ phi.T), FX["mpk_K"], rtol=1e-13)
    mpk1 = Sparse_GP.MPK_GP(active_dims=np.arange(3), poly_deg=1, sigma_n_init=0.1 * np.ones(1), Sigma_pos_par_init=np.array([0.4, 0.6, 0.8, 1.2]),
                            flg_offset=True, dtype=DT, device=CPU)
    assert np.array_equal(mpk1.get_phi(X).numpy(), FX["mpk1_phi"])
    np.testing.assert_allclose(mpk1.get_Sigma().detach().numpy(), FX["mpk1_Sigma"], rtol=1e-15)
    np.testing.assert_allclose(mpk1.get_parameters_inv_lemma(X, torch.tensor(FX["mpk1_Y"], dtype=DT)).detach().numpy(), FX["mpk1_w_lemma"], rtol=1e-10)
    assert [n for n, _ in mpk1.named_parameters()] == ["sigma_n_log", "mean_par", "Sigma_pos_par"]  # state_dict order of the reference


def test_pendulum_ode_matches_the_reference():
    got = np.array([ode_systems.pend([0.3, -0.7], 0.0, 1.2), ode_systems.pend([2.0, 0.1], 0.0, np.array([[-0.4]]))])
    np.testing.assert_allclose(got, FX["pend"], rtol=1e-15)
