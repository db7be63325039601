"""CPU: the package can be imported through the reference's own module paths and both paths give the
same module objects (so isinstance checks and class handles agree)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_style_imports_alias_the_canonical_modules():
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import gpr_lib.GP_prior.Stationary_GP as SGP\n"
        "import gpr_lib.GP_prior.GP_prior as GP\n"
        "import gpr_lib.GP_prior.Sparse_GP as SP\n"
        "import gpr_lib.Likelihood.Gaussian_likelihood as Likelihood\n"
        "import model_learning.Model_learning as ML\n"
        "import policy_learning.Policy as Policy\n"
        "import policy_learning.Cost_function as Cost_function\n"
        "import policy_learning.MC_PILCO as MC_PILCO\n"
        "import simulation_class.ode_systems as f_ode\n"
        "import mc_pilco_amd.gpr_lib.GP_prior.Stationary_GP as C1\n"
        "import mc_pilco_amd.policy_learning.MC_PILCO as C2\n"
        "assert SGP is C1 and MC_PILCO is C2\n"
        "assert ML.Speed_Model_learning_RBF_angle_state.__mro__[1] is ML.Model_learning\n"
        "for n in ['RBF']: assert hasattr(SGP, n)\n"
        "for n in ['Sum_Independent_GP', 'GP_prior', 'Combine_GP']: assert hasattr(GP, n)\n"
        "for n in ['MPK_GP', 'get_Volterra_MPK_GP', 'Linear_GP']: assert hasattr(SP, n)\n"
        "for n in ['Model_learning', 'Model_learning_RBF', 'Model_learning_RBF_angle_state', 'Model_learning_RBF_MPK_angle_state',"
        " 'Speed_Model_learning_RBF_angle_state', 'Speed_Model_learning_RBF_MPK_angle_state']: assert hasattr(ML, n)\n"
        "for n in ['Sum_of_gaussians', 'Sum_of_gaussians_with_angles', 'Sum_of_gaussians_with_target_trajectory', 'Random_exploration']:"
        " assert hasattr(Policy, n)\n"
        "for n in ['Expected_cost', 'Cart_pole_cost', 'Expected_saturated_distance_from_trajectory', 'Expected_distance',"
        " 'Expected_saturated_distance']: assert hasattr(Cost_function, n)\n"
        "assert hasattr(MC_PILCO.MC_PILCO, 'reinforce') and hasattr(MC_PILCO.MC_PILCO, 'apply_policy') and hasattr(MC_PILCO.MC_PILCO, 'reinforce_policy')\n"
        "assert hasattr(Likelihood, 'Marginal_log_likelihood') and callable(f_ode.cartpole)\n"
        "print('OK')\n"
    ) % os.path.join(ROOT, "mc-pilco_amd")
    out = subprocess.check_output([sys.executable, "-c", code], stderr=subprocess.STDOUT).decode()
    assert out.strip().endswith("OK"), out


def test_signatures_match_the_reference_surface():
    """Keyword names (including the reference's spellings) of the constructors / methods the launch scripts use."""
    import inspect

    from mc_pilco_amd.gpr_lib.GP_prior import Sparse_GP, Stationary_GP
    from mc_pilco_amd.model_learning import Model_learning as ML
    from mc_pilco_amd.policy_learning import MC_PILCO, Cost_function, Policy

    def names(f):
        return [p for p in inspect.signature(f).parameters if p != "self"]

    assert names(Stationary_GP.RBF.__init__) == ["active_dims", "lengthscales_init", "flg_train_lengthscales", "sigma_n_init", "flg_train_sigma_n",
                                                  "lambda_init", "flg_train_lambda", "mean_init", "flg_train_mean", "name", "dtype", "sigma_n_num",
                                                  "device"]
    assert names(Sparse_GP.get_Volterra_MPK_GP) == ["active_dims", "poly_deg", "sigma_n_init", "flg_train_sigma_n", "Sigma_pos_par_init_list",
                                                    "flg_train_Sigma_pos_par_list", "name", "dtype", "sigma_n_num", "device"]
    assert names(ML.Speed_Model_learning_RBF_angle_state.__init__) == ["num_gp", "init_dict_list", "T_sampling", "angle_indeces", "not_angle_indeces",
                                                                        "vel_indeces", "not_vel_indeces", "approximation_mode", "approximation_dict",
                                                                        "dtype", "device", "flg_norm"]
    assert names(ML.Model_learning.get_next_state) == ["current_state", "current_input", "particle_pred"]
    assert names(Policy.Sum_of_gaussians_with_angles.__init__)[:5] == ["state_dim", "input_dim", "num_basis", "angle_indices", "non_angle_indices"]
    assert names(Policy.Sum_of_gaussians.forward) == ["states", "t", "p_dropout"]
    assert names(Policy.Sum_of_gaussians.reinit) == ["lenghtscales_par", "centers_par", "weight_par"]
    assert names(Cost_function.Cart_pole_cost.__init__) == ["target_state", "lengthscales", "angle_index", "pos_index"]
    assert names(MC_PILCO.MC_PILCO.__init__) == ["T_sampling", "state_dim", "input_dim", "f_sim", "f_model_learning", "model_learning_par",
                                                 "f_rand_exploration_policy", "rand_exploration_policy_par", "f_control_policy", "control_policy_par",
                                                 "f_cost_function", "cost_function_par", "std_meas_noise", "log_path", "dtype", "device"]
    assert names(MC_PILCO.MC_PILCO.apply_policy) == ["particles_initial_state_mean", "particles_initial_state_var", "flg_particles_init_uniform",
                                                     "particles_init_up_bound", "particles_init_low_bound", "flg_particles_init_multi_gauss",
                                                     "num_particles", "T_control", "p_dropout"]
    rp = names(MC_PILCO.MC_PILCO.reinforce_policy)
    assert rp[:12] == ["T_control", "num_particles", "trial_index", "particles_initial_state_mean", "particles_initial_state_var",
                       "flg_particles_init_uniform", "particles_init_up_bound", "particles_init_low_bound", "flg_particles_init_multi_gauss",
                       "opt_steps_list", "lr_list", "f_optimizer"]
    assert "policy_reinit_dict" in rp and "p_dropout_list" in rp and "min_step" in rp
    assert names(MC_PILCO.MC_PILCO4PMS.__init__) == ["T_sampling", "state_dim", "input_dim", "f_sim", "f_model_learning", "model_learning_par",
                                                     "f_rand_exploration_policy", "rand_exploration_policy_par", "f_control_policy",
                                                     "control_policy_par", "f_cost_function", "cost_function_par", "pos_indeces", "vel_indeces",
                                                     "std_meas_noise", "log_path", "filtering_dict", "std_meas_noise_sim", "dtype", "device"]
    assert names(MC_PILCO.MC_PILCO4PMS.apply_policy) == names(MC_PILCO.MC_PILCO.apply_policy)
    assert names(MC_PILCO.MC_PILCO4PMS.get_velocities) == ["meas_states", "input_samples", "noiseless_samples", "noisy_samples"]


def test_pms_simulator_filter_matches_the_restated_formula():
    """PMS_Model.rollout: measured positions, backward-difference velocities, first-order Butterworth filter."""
    import numpy as np

    from mc_pilco_amd.simulation_class import model as sim
    from oracle import mcpilco_oracle as orc

    np.random.seed(3)
    f = lambda y, t, u: np.array([y[1], -y[0] + u[0]])
    m = sim.PMS_Model(f, {"fc": 0.5})
    meas, inputs, clean, noisy = m.rollout(s0=np.array([0.1, 0.0]), policy=lambda x, t: np.array([0.3 * np.sin(t)]), T=0.5, dt=0.05,
                                           noise=np.array([0.01, 0.0]), vel_indeces=[1], pos_indeces=[0])
    b, a = orc.butter1(0.5)
    assert meas.shape == (11, 2) and inputs.shape == (11, 1)
    for k in range(1, 11):
        nv = (meas[k, 0] - meas[k - 1, 0]) / 0.05
        assert abs(noisy[k, 1] - nv) < 1e-14
        prev_nv = noisy[k - 1, 1]
        assert abs(meas[k, 1] - (b[0] * nv + b[1] * prev_nv - a[1] * meas[k - 1, 1]) / a[0]) < 1e-12


def test_hand_off_timeout_switches_to_the_unsharded_kernels():
    """A GP-sharded rollout whose workgroups never met reports MCP_STATUS_SYNC: the optimisation loop repeats the step with the
    GP-sharded launch forms switched off (never a silently wrong trajectory, never a rank-local raise); a NaN cost alone is
    data: the reference re-samples (MC_PILCO.py:497)."""
    import torch

    import mcp_boot  # noqa: F401
    from mc_pilco_amd import hipabi
    from mc_pilco_amd.policy_learning.MC_PILCO import MC_PILCO

    obj = MC_PILCO.__new__(MC_PILCO)  # only the status logic is exercised
    obj.dtype, obj.gp_sharding = torch.float64, True
    obj.last_status = torch.zeros(1, dtype=torch.int32)
    assert obj._rollout_failed(obj._step_flags(torch.tensor(1.5, dtype=torch.float64))) is False
    assert obj._rollout_failed(obj._step_flags(torch.tensor(float("nan"), dtype=torch.float64))) is True
    obj.last_status = torch.tensor([hipabi.STATUS_NAN], dtype=torch.int32)
    assert obj._rollout_failed(obj._step_flags(torch.tensor(float("nan"), dtype=torch.float64))) is True
    assert obj.gp_sharding is True
    obj.last_status = torch.tensor([hipabi.STATUS_SYNC], dtype=torch.int32)
    assert obj._rollout_failed(obj._step_flags(torch.tensor(1.5, dtype=torch.float64))) is True  # the step is repeated ...
    assert obj.gp_sharding is False                                                                # ... on the unsharded kernels
    # flags that arrive summed over ranks (sharding.StepReducer) are read the same way
    obj.gp_sharding = True
    assert obj._rollout_failed(torch.tensor([0.0, 3.0, 0.0], dtype=torch.float64)) is True and obj.gp_sharding is False
    assert obj._rollout_failed(torch.tensor([0.0, 0.0, 0.0], dtype=torch.float64)) is False
