"""Synthetic MC-PILCO workloads of BASELINE.json's shapes, built on the GPU through the HIP
pretrain path (Gram -> Cholesky -> inverse -> alpha -> pack).  Shared by bench.py,
__graft_entry__.smoke() and the tests; ``numpy_problem`` gives the same problem as plain arrays
so the CPU oracle can be run on identical inputs.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import ops
from . import synthetic as sy

DT = torch.float64

CONFIGS = {
    # name: (system, poly_deg, N, M, T)   -- BASELINE.json configs
    "c1": ("cartpole", 0, 300, 400, 150),   # configs[0]/[1]: cart-pole SE kernel, M=400, T=150, N~300
    "c1_script": ("cartpole", 0, 300, 400, 60),  # the launch script's own horizon int(3.0/0.05)
    "c3": ("cartpole", 2, 300, 4000, 150),  # configs[2]: SE + polynomial(2), M=4000
    "c4": ("cartpole", 2, 300, 4000, 150),  # configs[3]: the c3 model, M=32000 over 8 GPUs = 4000 particles per GPU (bench.py --gpus N)
    "c5": ("ur5", 1, 400, 2000, 300),       # configs[4]: UR5 12-D state, 6 GPs, SE + polynomial(1)
    # the reference's own small-swarm launch scripts (their shapes; N = the data of the last trials)
    "c2_script": ("cartpole", 2, 300, 400, 60),   # test_mcpilco_cartpole.py:51-53,101,199: SE + polynomial(2), M=400, T=3.0/0.05
    "pms_script": ("cartpole", 0, 300, 400, 90),  # test_mcpilco4pms_cartpole.py:51-53,155-157,171: SE, M=400, T=3.0/(1/30), measured states
    "ur5_script": ("ur5", 1, 400, 200, 200),      # test_mcpilco_ur5_mujoco.py:58-59,102,195: 6 GPs, D=24, SE + polynomial(1), M=200, T=4.0/0.02
    "pms_script_n450": ("cartpole", 0, 450, 400, 90),  # the same script's LAST trial: exact GP, 90 samples per trial, no subset (:50-53,64-88) -> N = 450
    "c1_script_n360": ("cartpole", 0, 360, 400, 60),   # the cart-pole scripts' last trial without a subset: 60 + 5 x 60 samples
    "c2_script_n360": ("cartpole", 2, 360, 400, 60),
    "c2p1_script": ("cartpole", 1, 300, 400, 60),  # the same with a degree-1 Volterra term (tests: the lean kernel's MAXDEG = 1 instantiations)
    "ur5_se": ("ur5", 0, 400, 200, 200),            # (tests: the wide class's SE-only instantiations)
    "tiny": ("cartpole", 0, 48, 16, 6),
    "tiny_ur5": ("ur5", 1, 40, 8, 5),
}
# per-workload overrides: sampling time of the data / model, measurement model of MC_PILCO4PMS (pos, vel, noise std, filter cutoff)
OPTIONS = {
    "pms_script": dict(Ts=1.0 / 30.0, pms=dict(pos=[0, 2], vel=[1, 3], std=3e-3, fc=0.5)),
    "pms_script_n450": dict(Ts=1.0 / 30.0, pms=dict(pos=[0, 2], vel=[1, 3], std=3e-3, fc=0.5)),
}


def numpy_problem(name, N=None, seed=1):
    """Plain-array description of a workload: training data, hyper-parameters, policy init."""
    system, deg, N0, M, T = CONFIGS[name]
    N = N or N0
    rng = np.random.RandomState(seed + 100)
    opt = OPTIONS.get(name, {})
    if system == "cartpole":
        c = dict(sy.CARTPOLE, Ts=opt.get("Ts", sy.CARTPOLE["Ts"]))
        n_roll = (N + 59) // 60
        Z, Ys = sy.gp_io(sy.cartpole_rollouts(n_roll=n_roll, seed=seed, Ts=c["Ts"]), c["angle"], c["not_angle"], c["vel"])
        pol = sy.cartpole_policy_init(B=c["B"], u_max=c["u_max"], seed=seed)
        kind, extra = "angles", dict(angle=[2], non_angle=[0, 1, 3])
        target = None
    else:
        c = sy.UR5
        n_roll = (N + 199) // 200
        Z, Ys = sy.gp_io(sy.ur5_rollouts(n_roll=n_roll, seed=seed), c["angle"], c["not_angle"], c["vel"])
        pol = sy.ur5_policy_init(B=c["B"], seed=seed)
        kind, extra = "traj", {}
        target = sy.ur5_target_traj(T=T, Ts=c["Ts"])
    Z, Ys = Z[:N], [y[:N] for y in Ys]
    poly = None
    if deg >= 1:
        # "trained-like" small polynomial weights (exp(par) = 0.01), slightly perturbed per GP
        poly = []
        for _g in range(c["G"]):
            w = [0.01 * (0.8 + 0.4 * rng.rand(c["D"] + 1))]
            if deg >= 2:
                w.append(0.01 * (0.8 + 0.4 * rng.rand(2 * c["D"])))
            poly.append(w)
    return dict(name=name, system=system, cfg=c, deg=deg, N=N, M=M, T=T, Z=Z, Ys=Ys, poly=poly, policy=pol, policy_kind=kind,
                policy_extra=extra, target_traj=target, pms=opt.get("pms"))


@dataclass
class Workload:
    name: str
    model: ops.PackedModel
    policy: ops.PackedPolicy
    cost: ops.PackedCost
    params: List[torch.Tensor]  # [log_lengthscales [1,P], centers [B,P], weight [U,B]] leaf tensors
    x0_mean: torch.Tensor
    x0_std: torch.Tensor
    M: int
    T: int
    p_drop: float
    problem: dict
    meas: Optional[ops.MeasSpec] = None  # measurement model between particles and policy (MC_PILCO4PMS), None: the true state

    def sample_x0(self, M=None, generator=None):
        M = M or self.M
        e = torch.randn(M, self.x0_mean.numel(), dtype=DT, device=self.x0_mean.device, generator=generator)
        return torch.addcmul(self.x0_mean, self.x0_std, e)


def spec_for(c, sigma_n, poly_w):
    w1 = w20 = w21 = None
    if poly_w:
        w1 = ops.mpk_weights(np.log(poly_w[0]), 1)[0]
        if len(poly_w) > 1:
            w20, w21 = ops.mpk_weights(np.log(poly_w[1]), 2)
    return ops.KernelSpec(torch.as_tensor(c["lengthscales"], dtype=DT), float(c["lam"]), float(sigma_n) ** 2, 0.0, w1, w20, w21)


def pretrain_packed(spec, Z, Y, device):
    """GP_prior.forward + get_alpha on the device, then the kernels' packed layout."""
    Zg = torch.as_tensor(Z, dtype=DT).to(device).contiguous()
    K = ops.cov_build(spec, Zg, None, noise=True)
    U, logdet, status = ops.chol_factor(K)
    if int(status.item()) != 0:
        raise RuntimeError("Gram matrix is not positive definite")
    _, Kinv = ops.chol_inverse(U)
    alpha = ops.gp_alpha(Kinv, torch.as_tensor(Y, dtype=DT).to(device), spec.mean)
    return ops.PackedGP(spec, Zg, alpha, Kinv)


def build(name, device=None, M=None, T=None, N=None, p_drop=0.25, seed=1) -> Workload:
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    pb = numpy_problem(name, N=N, seed=seed)
    c = pb["cfg"]
    M = M or pb["M"]
    T = T or pb["T"]
    if pb["target_traj"] is not None and pb["target_traj"].shape[0] != T:
        pb["target_traj"] = sy.ur5_target_traj(T=T, Ts=c["Ts"])
    gps = []
    for g in range(c["G"]):
        spec = spec_for(c, c["sigma_n"], None if pb["poly"] is None else pb["poly"][g])
        gps.append(pretrain_packed(spec, pb["Z"], pb["Ys"][g], device))
    model = ops.PackedModel(gps, c["S"], c["U"], c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])
    pi = pb["policy"]
    log_ls = torch.log(torch.as_tensor(pi["lengthscales"], dtype=DT)).reshape(1, -1).to(device).requires_grad_(True)
    centers = torch.as_tensor(pi["centers"], dtype=DT).to(device).contiguous().requires_grad_(True)
    weight = torch.as_tensor(pi["weight"], dtype=DT).to(device).contiguous().requires_grad_(True)
    policy = ops.PackedPolicy(pb["policy_kind"], c["S"], log_ls, centers, weight, c["u_max"], True, target_traj=pb["target_traj"],
                              **pb["policy_extra"])
    if pb["system"] == "cartpole":
        cost = ops.PackedCost("cartpole", c["S"], device, target_state=c["cost_target"], lengthscales=c["cost_ls"],
                              angle_index=c["cost_angle_index"], pos_index=c["cost_pos_index"])
    else:
        cost = ops.PackedCost("traj", c["S"], device, target_traj=pb["target_traj"], lengthscales=c["cost_ls"], used=None)
    x0m = torch.as_tensor(c["x0_mean"], dtype=DT).to(device).reshape(1, -1)
    x0s = torch.sqrt(torch.as_tensor(c["x0_var"], dtype=DT)).to(device).reshape(1, -1)
    meas = None
    if pb.get("pms"):
        from scipy import signal

        q = pb["pms"]
        bb, aa = signal.butter(1, q["fc"])
        meas = ops.MeasSpec(pos=q["pos"], vel=q["vel"], std_pos=[q["std"]] * len(q["pos"]), b=bb, a=aa)
    return Workload(name, model, policy, cost, [log_ls, centers, weight], x0m, x0s, M, T, p_drop, pb, meas)


def policy_grad_step(w: Workload, x0, noise: ops.NoiseSpec, group=None):
    """One iteration of reinforce_policy's hot loop on the HIP path: fused rollout -> expected
    cost -> reverse-time adjoint.  Leaves gradients in w.params[i].grad (this rank's particles;
    already scaled by 1/M_total).  Returns (cost, std, status)."""
    for p in w.params:
        p.grad = None
    states, inputs, status = ops.rollout(w.model, w.policy, noise, x0, w.T, w.p_drop)
    cost, std = ops.expected_cost(w.cost, states, group)
    cost.backward()
    return cost.detach(), std.detach(), status


def flops_per_particle_step(w: Workload, forward_only=False):
    """SURVEY.md 8d algorithmic flops per particle-step (forward + backward; ``forward_only``: the forward terms alone -- the GP
    sums of the first bracket and the policy's forward half)."""
    P, B, U, D = w.policy.P, w.policy.B, w.policy.U, w.model.D
    f = (3 if forward_only else 6) * B * (P + U + 2)
    for gp in w.model.gps:
        N = gp.N
        deg = gp.spec.poly_deg
        fpoly = 0
        if deg >= 1:
            fpoly += (2 * (D + 1) + 2) * N
        if deg >= 2:
            fpoly += (4 * D + 3) * N
        f += 2 * N * N + (3 * D + 8) * N + fpoly  # forward
        if not forward_only:
            f += (6 * D + 6) * N + fpoly  # backward
    return f


def dropin_c1(device, pms=False, num_particles=400, T_control=7.5):
    """The headline workload (cart-pole SE, N=300, M=400, T=150, B=200) on the DROP-IN classes: an ``MC_PILCO`` object with a
    pretrained model and the keyword arguments of ``reinforce_policy`` -- what bench.py's ``loop_ms_per_step`` and
    tools/time_reinforce_policy.py time (the same work as one bench step plus the loop's monitors, NaN check and printing)."""
    import contextlib
    import io

    from .model_learning import Model_learning as ML
    from .policy_learning import MC_PILCO, Cost_function, Policy

    dt = DT
    c = sy.CARTPOLE
    Tt = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=device)
    rbf = dict(active_dims=np.arange(6), lengthscales_init=np.asarray(c["lengthscales"], dtype=float), flg_train_lengthscales=True,
               lambda_init=np.ones(1), flg_train_lambda=False, sigma_n_init=c["sigma_n"] * np.ones(1), sigma_n_num=None, flg_train_sigma_n=True,
               dtype=dt, device=device)
    mlp = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
               not_vel_indeces=c["not_vel"], dtype=dt, device=device, init_dict_list=[rbf] * 2)
    pi = sy.cartpole_policy_init(B=200, seed=1)
    ppar = dict(state_dim=4, input_dim=1, num_basis=200, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True, u_max=c["u_max"],
                flg_drop=True, dtype=dt, device=device)
    kw = dict(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None, f_model_learning=ML.Speed_Model_learning_RBF_angle_state,
              model_learning_par=mlp, f_rand_exploration_policy=Policy.Random_exploration,
              rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=10.0, dtype=dt), f_control_policy=Policy.Sum_of_gaussians_with_angles,
              control_policy_par=ppar, f_cost_function=Cost_function.Cart_pole_cost,
              cost_function_par=dict(target_state=Tt(c["cost_target"]), lengthscales=Tt(c["cost_ls"]), angle_index=2, pos_index=0), log_path=None,
              dtype=dt, device=device)
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
    args = dict(T_control=T_control, num_particles=num_particles, trial_index=0, particles_initial_state_mean=Tt(c["x0_mean"]),
                particles_initial_state_var=Tt(c["x0_var"]), flg_particles_init_uniform=False, particles_init_up_bound=None,
                particles_init_low_bound=None, flg_particles_init_multi_gauss=False, lr_list=[0.01],
                f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)", num_step_print=50, p_dropout_list=[0.25],
                policy_reinit_dict=dict(lenghtscales_par=np.ones(5), centers_par=np.array([np.pi, np.pi, np.pi, 1.0, 1.0]), weight_par=10.0))
    return obj, args


def time_reinforce_policy(device, steps=100, pms=False, T_control=7.5):
    """(seconds per optimizer step of MC_PILCO.reinforce_policy on the drop-in classes, first cost, last cost); T_control 7.5 s =
    150 steps (c1), 3.0 s = the launch script's 60."""
    import contextlib
    import io
    import time

    obj, args = dropin_c1(device, pms=pms, T_control=T_control)
    with contextlib.redirect_stdout(io.StringIO()):
        obj.reinforce_policy(opt_steps_list=[10], **args)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = obj.reinforce_policy(opt_steps_list=[steps], **args)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    return el / steps, float(out[0][0]), float(out[0][-1])


def time_fit_model(device, N=300, epochs=100):
    """Seconds per epoch per GP of GP hyper-parameter training on the device at the cart-pole size: Model_learning.reinforce_model
    (Model_learning.py:398-421 -> GP_prior.fit_model, GP_prior.py:179-230) on the drop-in classes, 2 GPs, Adam, full batch."""
    import contextlib
    import io
    import time

    from .gpr_lib.Likelihood import Gaussian_likelihood as Likelihood
    from .model_learning import Model_learning as ML

    c = sy.CARTPOLE
    rbf = dict(active_dims=np.arange(6), lengthscales_init=np.ones(6), flg_train_lengthscales=True, lambda_init=np.ones(1), flg_train_lambda=False,
               sigma_n_init=np.ones(1), sigma_n_num=None, flg_train_sigma_n=True, dtype=DT, device=device)
    par = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
               not_vel_indeces=c["not_vel"], dtype=DT, device=device, init_dict_list=[rbf] * 2)
    with contextlib.redirect_stdout(io.StringIO()):
        ml = ML.Speed_Model_learning_RBF_angle_state(**par)
        for xs, us in sy.cartpole_rollouts(n_roll=(N + 59) // 60):
            ml.add_data(np.asarray(xs), np.asarray(us))
        opt = dict(f_optimizer="lambda p : torch.optim.Adam(p, lr=0.01)", criterion=Likelihood.Marginal_log_likelihood, N_epoch=10, N_epoch_print=100000)
        ml.reinforce_model([opt, opt])  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt["N_epoch"] = epochs
        ml.reinforce_model([opt, opt])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    return el / (2 * epochs), int(ml.gp_inputs.shape[0])


def time_fit_model_ur5(device, N=400, epochs=30):
    """The same for the UR5-shaped model (6 GPs, D=24, SE + polynomial(1), test_mcpilco_ur5_mujoco.py:71-117): seconds per epoch for
    ALL six GPs (one epoch of each), and N."""
    import contextlib
    import io
    import time

    from .gpr_lib.Likelihood import Gaussian_likelihood as Likelihood
    from .model_learning import Model_learning as ML

    c = sy.UR5
    D = c["D"]
    rbf = dict(active_dims=np.arange(D), lengthscales_init=np.ones(D), flg_train_lengthscales=True, lambda_init=np.ones(1), flg_train_lambda=False,
               sigma_n_init=np.ones(1), sigma_n_num=None, flg_train_sigma_n=True, dtype=DT, device=device)
    mpk = dict(active_dims=np.arange(D), poly_deg=1, Sigma_pos_par_init_list=[np.ones(D + 1)], flg_train_Sigma_pos_par_list=[True], dtype=DT,
               device=device)
    par = dict(num_gp=6, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
               not_vel_indeces=c["not_vel"], dtype=DT, device=device, init_dict_list=[[rbf, mpk]] * 6)
    with contextlib.redirect_stdout(io.StringIO()):
        ml = ML.Speed_Model_learning_RBF_MPK_angle_state(**par)
        for xs, us in sy.ur5_rollouts(n_roll=(N + 199) // 200):
            ml.add_data(np.asarray(xs), np.asarray(us))
        opt = dict(f_optimizer="lambda p : torch.optim.Adam(p, lr=0.01)", criterion=Likelihood.Marginal_log_likelihood, N_epoch=5, N_epoch_print=100000)
        ml.reinforce_model([opt] * 6)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt["N_epoch"] = epochs
        ml.reinforce_model([opt] * 6)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    return el / epochs, int(ml.gp_inputs.shape[0])


def time_pretrain(device, shape="cartpole", reps=5):
    """``Model_learning.pretrain_gp`` (Model_learning.py:163-208: get_SOD GP_prior.py:232-257, then forward / get_alpha / get_estimate on
    the subset) on the drop-in classes, per GP, with the launch scripts' SOD settings, and the stages on their own (HIP events on the
    launch stream, median of ``reps``):
      cartpole  N = 300, D = 6, SE, 2 GPs, relative threshold 0.5 (sigma_n = 0.36: SURVEY 8c's pruning case, ~264 of 300 rows stay);
      ur5       N = 600, D = 24, SE + polynomial(1), 6 GPs, absolute threshold 0.001 (test_mcpilco_ur5_mujoco.py:82-86).
    Returns a dict (seconds per GP for the whole call incl. its host syncs, microseconds per stage, rows kept)."""
    import contextlib
    import io
    import time

    from .model_learning import Model_learning as ML

    quiet = lambda: contextlib.redirect_stdout(io.StringIO())
    if shape == "cartpole":
        c, N, G = sy.CARTPOLE, 300, 2
        rbf = dict(active_dims=np.arange(6), lengthscales_init=np.asarray(c["lengthscales"], dtype=float), flg_train_lengthscales=True,
                   lambda_init=np.ones(1), flg_train_lambda=False, sigma_n_init=0.36 * np.ones(1), sigma_n_num=None, flg_train_sigma_n=True, dtype=DT,
                   device=device)
        par = dict(num_gp=G, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
                   not_vel_indeces=c["not_vel"], dtype=DT, device=device, init_dict_list=[rbf] * G, approximation_mode="SOD",
                   approximation_dict={"SOD_threshold_mode": "relative", "SOD_threshold": 0.5, "flg_SOD_permutation": False})
        with quiet():
            ml = ML.Speed_Model_learning_RBF_angle_state(**par)
            for xs, us in sy.cartpole_rollouts(n_roll=5):
                ml.add_data(np.asarray(xs), np.asarray(us))
    else:
        c, N, G = sy.UR5, 600, 6
        D = c["D"]
        rbf = dict(active_dims=np.arange(D), lengthscales_init=np.asarray(c["lengthscales"], dtype=float), flg_train_lengthscales=True,
                   lambda_init=np.ones(1), flg_train_lambda=False, sigma_n_init=c["sigma_n"] * np.ones(1), sigma_n_num=None, flg_train_sigma_n=True,
                   dtype=DT, device=device)
        mpk = dict(active_dims=np.arange(D), poly_deg=1, Sigma_pos_par_init_list=[0.05 * np.ones(D + 1)], flg_train_Sigma_pos_par_list=[True], dtype=DT,
                   device=device)
        par = dict(num_gp=G, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
                   not_vel_indeces=c["not_vel"], dtype=DT, device=device, init_dict_list=[[rbf, mpk]] * G, approximation_mode="SOD",
                   approximation_dict={"SOD_threshold_mode": "absolute", "SOD_threshold": [0.001] * G, "flg_SOD_permutation": False})
        with quiet():
            ml = ML.Speed_Model_learning_RBF_MPK_angle_state(**par)
            for xs, us in sy.ur5_rollouts(n_roll=3):
                ml.add_data(np.asarray(xs), np.asarray(us))
    assert int(ml.gp_inputs.shape[0]) == N
    whole = []
    with quiet(), torch.no_grad():
        for r in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for g in range(G):
                ml.pretrain_gp(g)
            torch.cuda.synchronize()
            if r:
                whole.append((time.perf_counter() - t0) / G)
    kept = [len(ix) for ix in ml.SOD_indices]
    # the stages on their own (GP 0): device time between HIP events
    gp = ml.gp_list[0]
    X = gp._cols(ml.gp_inputs)
    spec = gp.kernel_spec()
    thr = float(0.5 * torch.sqrt(gp.get_sigma_n_2().detach())) if shape == "cartpole" else 0.001
    idx = ml.SOD_indices[0]
    Xs, Ys = X[idx, :].contiguous(), ml.gp_output_list[0][idx, :].contiguous()
    stages = {}

    def timed(name, fn):
        ts, out = [], None
        for _ in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        stages[name] = sorted(ts[1:])[len(ts[1:]) // 2]
        return out

    import ctypes as C

    from . import hipabi as abi

    nb = abi.lib().mcp_sod_workspace_bytes(N)
    ws = torch.empty((nb + 7) // 8, dtype=DT, device=device)
    ix = torch.zeros(N, dtype=torch.int32, device=device)
    cnt = torch.zeros(1, dtype=torch.int32, device=device)
    kc = spec.to_c(device)
    timed("sod_select", lambda: abi.check(abi.lib().mcp_sod_select(C.byref(kc), N, abi.ptr(X), thr, abi.ptr(ix), abi.ptr(cnt), abi.ptr(ws), nb,
                                                                   abi.stream()), "mcp_sod_select"))
    K = timed("gram", lambda: ops.cov_build(spec, Xs, None, noise=True))
    U, _, _ = timed("cholesky", lambda: ops.chol_factor(K))
    _, Kinv = timed("inverse", lambda: ops.chol_inverse(U))
    alpha = timed("alpha", lambda: ops.gp_alpha(Kinv, Ys, 0.0))
    timed("pack", lambda: ops.PackedGP(spec, Xs, alpha, Kinv))
    timed("posterior_all_rows", lambda: ops.posterior(ops.PackedGP(spec, Xs, alpha, Kinv), X))
    whole.sort()
    return {"N": N, "gps": G, "rows_kept": kept, "s_per_gp": whole[len(whole) // 2], "stage_us": stages,
            "stage_us_total": sum(stages.values())}
