"""Generates the golden vectors in this directory by IMPORTING THE REFERENCE (read-only at
/root/reference) and running its own classes on small deterministic inputs.

Run once in the build container (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tests/golden/make_golden.py

Only inputs and outputs (arrays) are stored -- no reference source text.  The noise the
reference drew internally is recovered by re-seeding torch and replaying its draw order
(SURVEY.md 8c); the script asserts that the replay reproduces the reference's x0 bit-exactly
before storing eps / masks.
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
sys.path.insert(1, os.path.dirname(os.path.dirname(HERE)))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

with contextlib.redirect_stdout(io.StringIO()):
    import gpr_lib.Utils.Parameters_covariance_functions  # noqa: F401  (needed before MPK GPs are built)
    import gpr_lib.Likelihood.Gaussian_likelihood  # noqa: F401
    import gpr_lib.GP_prior.GP_prior as RGP
    import gpr_lib.GP_prior.Sparse_GP as RSP
    import gpr_lib.GP_prior.Stationary_GP as RSGP
    import model_learning.Model_learning as RML
    import policy_learning.Cost_function as RC
    import policy_learning.MC_PILCO as RMC
    import policy_learning.Policy as RP

import mcp_boot  # noqa: E402,F401
from mc_pilco_amd import synthetic as sy  # noqa: E402

dtype = torch.float64
dev = torch.device("cpu")
torch.set_num_threads(1)
quiet = contextlib.redirect_stdout(io.StringIO())


def T(a):
    return torch.tensor(np.asarray(a), dtype=dtype)


def N(t):
    return t.detach().cpu().numpy().copy()


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np.asarray(v) for k, v in kw.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in kw.items()})


def rbf_dict(D, ls, sigma_n, lam=1.0):
    return dict(
        active_dims=np.arange(D), lengthscales_init=np.asarray(ls, dtype=float), flg_train_lengthscales=True,
        lambda_init=lam * np.ones(1), flg_train_lambda=False, sigma_n_init=sigma_n * np.ones(1),
        sigma_n_num=None, flg_train_sigma_n=True, dtype=dtype, device=dev,
    )


def mpk_dict(D, deg, weights):
    return dict(
        active_dims=np.arange(D), poly_deg=deg, Sigma_pos_par_init_list=weights,
        flg_train_Sigma_pos_par_list=[True] * deg, dtype=dtype, device=dev,
    )


def poly_weights(D, deg, rng, scale):
    w = [scale * (0.5 + rng.rand(D + 1))]
    for k in range(2, deg + 1):
        w.append(scale * (0.5 + rng.rand(k * D)))
    return w


# ---------------------------------------------------------------------------------------
# (1)-(3)  Gram, forward (Cholesky / inverse / logdet), alpha, posterior  -- three kernels
# ---------------------------------------------------------------------------------------
def kernel_fixture(name, D, deg, Z, Y, ztest, ls, sigma_n, pw):
    with quiet:
        if deg == 0:
            gp = RSGP.RBF(**rbf_dict(D, ls, sigma_n))
        else:
            gp = RGP.Sum_Independent_GP(RSGP.RBF(**rbf_dict(D, ls, sigma_n)), RSP.get_Volterra_MPK_GP(**mpk_dict(D, deg, pw)))
    X = T(Z)
    Yt = T(Y)
    Xs = T(ztest)
    with torch.no_grad(), quiet:
        K_noise = gp.get_covariance(X, flg_noise=True)
        K_cross = gp.get_covariance(Xs, X)
        diag = gp.get_diag_covariance(Xs)
        mX, K, Kinv, logdet = gp(X)
        alpha, _, _ = gp.get_alpha(X, Yt)
        mu, var = gp.get_estimate_from_alpha(X, Xs, alpha, mX, K_X_inv=Kinv)
    out = dict(
        X=Z, Y=Y, Xs=ztest, lengthscales=ls, sigma_n=sigma_n, lam=1.0, deg=deg,
        K_noise=N(K_noise), K_cross=N(K_cross), diag=N(diag), Kinv=N(Kinv), logdet=N(logdet),
        alpha=N(alpha), mX=N(mX), mu=N(mu), var=N(var),
    )
    for k, w in enumerate(pw or []):
        out["poly_w%d" % (k + 1)] = w
    save(name, **out)


ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
if ONLY is not None:
    _real_save = save

    def save(name, **kw):  # noqa: F811  regenerate a single fixture without touching the others
        if name.startswith(ONLY):
            _real_save(name, **kw)


rng = np.random.RandomState(7)
cp = sy.cartpole_rollouts()
Zc, Yc = sy.gp_io(cp, sy.CARTPOLE["angle"], sy.CARTPOLE["not_angle"], sy.CARTPOLE["vel"])
sel = rng.permutation(Zc.shape[0])[:64]
ztest_c = Zc[rng.permutation(Zc.shape[0])[:32]] + 0.05 * rng.randn(32, 6)
kernel_fixture("kern_se", 6, 0, Zc[sel], Yc[0][sel], ztest_c, sy.CARTPOLE["lengthscales"], 0.03, None)
kernel_fixture("kern_se_poly2", 6, 2, Zc[sel], Yc[1][sel], ztest_c, sy.CARTPOLE["lengthscales"], 0.03, poly_weights(6, 2, rng, 0.05))
ur = sy.ur5_rollouts()
Zu, Yu = sy.gp_io(ur, sy.UR5["angle"], sy.UR5["not_angle"], sy.UR5["vel"])
selu = rng.permutation(Zu.shape[0])[:64]
ztest_u = Zu[rng.permutation(Zu.shape[0])[:32]] + 0.01 * rng.randn(32, 24)
kernel_fixture("kern_se_poly1_d24", 24, 1, Zu[selu], Yu[2][selu], ztest_u, sy.UR5["lengthscales"], 0.01, poly_weights(24, 1, rng, 0.05))


# ---------------------------------------------------------------------------------------
# (4) get_SOD index lists (relative threshold through pretrain_gp, absolute directly)
# ---------------------------------------------------------------------------------------
def sod_fixture():
    D = 6
    sig = 0.36
    with quiet:
        gp = RSGP.RBF(**rbf_dict(D, sy.CARTPOLE["lengthscales"], sig))
    X = T(Zc[:150])
    Y = T(Yc[0][:150])
    with torch.no_grad(), quiet:
        thr_rel = 0.5 * torch.sqrt(gp.get_sigma_n_2())
        idx_rel = [int(i) for i in gp.get_SOD(X, Y, thr_rel)]
        idx_abs = [int(i) for i in gp.get_SOD(X, Y, 0.25)]
        # margins |sqrt(var)-thr| along the greedy path (how close any decision was to flipping)
        margins = []
        for thr, idx in ((float(thr_rel), idx_rel), (0.25, idx_abs)):
            keep = [0]
            mm = np.inf
            for i in range(1, X.shape[0]):
                _, var, _ = gp.get_estimate(X[keep, :], Y[keep, :], X[i : i + 1, :])
                mm = min(mm, abs(float(torch.sqrt(var)) - thr))
                if float(torch.sqrt(var)) > thr:
                    keep.append(i)
            assert keep == idx
            margins.append(mm)
    save("sod", X=N(X), Y=N(Y), lengthscales=sy.CARTPOLE["lengthscales"], sigma_n=sig, thr_rel=float(thr_rel), thr_rel_factor=0.5,
         idx_rel=np.array(idx_rel), thr_abs=0.25, idx_abs=np.array(idx_abs), min_margin=np.array(margins))


sod_fixture()


# ---------------------------------------------------------------------------------------
# model builders (reference objects with fixed, trained-like hyper-parameters)
# ---------------------------------------------------------------------------------------
def build_cartpole_model(n_train, deg, sod, pw_list=None):
    c = sy.CARTPOLE
    par = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"],
               vel_indeces=c["vel"], not_vel_indeces=c["not_vel"], dtype=dtype, device=dev)
    if sod:
        par["approximation_mode"] = "SOD"
        par["approximation_dict"] = {"SOD_threshold_mode": "relative", "SOD_threshold": 0.5, "flg_SOD_permutation": False}
    sig = 0.2 if sod else c["sigma_n"]
    with quiet:
        if deg == 0:
            par["init_dict_list"] = [rbf_dict(6, c["lengthscales"], sig)] * 2
            ml = RML.Speed_Model_learning_RBF_angle_state(**par)
        else:
            par["init_dict_list"] = [[rbf_dict(6, c["lengthscales"], sig), mpk_dict(6, deg, pw_list[g])] for g in range(2)]
            ml = RML.Speed_Model_learning_RBF_MPK_angle_state(**par)
        x = np.concatenate([r[0] for r in cp], 0)[: n_train + 1]
        u = np.concatenate([r[1] for r in cp], 0)[: n_train + 1]
        ml.add_data(x, u)
        with torch.no_grad():
            for g in range(2):
                ml.pretrain_gp(g)
        ml.set_eval_mode()
    return ml, x, u, sig


def build_ur5_model(n_train, pw_list):
    c = sy.UR5
    par = dict(num_gp=6, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"],
               vel_indeces=c["vel"], not_vel_indeces=c["not_vel"], dtype=dtype, device=dev)
    par["init_dict_list"] = [[rbf_dict(24, c["lengthscales"], c["sigma_n"]), mpk_dict(24, 1, pw_list[g])] for g in range(6)]
    with quiet:
        ml = RML.Speed_Model_learning_RBF_MPK_angle_state(**par)
        x = ur[0][0][: n_train + 1]
        u = ur[0][1][: n_train + 1]
        ml.add_data(x, u)
        with torch.no_grad():
            for g in range(6):
                ml.pretrain_gp(g)
        ml.set_eval_mode()
    return ml, x, u


def model_arrays(ml, prefix=""):
    out = {}
    for g in range(ml.num_gp):
        out[prefix + "Xtr%d" % g] = N(ml.gp_inputs_tr_list[g])
        out[prefix + "alpha%d" % g] = N(ml.alpha_list[g])
        out[prefix + "Kinv%d" % g] = N(ml.K_X_inv_list[g])
        if getattr(ml, "SOD_indices", None) is not None and ml.approximation_mode == "SOD":
            out[prefix + "sod%d" % g] = np.array([int(i) for i in ml.SOD_indices[g]])
    return out


# ---------------------------------------------------------------------------------------
# (5) one get_next_state step (injected eps, and particle_pred=False)
# ---------------------------------------------------------------------------------------
def step_fixture():
    ml, xtr, utr, sig = build_cartpole_model(120, 0, sod=False)
    M = 24
    rs = np.random.RandomState(3)
    x = T(np.concatenate([xtr[rs.permutation(100)[:M]] + 0.02 * rs.randn(M, 4)], 0))
    u = T(10 * (2 * rs.rand(M, 1) - 1))
    torch.manual_seed(11)
    with torch.no_grad():
        nxt, mu, var = ml.get_next_state(x, u)
    torch.manual_seed(11)
    eps = torch.empty(M, 2, dtype=dtype).normal_()
    with torch.no_grad():
        nxt_mean, _, _ = ml.get_next_state(x, u, particle_pred=False)
    save("step_se", states_tr=xtr, inputs_tr=utr, sigma_n=sig, x=N(x), u=N(u), eps=N(eps), next=N(nxt), mu=N(mu), var=N(var),
         next_mean=N(nxt_mean), **model_arrays(ml))


step_fixture()


# ---------------------------------------------------------------------------------------
# (6) policy forward, three classes, p in {0, 0.25}
# ---------------------------------------------------------------------------------------
def policy_fixture():
    rs = np.random.RandomState(5)
    M, B = 16, 40
    out = {}
    # plain
    with quiet:
        pol = RP.Sum_of_gaussians(state_dim=4, input_dim=2, num_basis=B, lengthscales_init=0.8 + rs.rand(4),
                                  centers_init=2 * rs.randn(B, 4), weight_init=rs.randn(2, B), flg_squash=True, u_max=[3.0, 1.5],
                                  flg_drop=True, dtype=dtype, device=dev)
    x = T(rs.randn(M, 4))
    with torch.no_grad():
        out["plain_u0"] = N(pol(x, t=0, p_dropout=0.0))
        torch.manual_seed(21)
        out["plain_u25"] = N(pol(x, t=0, p_dropout=0.25))
        torch.manual_seed(21)
        out["plain_mask"] = N(torch.empty(M, 1, B, dtype=dtype).bernoulli_(0.75).reshape(M, B))
    out.update(plain_x=N(x), plain_ls=N(torch.exp(pol.log_lengthscales)), plain_centers=N(pol.centers), plain_weight=N(pol.f_linear.weight),
               plain_umax=np.array([3.0, 1.5]))
    # angles (cart-pole)
    pi = sy.cartpole_policy_init(B=B, seed=2)
    with quiet:
        pol = RP.Sum_of_gaussians_with_angles(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]),
                                              non_angle_indices=np.array([0, 1, 3]), lengthscales_init=pi["lengthscales"] * 1.3,
                                              centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True, u_max=10.0,
                                              flg_drop=True, dtype=dtype, device=dev)
    x = T(np.concatenate([rs.randn(M, 2), 3 * rs.randn(M, 1), rs.randn(M, 1)], 1))
    with torch.no_grad():
        out["ang_u0"] = N(pol(x, t=3, p_dropout=0.0))
        torch.manual_seed(22)
        out["ang_u25"] = N(pol(x, t=3, p_dropout=0.25))
        torch.manual_seed(22)
        out["ang_mask"] = N(torch.empty(M, 1, B, dtype=dtype).bernoulli_(0.75).reshape(M, B))
    out.update(ang_x=N(x), ang_ls=N(torch.exp(pol.log_lengthscales)), ang_centers=N(pol.centers), ang_weight=N(pol.f_linear.weight), ang_umax=10.0)
    # target trajectory (UR5-shaped)
    tt = sy.ur5_target_traj(T=10)
    piu = sy.ur5_policy_init(B=B, seed=3)
    with quiet:
        pol = RP.Sum_of_gaussians_with_target_trajectory(state_dim=24, input_dim=6, num_basis=B, target_traj=tt,
                                                         lengthscales_init=piu["lengthscales"], centers_init=piu["centers"],
                                                         weight_init=piu["weight"], flg_squash=True, u_max=[1.0] * 6, flg_drop=True,
                                                         dtype=dtype, device=dev)
    x = T(tt[4:5] + 0.3 * rs.randn(M, 12))
    with torch.no_grad():
        out["traj_u0"] = N(pol(x, t=4, p_dropout=0.0))
        torch.manual_seed(23)
        out["traj_u25"] = N(pol(x, t=4, p_dropout=0.25))
        torch.manual_seed(23)
        out["traj_mask"] = N(torch.empty(M, 1, B, dtype=dtype).bernoulli_(0.75).reshape(M, B))
    out.update(traj_x=N(x), traj_ls=N(torch.exp(pol.log_lengthscales)), traj_centers=N(pol.centers), traj_weight=N(pol.f_linear.weight),
               traj_umax=np.ones(6), traj_target=tt, traj_t=4)
    save("policy", **out)


policy_fixture()


# ---------------------------------------------------------------------------------------
# (7) costs
# ---------------------------------------------------------------------------------------
def cost_fixture():
    rs = np.random.RandomState(9)
    Tn, M = 7, 12
    st = T(np.concatenate([rs.randn(Tn, M, 2), 2.5 * rs.randn(Tn, M, 1), rs.randn(Tn, M, 1)], 2))
    cf = RC.Cart_pole_cost(target_state=T([np.pi, 0.0]), lengthscales=T([3.0, 1.0]), angle_index=2, pos_index=0)
    st.requires_grad_(True)
    c, s = cf(st, None, 0)
    c.backward()
    out = dict(cp_states=N(st), cp_cost=N(c), cp_std=N(s), cp_grad=N(st.grad))
    tt = sy.ur5_target_traj(T=Tn)
    st2 = T(tt.reshape(Tn, 1, 12) + 0.4 * rs.randn(Tn, M, 12))
    st2.requires_grad_(True)
    cf2 = RC.Expected_saturated_distance_from_trajectory(target_traj=T(tt), lengthscales=T(sy.UR5["cost_ls"]), used_indeces=list(range(12)))
    c2, s2 = cf2(st2, None, 0)
    c2.backward()
    out.update(tr_states=N(st2), tr_target=tt, tr_ls=np.array(sy.UR5["cost_ls"]), tr_cost=N(c2), tr_std=N(s2), tr_grad=N(st2.grad))
    save("cost", **out)


cost_fixture()


# ---------------------------------------------------------------------------------------
# (8) full apply_policy + cost + backward through MC_PILCO, noise recovered by replay
# ---------------------------------------------------------------------------------------
def make_mcpilco(ml_par_builder, policy_cls, policy_par, cost_cls, cost_par, S, U, Ts):
    with quiet:
        obj = RMC.MC_PILCO(
            T_sampling=Ts, state_dim=S, input_dim=U, f_sim=lambda y, t, u: None,
            f_model_learning=lambda **kw: ml_par_builder, model_learning_par={},
            f_rand_exploration_policy=RP.Random_exploration,
            rand_exploration_policy_par=dict(state_dim=S, input_dim=U, u_max=1.0, dtype=dtype, device=dev),
            f_control_policy=policy_cls, control_policy_par=policy_par, f_cost_function=cost_cls, cost_function_par=cost_par,
            log_path=None, dtype=dtype, device=dev,
        )
    return obj


def replay_noise(seed, M, S, G, B, Tn, p):
    torch.manual_seed(seed)
    eps0 = torch.empty(M, S, dtype=dtype).normal_()
    masks = []
    if p > 0:
        masks.append(torch.empty(M, 1, B, dtype=dtype).bernoulli_(1 - p).reshape(M, B))
    eps = []
    for _ in range(1, Tn):
        eps.append(torch.empty(M, G, dtype=dtype).normal_())
        if p > 0:
            masks.append(torch.empty(M, 1, B, dtype=dtype).bernoulli_(1 - p).reshape(M, B))
    return eps0, torch.stack(eps), (torch.stack(masks) if p > 0 else None)


def rollout_fixture(name, kind, M, Tn, p, seed, B):
    if kind in ("se", "se_sod", "se_poly2"):
        c = sy.CARTPOLE
        rs = np.random.RandomState(13)
        pw = [poly_weights(6, 2, rs, 0.02) for _ in range(2)] if kind == "se_poly2" else None
        ml, xtr, utr, sig = build_cartpole_model(100 if kind != "se_sod" else 140, 2 if kind == "se_poly2" else 0, sod=(kind == "se_sod"), pw_list=pw)
        pi = sy.cartpole_policy_init(B=B, seed=4)
        ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                    lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True,
                    u_max=c["u_max"], flg_drop=True, dtype=dtype, device=dev)
        obj = make_mcpilco(ml, RP.Sum_of_gaussians_with_angles, ppar, RC.Cart_pole_cost,
                           dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0), 4, 1, c["Ts"])
        S, U, G = 4, 1, 2
        x0m, x0v = T(c["x0_mean"]), T(np.array([1e-2, 1e-2, 4e-2, 1e-2]))
        extra = {}
    else:  # ur5
        c = sy.UR5
        rs = np.random.RandomState(17)
        pw = [poly_weights(24, 1, rs, 0.02) for _ in range(6)]
        ml, xtr, utr = build_ur5_model(80, pw)
        sig = c["sigma_n"]
        tt = sy.ur5_target_traj(T=Tn)
        pi = sy.ur5_policy_init(B=B, seed=5)
        ppar = dict(state_dim=24, input_dim=6, num_basis=B, target_traj=tt, lengthscales_init=pi["lengthscales"],
                    centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True, u_max=c["u_max"], flg_drop=True,
                    dtype=dtype, device=dev)
        obj = make_mcpilco(ml, RP.Sum_of_gaussians_with_target_trajectory, ppar, RC.Expected_saturated_distance_from_trajectory,
                           dict(target_traj=T(tt), lengthscales=T(c["cost_ls"]), used_indeces=list(range(12))), 12, 6, c["Ts"])
        S, U, G = 12, 6, 6
        x0m, x0v = T(xtr[0]), T(1e-4 * np.ones(12))
        extra = dict(target_traj=tt)
    pol = obj.control_policy
    torch.manual_seed(seed)
    st, inp = obj.apply_policy(particles_initial_state_mean=x0m, particles_initial_state_var=x0v, flg_particles_init_uniform=False,
                               particles_init_up_bound=None, particles_init_low_bound=None, flg_particles_init_multi_gauss=False,
                               num_particles=M, T_control=Tn, p_dropout=p)
    cost, std = obj.cost_function(st, inp, 0)
    cost.backward()
    eps0, eps, masks = replay_noise(seed, M, S, G, B, Tn, p)
    x0 = x0m.reshape(1, -1) + torch.sqrt(x0v).reshape(1, -1) * eps0
    assert torch.equal(x0, st[0].detach()), "noise replay does not reproduce the reference's x0"
    out = dict(states_tr=xtr, inputs_tr=utr, sigma_n=sig, x0_mean=N(x0m), x0_var=N(x0v), eps0=N(eps0), eps=N(eps), p_drop=p,
               states=N(st), inputs=N(inp), cost=N(cost), std=N(std),
               pol_ls=N(torch.exp(pol.log_lengthscales)), pol_centers=N(pol.centers), pol_weight=N(pol.f_linear.weight),
               g_log_ls=N(pol.log_lengthscales.grad), g_centers=N(pol.centers.grad), g_weight=N(pol.f_linear.weight.grad), **extra)
    if masks is not None:
        out["masks"] = N(masks).astype(np.uint8)
    if pw is not None:
        for g, w in enumerate(pw):
            for k, wk in enumerate(w):
                out["poly_w%d_gp%d" % (k + 1, g)] = wk
    out.update(model_arrays(ml))
    save(name, **out)


rollout_fixture("rollout_se", "se", M=24, Tn=10, p=0.25, seed=101, B=48)
rollout_fixture("rollout_se_nodrop", "se", M=16, Tn=8, p=0.0, seed=102, B=32)
rollout_fixture("rollout_se_sod", "se_sod", M=16, Tn=8, p=0.25, seed=103, B=32)
rollout_fixture("rollout_se_poly2", "se_poly2", M=20, Tn=10, p=0.25, seed=104, B=40)
rollout_fixture("rollout_ur5", "ur5", M=16, Tn=8, p=0.25, seed=105, B=36)
rollout_fixture("rollout_se_long", "se", M=32, Tn=60, p=0.25, seed=106, B=64)


# ---------------------------------------------------------------------------------------
# (8b) MC_PILCO4PMS.apply_policy (MC_PILCO.py:808-906): the policy sees noisy positions, finite-difference velocities
#      and a first-order Butterworth filter; RNG order: x0, mask_0, then per step eps_t, position noise, mask_t
# ---------------------------------------------------------------------------------------
def pms_fixture(name="rollout_pms", M=16, Tn=9, p=0.25, seed=107, B=32):
    from scipy import signal
    c = sy.CARTPOLE
    ml, xtr, utr, sig = build_cartpole_model(100, 0, sod=False)
    pi = sy.cartpole_policy_init(B=B, seed=6)
    ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True,
                u_max=c["u_max"], flg_drop=True, dtype=dtype, device=dev)
    pos, vel, fc = [0, 2], [1, 3], 0.5
    std_meas = np.array([0.01, 0.02, 0.015, 0.03])
    with quiet:
        obj = RMC.MC_PILCO4PMS(
            T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None,
            f_model_learning=lambda **kw: ml, model_learning_par={},
            f_rand_exploration_policy=RP.Random_exploration,
            rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=1.0, dtype=dtype, device=dev),
            f_control_policy=RP.Sum_of_gaussians_with_angles, control_policy_par=ppar, f_cost_function=RC.Cart_pole_cost,
            cost_function_par=dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0),
            pos_indeces=pos, vel_indeces=vel, std_meas_noise=std_meas, log_path=None, filtering_dict={"fc": fc}, dtype=dtype, device=dev)
    pol = obj.control_policy
    x0m, x0v = T(c["x0_mean"]), T(np.array([1e-2, 1e-2, 4e-2, 1e-2]))
    torch.manual_seed(seed)
    st, inp = obj.apply_policy(particles_initial_state_mean=x0m, particles_initial_state_var=x0v, flg_particles_init_uniform=False,
                               particles_init_up_bound=None, particles_init_low_bound=None, flg_particles_init_multi_gauss=False,
                               num_particles=M, T_control=Tn, p_dropout=p)
    cost, std = obj.cost_function(st, inp, 0)
    cost.backward()
    # replay the draw order
    torch.manual_seed(seed)
    eps0 = torch.empty(M, 4, dtype=dtype).normal_()
    masks = [torch.empty(M, 1, B, dtype=dtype).bernoulli_(1 - p).reshape(M, B)]
    eps, pn = [], []
    for _ in range(1, Tn):
        eps.append(torch.empty(M, 2, dtype=dtype).normal_())
        pn.append(torch.randn(M, len(pos), dtype=dtype))
        masks.append(torch.empty(M, 1, B, dtype=dtype).bernoulli_(1 - p).reshape(M, B))
    x0 = x0m.reshape(1, -1) + torch.sqrt(x0v).reshape(1, -1) * eps0
    assert torch.allclose(x0, st[0].detach(), rtol=0, atol=1e-15), "noise replay does not reproduce the reference's x0"
    b, a = signal.butter(1, fc)
    out = dict(states_tr=xtr, inputs_tr=utr, sigma_n=sig, x0_mean=N(x0m), x0_var=N(x0v), x0=N(st[0]), eps=N(torch.stack(eps)),
               pos_noise=N(torch.stack(pn)), masks=N(torch.stack(masks)).astype(np.uint8), p_drop=p, states=N(st), inputs=N(inp),
               cost=N(cost), std=N(std), pos_indeces=np.array(pos), vel_indeces=np.array(vel), std_meas_noise=std_meas, fc=fc,
               butter_b=np.asarray(b), butter_a=np.asarray(a),
               pol_ls=N(torch.exp(pol.log_lengthscales)), pol_centers=N(pol.centers), pol_weight=N(pol.f_linear.weight),
               g_log_ls=N(pol.log_lengthscales.grad), g_centers=N(pol.centers.grad), g_weight=N(pol.f_linear.weight.grad))
    out.update(model_arrays(ml))
    save(name, **out)


pms_fixture()


# ---------------------------------------------------------------------------------------
# (9) multi-Gaussian and uniform initial distributions (indices bit-exact)
# ---------------------------------------------------------------------------------------
def init_fixture():
    ml, xtr, utr, sig = build_cartpole_model(60, 0, sod=False)
    c = sy.CARTPOLE
    pi = sy.cartpole_policy_init(B=16, seed=4)
    ppar = dict(state_dim=4, input_dim=1, num_basis=16, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True,
                u_max=c["u_max"], flg_drop=True, dtype=dtype, device=dev)
    obj = make_mcpilco(ml, RP.Sum_of_gaussians_with_angles, ppar, RC.Cart_pole_cost,
                       dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0), 4, 1, c["Ts"])
    means = T([[0.0, 0.0, 0.0, 0.0], [0.5, 0.0, 1.0, 0.0], [-0.5, 0.1, -1.0, 0.0]])
    vars_ = T([[1e-4] * 4, [1e-3] * 4, [4e-4] * 4])
    M = 20
    torch.manual_seed(301)
    with torch.no_grad():
        st, _ = obj.apply_policy(particles_initial_state_mean=means, particles_initial_state_var=vars_, flg_particles_init_uniform=False,
                                 particles_init_up_bound=None, particles_init_low_bound=None, flg_particles_init_multi_gauss=True,
                                 num_particles=M, T_control=1, p_dropout=0.0)
    torch.manual_seed(301)
    idx = torch.randint(0, 3, [M])
    e0 = torch.empty(M, 4, dtype=dtype).normal_()
    assert torch.equal(means[idx] + torch.sqrt(vars_[idx]) * e0, st[0])
    lb, ub = T([-1.0, -0.5, -3.0, -0.5]), T([1.0, 0.5, 3.0, 0.5])
    torch.manual_seed(302)
    with torch.no_grad():
        st_u, _ = obj.apply_policy(particles_initial_state_mean=means[0], particles_initial_state_var=vars_[0], flg_particles_init_uniform=True,
                                   particles_init_up_bound=ub, particles_init_low_bound=lb, flg_particles_init_multi_gauss=False,
                                   num_particles=M, T_control=1, p_dropout=0.0)
    save("init_dists", means=N(means), vars=N(vars_), mg_seed=301, mg_idx=N(idx), mg_eps0=N(e0), mg_x0=N(st[0]), lb=N(lb), ub=N(ub), un_seed=302, un_x0=N(st_u[0]))


init_fixture()


# ---------------------------------------------------------------------------------------
# (10) marginal likelihood and its gradient w.r.t. every hyper-parameter (GP_prior.fit_model's objective)
# ---------------------------------------------------------------------------------------
def nll_fixture(name, D, deg, Z, Y, ls, sigma_n, pw):
    with quiet:
        if deg == 0:
            gp = RSGP.RBF(**dict(rbf_dict(D, ls, sigma_n), flg_train_lambda=True))
        else:
            gp = RGP.Sum_Independent_GP(RSGP.RBF(**dict(rbf_dict(D, ls, sigma_n), flg_train_lambda=True)),
                                        RSP.get_Volterra_MPK_GP(**mpk_dict(D, deg, pw)))
    crit = gpr_lib.Likelihood.Gaussian_likelihood.Marginal_log_likelihood()
    with quiet:
        loss = crit(gp(T(Z)), T(Y))
    loss.backward()
    out = dict(X=Z, Y=Y, lengthscales=ls, sigma_n=sigma_n, deg=deg, loss=N(loss))
    for n_, p_ in gp.named_parameters():
        if p_.grad is not None:
            out["grad__" + n_] = N(p_.grad)
    for k, w in enumerate(pw or []):
        out["poly_w%d" % (k + 1)] = w
    save(name, **out)


rs_n = np.random.RandomState(21)
nll_fixture("nll_se", 6, 0, Zc[:80], Yc[0][:80], sy.CARTPOLE["lengthscales"], 0.05, None)
nll_fixture("nll_se_poly2", 6, 2, Zc[:80], Yc[1][:80], sy.CARTPOLE["lengthscales"], 0.05, poly_weights(6, 2, rs_n, 0.05))
nll_fixture("nll_se_poly1_d24", 24, 1, Zu[:64], Yu[1][:64], sy.UR5["lengthscales"], 0.01, poly_weights(24, 1, rs_n, 0.05))
print("done")
