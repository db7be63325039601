"""Round-2 golden vectors, generated like make_golden.py by IMPORTING THE REFERENCE (read-only at /root/reference) and
running its own classes.  Only arrays / plain pickled data are stored -- no reference source text.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tests/golden/make_golden_r2.py

  policy_opt_trace      MC_PILCO.reinforce_policy (policy_learning/MC_PILCO.py:375-613): 6 plain Adam steps, and a run whose
                        monitors force two learning-rate halvings, two dropout reductions and the early exit
  delta_model_step      Model_learning_RBF_angle_state (model_learning/Model_learning.py:471-493, 528-580): one get_next_state
  simple_costs          Expected_distance / Expected_saturated_distance (policy_learning/Cost_function.py:39-101)
  mean_rollout          MC_PILCO.rollout (MC_PILCO.py:347-373): mean-only prediction of a recorded trajectory
  sod_permutation       GP_prior.get_SOD(flg_permutation=True) (gpr_lib/GP_prior/GP_prior.py:244-247), seeded
  options               policy scale_factor, per-trial cost lengthscales;  rollout_bias: policy with flg_bias, full rollout + gradient
  ref_log.pkl (+ ref_log_expect.npz)   a log.pkl with the reference's keys and state_dicts, written from reference objects, for
                        MC_PILCO.load_model_from_log (MC_PILCO.py:711-751) of the drop-in
"""
import contextlib
import io
import os
import pickle as pkl
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
    import gpr_lib.Utils.Parameters_covariance_functions  # noqa: F401
    import gpr_lib.Likelihood.Gaussian_likelihood  # noqa: F401
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
c = sy.CARTPOLE
cp = sy.cartpole_rollouts()


def T(a):
    return torch.tensor(np.asarray(a), dtype=dtype)


def N(t):
    return t.detach().cpu().numpy().copy()


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np.asarray(v) for k, v in kw.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in kw.items()})


def rbf_dict(D, ls, sigma_n, lam=1.0):
    return dict(active_dims=np.arange(D), lengthscales_init=np.asarray(ls, dtype=float), flg_train_lengthscales=True,
                lambda_init=lam * np.ones(1), flg_train_lambda=False, sigma_n_init=sigma_n * np.ones(1), sigma_n_num=None,
                flg_train_sigma_n=True, dtype=dtype, device=dev)


def speed_model(n_train, sig=None):
    sig = c["sigma_n"] if sig is None else sig
    par = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
               not_vel_indeces=c["not_vel"], dtype=dtype, device=dev, init_dict_list=[rbf_dict(6, c["lengthscales"], sig)] * 2)
    with quiet:
        ml = RML.Speed_Model_learning_RBF_angle_state(**par)
        x = np.concatenate([r[0] for r in cp], 0)[: n_train + 1]
        u = np.concatenate([r[1] for r in cp], 0)[: n_train + 1]
        ml.add_data(x, u)
        with torch.no_grad():
            for g in range(2):
                ml.pretrain_gp(g)
        ml.set_eval_mode()
    return ml, x, u, sig


def mcpilco(ml, B, seed_pol):
    pi = sy.cartpole_policy_init(B=B, seed=seed_pol)
    ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True,
                u_max=c["u_max"], flg_drop=True, dtype=dtype, device=dev)
    with quiet:
        obj = RMC.MC_PILCO(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None,
                           f_model_learning=lambda **kw: ml, model_learning_par={}, f_rand_exploration_policy=RP.Random_exploration,
                           rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=1.0, dtype=dtype, device=dev),
                           f_control_policy=RP.Sum_of_gaussians_with_angles, control_policy_par=ppar, f_cost_function=RC.Cart_pole_cost,
                           cost_function_par=dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0),
                           log_path=None, dtype=dtype, device=dev)
    return obj, pi


# ---------------------------------------------------------------------------------------
# reinforce_policy traces
# ---------------------------------------------------------------------------------------
def policy_opt_trace():
    out = {}
    ml, xtr, utr, sig = speed_model(100)
    out.update(states_tr=xtr, inputs_tr=utr, sigma_n=sig)
    B, M = 32, 24
    x0m, x0v = T(c["x0_mean"]), T(np.array([1e-2, 1e-2, 4e-2, 1e-2]))
    common = dict(T_control=0.5, num_particles=M, trial_index=0, particles_initial_state_mean=x0m, particles_initial_state_var=x0v,
                  flg_particles_init_uniform=False, particles_init_up_bound=None, particles_init_low_bound=None,
                  flg_particles_init_multi_gauss=False, f_optimizer="lambda p, lr : torch.optim.Adam(p, lr)", num_step_print=100,
                  policy_reinit_dict=None)
    assert int(0.5 / c["Ts"]) == 10
    runs = {
        # (a) six plain Adam steps
        "plain": dict(opt_steps_list=[6], lr_list=[0.01], p_dropout_list=[0.25], seed=401),
        # (b) monitors forced: every window of 2 ratios is below min_diff_cost -> lr 0.01 -> 0.005 (step 1) -> 0.004 = lr_min (step 4)
        #     -> exit (step 7); dropout 0.25 -> 0.125 -> 0.0 (from then on no mask is drawn: the RNG stream changes)
        "forced": dict(opt_steps_list=[12], lr_list=[0.01], p_dropout_list=[0.25], seed=402, alpha_diff_cost=0.9, lr_reduction_ratio=0.5,
                       lr_min=0.004, p_drop_reduction=0.125, min_diff_cost=1e9, num_min_diff_cost=2, min_step=0),
    }
    for tag, kw in runs.items():
        obj, pi = mcpilco(ml, B, 8)
        seed = kw.pop("seed")
        torch.manual_seed(seed)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            cost_list, std_list, st, inp = obj.reinforce_policy(**common, **kw)
        pol = obj.control_policy
        out.update({tag + "_seed": seed, tag + "_cost_list": cost_list, tag + "_std_list": std_list, tag + "_last_states": st, tag + "_last_inputs": inp,
                    tag + "_final_log_ls": N(pol.log_lengthscales), tag + "_final_centers": N(pol.centers), tag + "_final_weight": N(pol.f_linear.weight),
                    tag + "_n_lr_reductions": buf.getvalue().count("REDUCING THE LEARNING RATE"),
                    tag + "_early_exit": int("EXIT FROM OPTIMIZATION" in buf.getvalue())})
        print(tag, "steps done", len(cost_list), "lr reductions", out[tag + "_n_lr_reductions"], "exit", out[tag + "_early_exit"])
    out.update(pol_ls=pi["lengthscales"], pol_centers=pi["centers"], pol_weight=pi["weight"], x0_mean=N(x0m), x0_var=N(x0v), T_control=0.5, M=M)
    assert len(out["forced_cost_list"]) == 8 and out["forced_n_lr_reductions"] == 2 and out["forced_early_exit"] == 1
    save("policy_opt_trace", **out)


policy_opt_trace()


# ---------------------------------------------------------------------------------------
# delta-state model: one step
# ---------------------------------------------------------------------------------------
def delta_model_step():
    sig = 0.03
    par = dict(num_gp=4, angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], dtype=dtype, device=dev,
               init_dict_list=[rbf_dict(6, c["lengthscales"], sig)] * 4)
    with quiet:
        ml = RML.Model_learning_RBF_angle_state(**par)
        x = np.concatenate([r[0] for r in cp], 0)[:91]
        u = np.concatenate([r[1] for r in cp], 0)[:91]
        ml.add_data(x, u)
        with torch.no_grad():
            for g in range(4):
                ml.pretrain_gp(g)
        ml.set_eval_mode()
    rs = np.random.RandomState(31)
    M = 20
    xs = T(x[rs.permutation(80)[:M]] + 0.02 * rs.randn(M, 4))
    us = T(10 * (2 * rs.rand(M, 1) - 1))
    torch.manual_seed(12)
    with torch.no_grad():
        nxt, mu, var = ml.get_next_state(xs, us)
    torch.manual_seed(12)
    eps = torch.empty(M, 4, dtype=dtype).normal_()
    assert torch.allclose(nxt, xs + mu + torch.sqrt(var) * eps, rtol=0, atol=1e-15)
    with torch.no_grad():
        nxt_mean, _, _ = ml.get_next_state(xs, us, particle_pred=False)
    out = dict(states_tr=x, inputs_tr=u, sigma_n=sig, x=N(xs), u=N(us), eps=N(eps), next=N(nxt), mu=N(mu), var=N(var), next_mean=N(nxt_mean),
               gp_inputs=N(ml.gp_inputs))
    for g in range(4):
        out["alpha%d" % g] = N(ml.alpha_list[g])
        out["gp_output%d" % g] = N(ml.gp_output_list[g])
    save("delta_model_step", **out)


delta_model_step()


# ---------------------------------------------------------------------------------------
# simple costs
# ---------------------------------------------------------------------------------------
def simple_costs():
    rs = np.random.RandomState(33)
    Tn, M = 6, 10
    st = T(rs.randn(Tn, M, 4))
    target = T([[0.3, -0.2, 1.0]])
    ls = T([1.5, 0.7, 2.0])
    active = [0, 1, 2]
    out = dict(states=N(st), target=N(target), lengthscales=N(ls), active_dims=np.array(active))
    for tag, cls in (("dist", RC.Expected_distance), ("sat", RC.Expected_saturated_distance)):
        s = st.clone().requires_grad_(True)
        cf = cls(target_state=target, lengthscales=ls, active_dims=active)
        cst, sd = cf(s, None, 0)
        cst.backward()
        out.update({tag + "_cost": N(cst), tag + "_std": N(sd), tag + "_grad": N(s.grad)})
    save("simple_costs", **out)


simple_costs()


# ---------------------------------------------------------------------------------------
# mean-only rollout of a recorded trajectory
# ---------------------------------------------------------------------------------------
def mean_rollout():
    ml, xtr, utr, sig = speed_model(100)
    obj, pi = mcpilco(ml, 16, 8)
    x_rec, u_rec = cp[2][0][:25], cp[2][1][:25]
    obj.state_samples_history = [x_rec]
    obj.input_samples_history = [u_rec]
    with torch.no_grad(), quiet:
        traj = obj.rollout(data_collection_index=0)
        traj12 = obj.rollout(data_collection_index=0, T_rollout=12)
    save("mean_rollout", states_tr=xtr, inputs_tr=utr, sigma_n=sig, x_rec=x_rec, u_rec=u_rec, traj=traj, traj12=traj12)


mean_rollout()


# ---------------------------------------------------------------------------------------
# SOD with a seeded permutation
# ---------------------------------------------------------------------------------------
def sod_permutation():
    Zc, Yc = sy.gp_io(cp, c["angle"], c["not_angle"], c["vel"])
    sig = 0.36
    with quiet:
        gp = RSGP.RBF(**rbf_dict(6, c["lengthscales"], sig))
    X, Y = T(Zc[:120]), T(Yc[0][:120])
    torch.manual_seed(55)
    with torch.no_grad(), quiet:
        thr = 0.5 * torch.sqrt(gp.get_sigma_n_2())
        idx = [int(i) for i in gp.get_SOD(X, Y, thr, flg_permutation=True)]
    torch.manual_seed(55)
    perm = torch.arange(1, 120)[torch.randperm(119)]
    # smallest decision margin along the permuted greedy path
    keep, mm = [0], np.inf
    with torch.no_grad(), quiet:
        for i in [int(v) for v in perm]:
            _, var, _ = gp.get_estimate(X[keep, :], Y[keep, :], X[i:i + 1, :])
            mm = min(mm, abs(float(torch.sqrt(var)) - float(thr)))
            if float(torch.sqrt(var)) > float(thr):
                keep.append(i)
    assert keep == idx
    save("sod_permutation", X=N(X), Y=N(Y), lengthscales=c["lengthscales"], sigma_n=sig, thr=float(thr), seed=55, perm=N(perm), idx=np.array(idx),
         min_margin=mm)


sod_permutation()


# ---------------------------------------------------------------------------------------
# a log.pkl in the reference's format, from reference objects
# ---------------------------------------------------------------------------------------
def ref_log():
    par = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"],
               not_vel_indeces=c["not_vel"], dtype=dtype, device=dev,
               init_dict_list=[rbf_dict(6, np.array(c["lengthscales"]) * (1.0 + 0.1 * g), 0.03 + 0.01 * g) for g in range(2)])
    with quiet:
        ml = RML.Speed_Model_learning_RBF_angle_state(**par)
    # two data collections (exploration + trial 0), as MC_PILCO.reinforce logs them (MC_PILCO.py:161-172, 214-221, 245-250)
    hist_x = [cp[0][0][:31], cp[1][0][:41]]
    hist_u = [cp[0][1][:31], cp[1][1][:41]]
    with quiet:
        for x, u in zip(hist_x, hist_u):
            ml.add_data(x, u)
        with torch.no_grad():
            for g in range(2):
                ml.pretrain_gp(g)
    obj, pi = mcpilco(ml, 16, 8)
    log = {
        "parameters_gp_0": [{k: v.clone() for k, v in ml.gp_list[g].state_dict().items()} for g in range(2)],
        "gp_inputs_0": ml.gp_inputs.clone(), "gp_output_list_0": [y.clone() for y in ml.gp_output_list],
        "state_samples_history": hist_x, "input_samples_history": hist_u, "noiseless_states_history": [x.copy() for x in hist_x],
        "cost_trial_list": [np.array([3.0, 2.5])], "std_cost_trial_list": [np.array([0.1, 0.2])],
        "parameters_trial_list": [{k: v.clone() for k, v in obj.control_policy.state_dict().items()}],
        "particles_states_list": [np.zeros((2, 3, 4))], "particles_inputs_list": [np.zeros((2, 3, 1))],
    }
    pkl.dump(log, open(os.path.join(HERE, "ref_log.pkl"), "wb"))
    print("wrote ref_log.pkl", sorted(log.keys()))
    save("ref_log_expect", alpha0=N(ml.alpha_list[0]), alpha1=N(ml.alpha_list[1]), gp_inputs=N(ml.gp_inputs),
         ls0=N(torch.exp(ml.gp_list[0].log_lengthscales_par)), ls1=N(torch.exp(ml.gp_list[1].log_lengthscales_par)),
         pol_centers=N(obj.control_policy.centers))


ref_log()


# ---------------------------------------------------------------------------------------
# options no launch script uses: policy scale_factor (Policy.py:220-222, 252), per-trial cost lengthscales (Cost_function.py:136-141)
# ---------------------------------------------------------------------------------------
def options():
    rs = np.random.RandomState(41)
    M, B = 12, 24
    sf = np.array([2.0, 0.5, 3.0, 1.5])
    with quiet:
        pol = RP.Sum_of_gaussians(state_dim=4, input_dim=2, num_basis=B, lengthscales_init=0.8 + rs.rand(4), centers_init=rs.randn(B, 4),
                                  weight_init=rs.randn(2, B), flg_squash=True, u_max=[3.0, 1.5], scale_factor=sf, flg_drop=True, dtype=dtype,
                                  device=dev)
    x = T(2.0 * rs.randn(M, 4))
    wsum = T(rs.randn(M, 2))
    u = pol(x, t=0, p_dropout=0.0)
    (u * wsum).sum().backward()
    out = dict(sf_x=N(x), sf_scale=sf, sf_ls=N(torch.exp(pol.log_lengthscales)), sf_centers=N(pol.centers), sf_weight=N(pol.f_linear.weight),
               sf_umax=np.array([3.0, 1.5]), sf_u=N(u), sf_wsum=N(wsum), sf_g_log_ls=N(pol.log_lengthscales.grad), sf_g_centers=N(pol.centers.grad),
               sf_g_weight=N(pol.f_linear.weight.grad))
    Tn = 6
    tt = sy.ur5_target_traj(T=Tn)
    st = T(tt.reshape(Tn, 1, 12) + 0.4 * rs.randn(Tn, 10, 12)).requires_grad_(True)
    ls_all = T(np.stack([np.array(sy.UR5["cost_ls"]), 0.5 + rs.rand(12)]))
    cf = RC.Expected_saturated_distance_from_trajectory(target_traj=T(tt), lengthscales=ls_all, flg_var_lengthscales=True, used_indeces=list(range(12)))
    cst, sd = cf(st, None, 1)
    cst.backward()
    out.update(vl_states=N(st), vl_target=tt, vl_ls_all=N(ls_all), vl_trial=1, vl_cost=N(cst), vl_std=N(sd), vl_grad=N(st.grad))
    save("options", **out)


options()


# ---------------------------------------------------------------------------------------
# policy with an output bias (flg_bias, Policy.py:203-212): full apply_policy + cost + backward, noise recovered by replay
# ---------------------------------------------------------------------------------------
def rollout_bias(M=16, Tn=8, p=0.25, seed=108, B=32):
    ml, xtr, utr, sig = speed_model(100)
    pi = sy.cartpole_policy_init(B=B, seed=9)
    ppar = dict(state_dim=4, input_dim=1, num_basis=B, angle_indices=np.array([2]), non_angle_indices=np.array([0, 1, 3]),
                lengthscales_init=pi["lengthscales"], centers_init=pi["centers"], weight_init=pi["weight"], flg_squash=True, u_max=c["u_max"],
                flg_bias=True, bias_init=np.array([1.7]), flg_train_bias=True, flg_drop=True, dtype=dtype, device=dev)
    with quiet:
        obj = RMC.MC_PILCO(T_sampling=c["Ts"], state_dim=4, input_dim=1, f_sim=lambda y, t, u: None, f_model_learning=lambda **kw: ml,
                           model_learning_par={}, f_rand_exploration_policy=RP.Random_exploration,
                           rand_exploration_policy_par=dict(state_dim=4, input_dim=1, u_max=1.0, dtype=dtype, device=dev),
                           f_control_policy=RP.Sum_of_gaussians_with_angles, control_policy_par=ppar, f_cost_function=RC.Cart_pole_cost,
                           cost_function_par=dict(target_state=T(c["cost_target"]), lengthscales=T(c["cost_ls"]), angle_index=2, pos_index=0),
                           log_path=None, dtype=dtype, device=dev)
    pol = obj.control_policy
    x0m, x0v = T(c["x0_mean"]), T(np.array([1e-2, 1e-2, 4e-2, 1e-2]))
    torch.manual_seed(seed)
    st, inp = obj.apply_policy(particles_initial_state_mean=x0m, particles_initial_state_var=x0v, flg_particles_init_uniform=False,
                               particles_init_up_bound=None, particles_init_low_bound=None, flg_particles_init_multi_gauss=False,
                               num_particles=M, T_control=Tn, p_dropout=p)
    cost, std = obj.cost_function(st, inp, 0)
    cost.backward()
    torch.manual_seed(seed)
    eps0 = torch.empty(M, 4, dtype=dtype).normal_()
    masks = [torch.empty(M, 1, B, dtype=dtype).bernoulli_(1 - p).reshape(M, B)]
    eps = []
    for _ in range(1, Tn):
        eps.append(torch.empty(M, 2, dtype=dtype).normal_())
        masks.append(torch.empty(M, 1, B, dtype=dtype).bernoulli_(1 - p).reshape(M, B))
    x0 = x0m.reshape(1, -1) + torch.sqrt(x0v).reshape(1, -1) * eps0
    assert torch.equal(x0, st[0].detach()), "noise replay does not reproduce the reference's x0"
    save("rollout_bias", states_tr=xtr, inputs_tr=utr, sigma_n=sig, x0_mean=N(x0m), x0_var=N(x0v), eps=N(torch.stack(eps)),
         masks=N(torch.stack(masks)).astype(np.uint8), p_drop=p, seed=seed, states=N(st), inputs=N(inp), cost=N(cost), std=N(std),
         pol_ls=N(torch.exp(pol.log_lengthscales)), pol_centers=N(pol.centers), pol_weight=N(pol.f_linear.weight), pol_bias=N(pol.f_linear.bias),
         g_log_ls=N(pol.log_lengthscales.grad), g_centers=N(pol.centers.grad), g_weight=N(pol.f_linear.weight.grad), g_bias=N(pol.f_linear.bias.grad))


rollout_bias()
print("done")
