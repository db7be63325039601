"""Shared builders: fixture arrays -> oracle objects (tests only)."""
import numpy as np
import torch

from mc_pilco_amd import synthetic as sy
from oracle import mcpilco_oracle as orc

DT = torch.float64


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=DT)


def hyper(ls, sigma_n, lam=1.0, poly_w=None):
    return orc.GPHyper(
        log_ls=torch.log(T(ls)).reshape(-1),
        log_lambda=torch.log(T([lam])),
        log_sigma_n=torch.log(T([float(sigma_n)])),
        poly_log_par=None if poly_w is None else [torch.log(T(w)).reshape(-1) for w in poly_w],
    )


def kind_cfg(kind):
    return sy.UR5 if kind == "ur5" else sy.CARTPOLE


def fixture_poly(fx, g):
    ws = []
    k = 1
    while "poly_w%d_gp%d" % (k, g) in fx:
        ws.append(fx["poly_w%d_gp%d" % (k, g)])
        k += 1
    return ws or None


def oracle_model(fx, kind, from_cache=True, sod=False):
    """SpeedModel from a rollout/step fixture.  from_cache=True uses the reference's cached
    operands (alpha, Kinv, X_tr) stored in the fixture; False re-derives them with the oracle's
    own pretrain (Gram -> Cholesky -> inverse [-> SOD])."""
    c = kind_cfg(kind)
    G = c["G"]
    hyp = [hyper(c["lengthscales"], float(fx["sigma_n"]), c["lam"], fixture_poly(fx, g)) for g in range(G)]
    caches = []
    if from_cache:
        for g in range(G):
            caches.append(orc.GPCache(T(fx["Xtr%d" % g]), T(fx["alpha%d" % g]), T(fx["Kinv%d" % g]), torch.zeros(fx["Xtr%d" % g].shape[0], 1, dtype=DT),
                                      None if "sod%d" % g not in fx else [int(i) for i in fx["sod%d" % g]]))
    else:
        Z, Ys = orc.speed_model_io(fx["states_tr"], fx["inputs_tr"], c["angle"], c["not_angle"], c["vel"])
        for g in range(G):
            caches.append(orc.pretrain_gp(hyp[g], Z, Ys[g], "relative" if sod else None, 0.5 if sod else None))
    return orc.SpeedModel(hyp, caches, c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])


def oracle_policy(fx, kind):
    c = kind_cfg(kind)
    if kind == "ur5":
        return orc.PolicyPar(torch.log(T(fx["pol_ls"])), T(fx["pol_centers"]), T(fx["pol_weight"]), c["u_max"], "traj", target_traj=T(fx["target_traj"]))
    return orc.PolicyPar(torch.log(T(fx["pol_ls"])), T(fx["pol_centers"]), T(fx["pol_weight"]), c["u_max"], "angles", angle=[2], non_angle=[0, 1, 3])


def oracle_cost_fn(fx, kind):
    c = kind_cfg(kind)
    if kind == "ur5":
        tt = T(fx["target_traj"])
        ls = T(c["cost_ls"])
        return lambda st: orc.traj_cost(st, tt, ls)
    tg = T(c["cost_target"])
    ls = T(c["cost_ls"])
    return lambda st: orc.cart_pole_cost(st, tg, ls, c["cost_angle_index"], c["cost_pos_index"])


ROLLOUT_FIXTURES = [
    ("rollout_se", "se"),
    ("rollout_se_nodrop", "se"),
    ("rollout_se_sod", "se"),
    ("rollout_se_poly2", "se"),
    ("rollout_ur5", "ur5"),
    ("rollout_se_long", "se"),
]
