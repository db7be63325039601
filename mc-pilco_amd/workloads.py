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
    "c5": ("ur5", 1, 400, 2000, 300),       # configs[4]: UR5 12-D state, 6 GPs, SE + polynomial(1)
    "tiny": ("cartpole", 0, 48, 16, 6),
    "tiny_ur5": ("ur5", 1, 40, 8, 5),
}


def numpy_problem(name, N=None, seed=1):
    """Plain-array description of a workload: training data, hyper-parameters, policy init."""
    system, deg, N0, M, T = CONFIGS[name]
    N = N or N0
    rng = np.random.RandomState(seed + 100)
    if system == "cartpole":
        c = sy.CARTPOLE
        n_roll = (N + 59) // 60
        Z, Ys = sy.gp_io(sy.cartpole_rollouts(n_roll=n_roll, seed=seed), c["angle"], c["not_angle"], c["vel"])
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
                policy_extra=extra, target_traj=target)


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

    def sample_x0(self, M=None, generator=None):
        M = M or self.M
        e = torch.randn(M, self.x0_mean.numel(), dtype=DT, device=self.x0_mean.device, generator=generator)
        return self.x0_mean + self.x0_std * e


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
    return Workload(name, model, policy, cost, [log_ls, centers, weight], x0m, x0s, M, T, p_drop, pb)


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


def flops_per_particle_step(w: Workload):
    """SURVEY.md 8d algorithmic flops per particle-step (forward + backward)."""
    P, B, U, D = w.policy.P, w.policy.B, w.policy.U, w.model.D
    f = 6 * B * (P + U + 2)
    for gp in w.model.gps:
        N = gp.N
        deg = gp.spec.poly_deg
        fpoly = 0
        if deg >= 1:
            fpoly += (2 * (D + 1) + 2) * N
        if deg >= 2:
            fpoly += (4 * D + 3) * N
        f += 2 * N * N + (3 * D + 8) * N + fpoly  # forward
        f += (6 * D + 6) * N + fpoly  # backward
    return f
