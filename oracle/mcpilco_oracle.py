"""CPU oracle for the MC-PILCO particle-rollout / GP-dynamics hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker (or as the timed CPU baseline) -- never as a fallback
for the HIP path.

This file is a from-scratch, functional restatement (PyTorch CPU, float64) of the
algorithm in the reference repository (paths relative to the reference root, cited per
function as ``file:line``).  Parity is PINNED: ``tests/golden/*.npz`` were produced by
importing the reference itself (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks every function below against them.

Conventions: all tensors float64, row-major.  M particles, S state dim, U input dim,
G GPs, N training points, D GP-input dim, B basis functions, P policy-feature dim.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

DT = torch.float64


# --------------------------------------------------------------------------------------
# kernel hyper-parameters
# --------------------------------------------------------------------------------------
@dataclass
class GPHyper:
    """Hyper-parameters of one GP of the dynamics model.

    SE part:  gpr_lib/GP_prior/Stationary_GP.py:112-181 (class RBF)
    poly part (optional): gpr_lib/GP_prior/Sparse_GP.py:559-737 (MPK_GP, get_Volterra_MPK_GP)
    sum:  gpr_lib/GP_prior/GP_prior.py:299-347 (Sum_Independent_GP)
    """

    log_ls: torch.Tensor  # [D]   log lengthscales (ARD)
    log_lambda: torch.Tensor  # [1]
    log_sigma_n: torch.Tensor  # [1]   log noise std
    mean: torch.Tensor = field(default_factory=lambda: torch.zeros(1, dtype=DT))
    sigma_n_num: float = 0.0
    # polynomial (Volterra MPK) part: list over degrees k=1..deg of the raw (log) parameter
    # vectors; entry k-1 has D+1 values for k==1 (offset feature) and k*D values for k>=2.
    poly_log_par: Optional[List[torch.Tensor]] = None

    @property
    def D(self) -> int:
        return int(self.log_ls.numel())

    def sigma_n_2(self) -> torch.Tensor:
        # GP_prior.py:87-89
        return torch.exp(self.log_sigma_n) ** 2 + self.sigma_n_num**2


def se_sqdist(A: torch.Tensor, Bm: torch.Tensor, log_ls: torch.Tensor) -> torch.Tensor:
    """Lengthscale-weighted squared distances, expanded form.

    Stationary_GP.py:65-109 -- ||a/l||^2 + ||b/l||^2 - 2 (a/l)(b/l)^T (may be -1e-16).
    """
    ls = torch.exp(log_ls)
    a = A / ls
    b = Bm / ls
    a2 = (a * a).sum(1, keepdim=True)
    b2 = (b * b).sum(1, keepdim=True)
    return a2 + b2.t() - 2.0 * (a @ b.t())


def mpk_scales(par: torch.Tensor, k: int) -> List[torch.Tensor]:
    """Per-factor diagonal weights s_d of MPK_k:  s_d = (k-d) * exp(par[d*n:(d+1)*n]).

    Sparse_GP.py:613-623 -- the loop ``for deg in range(current_deg, poly_deg)`` re-adds the
    *same* slice (k-d) times; the Sigma is diag(s_d**2)
    (gpr_lib/Utils/Parameters_covariance_functions.py:18-27, flg_ARD=True).
    """
    n = par.numel() // k
    return [(k - d) * torch.exp(par[d * n : (d + 1) * n]) for d in range(k)]


def poly_cov(h: GPHyper, X1: torch.Tensor, X2: torch.Tensor) -> torch.Tensor:
    """Volterra MPK covariance  sum_k prod_{d<k} phi(X1) diag(s_kd^2) phi(X2)^T.

    Sparse_GP.py:426-441 (Linear_GP.get_covariance), :625-646 (product over degrees),
    :671-737 (phi=[x,1] only for MPK_1).
    """
    out = torch.zeros(X1.shape[0], X2.shape[0], dtype=DT)
    for k, par in enumerate(h.poly_log_par, start=1):
        if k == 1:
            p1 = torch.cat([X1, torch.ones(X1.shape[0], 1, dtype=DT)], 1)
            p2 = torch.cat([X2, torch.ones(X2.shape[0], 1, dtype=DT)], 1)
        else:
            p1, p2 = X1, X2
        term = torch.ones_like(out)
        for s in mpk_scales(par, k):
            term = term * ((p1 * (s * s)) @ p2.t())
        out = out + term
    return out


def poly_diag(h: GPHyper, X: torch.Tensor) -> torch.Tensor:
    """Sparse_GP.py:443-453, :658-668 -- diagonal of the Volterra MPK covariance."""
    out = torch.zeros(X.shape[0], dtype=DT)
    for k, par in enumerate(h.poly_log_par, start=1):
        p = torch.cat([X, torch.ones(X.shape[0], 1, dtype=DT)], 1) if k == 1 else X
        term = torch.ones_like(out)
        for s in mpk_scales(par, k):
            term = term * ((p * (s * s)) * p).sum(1)
        out = out + term
    return out


def gp_cov(h: GPHyper, X1: torch.Tensor, X2: Optional[torch.Tensor] = None, noise: bool = False) -> torch.Tensor:
    """Covariance of the (sum) GP.

    Stationary_GP.py:162-170 (lambda * exp(-dist), no 1/2);  GP_prior.py:314-335 (children
    summed with flg_noise=False, sigma_n^2 I added once, sigma_n^2 = the SE child's).
    """
    Xb = X1 if X2 is None else X2
    K = torch.exp(h.log_lambda) * torch.exp(-se_sqdist(X1, Xb, h.log_ls))
    if h.poly_log_par is not None:
        K = K + poly_cov(h, X1, Xb)
    if noise:
        K = K + h.sigma_n_2() * torch.eye(X1.shape[0], dtype=DT)
    return K


def gp_diag(h: GPHyper, X: torch.Tensor) -> torch.Tensor:
    """Stationary_GP.py:172-181; GP_prior.py:337-347 (noise-free prior variance)."""
    d = torch.exp(h.log_lambda) * torch.ones(X.shape[0], dtype=DT)
    if h.poly_log_par is not None:
        d = d + poly_diag(h, X)
    return d


def gp_mean(h: GPHyper, X: torch.Tensor) -> torch.Tensor:
    """Stationary_GP.py:157-160; GP_prior.py:306-312 (only the first child's mean)."""
    return h.mean.reshape(1, 1).repeat(X.shape[0], 1)


# --------------------------------------------------------------------------------------
# Gram / Cholesky / alpha / posterior
# --------------------------------------------------------------------------------------
def gp_forward(h: GPHyper, X: torch.Tensor):
    """GP_prior.py:91-115 -- K+sigma^2 I, U=chol (upper), logdet, U^-1, K^-1=U^-1 U^-T."""
    K = gp_cov(h, X, None, noise=True)
    Uc = torch.linalg.cholesky(K, upper=True)
    logdet = 2.0 * torch.log(torch.diagonal(Uc)).sum()
    Ui = torch.linalg.inv(Uc)
    Kinv = Ui @ Ui.t()
    return gp_mean(h, X), K, Kinv, logdet


def gp_alpha(h: GPHyper, X: torch.Tensor, Y: torch.Tensor):
    """GP_prior.py:130-135."""
    mX, _, Kinv, _ = gp_forward(h, X)
    return Kinv @ (Y - mX), mX, Kinv


def gp_estimate_from_alpha(h: GPHyper, X, Xs, alpha, Kinv=None):
    """GP_prior.py:137-155 -- mu = m + k a ; var = diag k(z,z) - rowsum((k Kinv) * k)."""
    Ks = gp_cov(h, Xs, X)
    mu = gp_mean(h, Xs) + Ks @ alpha
    if Kinv is None:
        return mu
    var = gp_diag(h, Xs) - ((Ks @ Kinv) * Ks).sum(1)
    return mu, var


def gp_estimate(h: GPHyper, X, Y, Xs):
    """GP_prior.py:157-171."""
    alpha, mX, Kinv = gp_alpha(h, X, Y)
    mu, var = gp_estimate_from_alpha(h, X, Xs, alpha, Kinv)
    return mu, var, alpha, mX, Kinv


def gp_get_sod(h: GPHyper, X, Y, threshold, perm: Optional[torch.Tensor] = None) -> List[int]:
    """Greedy subset-of-data selection, GP_prior.py:232-257.

    Start from sample 0; candidate i joins when sqrt(var_i | current subset) > threshold,
    the subset posterior being refactored from scratch for every candidate.  Returns plain
    python ints (the reference returns a mix of int and 0-dim long tensors).
    """
    keep = [0]
    order = range(1, X.shape[0]) if perm is None else [int(i) for i in perm]
    for i in order:
        _, var, *_ = gp_estimate(h, X[keep, :], Y[keep, :], X[i : i + 1, :])
        if bool(torch.sqrt(var) > threshold):
            keep.append(i)
    return keep


@dataclass
class GPCache:
    """What ``Model_learning.pretrain_gp`` caches per GP (Model_learning.py:163-208)."""

    X: torch.Tensor  # [N,D] training inputs kept (SOD subset or all)
    alpha: torch.Tensor  # [N,1]
    Kinv: torch.Tensor  # [N,N]
    mX: torch.Tensor  # [N,1]
    sod: Optional[List[int]] = None


def pretrain_gp(h: GPHyper, X, Y, sod_mode: Optional[str] = None, sod_threshold=None) -> GPCache:
    """Model_learning.py:163-208.  sod_mode in {None, "relative", "absolute"}."""
    if sod_mode is None:
        _, _, alpha, mX, Kinv = gp_estimate(h, X, Y, X)
        return GPCache(X, alpha, Kinv, mX, None)
    thr = sod_threshold * torch.sqrt(h.sigma_n_2()) if sod_mode == "relative" else sod_threshold
    idx = gp_get_sod(h, X, Y, thr)
    _, _, alpha, mX, Kinv = gp_estimate(h, X[idx, :], Y[idx, :], X)
    return GPCache(X[idx, :], alpha, Kinv, mX, idx)


# --------------------------------------------------------------------------------------
# dynamics model step (speed-integration model)
# --------------------------------------------------------------------------------------
@dataclass
class SpeedModel:
    """Speed_Model_learning_RBF(_MPK)_angle_state, Model_learning.py:619-760."""

    hyp: List[GPHyper]
    cache: List[GPCache]
    Ts: float
    angle: Sequence[int]
    not_angle: Sequence[int]
    vel: Sequence[int]
    not_vel: Sequence[int]


def gp_features(x: torch.Tensor, u: torch.Tensor, angle, not_angle) -> torch.Tensor:
    """Model_learning.py:670-683 -- z = [x_notangle, sin(x_angle), cos(x_angle), u]."""
    return torch.cat([x[:, list(not_angle)], torch.sin(x[:, list(angle)]), torch.cos(x[:, list(angle)]), u], 1)


def one_step_gp_out(m: SpeedModel, x, u):
    """Model_learning.py:231-242, :265-289/:315-336 -- per-GP mean [M,1] and var [M,1]."""
    z = gp_features(x, u, m.angle, m.not_angle)
    mus, vrs = [], []
    for h, c in zip(m.hyp, m.cache):
        mu, var = gp_estimate_from_alpha(h, c.X, z, c.alpha, c.Kinv)
        mus.append(mu)
        vrs.append(var.reshape(-1, 1))
    return z, mus, vrs


def next_state(m: SpeedModel, x, u, eps: Optional[torch.Tensor], particle_pred: bool = True):
    """Model_learning.py:210-229 and :685-718.

    delta = mu + sqrt(var)*eps (``Normal(mu, sqrt(var)).rsample()``), or mu when
    particle_pred is False;  v' = v + delta ; q' = q + Ts v + Ts/2 delta.
    ``eps`` None -> drawn here with the same call the reference makes
    (``torch.empty(M,G).normal_()`` inside Normal.rsample).
    """
    _, mus, vrs = one_step_gp_out(m, x, u)
    dmu = torch.cat(mus, 1)
    dvar = torch.cat(vrs, 1)
    if particle_pred:
        if eps is None:
            eps = torch.empty(dmu.shape, dtype=DT).normal_()
        delta = dmu + torch.sqrt(dvar) * eps
    else:
        delta = dmu
    nxt = torch.zeros_like(x)
    nxt[:, list(m.vel)] = x[:, list(m.vel)] + delta
    nxt[:, list(m.not_vel)] = x[:, list(m.not_vel)] + m.Ts * x[:, list(m.vel)] + m.Ts / 2 * delta
    return nxt, dmu, dvar


# --------------------------------------------------------------------------------------
# policy
# --------------------------------------------------------------------------------------
@dataclass
class PolicyPar:
    """Sum_of_gaussians family, policy_learning/Policy.py:153-403."""

    log_ls: torch.Tensor  # [1,P]
    centers: torch.Tensor  # [B,P]
    weight: torch.Tensor  # [U,B]   (f_linear.weight, no bias)
    u_max: object  # float or list[U]
    kind: str = "plain"  # "plain" | "angles" | "traj"
    angle: Sequence[int] = ()
    non_angle: Sequence[int] = ()
    target_traj: Optional[torch.Tensor] = None  # [T,S]
    squash: bool = True
    bias: Optional[torch.Tensor] = None  # [U] f_linear.bias (flg_bias, Policy.py:203-212)
    scale_factor: Optional[torch.Tensor] = None  # [P]; states / scale_factor before the RBF layer (plain class only, Policy.py:220-222, 252)


def policy_features(pp: PolicyPar, x: torch.Tensor, t: Optional[int]) -> torch.Tensor:
    """Policy.py:326-333 ([x_nonangle, cos, sin] -- cos BEFORE sin) and :397-399 ([x, x*_t - x])."""
    if pp.kind == "angles":
        return torch.cat([x[:, list(pp.non_angle)], torch.cos(x[:, list(pp.angle)]), torch.sin(x[:, list(pp.angle)])], 1)
    if pp.kind == "traj":
        tgt = pp.target_traj[t, :].reshape(1, -1).expand(x.shape[0], -1)
        return torch.cat([x, tgt - x], 1)
    return x


def policy_forward(pp: PolicyPar, x, t=None, mask: Optional[torch.Tensor] = None, p_drop: float = 0.0):
    """Policy.py:242-265 + squashing :52-60.

    phi_b = exp(-(||s/l||^2 + ||c_b/l||^2 - 2 (s/l)(c_b/l)^T)); dropout multiplies by
    mask/(1-p) (torch.nn.functional.dropout, training=True; no RNG draw at p == 0);
    u = u_max * tanh((phi W^T)/u_max).  ``mask`` [M,B] of {0,1}; None with p>0 -> drawn here
    exactly like F.dropout does on CPU (``empty_like(phi).bernoulli_(1-p)``).
    """
    s = policy_features(pp, x, t)
    if pp.scale_factor is not None:
        s = s / pp.scale_factor.reshape(1, -1)
    ls = torch.exp(pp.log_ls)
    a = s / ls
    c = pp.centers / ls
    dist = (a * a).sum(1, keepdim=True) + (c * c).sum(1, keepdim=True).t() - 2.0 * (a @ c.t())
    phi = torch.exp(-dist)
    if p_drop > 0.0:
        if mask is None:
            mask = torch.empty(phi.shape[0], 1, phi.shape[1], dtype=DT).bernoulli_(1 - p_drop).reshape(phi.shape)
        phi = phi * (mask / (1.0 - p_drop))
    lin = phi @ pp.weight.t()
    if pp.bias is not None:
        lin = lin + pp.bias.reshape(1, -1)
    if not pp.squash:
        return lin
    um = pp.u_max if isinstance(pp.u_max, (int, float)) else torch.tensor(pp.u_max, dtype=DT)
    return um * torch.tanh(lin / um)


# --------------------------------------------------------------------------------------
# cost
# --------------------------------------------------------------------------------------
def cart_pole_cost(states, target_state, lengthscales, angle_index, pos_index):
    """Cost_function.py:170-182.  target_state=[theta*, x*], lengthscales=[l_theta, l_x]."""
    x = states[:, :, pos_index]
    th = states[:, :, angle_index]
    return 1 - torch.exp(-(((torch.abs(th) - target_state[0]) / lengthscales[0]) ** 2) - ((x - target_state[1]) / lengthscales[1]) ** 2)


def traj_cost(states, target_traj, lengthscales, used=None):
    """Cost_function.py:124-147 (flg_var_lengthscales=False) -- 1-exp(-sum(((x-x*_t)/l)^2))."""
    used = list(range(states.shape[2])) if used is None else list(used)
    tg = target_traj.reshape(target_traj.shape[0], 1, -1)
    return 1 - torch.exp(-(((states[:, :, used] - tg[:, :, used]) / lengthscales) ** 2).sum(2))


def expected_cost(costs: torch.Tensor):
    """Cost_function.py:25-36 -- (sum_t mean_m c, sum_t std_m c) with unbiased, detached std."""
    return costs.mean(1).sum(), costs.detach().std(1).sum()


# --------------------------------------------------------------------------------------
# rollout
# --------------------------------------------------------------------------------------
def sample_x0(mean: torch.Tensor, var: torch.Tensor, M: int, eps0: Optional[torch.Tensor] = None):
    """policy_learning/MC_PILCO.py:650-657 -- MultivariateNormal(mean, diag(var)).rsample()
    == mean + sqrt(var) * eps0 with eps0 = torch.empty(M,S).normal_() (bit-exact)."""
    if eps0 is None:
        eps0 = torch.empty(M, mean.numel(), dtype=DT).normal_()
    return mean.reshape(1, -1) + torch.sqrt(var).reshape(1, -1) * eps0


def apply_policy(
    m: SpeedModel,
    pp: PolicyPar,
    x0: torch.Tensor,
    T: int,
    p_drop: float = 0.0,
    eps: Optional[torch.Tensor] = None,
    masks: Optional[torch.Tensor] = None,
    particle_pred: bool = True,
):
    """policy_learning/MC_PILCO.py:615-674 (the T-loop after x0 has been sampled).

    eps [T-1,M,G] / masks [T,M,B] given -> injected;  None -> drawn from the torch CPU
    generator in the reference's order: mask_0, then for t=1..T-1: eps_t, mask_t.
    Returns states [T,M,S], inputs [T,M,U].
    """
    xs = [x0]
    us = [policy_forward(pp, x0, 0, None if masks is None else masks[0], p_drop)]
    for t in range(1, T):
        e = None if eps is None else eps[t - 1]
        x, _, _ = next_state(m, xs[-1], us[-1], e, particle_pred)
        xs.append(x)
        us.append(policy_forward(pp, x, t, None if masks is None else masks[t], p_drop))
    return torch.stack(xs), torch.stack(us)


def butter1(fc: float):
    """First-order digital Butterworth low-pass, cutoff ``fc`` as a fraction of Nyquist (what ``scipy.signal.butter(1, fc)``
    returns, MC_PILCO.py:859): bilinear transform of 1/(s+1) pre-warped to tan(pi fc / 2)."""
    import math
    w = math.tan(math.pi * fc / 2.0)
    return [w / (1.0 + w), w / (1.0 + w)], [1.0, (w - 1.0) / (w + 1.0)]


def apply_policy_pms(
    m: SpeedModel,
    pp: PolicyPar,
    x0: torch.Tensor,
    T: int,
    pos: Sequence[int],
    vel: Sequence[int],
    std_pos: torch.Tensor,
    fc: float,
    p_drop: float = 0.0,
    eps: Optional[torch.Tensor] = None,
    masks: Optional[torch.Tensor] = None,
    pos_noise: Optional[torch.Tensor] = None,
):
    """policy_learning/MC_PILCO.py:808-906 (MC_PILCO4PMS.apply_policy, the T-loop after x0 has been sampled).

    The particles evolve on their true states; the policy is fed a *measured* state: positions plus Gaussian noise
    (:881-885), velocities by backward difference of the noisy positions (:888-891) passed through the first-order
    Butterworth filter (:895-899); at t=0 measured = true (:856).  eps [T-1,M,G], pos_noise [T-1,M,len(pos)] (standard
    normal, scaled here by std_pos) and masks [T,M,B] are injected; None draws from the torch CPU generator in the
    reference's order (mask_0; per step: eps_t, position noise, mask_t).
    """
    b, a = butter1(fc)
    Ts = m.Ts
    pos, vel = list(pos), list(vel)
    xs = [x0]
    noisy_prev = x0.clone()
    meas_prev = x0.clone()
    us = [policy_forward(pp, meas_prev, 0, None if masks is None else masks[0], p_drop)]
    for t in range(1, T):
        e = None if eps is None else eps[t - 1]
        x, _, _ = next_state(m, xs[-1], us[-1], e, True)
        xs.append(x)
        n = torch.randn(x.shape[0], len(pos), dtype=DT) if pos_noise is None else pos_noise[t - 1]
        noisy = x.clone()
        noisy[:, pos] = noisy[:, pos] + std_pos * n
        noisy[:, vel] = (noisy[:, pos] - noisy_prev[:, pos]) / Ts
        meas = noisy.clone()
        meas[:, vel] = (b[0] * noisy[:, vel] + b[1] * noisy_prev[:, vel] - a[1] * meas_prev[:, vel]) / a[0]
        us.append(policy_forward(pp, meas, t, None if masks is None else masks[t], p_drop))
        noisy_prev, meas_prev = noisy, meas
    return torch.stack(xs), torch.stack(us)


def draw_noise(M: int, S: int, G: int, B: int, T: int, p_drop: float):
    """Draws (eps0 [M,S], eps [T-1,M,G], masks [T,M,B] or None) from the torch CPU generator
    in exactly the order ``MC_PILCO.apply_policy`` consumes it (verified by
    tests/golden/make_golden.py against the reference run with the same seed)."""
    eps0 = torch.empty(M, S, dtype=DT).normal_()
    masks = [] if p_drop > 0 else None
    if masks is not None:
        masks.append(torch.empty(M, 1, B, dtype=DT).bernoulli_(1 - p_drop).reshape(M, B))
    eps = []
    for _ in range(1, T):
        eps.append(torch.empty(M, G, dtype=DT).normal_())
        if masks is not None:
            masks.append(torch.empty(M, 1, B, dtype=DT).bernoulli_(1 - p_drop).reshape(M, B))
    eps = torch.stack(eps) if eps else torch.zeros(0, M, G, dtype=DT)
    return eps0, eps, (torch.stack(masks) if masks is not None else None)


def policy_grad_step(m: SpeedModel, pp: PolicyPar, x0, T, cost_fn, p_drop=0.0, eps=None, masks=None):
    """One iteration of MC_PILCO.reinforce_policy's hot loop, MC_PILCO.py:484-522:
    apply_policy -> expected cost -> backward.  Returns (cost, std, grads dict, states, inputs)."""
    prm = [pp.log_ls, pp.centers, pp.weight]
    for p in prm:
        p.requires_grad_(True)
        p.grad = None
    states, inputs = apply_policy(m, pp, x0, T, p_drop, eps, masks)
    cost, std = expected_cost(cost_fn(states))
    cost.backward()
    g = {"log_ls": pp.log_ls.grad.clone(), "centers": pp.centers.grad.clone(), "weight": pp.weight.grad.clone()}
    for p in prm:
        p.requires_grad_(False)
        p.grad = None
    return cost.detach(), std.detach(), g, states.detach(), inputs.detach()


# --------------------------------------------------------------------------------------
# data -> GP input/output
# --------------------------------------------------------------------------------------
def speed_model_io(states, inputs, angle, not_angle, vel):
    """Model_learning.py:465-469, :662-683 -- GP inputs z[:-1] and per-GP targets x[1:,v]-x[:-1,v]."""
    x = torch.as_tensor(states, dtype=DT)
    u = torch.as_tensor(inputs, dtype=DT)
    z = gp_features(x, u, angle, not_angle)[:-1, :]
    ys = [(x[1:, i] - x[:-1, i]).reshape(-1, 1) for i in vel]
    return z, ys


# --------------------------------------------------------------------------------------
# delta-state dynamics model (Model_learning / Model_learning_RBF_angle_state)
# --------------------------------------------------------------------------------------
@dataclass
class DeltaModel:
    """Model_learning_RBF_angle_state, Model_learning.py:528-580: GP i predicts x_{t+1}[i] - x_t[i] for EVERY state component
    from z = [x_notangle, sin(x_angle), cos(x_angle), u] (angle = () gives the plain Model_learning_RBF, z = [x, u])."""

    hyp: List[GPHyper]
    cache: List[GPCache]
    angle: Sequence[int] = ()
    not_angle: Sequence[int] = ()


def delta_model_io(states, inputs, angle, not_angle):
    """Model_learning.py:450-469 with :564-579 -- GP inputs z[:-1] and per-state targets x[1:,i]-x[:-1,i]."""
    x = torch.as_tensor(states, dtype=DT)
    u = torch.as_tensor(inputs, dtype=DT)
    z = (gp_features(x, u, angle, not_angle) if len(angle) else torch.cat([x, u], 1))[:-1, :]
    return z, [(x[1:, i] - x[:-1, i]).reshape(-1, 1) for i in range(x.shape[1])]


def delta_next_state(m: DeltaModel, x, u, eps: Optional[torch.Tensor], particle_pred: bool = True):
    """Model_learning.py:210-229 and :471-493 -- x' = x + delta, delta = mu + sqrt(var) eps (Normal.rsample) or mu."""
    z = gp_features(x, u, m.angle, m.not_angle) if len(m.angle) else torch.cat([x, u], 1)
    mus, vrs = [], []
    for h, c in zip(m.hyp, m.cache):
        mu, var = gp_estimate_from_alpha(h, c.X, z, c.alpha, c.Kinv)
        mus.append(mu)
        vrs.append(var.reshape(-1, 1))
    dmu, dvar = torch.cat(mus, 1), torch.cat(vrs, 1)
    if particle_pred:
        if eps is None:
            eps = torch.empty(dmu.shape, dtype=DT).normal_()
        return x + dmu + torch.sqrt(dvar) * eps, dmu, dvar
    return x + dmu, dmu, dvar


# --------------------------------------------------------------------------------------
# simple costs
# --------------------------------------------------------------------------------------
def distance_cost(states, target_state, lengthscales, active_dims):
    """Cost_function.py:53-63 (distance_from_target) -- sum_i ((x_i - x*_i)/l_i)^2 in the reference's expanded form
    ||x/l||^2 + ||x*/l||^2 - 2 (x/l)(x*/l)^T, [T,M]."""
    a = states[:, :, list(active_dims)] / lengthscales
    b = (target_state / lengthscales).reshape(1, -1)
    d = (a * a).sum(2, keepdim=True) + (b * b).sum(1, keepdim=True).t() - 2.0 * torch.matmul(a, b.t())
    return d[:, :, 0]


def saturated_distance_cost(states, target_state, lengthscales, active_dims):
    """Cost_function.py:80-101 (saturated_distance_from_target) -- 1 - exp(-distance)."""
    return 1 - torch.exp(-distance_cost(states, target_state, lengthscales, active_dims))


# --------------------------------------------------------------------------------------
# mean-only rollout of a recorded input sequence
# --------------------------------------------------------------------------------------
def mean_rollout(m: SpeedModel, x_rec, u_rec, T_rollout: Optional[int] = None):
    """MC_PILCO.py:347-373 -- from the first recorded state, x_{t} = get_next_state(x_{t-1}, u_rec[t-1], particle_pred=False)."""
    x_rec = torch.as_tensor(x_rec, dtype=DT)
    u_rec = torch.as_tensor(u_rec, dtype=DT)
    n = x_rec.shape[0] if T_rollout is None else T_rollout
    traj = torch.zeros(n, x_rec.shape[1], dtype=DT)
    traj[0:1] = x_rec[0:1]
    for t in range(1, n):
        traj[t : t + 1], _, _ = next_state(m, traj[t - 1 : t], u_rec[t - 1 : t], None, particle_pred=False)
    return traj


# --------------------------------------------------------------------------------------
# the optimizer loop
# --------------------------------------------------------------------------------------
def policy_reinit(pp: PolicyPar, lenghtscales_par, centers_par, weight_par):
    """Sum_of_gaussians.reinit, policy_learning/Policy.py:229-240: lengthscales reset, centres uniform in +-centers_par, weights
    uniform in +-weight_par/2 -- two ``torch.rand`` draws, in this order."""
    B, P = pp.centers.shape
    U = pp.weight.shape[0]
    pp.log_ls.data = torch.log(torch.as_tensor(lenghtscales_par, dtype=DT)).reshape(1, -1)
    pp.centers.data = torch.as_tensor(centers_par, dtype=DT) * 2 * (torch.rand(B, P, dtype=DT) - 0.5)
    pp.weight.data = weight_par * (torch.rand(U, B, dtype=DT) - 0.5)


def reinforce_policy(
    m: SpeedModel,
    pp: PolicyPar,
    x0_mean,
    x0_var,
    M: int,
    T: int,
    cost_fn,
    num_opt_steps: int,
    lr: float,
    p_dropout: float = 0.0,
    alpha_diff_cost: float = 0.99,
    lr_reduction_ratio: float = 0.5,
    lr_min: float = 0.001,
    p_drop_reduction: float = 0.0,
    min_diff_cost: float = 0.1,
    num_min_diff_cost: int = 200,
    min_step: float = float("inf"),
    make_optimizer=None,
    nan_calls=(),
    policy_reinit_dict=None,
):
    """MC_PILCO.reinforce_policy, policy_learning/MC_PILCO.py:375-613, for a Gaussian initial distribution: warm-up rollout that
    seeds the monitor (:430-456, under no_grad; a NaN cost there re-initialises the policy, up to 10 times), then per step: up to
    ten rollouts until the cost is not NaN (:479-501; x0, mask_0, then eps_t, mask_t from the torch CPU generator, :615-674) ->
    expected cost -> exponential statistics of the cost difference (:508-519) -> backward -> optimizer step (:522-525, taken on the
    NaN cost too when all ten attempts failed) -> learning-rate halving / dropout reduction / early exit (:543-566; the window
    slice keeps Python's negative-index behaviour) -> after ten NaN attempts: policy re-initialised, step counters, cost lists,
    the first-order statistics, optimizer, learning rate's ``min_diff`` / ``min_step`` and dropout reset (:573-607; like the
    reference NOT the second-moment statistic nor the previous cost, which stay NaN -- monitors only).
    ``nan_calls``: indices of expected-cost evaluations (counted from the warm-up one) whose cost is forced to NaN -- how the
    fixtures drive these branches (tests/golden/make_golden_r3.py).  The policy parameters in ``pp`` are updated in place.
    Returns (cost_list, std_list, info) with info = dict(lr_reductions=[steps], exit_step, last_states, last_inputs, n_retry,
    n_reinit, n_init_reinit, after_reinit)."""
    prm = [pp.log_ls, pp.centers, pp.weight]
    for q in prm:
        q.requires_grad_(True)
        q.grad = None
    make_optimizer = make_optimizer or (lambda p, lr: torch.optim.Adam(p, lr))
    nan_calls = set(int(i) for i in nan_calls)
    calls = [0]
    info = {"lr_reductions": [], "exit_step": None, "n_retry": 0, "n_reinit": 0, "n_init_reinit": 0, "after_reinit": None}

    def rollout(p_drop):
        x0 = sample_x0(x0_mean, x0_var, M)
        return apply_policy(m, pp, x0, T, p_drop)

    def ecost(st):
        c, s = expected_cost(cost_fn(st))
        k = calls[0]
        calls[0] += 1
        return (c * float("nan") if k in nan_calls else c), s

    def reinit():
        policy_reinit(pp, **policy_reinit_dict)
        info["after_reinit"] = [q.detach().clone() for q in prm]

    lr0 = lr
    p_applied = p_dropout
    with torch.no_grad():
        attempts, flg_nan = 0, True
        while attempts < 10 and flg_nan:
            st, _ = rollout(p_applied)
            cost_tm1, _ = ecost(st)
            if torch.isnan(cost_tm1):
                attempts += 1
                info["n_init_reinit"] += 1
                reinit()
            else:
                flg_nan = False
    cost_list = torch.zeros(num_opt_steps, dtype=DT)
    std_list = torch.zeros(num_opt_steps, dtype=DT)
    es1 = torch.zeros(num_opt_steps + 1, dtype=DT)
    es2 = 0.0
    ratio = torch.zeros(num_opt_steps + 1, dtype=DT)
    cur_min_diff, cur_min_step = min_diff_cost, min_step
    opt = make_optimizer(prm, lr)
    step = done = 0
    st = inp = None
    while step < num_opt_steps:
        opt.zero_grad()
        attempts, flg_nan = 0, True
        while attempts < 10 and flg_nan:
            st, inp = rollout(p_applied)
            cost, std = ecost(st)
            if torch.isnan(cost):
                attempts += 1
                info["n_retry"] += 1
            else:
                flg_nan = False
        cost_list[step] = cost.detach()
        std_list[step] = std.detach()
        with torch.no_grad():
            es1[step + 1] = alpha_diff_cost * es1[step] + (1 - alpha_diff_cost) * (cost - cost_tm1)
            es2 = alpha_diff_cost * (es2 + (1 - alpha_diff_cost) * ((cost - cost_tm1 - es1[step]) ** 2))
            cost_tm1 = cost_list[step]
            ratio[step + 1] = alpha_diff_cost * ratio[step] + (1 - alpha_diff_cost) * (es1[step + 1] / es2.sqrt())
        cost.backward()
        opt.step()
        if step > cur_min_step:
            if int(torch.sum(torch.abs(ratio[step + 1 - num_min_diff_cost : step + 1]) < cur_min_diff)) >= num_min_diff_cost:
                if lr > lr_min:
                    lr = max(lr * lr_reduction_ratio, lr_min)
                    cur_min_diff = max(cur_min_diff / 2, 0.01)
                    cur_min_step = step + num_min_diff_cost
                    opt = make_optimizer(prm, lr)
                    p_applied = max(p_applied - p_drop_reduction, 0.0)
                    info["lr_reductions"].append(step)
                else:
                    info["exit_step"] = step
                    step = num_opt_steps
        step += 1
        done += 1
        if flg_nan:  # ten NaN rollouts in a row (:573-607)
            info["n_reinit"] += 1
            reinit()
            step = done = 0
            cur_min_step, cur_min_diff = min_step, min_diff_cost
            cost_list = torch.zeros(num_opt_steps, dtype=DT)
            std_list = torch.zeros(num_opt_steps, dtype=DT)
            es1 = torch.zeros(num_opt_steps + 1, dtype=DT)
            ratio = torch.zeros(num_opt_steps + 1, dtype=DT)
            lr = lr0
            opt = make_optimizer(prm, lr)
            p_applied = p_dropout
    for q in prm:
        q.requires_grad_(False)
        q.grad = None
    info["last_states"], info["last_inputs"] = st.detach(), inp.detach()
    info["cost_calls"] = calls[0]
    return cost_list[:done].detach(), std_list[:done].detach(), info


# --------------------------------------------------------------------------------------
# GP hyper-parameter training
# --------------------------------------------------------------------------------------
def marginal_nll(h: GPHyper, X, Y):
    """Marginal_log_likelihood, gpr_lib/Likelihood/Gaussian_likelihood.py:15-24, on GP_prior.forward's outputs
    (GP_prior.py:91-115):  1/2 ((Y - m)^T K^-1 (Y - m) + logdet K)."""
    mX, K, Kinv, logdet = gp_forward(h, X)
    r = Y - mX
    return 0.5 * (r.t() @ Kinv @ r + logdet).reshape(())


def fit_model(h: GPHyper, X, Y, n_epoch: int, lr: float, make_optimizer=None):
    """GP_prior.fit_model (GP_prior.py:179-230) as Model_learning.train_gp_likelihood drives it (Model_learning.py:398-421): ONE
    full batch per epoch, Adam on the trainable log-parameters.  ``h``'s tensors are updated in place.  Returns (losses [n_epoch],
    trajectory: list over epochs 0..n_epoch of the parameter tensors in the order [log_sigma_n, log_ls, log_lambda, *poly])."""
    prm = [h.log_sigma_n, h.log_ls, h.log_lambda] + list(h.poly_log_par or [])
    for q in prm:
        q.requires_grad_(True)
        q.grad = None
    opt = (make_optimizer or (lambda p: torch.optim.Adam(p, lr=lr)))(prm)
    losses, traj = [], [[q.detach().clone() for q in prm]]
    for _ in range(n_epoch):
        opt.zero_grad()
        loss = marginal_nll(h, X, Y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        traj.append([q.detach().clone() for q in prm])
    for q in prm:
        q.requires_grad_(False)
        q.grad = None
    return torch.tensor(losses, dtype=DT), traj
