"""Deterministic synthetic workloads shaped like the reference's experiments.

No dataset or checkpoint is available offline, so tests and ``bench.py`` use:

* cart-pole data from this module's own RK4 integration of the cart-pole equations
  (same equations and constants as the reference's simulator,
  ``simulation_class/ode_systems.py:34-68``: m1=m2=0.5, l=0.5, b=0.1, g=9.81), driven by
  uniform random inputs -- 5 rollouts x 61 samples -> N=300 GP rows, D=6, G=2
  (``test_mcpilco_cartpole_rbf_ker.py:50-62``);
* a UR5-shaped (12-D state, 6 inputs, D=24, G=6) smooth joint trajectory
  (``test_mcpilco_ur5_mujoco.py:57-69``);
* fixed "trained-like" hyper-parameters and the launch scripts' policy initialisation.

Everything is numpy float64; nothing here touches the GPU.
"""
import math

import numpy as np


def cartpole_ode(y, u, m1=0.5, m2=0.5, l=0.5, b=0.1, g=9.81):
    """d/dt [p, p_dot, theta, theta_dot] for force u on the cart (pole down = 0)."""
    _, dp, th, dth = y
    s, c = math.sin(th), math.cos(th)
    den = 4 * (m1 + m2) - 3 * m2 * c * c
    ddp = (2 * m2 * l * dth * dth * s + 3 * m2 * g * s * c + 4 * u - 4 * b * dp) / den
    ddth = (-3 * m2 * l * dth * dth * s * c - 6 * (m1 + m2) * g * s - 6 * (u - b * dp) * c) / (l * den)
    return np.array([dp, ddp, dth, ddth])


def cartpole_rollouts(n_roll=5, n_step=61, Ts=0.05, seed=1, u_max=10.0, noise=0.01):
    """List of (states [n_step,4], inputs [n_step,1]) with measurement noise."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n_roll):
        y = rng.randn(4) * 1e-2
        ys, us = [], []
        for _k in range(n_step):
            u = float(u_max * (2 * rng.rand() - 1))
            ys.append(y.copy())
            us.append([u])
            h = Ts / 4
            for _s in range(4):
                k1 = cartpole_ode(y, u)
                k2 = cartpole_ode(y + h / 2 * k1, u)
                k3 = cartpole_ode(y + h / 2 * k2, u)
                k4 = cartpole_ode(y + h * k3, u)
                y = y + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        out.append((np.array(ys) + noise * rng.randn(n_step, 4), np.array(us)))
    return out


def ur5_rollouts(n_roll=2, n_step=201, Ts=0.02, seed=1, noise=1e-3):
    """UR5-shaped smooth trajectories: states [q(6), q_dot(6)], inputs [6] (a PD-like torque).
    Not a robot simulation -- only the shapes / smoothness of the reference's UR5 data."""
    rng = np.random.RandomState(seed)
    out = []
    t = np.arange(n_step) * Ts
    for _ in range(n_roll):
        amp = 0.4 + 0.4 * rng.rand(6)
        om = 0.5 + 1.5 * rng.rand(6)
        ph = 2 * np.pi * rng.rand(6)
        q = amp * np.sin(np.outer(t, om) + ph)
        qd = amp * om * np.cos(np.outer(t, om) + ph)
        qdd = -amp * om * om * np.sin(np.outer(t, om) + ph)
        u = np.tanh(0.3 * qdd + 0.2 * qd + 0.5 * np.sin(q))
        x = np.concatenate([q, qd], 1) + noise * rng.randn(n_step, 12)
        out.append((x, u))
    return out


def ur5_target_traj(T=300, Ts=0.02):
    """Smooth 12-D target (q*, q_dot*) like ``envs/target_q_trajectory.csv`` (which has 200 rows)."""
    t = np.arange(T) * Ts
    om = np.array([0.6, 0.8, 1.0, 1.2, 0.7, 0.9])
    q = 0.5 * np.sin(np.outer(t, om))
    qd = 0.5 * om * np.cos(np.outer(t, om))
    return np.concatenate([q, qd], 1)


CARTPOLE = dict(
    S=4, U=1, G=2, D=6, Ts=0.05,
    angle=[2], not_angle=[0, 1, 3], vel=[1, 3], not_vel=[0, 2],
    lengthscales=np.array([2.0, 3.0, 6.0, 1.0, 1.0, 12.0]), lam=1.0, sigma_n=0.03,
    u_max=10.0, B=200, P=5,
    cost_target=[math.pi, 0.0], cost_ls=[3.0, 1.0], cost_angle_index=2, cost_pos_index=0,
    x0_mean=np.zeros(4), x0_var=1e-4 * np.ones(4),
)

UR5 = dict(
    S=12, U=6, G=6, D=24, Ts=0.02,
    angle=list(range(6)), not_angle=list(range(6, 12)), vel=list(range(6, 12)), not_vel=list(range(6)),
    lengthscales=np.concatenate([4.0 * np.ones(6), 3.0 * np.ones(6), 3.0 * np.ones(6), 5.0 * np.ones(6)]),
    lam=1.0, sigma_n=0.01, u_max=[1.0] * 6, B=400, P=24,
    cost_ls=[0.5] * 6 + [1.0] * 6,
    x0_mean=np.zeros(12), x0_var=1e-6 * np.ones(12),
)


def cartpole_policy_init(B=200, u_max=10.0, seed=1):
    """Launch-script initialisation, ``test_mcpilco_cartpole_rbf_ker.py:119-125``:
    centers uniform(-pi,pi) with consistent cos/sin columns, l=1, W ~ u_max*(U(0,1)-0.5)."""
    rng = np.random.RandomState(seed)
    ang = np.pi * 2 * (rng.rand(B, 1) - 0.5)
    na = np.pi * 2 * (rng.rand(B, 3) - 0.5)
    centers = np.concatenate([na, np.cos(ang), np.sin(ang)], 1)
    return dict(lengthscales=np.ones(5), centers=centers, weight=u_max * (rng.rand(1, B) - 0.5))


def ur5_policy_init(B=400, seed=1):
    """``test_mcpilco_ur5_mujoco.py:137-150``."""
    rng = np.random.RandomState(seed)
    centers = np.concatenate([np.pi / 2 * 2 * (rng.rand(B, 12) - 0.5), 0.1 * 2 * (rng.rand(B, 12) - 0.5)], 1)
    return dict(lengthscales=np.pi * np.ones(24), centers=centers, weight=2.0 * (rng.rand(6, B) - 0.5))


def gp_io(rollouts, angle, not_angle, vel):
    """Stacks rollouts into GP inputs z=[x_notangle, sin, cos, u] (rows 0..n-2 of each rollout)
    and per-GP targets x[t+1,v]-x[t,v]  (the speed-model data layout)."""
    Z, Y = [], [[] for _ in vel]
    for x, u in rollouts:
        z = np.concatenate([x[:, not_angle], np.sin(x[:, angle]), np.cos(x[:, angle]), u], 1)[:-1]
        Z.append(z)
        for g, v in enumerate(vel):
            Y[g].append((x[1:, v] - x[:-1, v]).reshape(-1, 1))
    return np.concatenate(Z, 0), [np.concatenate(y, 0) for y in Y]
