"""ODE-simulated "real system" (data source, 61 samples per trial; host side) -- the role of the reference's
``simulation_class/model.py:32-68``: integrate ``f_sim`` over each sampling interval with the policy's input
held constant, return noisy and noiseless state samples."""
import numpy as np
from scipy.integrate import odeint


class Model:
    def __init__(self, fcn):
        self.fcn = fcn

    def rollout(self, s0, policy, T, dt, noise):
        n = int(T / dt) + 1
        state_dim = len(s0)
        times = np.linspace(0, (n - 1) * dt, n)
        clean = np.zeros([n, state_dim])
        noisy = np.zeros([n, state_dim])
        clean[0] = np.asarray(s0, dtype=float)
        noisy[0] = clean[0] + np.asarray(noise) * np.random.randn(state_dim)
        u0 = np.asarray(policy(noisy[0], 0.0)).reshape(-1)
        inputs = np.zeros([n, u0.size])
        inputs[0] = u0
        for k in range(1, n):
            seg = odeint(self.fcn, clean[k - 1], [times[k - 1], times[k]], args=(inputs[k - 1],))
            clean[k] = seg[-1]
            noisy[k] = clean[k] + np.asarray(noise) * np.random.randn(state_dim)
            inputs[k] = np.asarray(policy(noisy[k], times[k])).reshape(-1)
        return noisy, inputs, clean
