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


class PMS_Model:
    """Partially measurable system -- the role of ``simulation_class/model.py:71-141``: only positions are measured (with
    noise); velocities are estimated online by backward differences of the measured positions and smoothed by a first-order
    Butterworth low-pass filter (cut-off ``filtering_dict["fc"]``), and the policy acts on that measured state."""

    def __init__(self, fcn, filtering_dict):
        self.fcn = fcn
        self.filtering_dict = filtering_dict

    def rollout(self, s0, policy, T, dt, noise, vel_indeces, pos_indeces):
        from scipy import signal

        n = int(T / dt) + 1
        state_dim = len(s0)
        times = np.linspace(0, T, n)
        b, a = signal.butter(1, self.filtering_dict["fc"])
        clean = np.zeros([n, state_dim])
        noisy = np.zeros([n, state_dim])
        meas = np.zeros([n, state_dim])
        clean[0] = noisy[0] = meas[0] = np.asarray(s0, dtype=float)
        u0 = np.asarray(policy(meas[0], 0.0)).reshape(-1)
        inputs = np.zeros([n, u0.size])
        for k in range(n - 1):
            inputs[k] = np.asarray(policy(meas[k], times[k])).reshape(-1)
            clean[k + 1] = odeint(self.fcn, clean[k], [times[k], times[k] + dt], args=(inputs[k],))[-1]
            noisy[k + 1] = clean[k + 1] + np.random.randn(state_dim) * noise
            meas[k + 1, pos_indeces] = noisy[k + 1, pos_indeces]
            noisy[k + 1, vel_indeces] = (meas[k + 1, pos_indeces] - meas[k, pos_indeces]) / dt
            meas[k + 1, vel_indeces] = (b[0] * noisy[k + 1, vel_indeces] + b[1] * noisy[k, vel_indeces] - a[1] * meas[k, vel_indeces]) / a[0]
        inputs[-1] = np.asarray(policy(meas[-1], T)).reshape(-1)
        return meas, inputs, clean, noisy
