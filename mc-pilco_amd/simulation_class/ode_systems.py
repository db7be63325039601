"""ODE right-hand sides of the simulated systems (data source only; host side).

``cartpole(y, t, u)``: cart-pole with the force on the cart, state [p, p_dot, theta, theta_dot], pole down at
theta = 0 -- same equations and constants as the reference's simulator (simulation_class/ode_systems.py:34-68);
``u`` may be a scalar or any array with one element.
``pend(y, t, u)``: torque-driven pendulum, state [theta, theta_dot], pole down at theta = 0 (ode_systems.py:16-31:
m = l = 1, b = 0.1, g = 9.81, I = m l^2 / 3)."""
import numpy as np

from mc_pilco_amd import synthetic


def cartpole(y, t, u):
    return list(synthetic.cartpole_ode(np.asarray(y, dtype=float), float(np.ravel(u)[0])))


def pend(y, t, u):
    theta, theta_dot = float(y[0]), float(y[1])
    m, l, b, g = 1.0, 1.0, 0.1, 9.81
    inertia = m * l ** 2 / 3.0
    return [theta_dot, (float(np.ravel(u)[0]) - b * theta_dot - 0.5 * m * l * g * np.sin(theta)) / inertia]
