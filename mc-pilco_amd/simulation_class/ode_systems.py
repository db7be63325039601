"""ODE right-hand sides of the simulated systems (data source only; host side).

``cartpole(y, t, u)``: cart-pole with the force on the cart, state [p, p_dot, theta, theta_dot], pole down at
theta = 0 -- same equations and constants as the reference's simulator (simulation_class/ode_systems.py:34-68);
``u`` may be a scalar or any array with one element."""
import numpy as np

from mc_pilco_amd import synthetic


def cartpole(y, t, u):
    return list(synthetic.cartpole_ode(np.asarray(y, dtype=float), float(np.ravel(u)[0])))
