"""Pins the CPU oracle at the sizes that ship (N = 300 / N = 400) against vectors the reference itself produced
(tests/golden/make_golden_r5.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import T, hyper
from oracle import mcpilco_oracle as orc


def relerr(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def _hyper(fx):
    pw = [fx[k] for k in ("poly_w1", "poly_w2") if k in fx]
    return hyper(fx["lengthscales"], float(fx["sigma_n"]), 1.0, pw or None)


def test_forward_alpha_posterior_at_n300(golden):
    """GP_prior.py:91-155 at N = 300 (condition number of K stored in the fixture, ~1e6): the factorisation-derived quantities to
    1e-9 (two LAPACK routes to the same inverse), the posterior from the reference's own cached operands to 1e-12."""
    fx = golden("kern_se_n300")
    h = _hyper(fx)
    X, Y, Xs = T(fx["X"]), T(fx["Y"]), T(fx["Xs"])
    mX, K, Kinv, logdet = orc.gp_forward(h, X)
    assert relerr(Kinv, fx["Kinv"]) < 1e-9
    assert abs(float(logdet) - float(fx["logdet"])) < 1e-11 * abs(float(fx["logdet"]))
    alpha, _, _ = orc.gp_alpha(h, X, Y)
    assert relerr(alpha, fx["alpha"]) < 1e-9
    mu, var = orc.gp_estimate_from_alpha(h, X, Xs, T(fx["alpha"]), T(fx["Kinv"]))
    assert relerr(mu, fx["mu"]) < 1e-12
    assert np.max(np.abs(var.numpy() - fx["var"])) < 1e-12


@pytest.mark.parametrize("name", ["nll_se_n300", "nll_se_poly2_n300", "nll_se_poly1_d24_n400"])
def test_marginal_likelihood_and_gradient_at_real_sizes(golden, name):
    """orc.marginal_nll + autograd against the reference's Marginal_log_likelihood + autograd at N = 300 / 400."""
    fx = golden(name)
    h = _hyper(fx)
    prm = [h.log_sigma_n, h.log_ls, h.log_lambda] + list(h.poly_log_par or [])
    for q in prm:
        q.requires_grad_(True)
    loss = orc.marginal_nll(h, T(fx["X"]), T(fx["Y"]))
    loss.backward()
    ref = float(np.asarray(fx["loss"]).reshape(-1)[0])
    assert abs(float(loss.detach()) - ref) < 1e-10 * abs(ref)
    pre = "gp_list.0." if h.poly_log_par is not None else ""
    ref = {"log_sigma_n": fx["grad__%ssigma_n_log" % pre], "log_ls": fx["grad__%slog_lengthscales_par" % pre],
           "log_lambda": fx["grad__%slog_lambda_par" % pre]}
    for k, q in zip(["log_sigma_n", "log_ls", "log_lambda"], prm[:3]):
        assert float(np.abs(q.grad.numpy().reshape(-1) - ref[k].reshape(-1)).max()) < 1e-8 * max(1.0, float(np.abs(ref[k]).max())), k
    for d, q in enumerate(prm[3:]):
        r = fx["grad__gp_list.1.gp_list.%d.Sigma_pos_par" % d]
        assert float(np.abs(q.grad.numpy().reshape(-1) - r.reshape(-1)).max()) < 1e-8 * max(1.0, float(np.abs(r).max())), d


@pytest.mark.parametrize("name", ["sod_n300", "sod_ur5_n400"])
def test_sod_index_lists_at_real_sizes(golden, name):
    """get_SOD (GP_prior.py:232-257) at N = 300 (relative threshold, 264 of 300 kept) and on the UR5 shape at N = 400 (absolute
    threshold, SE + polynomial(1), 304 of 400 kept): index lists exact."""
    fx = golden(name)
    h = _hyper(fx)
    X, Y = T(fx["X"]), T(fx["Y"])
    if "thr_rel_factor" in fx:
        assert abs(float(fx["thr_rel_factor"]) * float(torch.sqrt(h.sigma_n_2())) - float(fx["thr"])) < 1e-15
    idx = orc.gp_get_sod(h, X, Y, float(fx["thr"]))
    assert idx == [int(i) for i in fx["idx"]]
    assert 0 < len(idx) < X.shape[0]
    assert float(fx["min_margin"]) > 1e-7  # (no decision of the reference's run was within rounding of flipping)
