"""Pins the CPU oracle against the round-6 vectors the reference itself produced (tests/golden/make_golden_r6.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import T, hyper
from oracle import mcpilco_oracle as orc


def _hyper(fx):
    pw = [fx[k] for k in ("poly_w1", "poly_w2") if k in fx]
    return hyper(fx["lengthscales"], float(fx["sigma_n"]), 1.0, pw or None)


def test_sod_index_list_on_the_ur5_pretrain_shape(golden):
    """get_SOD (GP_prior.py:232-257) at N = 600, D = 24, SE + polynomial(1), absolute threshold: 406 of 600 kept, list exact."""
    fx = golden("sod_ur5_n600")
    h = _hyper(fx)
    nt = torch.get_num_threads()
    torch.set_num_threads(min(4, nt))
    try:
        idx = orc.gp_get_sod(h, T(fx["X"]), T(fx["Y"]), float(fx["thr"]))
    finally:
        torch.set_num_threads(nt)
    assert idx == [int(i) for i in fx["idx"]]
    assert 0.10 * 600 < 600 - len(idx) < 0.45 * 600
    assert float(fx["min_margin"]) > 1e-7  # (no decision of the reference's run was within rounding of flipping)


@pytest.mark.parametrize("name", ["fwd_autograd_se_n300", "fwd_autograd_se_poly2_n300"])
def test_forward_autograd_at_n300(golden, name):
    """orc.gp_forward (GP_prior.py:91-115) + autograd under the fixture's criterion against the reference's: loss rel 1e-10, gradients
    1e-8 max(1, |g|_max)."""
    fx = golden(name)
    h = _hyper(fx)
    prm = [h.log_sigma_n, h.log_ls, h.log_lambda] + list(h.poly_log_par or [])
    for q in prm:
        q.requires_grad_(True)
    X, Y = T(fx["X"]), T(fx["Y"])
    mX, K, Kinv, logdet = orc.gp_forward(h, X)
    r = Y - mX
    loss = (0.5 * (r.t() @ Kinv @ r) + 0.3 * logdet + 1e-3 * torch.trace(K) + 0.05 * (Kinv * Kinv).sum() / Y.shape[0]).reshape(())
    loss.backward()
    ref = float(np.asarray(fx["loss"]).reshape(-1)[0])
    assert abs(float(loss.detach()) - ref) < 1e-10 * abs(ref)
    pre = "gp_list.0." if h.poly_log_par is not None else ""
    want = [fx["grad__%ssigma_n_log" % pre], fx["grad__%slog_lengthscales_par" % pre], fx["grad__%slog_lambda_par" % pre]]
    want += [fx["grad__gp_list.1.gp_list.%d.Sigma_pos_par" % d] for d in range(len(prm) - 3)]
    for q, g in zip(prm, want):
        assert float(np.abs(q.grad.numpy().reshape(-1) - g.reshape(-1)).max()) < 1e-8 * max(1.0, float(np.abs(g).max()))
