"""Marginal-likelihood loss and analytic gradient for GP hyper-parameter training on the device.

Replaces what autograd does in the reference's ``GP_prior.fit_model`` (gpr_lib/GP_prior/GP_prior.py:179-230) through
``forward`` (Gram, Cholesky, inverse) and ``Marginal_log_likelihood`` (gpr_lib/Likelihood/Gaussian_likelihood.py:15-24):
    L = 1/2 ((Y-m)^T K^-1 (Y-m) + log det K),      dL/dtheta = 1/2 tr((K^-1 - a a^T) dK/dtheta),  a = K^-1 (Y-m).
The HIP kernel ``mcp_nll_grad`` returns the gradient w.r.t. the kernel's log-parameters; this module scatters it into
``param.grad`` of the GP object's own parameters (RBF: log_lengthscales_par, log_lambda_par, sigma_n_log, mean_par;
MPK_k: Sigma_pos_par), honouring ``requires_grad``.
"""
import ctypes as C

import torch

from . import hipabi as abi
from . import ops

DT = torch.float64


def _leaves(gp):
    from .gpr_lib.GP_prior.GP_prior import Combine_GP

    return list(gp._leaves()) if isinstance(gp, Combine_GP) else [gp]


def nll_loss_and_grad(gp, X, Y):
    """Returns the loss (0-dim tensor) and sets ``.grad`` of every trainable parameter of ``gp``."""
    from .gpr_lib.GP_prior import Sparse_GP, Stationary_GP

    dev = gp.device
    spec = gp.kernel_spec_dev()  # hyper-parameters stay on the device: the epoch has no device->host round trip
    Xc = gp._cols(X)
    N, D = Xc.shape
    K = ops.cov_build(spec, Xc, None, noise=gp.GP_with_noise)
    U, logdet, status = ops.chol_factor(K)
    # the not-positive-definite flag is checked by the caller at its next host sync point (``check_status``): reading it
    # here would stall the launch queue in the middle of every epoch
    prev = getattr(gp, "_nll_status", None)
    gp._nll_status = status if prev is None else torch.bitwise_or(prev, status)
    _, Kinv = ops.chol_inverse(U)
    r = (Y.to(dev) - gp.get_mean(X)).reshape(-1).contiguous()
    alpha = ops.gp_alpha(Kinv, r, 0.0).reshape(-1).contiguous()
    loss = 0.5 * (torch.dot(r, alpha) + logdet)
    nbytes = abi.lib().mcp_nll_workspace_bytes(N, D)
    ws = torch.empty((nbytes + 7) // 8, dtype=DT, device=dev)
    g = torch.empty(4 * D + 3, dtype=DT, device=dev)
    kc = spec.to_c(dev)
    abi.check(abi.lib().mcp_nll_grad(C.byref(kc), N, abi.ptr(Xc), abi.ptr(Kinv), N, abi.ptr(alpha), abi.ptr(g), abi.ptr(ws), nbytes,
                                     abi.stream()), "mcp_nll_grad")

    def put(p, val):
        if p.requires_grad:
            p.grad = val.reshape(p.shape).to(p.dtype).clone()

    first = True
    for leaf in _leaves(gp):
        if leaf.GP_with_noise:
            put(leaf.sigma_n_log, g[D + 1] * 2.0 * torch.exp(2.0 * leaf.sigma_n_log.detach()))
        if isinstance(leaf, Stationary_GP.RBF):
            put(leaf.log_lengthscales_par, g[0:D] if leaf.flg_ARD else g[0:D].sum())
            put(leaf.log_lambda_par, g[D])
            if first:  # only the first child's mean enters the model (GP_prior.py:306-312)
                put(leaf.mean_par, -alpha.sum())
            elif leaf.mean_par.requires_grad:
                leaf.mean_par.grad = torch.zeros_like(leaf.mean_par)
        elif isinstance(leaf, Sparse_GP.MPK_GP):
            if leaf.poly_deg == 1:
                put(leaf.Sigma_pos_par, g[D + 2:2 * D + 3] if leaf.flg_offset else g[D + 2:2 * D + 2])
            else:
                put(leaf.Sigma_pos_par, torch.cat([g[2 * D + 3:3 * D + 3], g[3 * D + 3:4 * D + 3]]))
        first = False
    return loss.detach()


def check_status(gp):
    """Raises if any Cholesky since the last check met a matrix that is not positive definite (torch.cholesky's error in the
    reference, GP_prior.py:106)."""
    st = getattr(gp, "_nll_status", None)
    gp._nll_status = None
    if st is not None and ops.status_flags(st)["not_spd"]:
        raise RuntimeError("cholesky: the covariance matrix is not positive-definite")
