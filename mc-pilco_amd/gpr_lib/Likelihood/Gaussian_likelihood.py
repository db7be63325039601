"""Marginal likelihood criterion -- drop-in for ``gpr_lib/Likelihood/Gaussian_likelihood.py:12-24``:
loss = 1/2 ((Y-m)^T K^-1 (Y-m) + log det K)   (the 2 pi term is dropped, as in the reference).

``forward`` is the reference's expression on the outputs of ``GP_prior.forward`` (an autograd graph when hyper-parameters are trainable:
``GP_prior._ForwardFunction``).  ``loss_and_grad`` is the route ``fit_model`` takes for this criterion: the loss and the analytic gradient
dL/dtheta = 1/2 tr((K^-1 - a a^T) dK/dtheta)  from the HIP kernels, written into ``param.grad`` without building a graph.
"""
import torch


class Marginal_log_likelihood(torch.nn.modules.loss._Loss):
    def forward(self, output_GP_prior, Y):
        m_X, _, K_X_inv, log_det = output_GP_prior
        r = Y - m_X
        return 0.5 * (torch.sum(r * (K_X_inv @ r)) + log_det)

    def loss_and_grad(self, gp, X, Y):
        from mc_pilco_amd import nll

        return nll.nll_loss_and_grad(gp, X, Y)
