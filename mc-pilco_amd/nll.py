"""Marginal-likelihood loss and analytic gradient for GP hyper-parameter training
(GP_prior.fit_model + Marginal_log_likelihood, gpr_lib/GP_prior/GP_prior.py:179-230,
gpr_lib/Likelihood/Gaussian_likelihood.py:12-24).  SURVEY 8f rank 1 -- not built yet."""


def nll_loss_and_grad(gp, X, Y):
    raise NotImplementedError("GP hyper-parameter training on the HIP path is not implemented yet (SURVEY.md 8f, rank 1); "
                              "load trained hyper-parameters into the GP objects (state_dict) and call pretrain_gp")
