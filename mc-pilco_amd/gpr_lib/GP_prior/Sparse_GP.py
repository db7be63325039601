"""Linear / multiplicative-polynomial kernels on the HIP path -- drop-in for the parts of
``gpr_lib/GP_prior/Sparse_GP.py`` the MC-PILCO launch scripts use:

  MPK_GP               Sparse_GP.py:559-668   product over degrees of phi^T diag(s_d^2) phi'
  get_Volterra_MPK_GP  Sparse_GP.py:671-737   MPK_1 (offset feature, optional noise) + MPK_2 ... + MPK_deg

Reference behaviour kept on purpose: the factor weights of MPK_k are s_d = (k-d)*exp(par[d*n:(d+1)*n])
(``get_Sigma`` re-adds the same slice k-d times, :613-623); phi = [x, 1] only when ``flg_offset``; the
MPK noise branch is dead in the reference (:644), so noise comes from the sum's other children.
The HIP kernels implement Volterra degree <= 2.  Out of scope: Poly_GP, get_SOR_GP (unused).
"""
import numpy as np
import torch

from mc_pilco_amd import ops

from . import GP_prior

from mc_pilco_amd.gpr_lib.Utils import Parameters_covariance_functions

__all__ = ["Linear_GP", "MPK_GP", "get_Volterra_MPK_GP", "get_pos_par_sqrt", "f_init_pos_par_sqrt", "get_pos_par_log", "f_init_pos_par_log"]


# positive-parameter transformations (Sparse_GP.py:15-32)
def get_pos_par_sqrt(par):
    return torch.sqrt(par ** 2)


def f_init_pos_par_sqrt(par):
    return par


def get_pos_par_log(par):
    return torch.exp(par)


def f_init_pos_par_log(par):
    return np.log(par)


class Linear_GP(GP_prior.GP_prior):
    """Dot-product kernel k(x, x') = phi(x)^T Sigma phi(x'), phi = x[active_dims] (+ a ones column with ``flg_offset``), Sigma =
    ``Sigma_function(f_transofrm_pos_par(Sigma_pos_par), Sigma_free_par, *Sigma_f_additional_par_list)`` -- Sparse_GP.py:295-490.
    On the HIP path Sigma must be diagonal (the kernels take one weight per feature: ``mcp_kernel.w1``); that is every use the
    reference makes of it (``diagonal_covariance`` under MPK_GP).  The regressor-space helpers are ordinary torch ops."""

    def __init__(self, active_dims, mean_init=None, flg_mean_trainable=False, flg_no_mean=False, sigma_n_init=None, flg_train_sigma_n=True,
                 Sigma_function=None, Sigma_f_additional_par_list=None, Sigma_pos_par_init=None, flg_train_Sigma_pos_par=True,
                 Sigma_free_par_init=None, flg_train_Sigma_free_par=True, flg_offset=False, f_transofrm_pos_par=get_pos_par_log,
                 f_init_pos_par=f_init_pos_par_log, name="", dtype=torch.float64, sigma_n_num=None, device=None):
        if active_dims is None:
            raise RuntimeError("Active_dims are needed")
        super().__init__(active_dims, sigma_n_init=sigma_n_init, flg_train_sigma_n=flg_train_sigma_n, name=name, dtype=dtype,
                         sigma_n_num=sigma_n_num, device=device)
        self.num_features = int(np.asarray(active_dims).size)
        self.flg_offset = flg_offset
        self.f_transofrm_pos_par = f_transofrm_pos_par
        self.f_init_pos_par = f_init_pos_par
        self.check_mean(mean_init, flg_mean_trainable, flg_no_mean)
        self.check_sigma_function(Sigma_function, Sigma_f_additional_par_list, Sigma_pos_par_init, flg_train_Sigma_pos_par, Sigma_free_par_init,
                                  flg_train_Sigma_free_par)

    def check_mean(self, mean_init, flg_mean_trainable, flg_no_mean):
        if mean_init is None:
            mean_init = np.zeros(1)
            flg_no_mean = True
        self.flg_no_mean = flg_no_mean
        self.mean_par = torch.nn.Parameter(torch.tensor(np.asarray(mean_init, dtype=float), dtype=self.dtype, device=self.device),
                                           requires_grad=flg_mean_trainable)

    def check_sigma_function(self, Sigma_function, Sigma_f_additional_par_list, Sigma_pos_par_init, flg_train_Sigma_pos_par, Sigma_free_par_init,
                             flg_train_Sigma_free_par):
        if Sigma_function is None:
            raise RuntimeError("Specify a Sigma function")
        self.Sigma_function = Sigma_function
        self.Sigma_f_additional_par_list = Sigma_f_additional_par_list
        par = lambda v, flg: None if v is None else torch.nn.Parameter(torch.tensor(np.asarray(v, dtype=float), dtype=self.dtype, device=self.device),
                                                                        requires_grad=flg)
        self.Sigma_pos_par = par(None if Sigma_pos_par_init is None else self.f_init_pos_par(Sigma_pos_par_init), flg_train_Sigma_pos_par)
        self.Sigma_free_par = par(Sigma_free_par_init, flg_train_Sigma_free_par)

    def get_phi(self, X):
        """Regression matrix of the inputs X: the active columns (+ ones)."""
        Xa = X[:, self.active_dims]
        return torch.cat([Xa, torch.ones(X.shape[0], 1, dtype=self.dtype, device=Xa.device)], 1) if self.flg_offset else Xa

    def get_Sigma(self):
        pos = None if self.Sigma_pos_par is None else self.f_transofrm_pos_par(self.Sigma_pos_par)
        return self.Sigma_function(pos, self.Sigma_free_par, *self.Sigma_f_additional_par_list)

    def get_Sigma_list(self):
        return [self.get_Sigma()]

    def get_mean(self, X):
        if self.flg_no_mean:
            return torch.zeros(X.shape[0], 1, dtype=self.dtype, device=self.device)
        return torch.matmul(self.get_phi(X.to(self.device)), self.mean_par)

    def kernel_spec(self) -> ops.KernelSpec:
        """phi^T diag(w) phi' as the kernels' degree-1 polynomial term (no SE part: lambda = 0)."""
        Sigma = self.get_Sigma().detach().to(torch.float64).cpu()
        if not self.flg_no_mean:
            raise NotImplementedError("a linear prior mean phi(X) w is not implemented by the HIP kernels")
        if float((Sigma - torch.diag(torch.diag(Sigma))).abs().max()) != 0.0:
            raise NotImplementedError("the HIP kernels implement diagonal Sigma matrices (one weight per feature)")
        w = torch.diag(Sigma)
        D = self.num_features
        w1 = w if self.flg_offset else torch.cat([w, torch.zeros(1, dtype=torch.float64)])
        return ops.KernelSpec(torch.ones(D, dtype=torch.float64), 0.0, float(self.get_sigma_n_2()) if self.GP_with_noise else 0.0, 0.0, w1, None, None)

    def kernel_spec_dev(self) -> ops.KernelSpec:
        """kernel_spec with the weights and the noise left on the device (mcp_kernel.scal); what GP_prior.forward's autograd route and the
        marginal-likelihood gradient read.  The structure checks of kernel_spec (diagonal Sigma: one host read) are made once per object."""
        if not self.flg_no_mean:
            raise NotImplementedError("a linear prior mean phi(X) w is not implemented by the HIP kernels")
        Sigma = self.get_Sigma().detach().to(torch.float64)
        if not self.__dict__.get("_sigma_is_diagonal", False):
            if float((Sigma - torch.diag(torch.diag(Sigma))).abs().max()) != 0.0:
                raise NotImplementedError("the HIP kernels implement diagonal Sigma matrices (one weight per feature)")
            self.__dict__["_sigma_is_diagonal"] = True
        D, dev = self.num_features, self.device
        z = torch.zeros(1, dtype=torch.float64, device=dev)
        w = torch.diag(Sigma)
        w1 = w if self.flg_offset else torch.cat([w, z])
        sig2 = self.get_sigma_n_2().detach().reshape(-1)[:1].to(torch.float64) if self.GP_with_noise else z
        nan = float("nan")
        return ops.KernelSpec(torch.ones(D, dtype=torch.float64, device=dev), nan, nan, nan, w1, None, None, scal=torch.cat([z, sig2, z]).contiguous())

    def get_parameters(self, X, Y, flg_print=False):
        """w_hat = Sigma phi(X)^T K^-1 (Y - m)  (valid when this is the only kernel of the model)."""
        m_X, _, K_X_inv = self.forward_for_estimate(X)
        w_hat = torch.matmul(self.get_Sigma(), torch.matmul(self.get_phi(X.to(self.device)).transpose(0, 1), torch.matmul(K_X_inv, Y.to(self.device) - m_X)))
        if flg_print:
            print(self.name + " linear parameters estimated: ", w_hat.data)
        return w_hat

    def get_parameters_inv_lemma(self, X, Y, flg_print=False):
        """The reference's expression through the matrix inversion lemma (Sparse_GP.py:470-490), term by term."""
        Y = Y.to(self.device) - self.get_mean(X)
        Phi = self.get_phi(X.to(self.device))
        s2 = self.get_sigma_n_2()
        cov = torch.inverse(torch.inverse(self.get_Sigma()) + s2 * torch.matmul(Phi.transpose(0, 1), Phi))
        w_hat = s2 * torch.matmul(torch.matmul(cov, Phi.transpose(0, 1)), Y)
        if flg_print:
            print(self.name + " linear parameters estimated: ", w_hat.data)
        return w_hat


class MPK_GP(Linear_GP):
    def __init__(self, active_dims, poly_deg, sigma_n_init=None, flg_train_sigma_n=True, Sigma_pos_par_init=None, flg_train_Sigma_pos_par=True,
                 flg_offset=True, name="", dtype=torch.float64, sigma_n_num=None, device=None):
        n_par = int(np.asarray(active_dims).size) + (1 if flg_offset else 0)
        super().__init__(active_dims=active_dims, mean_init=None, flg_mean_trainable=False, flg_no_mean=True, sigma_n_init=sigma_n_init,
                         flg_train_sigma_n=flg_train_sigma_n, Sigma_function=Parameters_covariance_functions.diagonal_covariance,
                         Sigma_f_additional_par_list=[n_par, True], Sigma_pos_par_init=None, flg_train_Sigma_pos_par=False,
                         Sigma_free_par_init=None, flg_train_Sigma_free_par=False, flg_offset=flg_offset, name=name, dtype=dtype,
                         sigma_n_num=sigma_n_num, device=device)
        self.poly_deg = int(poly_deg)
        init = np.asarray(Sigma_pos_par_init, dtype=float)
        self.Sigma_pos_par = torch.nn.Parameter(torch.tensor(np.log(init), dtype=dtype, device=self.device), requires_grad=flg_train_Sigma_pos_par)
        self.num_Sigma_pos_par = int(init.size / self.poly_deg)
        self.current_deg = 0  # (the reference's stateful cursor of get_Sigma, Sparse_GP.py:613-623,632-634)
        if self.num_Sigma_pos_par != n_par:
            raise RuntimeError("MPK_GP of degree %d over %d features needs %d parameters per factor" % (self.poly_deg, self.num_features, n_par))

    def get_Sigma_deg(self, current_deg):
        """Sigma of factor ``current_deg``: diag(s^2) with s = (poly_deg - current_deg) exp(par[current_deg]) -- the reference's loop
        re-adds the SAME slice once per remaining degree (Sparse_GP.py:648-656); reproduced on purpose."""
        n = self.num_Sigma_pos_par
        pos = (self.poly_deg - current_deg) * torch.exp(self.Sigma_pos_par[current_deg * n:(current_deg + 1) * n])
        return self.Sigma_function(pos, None, *self.Sigma_f_additional_par_list)

    def get_Sigma(self):
        return self.get_Sigma_deg(self.current_deg)

    def factor_weights(self):
        """Squared diagonal weights of each factor: list over d of [(k-d) exp(par_d)]^2."""
        return ops.mpk_weights(self.Sigma_pos_par.detach().cpu(), self.poly_deg)

    def kernel_spec_dev(self) -> ops.KernelSpec:
        """kernel_spec with everything left on the device (weights from the parameters by torch ops, noise in mcp_kernel.scal)."""
        D, dev = self.num_features, self.device
        w = ops.mpk_weights(self.Sigma_pos_par.detach().to(torch.float64), self.poly_deg)
        ones = torch.ones(D, dtype=torch.float64, device=dev)
        z = torch.zeros(1, dtype=torch.float64, device=dev)
        sig2 = self.get_sigma_n_2().detach().reshape(-1)[:1].to(torch.float64) if self.GP_with_noise else z
        scal = torch.cat([z, sig2, z]).contiguous()
        nan = float("nan")
        if self.poly_deg == 1:
            w1 = w[0] if self.flg_offset else torch.cat([w[0], z])
            return ops.KernelSpec(ones, nan, nan, nan, w1, None, None, scal=scal)
        if self.poly_deg == 2 and not self.flg_offset:
            return ops.KernelSpec(ones, nan, nan, nan, torch.zeros(D + 1, dtype=torch.float64, device=dev), w[0], w[1], scal=scal)
        raise NotImplementedError("the HIP kernels implement MPK degree 1 (with or without offset) and degree 2 (without offset)")

    def kernel_spec(self) -> ops.KernelSpec:
        D = self.num_features
        w = self.factor_weights()
        ones = torch.ones(D, dtype=torch.float64)
        sig2 = float(self.get_sigma_n_2()) if self.GP_with_noise else 0.0
        if self.poly_deg == 1:
            w1 = w[0] if self.flg_offset else torch.cat([w[0], torch.zeros(1, dtype=torch.float64)])
            return ops.KernelSpec(ones, 0.0, sig2, 0.0, w1, None, None)
        if self.poly_deg == 2 and not self.flg_offset:
            return ops.KernelSpec(ones, 0.0, sig2, 0.0, torch.zeros(D + 1, dtype=torch.float64), w[0], w[1])
        raise NotImplementedError("the HIP kernels implement MPK degree 1 (with or without offset) and degree 2 (without offset)")


def get_Volterra_MPK_GP(active_dims, poly_deg, sigma_n_init=None, flg_train_sigma_n=True, Sigma_pos_par_init_list=[],
                        flg_train_Sigma_pos_par_list=[], name="", dtype=torch.float64, sigma_n_num=None, device=None):
    """Sum of MPK_1 .. MPK_poly_deg: the first term carries the offset feature and the (optional) noise."""
    if poly_deg > 2:
        raise NotImplementedError("Volterra degree > 2 is not implemented by the HIP kernels")
    terms = [MPK_GP(active_dims, poly_deg=1, sigma_n_init=sigma_n_init, flg_train_sigma_n=flg_train_sigma_n,
                    Sigma_pos_par_init=Sigma_pos_par_init_list[0], flg_train_Sigma_pos_par=flg_train_Sigma_pos_par_list[0], flg_offset=True,
                    name="MPK_1", dtype=dtype, sigma_n_num=sigma_n_num, device=device)]
    for k in range(2, poly_deg + 1):
        terms.append(MPK_GP(active_dims, poly_deg=k, sigma_n_init=None, flg_train_sigma_n=False, Sigma_pos_par_init=Sigma_pos_par_init_list[k - 1],
                            flg_train_Sigma_pos_par=flg_train_Sigma_pos_par_list[k - 1], flg_offset=False, name="MPK_" + str(k), dtype=dtype,
                            sigma_n_num=None, device=device))
    return GP_prior.Sum_Independent_GP(*terms)
