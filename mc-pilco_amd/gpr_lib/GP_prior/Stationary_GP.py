"""Stationary GPs on the HIP path -- drop-in for ``gpr_lib/GP_prior/Stationary_GP.py``.

RBF:  k(a,b) = exp(log_lambda) * exp(-sum_d ((a_d-b_d)/l_d)^2)   (no factor 1/2, Stationary_GP.py:162-170),
constant prior mean ``mean_par`` (:157-160), diagonal lambda (+ noise) (:172-181).
Parameter names (``log_lengthscales_par``, ``log_lambda_par``, ``mean_par``, ``sigma_n_log``) are the
reference's, so ``state_dict``s and ``log.pkl`` files interoperate.
"""
import numpy as np
import torch

from mc_pilco_amd import ops

from . import GP_prior

__all__ = ["Stationary_GP", "RBF"]


class Stationary_GP(GP_prior.GP_prior):
    def __init__(self, active_dims, lengthscales_init=None, flg_train_lengthscales=True, sigma_n_init=None, flg_train_sigma_n=True, name="",
                 dtype=torch.float64, sigma_n_num=None, device=None):
        if active_dims is None:
            raise RuntimeError("Stationary_GP obj require active_dims")
        super().__init__(active_dims, sigma_n_init=sigma_n_init, flg_train_sigma_n=flg_train_sigma_n, name=name, dtype=dtype,
                         sigma_n_num=sigma_n_num, device=device)
        self.num_features = int(np.asarray(active_dims).size)
        if lengthscales_init is None:
            lengthscales_init = np.ones(self.num_features)
        lengthscales_init = np.asarray(lengthscales_init, dtype=float)
        self.flg_ARD = lengthscales_init.size != 1
        self.log_lengthscales_par = torch.nn.Parameter(torch.tensor(np.log(lengthscales_init), dtype=dtype, device=self.device),
                                                       requires_grad=flg_train_lengthscales)

    def lengthscales(self):
        ls = torch.exp(self.log_lengthscales_par.detach())
        return ls if self.flg_ARD else ls.reshape(-1)[:1].repeat(self.num_features)

    def get_weigted_distances(self, X1, X2=None):
        """Lengthscale-weighted squared distances [N1,N2] (convenience; not on the hot path)."""
        ls = self.lengthscales()
        A = self._cols(X1) / ls
        Bm = A if X2 is None else self._cols(X2) / ls
        return ((A[:, None, :] - Bm[None, :, :]) ** 2).sum(2)


class RBF(Stationary_GP):
    def __init__(self, active_dims, lengthscales_init=None, flg_train_lengthscales=True, sigma_n_init=None, flg_train_sigma_n=True,
                 lambda_init=None, flg_train_lambda=True, mean_init=None, flg_train_mean=False, name="", dtype=torch.float64,
                 sigma_n_num=None, device=None):
        super().__init__(active_dims, lengthscales_init=lengthscales_init, flg_train_lengthscales=flg_train_lengthscales,
                         sigma_n_init=sigma_n_init, flg_train_sigma_n=flg_train_sigma_n, name=name, dtype=dtype, sigma_n_num=sigma_n_num,
                         device=device)
        lambda_init = np.ones(1) if lambda_init is None else np.asarray(lambda_init, dtype=float)
        if lambda_init.size != 1:
            raise RuntimeError("Lambda must be a np array qith dimension 1")
        self.log_lambda_par = torch.nn.Parameter(torch.tensor(np.log(lambda_init), dtype=dtype, device=self.device), requires_grad=flg_train_lambda)
        mean_init = np.zeros(1) if mean_init is None else np.asarray(mean_init, dtype=float)
        self.mean_par = torch.nn.Parameter(torch.tensor(mean_init, dtype=dtype, device=self.device), requires_grad=flg_train_mean)

    def get_mean(self, X):
        return self.mean_par.detach().reshape(1, -1).repeat(X.shape[0], 1)

    def kernel_spec_dev(self) -> ops.KernelSpec:
        """The descriptor with its three scalars left ON THE DEVICE (mcp_kernel.scal): no device->host transfer, so an epoch of
        GP_prior.fit_model never waits for the GPU (the host enqueues the next epoch while the last one runs)."""
        z = torch.zeros(1, dtype=torch.float64, device=self.device)
        sig2 = self.get_sigma_n_2().detach().reshape(-1)[:1].to(torch.float64) if self.GP_with_noise else z
        scal = torch.cat([torch.exp(self.log_lambda_par.detach()).reshape(-1)[:1].to(torch.float64), sig2,
                          self.mean_par.detach().reshape(-1)[:1].to(torch.float64)]).contiguous()
        nan = float("nan")
        return ops.KernelSpec(self.lengthscales().to(torch.float64), nan, nan, nan, scal=scal)

    def kernel_spec(self) -> ops.KernelSpec:
        # the three scalars of the descriptor in ONE device->host transfer
        sig2_t = self.get_sigma_n_2().detach().reshape(-1)[:1] if self.GP_with_noise else torch.zeros(1, dtype=self.dtype, device=self.device)
        sc = torch.cat([sig2_t.to(torch.float64), torch.exp(self.log_lambda_par.detach()).reshape(-1)[:1].to(torch.float64),
                        self.mean_par.detach().reshape(-1)[:1].to(torch.float64)]).cpu()
        return ops.KernelSpec(self.lengthscales().to(torch.float64), float(sc[1]), float(sc[0]), float(sc[2]))
