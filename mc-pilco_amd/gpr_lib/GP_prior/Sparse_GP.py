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

__all__ = ["Linear_GP", "MPK_GP", "get_Volterra_MPK_GP"]


class Linear_GP(GP_prior.GP_prior):
    """Base of the dot-product kernels.  Only what MPK_GP needs is provided."""

    def __init__(self, active_dims, flg_offset=False, sigma_n_init=None, flg_train_sigma_n=True, name="", dtype=torch.float64, sigma_n_num=None,
                 device=None):
        if active_dims is None:
            raise RuntimeError("Active_dims are needed")
        super().__init__(active_dims, sigma_n_init=sigma_n_init, flg_train_sigma_n=flg_train_sigma_n, name=name, dtype=dtype,
                         sigma_n_num=sigma_n_num, device=device)
        self.num_features = int(np.asarray(active_dims).size)
        self.flg_offset = flg_offset
        self.flg_no_mean = True
        self.mean_par = torch.nn.Parameter(torch.zeros(1, dtype=dtype, device=self.device), requires_grad=False)

    def get_mean(self, X):
        return torch.zeros(X.shape[0], 1, dtype=self.dtype, device=self.device)


class MPK_GP(Linear_GP):
    def __init__(self, active_dims, poly_deg, sigma_n_init=None, flg_train_sigma_n=True, Sigma_pos_par_init=None, flg_train_Sigma_pos_par=True,
                 flg_offset=True, name="", dtype=torch.float64, sigma_n_num=None, device=None):
        super().__init__(active_dims, flg_offset=flg_offset, sigma_n_init=sigma_n_init, flg_train_sigma_n=flg_train_sigma_n, name=name, dtype=dtype,
                         sigma_n_num=sigma_n_num, device=device)
        self.poly_deg = int(poly_deg)
        init = np.asarray(Sigma_pos_par_init, dtype=float)
        self.Sigma_pos_par = torch.nn.Parameter(torch.tensor(np.log(init), dtype=dtype, device=self.device), requires_grad=flg_train_Sigma_pos_par)
        self.num_Sigma_pos_par = int(init.size / self.poly_deg)
        want = self.num_features + (1 if flg_offset else 0)
        if self.num_Sigma_pos_par != want:
            raise RuntimeError("MPK_GP of degree %d over %d features needs %d parameters per factor" % (self.poly_deg, self.num_features, want))

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
