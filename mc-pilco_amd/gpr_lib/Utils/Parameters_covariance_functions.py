"""Parameter-covariance builders -- only ``diagonal_covariance`` is on the hot path
(``gpr_lib/Utils/Parameters_covariance_functions.py:18-27``, used by MPK_GP)."""
import torch


def diagonal_covariance(pos_par=None, free_par=None, num_par=None, flg_ARD=False):
    """diag(pos_par^2); without ARD one shared value on a num_par x num_par diagonal."""
    if flg_ARD:
        if num_par != pos_par.shape[0]:
            raise RuntimeError("The number of positive parameters and num_par must be equal when flg_ARD=True")
        return torch.diag(pos_par ** 2)
    return pos_par ** 2 * torch.eye(num_par, dtype=pos_par.dtype, device=pos_par.device)
