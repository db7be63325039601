"""Diagnostic: time of one GP hyper-parameter training epoch on the device (GP_prior.fit_model with the analytic
marginal-likelihood gradient kernel) at the cart-pole size, next to the same epoch of the CPU oracle formulation
(Cholesky + autograd, the reference's way)."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mcp_boot, numpy as np, torch
from mc_pilco_amd import synthetic as sy
from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood
from mc_pilco_amd.model_learning import Model_learning as ML
dev = torch.device("cuda", 0); dt = torch.float64
c = sy.CARTPOLE
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rbf = dict(active_dims=np.arange(6), lengthscales_init=np.ones(6), flg_train_lengthscales=True, lambda_init=np.ones(1), flg_train_lambda=False,
           sigma_n_init=np.ones(1), sigma_n_num=None, flg_train_sigma_n=True, dtype=dt, device=dev)
par = dict(num_gp=2, T_sampling=c["Ts"], angle_indeces=c["angle"], not_angle_indeces=c["not_angle"], vel_indeces=c["vel"], not_vel_indeces=c["not_vel"],
           dtype=dt, device=dev, init_dict_list=[rbf] * 2)
rolls = sy.cartpole_rollouts(n_roll=(N + 59) // 60)
with contextlib.redirect_stdout(io.StringIO()):
    ml = ML.Speed_Model_learning_RBF_angle_state(**par)
    for xs, us in rolls:
        ml.add_data(np.asarray(xs), np.asarray(us))
    opt = dict(f_optimizer="lambda p : torch.optim.Adam(p, lr=0.01)", criterion=Likelihood.Marginal_log_likelihood, N_epoch=20, N_epoch_print=1000)
    ml.reinforce_model([opt, opt])  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt["N_epoch"] = 200
    ml.reinforce_model([opt, opt])
    torch.cuda.synchronize(); el = time.perf_counter() - t0
print("fit_model on the device: N=%d, D=6: %.2f ms per epoch per GP (200 epochs x 2 GPs in %.2f s)" % (ml.gp_inputs.shape[0], 1e3 * el / 400, el))
