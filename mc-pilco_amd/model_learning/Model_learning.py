"""GP dynamics models on the HIP path -- drop-in for ``model_learning/Model_learning.py``.

Same class names, constructor keywords (including the reference's spellings ``angle_indeces``,
``vel_indeces``), public attributes (``gp_list``, ``gp_inputs``, ``gp_output_list``, ``alpha_list``,
``m_X_list``, ``K_X_inv_list``, ``gp_inputs_tr_list``, ``SOD_indices``, ``norm_list``, ``num_gp``) and
method signatures.  Reference lines replaced:

  add_data / data_to_gp_IO         Model_learning.py:123-147, 450-469, 662-683
  reinforce_model / pretrain_gp    Model_learning.py:149-208   (Gram, Cholesky, inverse, alpha, SOD on the device)
  get_next_state / get_one_step_gp_out / get_*_gp_estimate   Model_learning.py:210-336
  get_next_state_from_gp_output    Model_learning.py:471-493 (delta model), 685-718 (speed integration)

``packed()`` returns the device-resident operands in the fused kernels' layout (mc_pilco_amd.ops.PackedModel);
``policy_learning.MC_PILCO.apply_policy`` hands it to the fused rollout.  Out of scope: the SOR approximation
and SP_Speed_Model_learning_Furuta (unused by every launch script).
"""
import numpy as np
import torch
from torch.distributions.normal import Normal

from mc_pilco_amd import ops
from mc_pilco_amd.gpr_lib.GP_prior import GP_prior as GP
from mc_pilco_amd.gpr_lib.GP_prior import Sparse_GP
from mc_pilco_amd.gpr_lib.GP_prior import Stationary_GP as SGP
from mc_pilco_amd.gpr_lib.Likelihood import Gaussian_likelihood as Likelihood  # noqa: F401  (optimizer strings may name it)


class Model_learning(torch.nn.Module):
    """Delta-state GP model: GP i predicts x_{t+1}[i] - x_t[i] from [x_t, u_t]."""

    def __init__(self, num_gp, init_dict_list, approximation_mode=None, approximation_dict=None, dtype=torch.float64,
                 device=torch.device("cuda"), flg_norm=False):
        super().__init__()
        self.num_samples = 0
        self.dtype = dtype
        self.device = torch.device(device)
        self.init_dict_list = init_dict_list
        self.num_gp = num_gp
        self.alpha_list = [None] * num_gp
        self.m_X_list = [None] * num_gp
        self.K_X_inv_list = [None] * num_gp
        self.gp_inputs_tr_list = [None] * num_gp
        self._packed_gps = [None] * num_gp
        self._packed_model = None
        self.approximation_mode = approximation_mode
        if approximation_mode is None:
            print("EXACT GP INFERENCE SELECTED")
            self.get_gp_estimate = self.get_exact_gp_estimate
        else:
            self.approximation_dict = approximation_dict
            print("GP APPROXIMATION SELECTED")
            print("APPROXIMATION MODE: ", approximation_mode)
            print("APPROXIMATION OPTIONS: ", approximation_dict)
            if approximation_mode == "SOD":
                self.SOD_indices = [None] * num_gp
                self.get_gp_estimate = self.get_SOD_gp_estimate
                self.SOD_threshold_mode = approximation_dict["SOD_threshold_mode"]
                self.SOD_threshold = approximation_dict["SOD_threshold"]
                self.flg_SOD_permutation = approximation_dict["flg_SOD_permutation"]
            else:
                raise NotImplementedError("approximation_mode %r is not implemented on the HIP path (SOR is unused by the launch scripts)"
                                          % (approximation_mode,))
        self.init_gp_models()
        self.flg_norm = flg_norm
        self.norm_list = [1.0] * self.num_gp

    # ---- GP objects ---------------------------------------------------------------------------------
    def init_gp_models(self):
        self.gp_list = torch.nn.ModuleList([self.get_gp(gp_index=i, init_dict=self.init_dict_list[i]) for i in range(self.num_gp)])

    def get_gp(self, gp_index, init_dict):
        raise NotImplementedError()

    def set_eval_mode(self):
        for gp in self.gp_list:
            gp.set_eval_mode()

    def set_training_mode(self):
        for gp in self.gp_list:
            gp.set_training_mode()

    def to(self, device):
        super().to(device)
        self.device = torch.device(device)
        for gp in self.gp_list:
            gp.to(device)

    def print_model(self):
        for i, gp in enumerate(self.gp_list):
            print("GP " + str(i + 1) + ":")
            gp.print_model()

    # ---- data ---------------------------------------------------------------------------------------------
    def _t(self, a):
        return torch.as_tensor(np.asarray(a), dtype=self.dtype).to(self.device)

    def add_data(self, new_state_samples, new_input_samples):
        """Appends one interaction (states [n,S], inputs [n,U] as numpy) as GP inputs / targets."""
        zin, youts = self.data_to_gp_IO(self._t(new_state_samples), self._t(new_input_samples))
        if self.num_samples == 0:
            self.dim_state = new_state_samples.shape[1]
            self.dim_input = new_input_samples.shape[1]
            self.gp_inputs, self.gp_output_list = zin, youts
            self.num_samples = new_state_samples.shape[0]
        else:
            self.gp_inputs = torch.cat([self.gp_inputs, zin])
            self.gp_output_list = [torch.cat([a, b], 0) for a, b in zip(self.gp_output_list, youts)]
            self.num_samples = self.gp_inputs.shape[0]

    def data_to_gp_input(self, states, inputs):
        return torch.cat([states, inputs], 1)

    def data_to_gp_output(self, states):
        return [(states[1:, i] - states[:-1, i]).reshape([-1, 1]) for i in range(self.dim_state)]

    def data_to_gp_IO(self, states, inputs):
        if not hasattr(self, "dim_state"):
            self.dim_state = states.shape[1]
        return self.data_to_gp_input(states, inputs)[:-1, :], self.data_to_gp_output(states)

    # ---- training / pretraining ---------------------------------------------------------------------------------
    def reinforce_model(self, optimization_opt_list=None):
        self.init_gp_models()
        texts = self._train_all_gps_at_once(optimization_opt_list)
        if texts is not None:
            for i in range(self.num_gp):  # (the sequential run's output order: GP i's training text, then its pretrain)
                print(texts[i], end="")
                with torch.no_grad():
                    self.pretrain_gp(gp_index=i)
            return
        for i in range(self.num_gp):
            self.train_gp(gp_index=i, optimization_opt_dict=optimization_opt_list[i])
            with torch.no_grad():
                self.pretrain_gp(gp_index=i)

    def _train_all_gps_at_once(self, optimization_opt_list):
        """The GPs of a model are independent (the reference trains them one after the other, Model_learning.py:149-161): when they
        share kernel structure, training options and the plain Adam, every epoch of ALL of them is one batched launch sequence
        (mcp_nll_epoch: the grid carries the GP index) + one Adam launch.  Returns the text the sequential run prints, per GP, or
        None: not applicable -- the caller trains them one by one."""
        from mc_pilco_amd import nll

        if type(self).train_gp is not Model_learning.train_gp or type(self).train_gp_likelihood is not Model_learning.train_gp_likelihood:
            return None  # (a subclass with its own training)
        opts = list(optimization_opt_list[:self.num_gp])
        if len(opts) < self.num_gp or self.num_gp < 1:
            return None
        o0 = opts[0]
        from mc_pilco_amd.gpr_lib.Likelihood.Gaussian_likelihood import Marginal_log_likelihood

        for o in opts:
            if (o.get("f_optimizer") != o0.get("f_optimizer") or o.get("N_epoch") != o0.get("N_epoch") or o.get("N_epoch_print") != o0.get("N_epoch_print")
                    or o.get("criterion") is not Marginal_log_likelihood or o.get("train_mode", "likelihood") != "likelihood"):
                return None
        if self.flg_norm:
            for i in range(self.num_gp):
                self.norm_list[i] = torch.max(torch.abs(self.gp_output_list[i]))
        f_optim = eval(o0["f_optimizer"])  # same optimizer strings as the reference
        optims = [f_optim(gp.parameters()) for gp in self.gp_list]
        fit = nll.BatchedFit(self.gp_list, self.gp_inputs, self.gp_output_list, [1.0 / float(self.norm_list[i]) for i in range(self.num_gp)], optims,
                             o0["N_epoch"], o0["N_epoch_print"])
        return fit.run() if fit.eligible else None

    def train_gp(self, gp_index, optimization_opt_dict):
        self.train_gp_likelihood(gp_index, optimization_opt_dict)

    def train_gp_likelihood(self, gp_index, optimization_opt_dict):
        if self.flg_norm:
            self.norm_list[gp_index] = torch.max(torch.abs(self.gp_output_list[gp_index]))
        # The reference wraps the data in a DataLoader with batch_size = N and shuffle = False (Model_learning.py:403-411): ONE full batch
        # per epoch, in order.  The same iteration as a one-element list: the DataLoader indexes the N samples one by one and stacks
        # them again every epoch -- 300 tiny device ops, 0.9 ms of host time per epoch, more than the epoch's kernels take.
        loader = [(self.gp_inputs, self.gp_output_list[gp_index] / self.norm_list[gp_index])]
        f_optim = eval(optimization_opt_dict["f_optimizer"])  # same optimizer strings as the reference
        self.gp_list[gp_index].fit_model(trainloader=loader, optimizer=f_optim(self.gp_list[gp_index].parameters()),
                                         criterion=optimization_opt_dict["criterion"](), N_epoch=optimization_opt_dict["N_epoch"],
                                         N_epoch_print=optimization_opt_dict["N_epoch_print"])

    def pretrain_gp(self, gp_index):
        """Caches alpha, m_X, K_X_inv (and the SOD subset) of GP ``gp_index`` and packs them for the kernels."""
        gp = self.gp_list[gp_index]
        X, Y = self.gp_inputs, self.gp_output_list[gp_index]
        if self.approximation_mode == "SOD":
            if self.SOD_threshold_mode == "relative":
                threshold = self.SOD_threshold * torch.sqrt(gp.get_sigma_n_2())
            elif self.SOD_threshold_mode == "absolute":
                threshold = self.SOD_threshold[gp_index]
            idx = gp.get_SOD(X=X, Y=Y, threshold=threshold, flg_permutation=self.flg_SOD_permutation)
            self.SOD_indices[gp_index] = idx
            Xtr, Ytr = X[idx, :], Y[idx, :]
        else:
            Xtr, Ytr = X, Y
        Y_hat, var, alpha, m_X, K_X_inv = gp.get_estimate(X=Xtr, Y=Ytr, X_test=X, flg_return_K_X_inv=True)
        self.K_X_inv_list[gp_index] = K_X_inv
        self.alpha_list[gp_index] = alpha
        self.m_X_list[gp_index] = m_X
        self.gp_inputs_tr_list[gp_index] = Xtr
        self._packed_gps[gp_index] = ops.PackedGP(gp.kernel_spec(), gp._cols(Xtr), alpha, K_X_inv)
        self._packed_model = None
        print("MSE gp " + str(gp_index) + ": ", torch.mean((Y - Y_hat) ** 2))

    def packed_gp(self, gp_index):
        if self._packed_gps[gp_index] is None:
            raise RuntimeError("GP %d has not been pretrained (call pretrain_gp / reinforce_model first)" % gp_index)
        return self._packed_gps[gp_index]

    def packed(self):
        raise NotImplementedError("only the speed-integration models have a fused-rollout layout")

    # ---- one-step prediction ------------------------------------------------------------------------------------------
    def get_next_state(self, current_state, current_input, particle_pred=True):
        """x_{t+1} samples (or means) with the mean and variance of the GP outputs."""
        _, _, mean_list, var_list = self.get_one_step_gp_out(states=current_state, inputs=current_input)
        var_list = [v * self.norm_list[i] ** 2 for i, v in enumerate(var_list)]
        return self.get_next_state_from_gp_output(current_state=current_state, current_input=current_input, gp_output_mean_list=mean_list,
                                                  gp_output_var_list=var_list, particle_pred=particle_pred)

    def get_one_step_gp_out(self, states, inputs):
        gp_inputs = self.data_to_gp_input(states=states, inputs=inputs)
        mean_list, var_list = self.get_gp_estimate(gp_inputs=gp_inputs, gp_index_list=range(self.num_gp))
        return gp_inputs, None, mean_list, var_list

    def get_gp_estimate_from_data(self, states, inputs, flg_pretrain=False, gp_index_list=None, flg_onestep=False):
        if gp_index_list is None:
            gp_index_list = range(self.num_gp)
        if flg_onestep:
            gp_inputs, outs = self.data_to_gp_input(states=states, inputs=inputs), None
        else:
            gp_inputs, outs = self.data_to_gp_IO(states=states, inputs=inputs)
        if flg_pretrain:
            for i in gp_index_list:
                self.pretrain_gp(gp_index=i)
        mean_list, var_list = self.get_gp_estimate(gp_inputs=gp_inputs, gp_index_list=gp_index_list)
        return gp_inputs, outs, mean_list, var_list

    def _estimates(self, gp_inputs, gp_index_list):
        means, variances = [], []
        for i in gp_index_list:
            mu, var = ops.posterior(self.packed_gp(i), self.gp_list[i]._cols(gp_inputs))
            means.append(mu)
            variances.append(var.reshape([-1, 1]))
        return means, variances

    def get_exact_gp_estimate(self, gp_inputs, gp_index_list=None):
        return self._estimates(gp_inputs, range(self.num_gp) if gp_index_list is None else gp_index_list)

    def get_SOD_gp_estimate(self, gp_inputs, gp_index_list):
        return self._estimates(gp_inputs, gp_index_list)

    def get_next_state_from_gp_output(self, current_state, current_input, gp_output_mean_list, gp_output_var_list, particle_pred=True):
        delta_mean = torch.cat(gp_output_mean_list, 1)
        delta_var = torch.cat(gp_output_var_list, 1)
        delta = Normal(delta_mean, torch.sqrt(delta_var)).rsample() if particle_pred else delta_mean
        return current_state + delta, delta_mean, delta_var


class Model_learning_RBF(Model_learning):
    def get_gp(self, gp_index, init_dict):
        return SGP.RBF(**init_dict)


class Model_learning_RBF_angle_state(Model_learning):
    """RBF GPs over [x_notangle, sin(angle), cos(angle), u]."""

    def __init__(self, num_gp, init_dict_list, angle_indeces, not_angle_indeces, approximation_mode=None, approximation_dict=None,
                 dtype=torch.float64, device=torch.device("cuda"), flg_norm=False):
        self.angle_indeces = angle_indeces
        self.not_angle_indeces = not_angle_indeces
        super().__init__(num_gp=num_gp, init_dict_list=init_dict_list, approximation_mode=approximation_mode,
                         approximation_dict=approximation_dict, dtype=dtype, device=device, flg_norm=flg_norm)

    def get_gp(self, gp_index, init_dict):
        return SGP.RBF(**init_dict)

    def data_to_gp_input(self, states, inputs):
        ang = states[:, self.angle_indeces]
        return torch.cat([states[:, self.not_angle_indeces], torch.sin(ang), torch.cos(ang), inputs], 1)


class Model_learning_RBF_MPK_angle_state(Model_learning_RBF_angle_state):
    def get_gp(self, gp_index, init_dict):
        return GP.Sum_Independent_GP(SGP.RBF(**init_dict[0]), Sparse_GP.get_Volterra_MPK_GP(**init_dict[1]))


class Speed_Model_learning_RBF_angle_state(Model_learning):
    """Speed-integration model: GP g predicts the change of velocity state vel_indeces[g]; the matching
    position not_vel_indeces[g] is integrated:  v' = v + d,  q' = q + Ts v + Ts/2 d."""

    def __init__(self, num_gp, init_dict_list, T_sampling, angle_indeces, not_angle_indeces, vel_indeces, not_vel_indeces,
                 approximation_mode=None, approximation_dict=None, dtype=torch.float64, device=torch.device("cuda"), flg_norm=False):
        self.vel_indeces = vel_indeces
        self.not_vel_indeces = not_vel_indeces
        self.angle_indeces = angle_indeces
        self.not_angle_indeces = not_angle_indeces
        self.T_sampling = T_sampling
        super().__init__(num_gp=num_gp, init_dict_list=init_dict_list, approximation_mode=approximation_mode,
                         approximation_dict=approximation_dict, dtype=dtype, device=device, flg_norm=flg_norm)

    def get_gp(self, gp_index, init_dict):
        return SGP.RBF(**init_dict)

    def data_to_gp_output(self, states):
        return [(states[1:, i] - states[:-1, i]).reshape([-1, 1]) for i in self.vel_indeces]

    def data_to_gp_input(self, states, inputs):
        ang = states[:, self.angle_indeces]
        return torch.cat([states[:, self.not_angle_indeces], torch.sin(ang), torch.cos(ang), inputs], 1)

    def get_next_state_from_gp_output(self, current_state, current_input, gp_output_mean_list, gp_output_var_list, particle_pred=True):
        dv_mean = torch.cat(gp_output_mean_list, 1)
        dv_var = torch.cat(gp_output_var_list, 1)
        dv = Normal(dv_mean, torch.sqrt(dv_var)).rsample() if particle_pred else dv_mean
        nxt = torch.zeros(current_state.shape, dtype=self.dtype, device=self.device)
        nxt[:, self.vel_indeces] = current_state[:, self.vel_indeces] + dv
        nxt[:, self.not_vel_indeces] = (current_state[:, self.not_vel_indeces] + self.T_sampling * current_state[:, self.vel_indeces]
                                        + self.T_sampling / 2 * dv)
        return nxt, dv_mean, dv_var

    def packed(self):
        """The whole model in the fused kernels' layout (built after every pretrain)."""
        if self._packed_model is None:
            gps = [self.packed_gp(i) for i in range(self.num_gp)]
            S = len(self.vel_indeces) + len(self.not_vel_indeces)
            U = gps[0].D - len(self.not_angle_indeces) - 2 * len(self.angle_indeces)
            scale = [float(n) ** 2 for n in self.norm_list]
            self._packed_model = ops.PackedModel(gps, S, U, self.T_sampling, self.angle_indeces, self.not_angle_indeces, self.vel_indeces,
                                                 self.not_vel_indeces, var_scale=scale)
        return self._packed_model


class Speed_Model_learning_RBF_MPK_angle_state(Speed_Model_learning_RBF_angle_state):
    """Speed-integration model with RBF + Volterra-polynomial kernels."""

    def get_gp(self, gp_index, init_dict):
        return GP.Sum_Independent_GP(SGP.RBF(**init_dict[0]), Sparse_GP.get_Volterra_MPK_GP(**init_dict[1]))
