"""Control policies on the HIP path -- drop-in for ``policy_learning/Policy.py``.

  Sum_of_gaussians                         Policy.py:153-265   RBF network + dropout + linear + tanh squashing (:52-60)
  Sum_of_gaussians_with_angles             Policy.py:268-335   features [x_nonangle, cos, sin]
  Sum_of_gaussians_with_target_trajectory  Policy.py:338-403   features [x, x*_t - x]

``forward(states, t=None, p_dropout=0.0)`` evaluates the policy with the fused kernel (T = 1) and is
differentiable with respect to ``log_lengthscales``, ``centers`` and ``f_linear.weight`` (names as in the
reference, so ``state_dict``s interoperate).  ``packed()`` exposes the live parameters to the fused rollout.
Dropout noise: ``noise_mode = "philox"`` (in-kernel counter-based generator, default) or ``"torch_cpu"`` (mask
drawn exactly like ``torch.nn.functional.dropout`` draws it on the CPU -- parity with the reference).
``scale_factor`` (plain ``Sum_of_gaussians`` only, as in the reference) is folded into the operands handed to the kernels.
``flg_bias`` adds ``f_linear.bias`` (not dropped out) to the linear layer inside the kernels; its gradient comes out of the adjoint sweep.
Exploration policies (Random_exploration, Sum_of_sinusoids :94-150, PD_controller :406-449) drive the simulated system once per
trial (61-201 samples): host-side, their ``np.random`` draws in the reference's order so that a seeded launch script collects the
same data.
"""
import numpy as np
import torch

from mc_pilco_amd import ops


class Policy(torch.nn.Module):
    def __init__(self, state_dim, input_dim, flg_squash=False, u_max=1, dtype=torch.float64, device=torch.device("cuda")):
        super().__init__()
        self.state_dim = state_dim
        self.input_dim = input_dim
        self.dtype = dtype
        self.device = torch.device(device)
        self.flg_squash = flg_squash
        self.u_max = u_max

    def forward(self, states, t=None, p_dropout=0.0):
        raise NotImplementedError()

    def forward_np(self, state, t=None):
        with torch.no_grad():
            u = self(states=torch.as_tensor(np.asarray(state), dtype=self.dtype).to(self.device), t=t)
        return u.detach().cpu().numpy()

    def to(self, device):
        super().to(device)
        self.device = torch.device(device)

    def squashing(self, u, u_max):
        """u_max tanh(u / u_max), u_max a scalar or one bound per input (Policy.py:52-60).  The fused kernels apply the same
        squashing inside the rollout; this method is the reference's public helper for user code."""
        if not np.isscalar(u_max):
            u_max = torch.tensor(u_max, dtype=self.dtype, device=self.device)
        return u_max * torch.tanh(u / u_max)

    def f_squash(self, x):
        """The reference's ``self.f_squash`` (Policy.py:28-33): squashing when ``flg_squash``, identity otherwise."""
        return self.squashing(x, self.u_max) if self.flg_squash else x

    def get_np_policy(self):
        return lambda state, t: self.forward_np(state, t)

    def reinit(self, scaling=1):
        raise NotImplementedError()


class Random_exploration(Policy):
    """Uniform random input in (-u_max, u_max); host-side, used only to collect data from the system."""

    def __init__(self, state_dim, input_dim, flg_squash=True, u_max=1.0, dtype=torch.float64, device=torch.device("cpu")):
        super().__init__(state_dim=state_dim, input_dim=input_dim, flg_squash=flg_squash, u_max=u_max, dtype=dtype, device=device)

    def forward(self, states, t=None, p_dropout=0.0):
        return torch.as_tensor(self.forward_np(states, t), dtype=self.dtype)

    def forward_np(self, state, t=None):
        return (np.asarray(self.u_max) * (2 * np.random.rand(self.input_dim) - 1)).reshape([-1, self.input_dim])


class Sum_of_sinusoids(Policy):
    """Exploration policy: u(t) = squash(sum_i A_i sin(omega_i t + phi_i)) with random amplitudes / frequencies / phases
    (Policy.py:94-150).  The draws are ``np.random``'s, in the reference's order: amplitudes (rand), omega (choice, rand),
    phases (choice, rand) -- a launch script seeded with ``np.random.seed`` gets the reference's exploration signal."""

    def __init__(self, state_dim, input_dim, num_sin, omega_min, omega_max, amplitude_min, amplitude_max, flg_squash=False, u_max=1,
                 dtype=torch.float64, device=torch.device("cpu")):
        super().__init__(state_dim=state_dim, input_dim=input_dim, flg_squash=flg_squash, u_max=u_max, dtype=dtype, device=device)
        self.num_sin = num_sin
        amplitude_min, amplitude_max = np.array(amplitude_min), np.array(amplitude_max)
        par = lambda a: torch.nn.Parameter(torch.tensor(a, dtype=self.dtype, device=self.device), requires_grad=False)
        self.amplitudes = par(amplitude_min + (amplitude_max - amplitude_min) * np.random.rand(num_sin, input_dim))
        self.omega = par(np.random.choice([-1, 1], [num_sin, input_dim]) * (omega_min + (omega_max - omega_min) * np.random.rand(num_sin, input_dim)))
        self.phases = par(np.random.choice([-1, 1], [num_sin, input_dim]) * (np.pi * (np.random.rand(num_sin, input_dim) - 0.5)))

    def forward(self, states, t, p_dropout=0.0):
        return self.f_squash(torch.sum(self.amplitudes * torch.sin(self.omega * t + self.phases), dim=0).reshape([-1, self.input_dim]))


class PD_controller(Policy):
    """u = squash(Kp e_q + Kd e_qdot) with e = target_traj[t] - state, gains = sqrt_*_gains^2 (Policy.py:406-449; the UR5 launch
    script's exploration controller).  Host-side helper for the system simulator; not part of the fused rollout."""

    def __init__(self, state_dim, input_dim, sqrt_Kp_gains, sqrt_Kd_gains, target_traj=None, flg_squash=True, u_max=1.0, flg_trainable=False,
                 dtype=torch.float64, device=torch.device("cpu")):
        super().__init__(state_dim=state_dim, input_dim=input_dim, flg_squash=flg_squash, u_max=u_max, dtype=dtype, device=device)
        self.target_traj = target_traj
        self.sqrt_Kp_gains = torch.nn.Parameter(torch.tensor(sqrt_Kp_gains, dtype=self.dtype, device=self.device), requires_grad=flg_trainable)
        self.sqrt_Kd_gains = torch.nn.Parameter(torch.tensor(sqrt_Kd_gains, dtype=self.dtype, device=self.device), requires_grad=flg_trainable)

    def forward(self, states, t, p_dropout=0.0):
        states = states.reshape([-1, self.state_dim])
        err = self.target_traj[t, :].reshape(1, -1) - states
        h = int(self.state_dim / 2)
        return self.f_squash(self.sqrt_Kp_gains ** 2 * err[:, 0:h] + self.sqrt_Kd_gains ** 2 * err[:, h:])


class Sum_of_gaussians(Policy):
    _kind = "plain"

    def __init__(self, state_dim, input_dim, num_basis, flg_train_lengthscales=True, lengthscales_init=None, flg_train_centers=True,
                 centers_init=None, centers_init_min=-1, centers_init_max=1, weight_init=None, flg_train_weight=True, flg_bias=False,
                 bias_init=None, flg_train_bias=False, flg_squash=False, u_max=1, scale_factor=None, flg_drop=True, dtype=torch.float64,
                 device=torch.device("cuda")):
        super().__init__(state_dim=state_dim, input_dim=input_dim, flg_squash=flg_squash, u_max=u_max, dtype=dtype, device=device)
        self.num_basis = num_basis
        if lengthscales_init is None:
            lengthscales_init = np.ones(state_dim)
        self.log_lengthscales = torch.nn.Parameter(torch.tensor(np.log(lengthscales_init), dtype=dtype, device=self.device).reshape([1, -1]),
                                                   requires_grad=flg_train_lengthscales)
        if centers_init is None:
            centers_init = centers_init_min + (centers_init_max - centers_init_min) * np.random.rand(num_basis, state_dim)
        self.centers = torch.nn.Parameter(torch.tensor(np.asarray(centers_init), dtype=dtype, device=self.device), requires_grad=flg_train_centers)
        self.f_linear = torch.nn.Linear(in_features=num_basis, out_features=input_dim, bias=bool(flg_bias))
        w = np.ones([input_dim, num_basis]) if weight_init is None else np.asarray(weight_init)
        self.f_linear.weight = torch.nn.Parameter(torch.tensor(w, dtype=dtype, device=self.device).contiguous(), requires_grad=flg_train_weight)
        if flg_bias:  # Policy.py:203-212: torch.nn.Linear's own initial bias unless bias_init is given; trained only with flg_train_bias
            b0 = self.f_linear.bias.detach().to(torch.float64).cpu().numpy() if bias_init is None else np.asarray(bias_init, dtype=float).reshape(-1)
            self.f_linear.bias = torch.nn.Parameter(torch.tensor(b0, dtype=dtype, device=self.device).contiguous(), requires_grad=flg_train_bias)
        # states / scale_factor before the RBF layer (Policy.py:220-222, 252): folded into the operands the kernels see --
        # ((s/f - c)/l)^2 = ((s - c f)/(l f))^2, i.e. centres c*f and log-lengthscales log l + log f (see packed())
        sf = np.ones(state_dim) if scale_factor is None else np.asarray(scale_factor, dtype=float).reshape(-1)
        self.scale_factor = torch.tensor(sf, dtype=dtype, device=self.device).reshape([1, -1])
        self._unit_scale = bool(np.all(sf == 1.0))
        self.flg_drop = flg_drop
        self.noise_mode = "philox"
        self.seed = 0
        self._calls = 0
        self._packed = None

    # ---- descriptors ---------------------------------------------------------------------------------------------
    def _system_state_dim(self):
        return self.state_dim

    def _pack_extra(self):
        return {}

    def packed(self) -> ops.PackedPolicy:
        """mcp_policy over the LIVE parameter tensors (optimizer updates are seen; rebuilt if a tensor was replaced)."""
        if not self._unit_scale:
            # derived operands, rebuilt per call (the parameters move every optimizer step); autograd carries the chain rule
            # back to log_lengthscales / centers through the two elementwise ops
            return ops.PackedPolicy(self._kind, self._system_state_dim(), self.log_lengthscales + torch.log(self.scale_factor),
                                    (self.centers * self.scale_factor).contiguous(), self.f_linear.weight, self.u_max, self.flg_squash,
                                    bias=self.f_linear.bias, **self._pack_extra())
        key = (self.log_lengthscales.data_ptr(), self.centers.data_ptr(), self.f_linear.weight.data_ptr(),
               None if self.f_linear.bias is None else self.f_linear.bias.data_ptr())
        if self._packed is None or self._packed[0] != key:
            pk = ops.PackedPolicy(self._kind, self._system_state_dim(), self.log_lengthscales, self.centers, self.f_linear.weight, self.u_max,
                                  self.flg_squash, bias=self.f_linear.bias, **self._pack_extra())
            self._packed = (key, pk)
        return self._packed[1]

    def reinit(self, lenghtscales_par, centers_par, weight_par):
        """Policy.py:229-240: lengthscales reset, centres uniform in +-centers_par, weights uniform in +-weight_par / 2 (two
        ``torch.rand`` draws, in this order).  ``draw_device`` (set by MC_PILCO in "reference" noise mode) = where the draws are made:
        on the CPU generator they are the reference's own numbers for the same seed."""
        dev, dt = self.device, self.dtype
        ddev = getattr(self, "draw_device", None) or dev
        self.log_lengthscales.data = torch.tensor(np.log(lenghtscales_par), dtype=dt, device=dev).reshape([1, -1])
        self.centers.data = torch.tensor(centers_par, dtype=dt, device=dev) * 2 * (torch.rand(self.num_basis, self.state_dim, dtype=dt, device=ddev).to(dev) - 0.5)
        self.f_linear.weight.data = weight_par * (torch.rand(self.input_dim, self.num_basis, dtype=dt, device=ddev).to(dev) - 0.5)
        self._packed = None

    # ---- evaluation -------------------------------------------------------------------------------------------------
    def dropout_noise(self, M, T, p_dropout):
        """NoiseSpec for T policy evaluations of M particles."""
        p = float(p_dropout) if self.flg_drop else 0.0
        self._calls += 1
        if p > 0.0 and self.noise_mode == "torch_cpu":
            masks = torch.stack([torch.empty(M, 1, self.num_basis, dtype=self.dtype).bernoulli_(1 - p).reshape(M, self.num_basis) for _ in range(T)])
            return ops.NoiseSpec(masks=masks.to(torch.uint8).to(self.device).contiguous()), p
        return ops.NoiseSpec(seed=self.seed, call=self._calls), p

    def forward(self, states, t=None, p_dropout=0.0):
        x = states.reshape([-1, self._system_state_dim()]).to(self.device)
        noise, p = self.dropout_noise(x.shape[0], 1, p_dropout)
        pk = self.packed()
        if pk.kind == "traj":
            pk = self._packed_at(int(t))
        _, inputs, _ = ops.rollout(None, pk, noise, x, 1, p)
        return inputs[0]


class Sum_of_gaussians_with_angles(Sum_of_gaussians):
    _kind = "angles"

    def __init__(self, state_dim, input_dim, num_basis, angle_indices, non_angle_indices, flg_train_lengthscales=True, lengthscales_init=None,
                 flg_train_centers=True, centers_init=None, centers_init_min=-1, centers_init_max=1, weight_init=None, flg_train_weight=True,
                 flg_bias=False, bias_init=None, flg_train_bias=False, flg_squash=False, u_max=1, flg_drop=True, dtype=torch.float64,
                 device=torch.device("cuda")):
        self.angle_indices = np.asarray(angle_indices)
        self.non_angle_indices = np.asarray(non_angle_indices)
        self.num_angle_indices = self.angle_indices.size
        self.num_non_angle_indices = self.non_angle_indices.size
        self._sys_dim = state_dim
        super().__init__(state_dim=state_dim + self.num_angle_indices, input_dim=input_dim, num_basis=num_basis,
                         flg_train_lengthscales=flg_train_lengthscales, lengthscales_init=lengthscales_init, flg_train_centers=flg_train_centers,
                         centers_init=centers_init, centers_init_min=centers_init_min, centers_init_max=centers_init_max, weight_init=weight_init,
                         flg_train_weight=flg_train_weight, flg_bias=flg_bias, bias_init=bias_init, flg_train_bias=flg_train_bias,
                         flg_squash=flg_squash, u_max=u_max, flg_drop=flg_drop, dtype=dtype, device=device)

    def _system_state_dim(self):
        return self._sys_dim

    def _pack_extra(self):
        return dict(angle=[int(i) for i in self.angle_indices], non_angle=[int(i) for i in self.non_angle_indices])


class Sum_of_gaussians_with_target_trajectory(Sum_of_gaussians):
    _kind = "traj"

    def __init__(self, state_dim, input_dim, num_basis, target_traj, flg_train_lengthscales=True, lengthscales_init=None, flg_train_centers=True,
                 centers_init=None, centers_init_min=-1, centers_init_max=1, weight_init=None, flg_train_weight=True, flg_bias=False,
                 bias_init=None, flg_train_bias=False, flg_squash=False, u_max=1, flg_drop=True, dtype=torch.float64, device=torch.device("cuda")):
        super().__init__(state_dim=state_dim, input_dim=input_dim, num_basis=num_basis, flg_train_lengthscales=flg_train_lengthscales,
                         lengthscales_init=lengthscales_init, flg_train_centers=flg_train_centers, centers_init=centers_init,
                         centers_init_min=centers_init_min, centers_init_max=centers_init_max, weight_init=weight_init,
                         flg_train_weight=flg_train_weight, flg_bias=flg_bias, bias_init=bias_init, flg_train_bias=flg_train_bias,
                         flg_squash=flg_squash, u_max=u_max, flg_drop=flg_drop, dtype=dtype, device=device)
        self.target_traj = torch.as_tensor(np.asarray(target_traj.detach().cpu() if isinstance(target_traj, torch.Tensor) else target_traj),
                                           dtype=self.dtype).to(self.device)

    def _system_state_dim(self):
        return self.state_dim // 2

    def _pack_extra(self):
        return dict(target_traj=self.target_traj)

    def _packed_at(self, t):
        """Descriptor whose target row 0 is x*_t (single-step evaluation at time t)."""
        return ops.PackedPolicy("traj", self._system_state_dim(), self.log_lengthscales, self.centers, self.f_linear.weight, self.u_max,
                                self.flg_squash, target_traj=self.target_traj[t:t + 1], bias=self.f_linear.bias)
