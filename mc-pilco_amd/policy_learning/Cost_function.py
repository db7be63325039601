"""Expected costs on the HIP path -- drop-in for ``policy_learning/Cost_function.py``.

  Expected_cost.forward                          Cost_function.py:25-36    sum_t mean_m c , sum_t std_m c (unbiased, detached)
  Cart_pole_cost / cart_pole_cost                Cost_function.py:150-182
  Expected_saturated_distance_from_trajectory    Cost_function.py:104-147

The two costs the launch scripts use run in the HIP cost kernels (forward and state gradient).  A generic
``Expected_cost(cost_function)`` with a user-supplied torch function keeps working (it is user code and runs
as ordinary torch ops on the GPU), as do the simple distance variants built on it (:39-101).
``forward(..., group=None)``: with a torch.distributed group the mean / std pool every rank's particles.
"""
import numpy as np
import torch

from mc_pilco_amd import ops


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


class Expected_cost(torch.nn.modules.loss._Loss):
    """sum over time of the particle mean of ``cost_function(states, inputs, trial_index)`` [T,M]."""

    def __init__(self, cost_function):
        super().__init__()
        self.cost_function = cost_function

    def forward(self, states_sequence, inputs_sequence, trial_index=None, group=None, counts=None):
        costs = self.cost_function(states_sequence, inputs_sequence, trial_index)
        if group is not None:
            import torch.distributed as dist

            # pooled over the ranks' particles; shards may be uneven: the total count is reduced, never assumed
            if counts is None:
                n = torch.tensor([float(costs.shape[1])], dtype=costs.dtype, device=costs.device)
                dist.all_reduce(n, group=group)
                n = float(n.item())
            else:
                n = float(sum(int(c) for c in counts))
            s1 = costs.sum(1)
            s1_all = s1.detach().clone()
            dist.all_reduce(s1_all, group=group)
            mean = s1_all / n
            m2 = ((costs.detach() - mean[:, None]) ** 2).sum(1)
            dist.all_reduce(m2, group=group)
            return torch.sum(s1) / n + (torch.sum(mean) - torch.sum(s1.detach()) / n), torch.sum(torch.sqrt(m2 / (n - 1)))
        return torch.sum(torch.mean(costs, 1)), torch.sum(torch.std(costs.detach(), 1))

    def local_moments(self, states_sequence, inputs_sequence, trial_index, m_total, shift=None, sums_out=None):
        """This rank's share of a particle-sharded cost in summable form (sharding.StepReducer): (share = sum_t sum_m c / m_total,
        differentiable;  sums [2T] = per time step sum_m (c - shift_t), sum_m (c - shift_t)^2, detached).  ``sums_out``: where to leave the
        sums (the slot of the step's all-reduce message, sharding.StepMessage.sums)."""
        costs = self.cost_function(states_sequence, inputs_sequence, trial_index)
        d = costs.detach() - (0.0 if shift is None else shift.reshape(-1, 1))
        sums = torch.cat([d.sum(1), (d * d).sum(1)])
        if sums_out is not None:
            sums_out.copy_(sums)
            sums = sums_out
        return costs.sum() / float(m_total), sums

    @staticmethod
    def from_sums(sums, n_total, shift=None, mean_out=None):
        """(cost, std) of the pooled swarm from the all-reduced sums."""
        T = sums.numel() // 2
        a, b = sums[:T], sums[T:]
        mean = a / n_total + (0.0 if shift is None else shift)
        if mean_out is not None:
            mean_out.copy_(mean)
        return torch.sum(mean), torch.sum(torch.sqrt(torch.clamp(b - a * a / n_total, min=0.0) / (n_total - 1)))


class _HipExpectedCost(Expected_cost):
    def __init__(self):
        super().__init__(None)
        self._packed = None

    def _pack(self, states):
        raise NotImplementedError()

    def forward(self, states_sequence, inputs_sequence=None, trial_index=None, group=None, counts=None):
        if self._packed is None or self._packed.device != states_sequence.device:
            self._packed = self._pack(states_sequence)
        return ops.expected_cost(self._packed, states_sequence, group, counts)

    def local_moments(self, states_sequence, inputs_sequence, trial_index, m_total, shift=None, sums_out=None):
        if self._packed is None or self._packed.device != states_sequence.device:
            self._packed = self._pack(states_sequence)
        return ops.local_cost(self._packed, states_sequence, m_total, shift, sums_out)

    @staticmethod
    def from_sums(sums, n_total, shift=None, mean_out=None):
        out = ops.cost_from_sums(sums, n_total, shift, mean_out)
        return out[0], out[1]


class Cart_pole_cost(_HipExpectedCost):
    """1 - exp(-((|theta|-theta*)/l_theta)^2 - ((x-x*)/l_x)^2);  target_state=[theta*, x*], lengthscales=[l_theta, l_x]."""

    def __init__(self, target_state, lengthscales, angle_index, pos_index):
        super().__init__()
        self.target_state, self.lengthscales = _np(target_state).reshape(-1), _np(lengthscales).reshape(-1)
        self.angle_index, self.pos_index = int(angle_index), int(pos_index)

    def _pack(self, states):
        return ops.PackedCost("cartpole", states.shape[2], states.device, target_state=self.target_state, lengthscales=self.lengthscales,
                              angle_index=self.angle_index, pos_index=self.pos_index)


class Expected_saturated_distance_from_trajectory(_HipExpectedCost):
    """1 - exp(-sum_i ((x_i - x*_{t,i}) / l_i)^2) over ``used_indeces``; target_traj must have one row per time step.
    ``flg_var_lengthscales``: ``lengthscales[trial_index]`` is the lengthscale vector of that trial (Cost_function.py:136-141)."""

    def __init__(self, target_traj, lengthscales, flg_var_lengthscales=False, used_indeces=None):
        super().__init__()
        self.flg_var_lengthscales = bool(flg_var_lengthscales)
        self.target_traj = _np(target_traj)
        self.lengthscales = [_np(l).reshape(-1) for l in lengthscales] if self.flg_var_lengthscales else _np(lengthscales).reshape(-1)
        self.used_indeces = None if used_indeces is None else [int(i) for i in used_indeces]
        self._packed_by_trial = {}

    def _pack(self, states, trial_index=None):
        ls = self.lengthscales[trial_index] if self.flg_var_lengthscales else self.lengthscales
        return ops.PackedCost("traj", states.shape[2], states.device, target_traj=self.target_traj, lengthscales=ls, used=self.used_indeces)

    def _select(self, states, trial_index):
        """Per-trial lengthscales: one packed descriptor per trial index, chosen before the base class evaluates."""
        if self.flg_var_lengthscales:
            key = (int(trial_index), str(states.device))
            if key not in self._packed_by_trial:
                self._packed_by_trial[key] = self._pack(states, int(trial_index))
            self._packed = self._packed_by_trial[key]

    def forward(self, states_sequence, inputs_sequence=None, trial_index=None, group=None, counts=None):
        self._select(states_sequence, trial_index)
        return super().forward(states_sequence, inputs_sequence, trial_index, group, counts)

    def local_moments(self, states_sequence, inputs_sequence, trial_index, m_total, shift=None, sums_out=None):
        self._select(states_sequence, trial_index)
        return super().local_moments(states_sequence, inputs_sequence, trial_index, m_total, shift, sums_out)


# ---- the two costs as plain functions (Cost_function.py:124-147, 170-182) -----------------------------------------------
# What a user hands to the generic ``Expected_cost(cost_function=...)``: ordinary differentiable torch ops on the GPU (user-code
# path; the classes above run the same formulas in the HIP cost kernels).
def saturated_distance_from_trajectory(states_sequence, inputs_sequence, trial_index, target_traj, lengthscales, flg_var_lengthscales,
                                       used_indeces):
    """1 - exp(-sum_i ((x_i - x*_{t,i}) / l_i)^2) over ``used_indeces`` (None: every state); ``flg_var_lengthscales``:
    ``lengthscales[trial_index]`` is the vector of that trial.  [T,M,S] -> [T,M]."""
    if used_indeces is None:
        used_indeces = list(range(states_sequence.shape[2]))
    tt = torch.as_tensor(target_traj, dtype=states_sequence.dtype, device=states_sequence.device)
    targets = tt.reshape(tt.shape[0], 1, -1).expand(states_sequence.shape)
    ls = lengthscales[trial_index] if flg_var_lengthscales else lengthscales
    ls = torch.as_tensor(ls, dtype=states_sequence.dtype, device=states_sequence.device)
    d = (states_sequence[:, :, used_indeces] - targets[:, :, used_indeces]) / ls
    return 1 - torch.exp(-(d * d).sum(2))


def cart_pole_cost(states_sequence, inputs_sequence, trial_index, target_state, lengthscales, angle_index, pos_index):
    """1 - exp(-((|theta| - theta*)/l_theta)^2 - ((x - x*)/l_x)^2);  target_state = [theta*, x*], lengthscales = [l_theta, l_x]."""
    x = states_sequence[:, :, pos_index]
    theta = states_sequence[:, :, angle_index]
    return 1 - torch.exp(-(((torch.abs(theta) - target_state[0]) / lengthscales[0]) ** 2) - ((x - target_state[1]) / lengthscales[1]) ** 2)


# ---- simple torch-level variants (Cost_function.py:39-101) -------------------------------------------------------------
def distance_from_target(states_sequence, inputs_sequence, trial_index, target_state, lengthscales, active_dims):
    d = (states_sequence[:, :, active_dims] - target_state) / lengthscales
    return (d * d).sum(2)


def saturated_distance_from_target(states_sequence, inputs_sequence, trial_index, target_state, lengthscales, active_dims):
    return 1 - torch.exp(-distance_from_target(states_sequence, inputs_sequence, trial_index, target_state, lengthscales, active_dims))


class Expected_distance(Expected_cost):
    def __init__(self, target_state, lengthscales, active_dims):
        super().__init__(lambda x, u, k: distance_from_target(x, u, k, target_state, lengthscales, active_dims))


class Expected_saturated_distance(Expected_cost):
    def __init__(self, target_state, lengthscales, active_dims):
        super().__init__(lambda x, u, k: saturated_distance_from_target(x, u, k, target_state, lengthscales, active_dims))
