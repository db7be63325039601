"""MC-PILCO driver on the HIP path -- drop-in for ``policy_learning/MC_PILCO.py`` (class ``MC_PILCO``).

  apply_policy       MC_PILCO.py:615-674   particle rollout  -> ONE fused HIP launch (mcp_rollout_fwd)
  reinforce_policy   MC_PILCO.py:375-613   optimizer loop: rollout, expected cost, backward (fused reverse-time adjoint,
                                           mcp_rollout_bwd), optimizer step, cost monitors, lr / dropout annealing, NaN retries
  reinforce          MC_PILCO.py:89-258    trial loop and ``log.pkl`` bookkeeping (same keys)
  rollout            MC_PILCO.py:347-373   mean-only single-trajectory prediction

Constructor injection is the plug-in mechanism, as in the reference: model / policy / cost classes and their kwargs.
Additions (all optional): ``noise_mode`` -- "philox" (in-kernel generator, default) or "reference" (noise drawn on the
CPU with the reference's torch calls in its order, for seed-for-seed parity); ``shard_particles(group)`` -- split the
particles over the ranks of a torch.distributed group (one all-gather of cost moments and one all-reduce of the
policy gradient per optimizer step; every rank then applies the identical update).
``MC_PILCO4PMS`` (MC_PILCO.py:755-958, partially measurable systems): the measurement filter between particles and policy is
part of the fused rollout kernels (``mcp_meas``); a step-wise path on the posterior / policy operators remains as fallback.
Out of scope: MC_PILCO_Experiment, MuJoCo environments.
"""
import copy
import pickle as pkl
import time

import numpy as np
import torch
from torch.distributions.multivariate_normal import MultivariateNormal
from torch.distributions.uniform import Uniform

from mc_pilco_amd import ops, sharding
from mc_pilco_amd.policy_learning import Cost_function as _Cost
from mc_pilco_amd.policy_learning import Policy as _Policy
from mc_pilco_amd.simulation_class import model as _sim


class MC_PILCO(torch.nn.Module):
    def __init__(self, T_sampling, state_dim, input_dim, f_sim, f_model_learning, model_learning_par, f_rand_exploration_policy,
                 rand_exploration_policy_par, f_control_policy, control_policy_par, f_cost_function, cost_function_par, std_meas_noise=None,
                 log_path=None, dtype=torch.float64, device=torch.device("cuda")):
        super().__init__()
        self.T_sampling = T_sampling
        self.dtype = dtype
        self.device = torch.device(device)
        self.state_dim = state_dim
        self.input_dim = input_dim
        print("\n\nGet the system...")
        self.system = _sim.Model(f_sim)
        self.std_meas_noise = np.zeros(state_dim) if std_meas_noise is None else std_meas_noise
        print("\n\nGet the learning object...")
        self.model_learning = f_model_learning(**model_learning_par)
        print("\n\nGet the exploration policy...")
        self.rand_exploration_policy = f_rand_exploration_policy(**rand_exploration_policy_par)
        print("\n\nGet the control policy...")
        self.control_policy = f_control_policy(**control_policy_par)
        print("\n\nGet the cost function...")
        self.cost_function = f_cost_function(**cost_function_par)
        self.state_samples_history = []
        self.input_samples_history = []
        self.noiseless_states_history = []
        self.num_data_collection = 0
        self.log_path = log_path
        if self.log_path is not None:
            self.log_dict = {}
        # HIP-path options
        self.noise_mode = "philox"
        self.seed = 0
        self._rollout_calls = 0
        self.dist_group = None
        self.last_status = None
        self.gp_sharding = True    # cleared for good once a GP-sharded launch reports MCP_STATUS_SYNC (co-residency was not there)
        self._reducer = None       # sharding.StepReducer: the one all-reduce of a sharded optimizer step
        self._step_msgs = {}       # sharding.StepMessage by (with gradients?): the persistent flat message of that all-reduce
        self._cost_shift = None    # previous step's pooled per-time-step mean cost (the shift of the summable cost moments)
        self.pipeline_depth = 1    # reinforce_policy reads an attempt's outcome this many attempts late (0: at once); see there
        self.capture_attempts = False  # True: reinforce_policy records an attempt into a HIP graph and replays it (pipelined loop only); see there
        self._call_dev = None      # device int64 [1]: the rollout counter of replayed attempts (mcp_noise.call_dev), None while attempts run eagerly
        self.attempts_replayed = 0  # attempts of the last reinforce_policy that ran as a graph replay (diagnostic)

    # ------------------------------------------------------------------------------------------------------------
    # particle sharding
    # ------------------------------------------------------------------------------------------------------------
    def shard_particles(self, group=None, transport="torch"):
        """Split ``num_particles`` over the ranks of ``group`` (default: the WORLD group).  Per optimizer step the ranks then meet
        in ONE all-reduce (sharding.StepReducer; ``transport`` "torch" = torch.distributed, "abi" = the C ABI's RCCL communicator).
        Every rank must seed torch identically (the launch scripts' ``torch.manual_seed(seed)``): x0 -- and in "reference"
        noise mode eps and the masks -- are drawn for ALL particles on every rank and sliced, so a sharded run simulates
        exactly the particles one GPU would."""
        import torch.distributed as dist

        self.dist_group = dist.group.WORLD if group is None else group
        self._reducer = sharding.StepReducer(self.dist_group, transport)

    def _world(self):
        if self.dist_group is None:
            return 1, 0
        import torch.distributed as dist

        return dist.get_world_size(self.dist_group), dist.get_rank(self.dist_group)

    # ------------------------------------------------------------------------------------------------------------
    # forward simulation of the particles
    # ------------------------------------------------------------------------------------------------------------
    def sample_initial_particles(self, mean, var, flg_uniform, up_bound, low_bound, flg_multi_gauss, num_particles):
        """x_0 ~ uniform / mixture of Gaussians / Gaussian.  In "reference" noise mode the draw is made on the CPU with
        the reference's own distribution calls (bit-exact for a given torch seed)."""
        if self.noise_mode != "reference" and not flg_uniform and not flg_multi_gauss:
            # same distribution as MultivariateNormal(mean, diag(var)) without building M covariance matrices and their Cholesky
            # factors on every optimizer step
            # (two launches per draw -- randn, addcmul -- instead of four: the standard deviations are kept per variance tensor and version)
            mean = mean.to(self.device).reshape(1, -1)
            hit = self.__dict__.get("_x0_std")
            if hit is None or hit[0] is not var or hit[1] != int(var._version):
                hit = self.__dict__["_x0_std"] = (var, int(var._version), torch.sqrt(var.to(self.device).reshape(1, -1)))
            return torch.addcmul(mean, hit[2], torch.randn(num_particles, mean.shape[1], dtype=mean.dtype, device=self.device))
        on = torch.device("cpu") if self.noise_mode == "reference" else self.device
        mean, var = mean.to(on), var.to(on)
        if flg_uniform:
            dist_ = Uniform(low_bound.to(on).repeat(num_particles, 1), up_bound.to(on).repeat(num_particles, 1))
        elif flg_multi_gauss:
            idx = torch.randint(0, mean.shape[0], [num_particles], device=on)
            dist_ = MultivariateNormal(loc=mean[idx, :], covariance_matrix=torch.diag_embed(var[idx, :]))
        else:
            dist_ = MultivariateNormal(loc=mean.repeat(num_particles, 1), covariance_matrix=torch.diag_embed(var.repeat(num_particles, 1)))
        return dist_.rsample().to(self.device)

    def _shard_slice(self, t, dim):
        """This rank's particles of a tensor drawn for the whole swarm (no-op on a single rank)."""
        off, cnt = self._shard
        return t if cnt == t.shape[dim] else t.narrow(dim, off, cnt).contiguous()

    def _rollout_noise(self, M, T, p_dropout):
        pol = self.control_policy
        p = float(p_dropout) if getattr(pol, "flg_drop", True) else 0.0
        G, B = self.model_learning.num_gp, pol.num_basis
        self._rollout_calls += 1
        if self.noise_mode == "reference":
            # the reference's draw order: mask_0, then for t = 1..T-1: eps_t, mask_t    -- drawn for the WHOLE swarm
            # (every rank draws the same numbers from the same seed) and sliced to this rank's particles
            Mt = self._m_total
            masks = [torch.empty(Mt, 1, B, dtype=self.dtype).bernoulli_(1 - p).reshape(Mt, B)] if p > 0 else None
            eps = []
            for _ in range(1, T):
                eps.append(torch.empty(Mt, G, dtype=self.dtype).normal_())
                if p > 0:
                    masks.append(torch.empty(Mt, 1, B, dtype=self.dtype).bernoulli_(1 - p).reshape(Mt, B))
            eps = self._shard_slice(torch.stack(eps) if eps else torch.zeros(0, Mt, G, dtype=self.dtype), 1).to(self.device).contiguous()
            mk = None if masks is None else self._shard_slice(torch.stack(masks).to(torch.uint8), 1).to(self.device).contiguous()
            return ops.NoiseSpec(eps=eps, masks=mk), p
        return self._philox_noise(), p

    def _philox_noise(self):
        """In-kernel noise keyed by (seed, rollout counter, global particle).  While an attempt is being recorded into a graph the counter is the
        device word the graph advances (by-value part 0); the host's ``_rollout_calls`` mirrors it either way."""
        if self._call_dev is not None:
            return ops.NoiseSpec(seed=self.seed, call=0, particle_offset=self._shard[0], call_dev=self._call_dev)
        return ops.NoiseSpec(seed=self.seed, call=self._rollout_calls, particle_offset=self._shard[0])

    def apply_policy(self, particles_initial_state_mean, particles_initial_state_var, flg_particles_init_uniform, particles_init_up_bound,
                     particles_init_low_bound, flg_particles_init_multi_gauss, num_particles, T_control, p_dropout=0.0):
        """Simulates ``num_particles`` particles for ``T_control`` steps under the control policy.
        Returns states [T,M,S] and inputs [T,M,U] (differentiable w.r.t. the policy parameters)."""
        world, rank = self._world()
        self._shard = sharding.shard_range(int(num_particles), world, rank)
        self._m_total = int(num_particles)
        M = self._shard[1]
        T = int(T_control)
        x0 = self._shard_slice(self.sample_initial_particles(particles_initial_state_mean, particles_initial_state_var, flg_particles_init_uniform,
                                                             particles_init_up_bound, particles_init_low_bound, flg_particles_init_multi_gauss,
                                                             self._m_total), 0)
        pol, ml = self.control_policy, self.model_learning
        if isinstance(pol, _Policy.Sum_of_gaussians) and hasattr(ml, "vel_indeces"):
            noise, p = self._rollout_noise(M, T, p_dropout)
            states, inputs, status = ops.rollout(ml.packed(), pol.packed(), noise, x0, T, p, gp_sharding=self.gp_sharding)
            self.last_status = status
            return states, inputs
        # generic (unfused) path: any model / policy object with the reference's step interface
        self.last_status = None  # (no fused launch: the flags of an earlier fused rollout do not describe this one)
        if world > 1:
            # its noise (torch draws inside get_next_state / the policy's dropout) is per LOCAL particle: identically seeded ranks
            # would simulate correlated shards, not the particles one GPU would
            raise NotImplementedError("particle sharding needs the fused rollout (Sum_of_gaussians policy + speed-integration model)")
        xs = [x0]
        us = [pol(x0, t=0, p_dropout=p_dropout)]
        for t in range(1, T):
            x, _, _ = ml.get_next_state(current_state=xs[-1], current_input=us[-1])
            xs.append(x)
            us.append(pol(x, t=t, p_dropout=p_dropout))
        return torch.stack(xs), torch.stack(us)

    def _cost(self, states, inputs, trial_index):
        """(cost, std) of the whole swarm WITHOUT running backward (warm-up of the cost monitors, user calls).  Sharded: one
        all-reduce of the summable cost moments."""
        if self.dist_group is None:
            return self.cost_function(states, inputs, trial_index)
        with torch.no_grad():
            cost, std, _ = self._cost_backward(states, inputs, trial_index, backward=False)
        return cost, std

    def _step_flags(self, cost):
        """Device vector [cost is NaN, a GP-sharded launch timed out (MCP_STATUS_SYNC), a predictive variance was <= 0
        (MCP_STATUS_NONPOS_VAR)] of the last fused rollout."""
        from mc_pilco_amd import hipabi

        st = self.last_status
        if st is None or st.device != cost.device:
            sync = nonpos = torch.zeros((), dtype=torch.bool, device=cost.device)
        else:
            sync = (st.reshape(-1)[0] & hipabi.STATUS_SYNC) != 0
            nonpos = (st.reshape(-1)[0] & hipabi.STATUS_NONPOS_VAR) != 0
        return torch.stack([torch.isnan(cost.detach()).reshape(()), sync.reshape(()), nonpos.reshape(())]).to(self.dtype)

    def _cost_backward(self, states, inputs, trial_index, backward=True, flags_as_vector=True):
        """Expected cost of the rollout and (``backward``) its gradient in the policy parameters' ``.grad``.
        Returns (cost, std, flags): flags is a device vector, > 0 where [the cost is NaN, a hand-off timed out] -- on EVERY rank
        alike, so all ranks take the same retry decision.

        Single process: the reference's two lines, ``cost_function(...)`` and ``cost.backward()`` (MC_PILCO.py:496,522).
        Sharded: each rank forms its share sum_t sum_m c / M_total, runs its own backward sweep (the gradient of the pooled mean
        needs nothing from the other ranks), and then ONE all-reduce sums [gradients | cost sums | flags]."""
        if self.dist_group is None:
            cost, std = self.cost_function(states, inputs, trial_index)
            if backward:
                # queued before anybody looks at the cost; gradients of a NaN rollout are discarded
                cost.backward(retain_graph=False)
            # (flags_as_vector False: the caller hands the rollout's status word and the cost to mcp_policy_step_commit itself)
            return cost, std, (self._step_flags(cost) if flags_as_vector else None)
        T = states.shape[0]
        if self._cost_shift is None or self._cost_shift.numel() != T:
            self._cost_shift = torch.zeros(T, dtype=self.dtype, device=states.device)
        cf = self.cost_function
        params = list(self.control_policy.parameters())
        # the step's message [gradients | cost sums | flags], kept per (policy size, horizon, with / without gradients): the adjoint sweep and
        # the cost kernels write straight into it and the all-reduce runs on it in place (sharding.StepMessage)
        n_grad = sum(q.numel() for q in params if q.requires_grad) if backward else 0
        msg = self._step_msgs.get(bool(backward))
        if msg is None or not msg.fits(n_grad, T, 3, states.device):
            msg = self._step_msgs[bool(backward)] = sharding.StepMessage(n_grad, T, 3, states.device, self.dtype)
        packed = self.control_policy.packed() if (backward and isinstance(self.control_policy, _Policy.Sum_of_gaussians)) else None
        share, sums = cf.local_moments(states, inputs, trial_index, self._m_total, self._cost_shift, sums_out=msg.sums)
        if backward:
            if packed is not None and packed.grad_numel() == n_grad:
                packed.grad_flat = msg.grad  # (mcp_rollout_bwd leaves log_ls | centers | weight | bias there: the order of `params`)
            try:
                share.backward(retain_graph=False)
            finally:
                if packed is not None:
                    packed.grad_flat = None
            for q in params:  # a parameter the cost does not reach still takes part in the message (every rank sends the same layout)
                if q.requires_grad and q.grad is None:
                    q.grad = torch.zeros_like(q)
        # one all-reduce; pooled cost / std; a NaN rollout neither poisons the next steps' shift nor goes unnoticed on the other ranks
        cost, std, flags, self._cost_shift = sharding.finish_step(cf, self._reducer, params if backward else [], sums, self._step_flags(share),
                                                                  self._m_total, self._cost_shift, msg=msg)
        return cost, std, flags

    # ------------------------------------------------------------------------------------------------------------
    # policy optimisation
    # ------------------------------------------------------------------------------------------------------------
    def _rollout_failed(self, flags):
        """Reads the step's flags (ONE device->host transfer; warm-up rollout and user code -- the optimizer loop itself reads the
        record of ``mcp_policy_step_commit``).  True when the cost is NaN (data, not an error: MC_PILCO.py:497) or when a GP-sharded
        launch timed out waiting for a partner workgroup; in the second case the GP-sharded launch forms are switched off for this
        object (the device was not giving the grid co-residency -- another process, CU masking), so the repeated step runs on the
        unsharded kernels: never a silently wrong trajectory, never a rank-local raise."""
        nan, sync, nonpos = (float(v) for v in flags.tolist())
        return self._judge_attempt(nan, sync, nonpos)

    def _judge_attempt(self, nan, sync, nonpos):
        if nonpos > 0:
            # The reference samples with Normal(mean, sqrt(var)).rsample() (Model_learning.py:704), whose argument validation raises
            # ValueError on a scale that is not > 0 -- a zero / negative predictive variance is a modelling error there, not a case of
            # the NaN retry, whatever the cost of that rollout turns out to be (sqrt of a negative variance makes it NaN).  The kernels
            # raise this flag for a FINITE variance <= 0 only; a NaN variance (divergence) is MCP_STATUS_NAN, the retry case.  Same on
            # every rank of a sharded run: the flag travels with the step's all-reduce.
            raise ValueError("Expected parameter scale of the particles' sampling distribution to be > 0: a GP's predictive variance was <= 0 "
                             "(MCP_STATUS_NONPOS_VAR)")
        if sync > 0:
            if self.gp_sharding:
                print("\nGP-sharded rollout: a partner workgroup never arrived (MCP_STATUS_SYNC) -- continuing on the unsharded kernels")
            self.gp_sharding = False
            return True
        return nan > 0

    @staticmethod
    def _plain_adam(opt):
        """(lr, beta1, beta2, eps) when ``opt`` is a torch.optim.Adam whose update is the textbook one (what every launch script
        builds: "lambda p, lr : torch.optim.Adam(p, lr)") -- the loop then runs the update itself, guarded on the device
        (mcp_adam_step_guarded), and never has to wait for a step's outcome.  None: any other optimizer; its own ``step()`` is called,
        after the host has seen that the attempt counts."""
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return None
        g = opt.param_groups[0]
        if (g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False) or g.get("differentiable", False)
                or g.get("decoupled_weight_decay", False) or isinstance(g["lr"], torch.Tensor)):
            return None
        ps = [q for q in g["params"] if q.requires_grad]
        if not ps or len(ps) > 32 or any(q.dtype != torch.float64 or not q.is_cuda or not q.is_contiguous() for q in ps):
            return None
        return float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])

    def reinforce_policy(self, T_control, num_particles, trial_index, particles_initial_state_mean, particles_initial_state_var,
                         flg_particles_init_uniform, particles_init_up_bound, particles_init_low_bound, flg_particles_init_multi_gauss,
                         opt_steps_list, lr_list, f_optimizer, num_step_print=10, policy_reinit_dict=None, p_dropout_list=None,
                         std_cost_filt_order=None, std_cost_filt_cutoff=None, max_std_cost=None, alpha_cost=0.99, alpha_input=0.99,
                         alpha_diff_cost=0.99, lr_reduction_ratio=0.5, lr_min=0.001, p_drop_reduction=0.0, min_diff_cost=0.1,
                         num_min_diff_cost=200, min_step=np.inf):
        """Monte-Carlo policy gradient: at most ``opt_steps_list[trial_index]`` optimizer steps (MC_PILCO.py:375-613).

        One ATTEMPT = rollout + expected cost + adjoint sweep; what the reference then decides on the host from ``torch.isnan(cost)``
        -- does the attempt count, the cost-difference monitors, the lr / exit condition -- is decided on the device by
        ``mcp_policy_step_commit`` (and, for a plain Adam, the update itself by ``mcp_adam_step_guarded``), which leaves a small record
        per attempt.  The host reads that record ``self.pipeline_depth`` attempts late (1: it has already enqueued the next attempt,
        so the GPU never idles while Python catches up; 0: at once -- the mode with host-drawn "reference" noise, with any other
        optimizer and in a particle-sharded run).  A failed attempt needs nothing from the host (the next attempt is the retry);
        where the host must act -- ten failures in a row, the lr / exit condition, the last step -- the device ignores the attempts
        enqueued meanwhile and the host rewinds the noise counters past them: both depths take exactly the same steps."""
        import ctypes as C

        from mc_pilco_amd import hipabi as abi

        dev, dt = self.device, self.dtype
        horizon = int(T_control / self.T_sampling)
        n_steps = opt_steps_list[trial_index]
        sim = dict(particles_initial_state_mean=particles_initial_state_mean, particles_initial_state_var=particles_initial_state_var,
                   flg_particles_init_uniform=flg_particles_init_uniform, flg_particles_init_multi_gauss=flg_particles_init_multi_gauss,
                   particles_init_up_bound=particles_init_up_bound, particles_init_low_bound=particles_init_low_bound,
                   num_particles=num_particles, T_control=horizon)
        p_drop0 = 0.0 if p_dropout_list is None else p_dropout_list[trial_index]
        if p_dropout_list is not None:
            print("\nDROPOUT ACTIVE:")
            print("p_dropout:", p_drop0)
        make_opt = eval(f_optimizer)  # the reference passes optimizers as strings, e.g. "lambda p, lr : torch.optim.Adam(p, lr)"
        # re-initialisations draw where the rest of the noise is drawn: in "reference" mode on the CPU generator (the reference's stream)
        self.control_policy.draw_device = torch.device("cpu") if self.noise_mode == "reference" else None

        # reference value for the cost-difference monitor (policy re-initialised while the cost is NaN)
        with torch.no_grad():
            for _ in range(10):
                st0, in0 = self.apply_policy(p_dropout=p_drop0, **sim)
                cost0, _, fl0 = self._cost_backward(st0, in0, trial_index, backward=False)
                sharded_before = self.gp_sharding
                if not self._rollout_failed(fl0):
                    break
                if sharded_before and not self.gp_sharding:
                    continue  # a hand-off time-out, not a NaN: same policy, unsharded kernels
                print("\nSE filter initialization: Cost is NaN - reinit the policy")
                self.control_policy.reinit(**policy_reinit_dict)

        # ---- device-side loop state (mcp_opt_state + the monitors' arrays) ------------------------------------------------------------
        lib = abi.lib()
        if dt != torch.float64:
            # the loop state, the cost lists and the Adam kernel are double precision on the device (mcp_opt_state, mcp_adam_step_guarded):
            # another dtype would be read through double* -- refuse it here rather than take garbage decisions
            raise RuntimeError("reinforce_policy on the HIP path works in torch.float64 (the kernels are fp64); got dtype %s" % dt)
        st = torch.zeros(7, dtype=torch.int64, device=dev)  # mcp_opt_state: step, attempt, pending, adam_t, total_attempts | es2, cost_prev
        st[5:].view(torch.float64)[1:2].copy_(cost0.detach().reshape(1))   # cost_tm1 = the warm-up cost (MC_PILCO.py:462)
        cost_list = torch.zeros(n_steps, device=dev, dtype=dt)
        std_list = torch.zeros(n_steps, device=dev, dtype=dt)
        es1 = torch.zeros(n_steps + 1, device=dev, dtype=dt)
        ratio = torch.zeros(n_steps + 1, device=dev, dtype=dt)
        hs = dict(lr=lr_list[trial_index], p_drop=p_drop0, min_diff=min_diff_cost, min_step=min_step, prev_cost=0.0)  # what the host holds
        params = [q for q in self.control_policy.parameters()]
        opt = make_opt(p=self.control_policy.parameters(), lr=hs["lr"])
        adam = self._plain_adam(opt)
        depth = int(getattr(self, "pipeline_depth", 1))
        if adam is None or self.noise_mode == "reference" or self.dist_group is not None:
            depth = 0
        ad = {}

        def fresh_adam_state():
            if adam is not None:
                ps = [q for q in opt.param_groups[0]["params"] if q.requires_grad]
                ad.update(ps=ps, m=[torch.zeros_like(q) for q in ps], v=[torch.zeros_like(q) for q in ps],
                          numel=(C.c_int64 * len(ps))(*[q.numel() for q in ps]))
                for k in ("m", "v", "ps"):
                    ad["c_" + k] = (abi.dptr * len(ps))(*[t.data_ptr() for t in ad[k]])
            st[3:4].zero_()

        fresh_adam_state()
        ring = [torch.empty(abi.OPT_RECORD_DOUBLES, dtype=dt).pin_memory() for _ in range(depth + 2)]
        rec_dev = torch.zeros(depth + 2, abi.OPT_RECORD_DOUBLES, dtype=dt, device=dev)
        seq = [0]

        # ---- one attempt = rollout -> cost -> adjoint -> guarded Adam -> commit: ~15 launches and as many host calls.  In the pipelined loop it
        # is recorded ONCE into a HIP graph and replayed (round 6): everything an attempt reads that changes from one attempt to the next lives in
        # device memory -- the parameters, the loop state, torch's generator offset (graph-safe) and the rollout counter of the in-kernel noise
        # (mcp_noise.call_dev, advanced by the graph itself) -- so a replay takes the same step the eager calls would, bit for bit.  Two graphs
        # alternate (each with its own trajectories and record row: an attempt voided while the host decides must not overwrite the outputs of
        # the one before it).  The first two attempts after every (re)start run eagerly (they warm the launch paths); a host decision that
        # changes a recorded value -- lr, dropout, new Adam moments, re-initialised parameters -- drops the graphs.
        pol_ = self.control_policy
        use_graph = (bool(getattr(self, "capture_attempts", False)) and depth > 0 and adam is not None and dev.type == "cuda"
                     and type(self).apply_policy is MC_PILCO.apply_policy  # (the measurement-model rollout of MC_PILCO4PMS keeps the eager loop)
                     and isinstance(pol_, _Policy.Sum_of_gaussians) and getattr(pol_, "_unit_scale", False)
                     and hasattr(self.model_learning, "vel_indeces") and isinstance(self.cost_function, _Cost._HipExpectedCost))
        cap = dict(on=use_graph, graphs=[None, None], outs=[None, None], eager=0, rec=torch.zeros(2, abi.OPT_RECORD_DOUBLES, dtype=dt, device=dev),
                   one=torch.ones(1, dtype=dt, device=dev), last_flat=None)
        self.attempts_replayed = 0

        def drop_graphs():
            cap["graphs"], cap["outs"], cap["eager"] = [None, None], [None, None], 0
            self._call_dev = None

        def commit(cost, std, flags, status, grads, rec_row):
            cptr, sptr = abi.ptr(cost.detach().reshape(1)), abi.ptr(std.detach().reshape(1))
            if adam is not None:
                abi.check(lib.mcp_adam_step_guarded(len(ad["ps"]), ad["c_ps"], grads, ad["c_m"], ad["c_v"], ad["numel"], float(hs["lr"]), adam[1],
                                                    adam[2], adam[3], abi.ptr(st), 0, n_steps, cptr, abi.ptr(flags), abi.ptr(status), abi.stream()),
                          "mcp_adam_step_guarded")
            abi.check(lib.mcp_policy_step_commit(abi.ptr(st), n_steps, cptr, sptr, abi.ptr(flags), abi.ptr(status), abi.ptr(cost_list),
                                                 abi.ptr(std_list), abi.ptr(es1), abi.ptr(ratio), float(alpha_diff_cost),
                                                 float(min(hs["min_step"], 1e300)), float(hs["min_diff"]), int(num_min_diff_cost),
                                                 abi.ptr(rec_row), abi.stream()), "mcp_policy_step_commit")

        def attempt_body(rec_row):
            """The reference's lines (MC_PILCO.py:484-525) on the drop-in classes: apply_policy -> cost_function -> cost.backward() -> step."""
            for q in params:
                q.grad = None
            states, inputs = self.apply_policy(p_dropout=hs["p_drop"], **sim)
            cost, std, flags = self._cost_backward(states, inputs, trial_index, flags_as_vector=self.dist_group is not None)
            status = None if (flags is not None or self.last_status is None) else self.last_status
            grads = None if adam is None else (abi.dptr * len(ad["ps"]))(*[None if q.grad is None else q.grad.data_ptr() for q in ad["ps"]])
            commit(cost, std, flags, status, grads, rec_row)
            return states, inputs, cost, None

        def attempt_body_raw(rec_row):
            """The same attempt as the operators underneath make it, without the autograd engine (whose stream bookkeeping does not survive a
            stream capture): x0 -> mcp_rollout_fwd -> mcp_cost_fwd / _finalize / _bwd -> mcp_rollout_bwd -> guarded Adam -> commit.  Identical
            launches with identical arguments, hence identical bits; this is the form that is recorded and replayed."""
            self._call_dev.add_(1)
            world, rank = self._world()
            self._shard = sharding.shard_range(int(sim["num_particles"]), world, rank)
            self._m_total = int(sim["num_particles"])
            M, T = self._shard[1], int(sim["T_control"])
            x0 = self.sample_initial_particles(sim["particles_initial_state_mean"], sim["particles_initial_state_var"], sim["flg_particles_init_uniform"],
                                               sim["particles_init_up_bound"], sim["particles_init_low_bound"], sim["flg_particles_init_multi_gauss"],
                                               self._m_total)
            noise, p = self._rollout_noise(M, T, hs["p_drop"])
            model, pk = self.model_learning.packed(), pol_.packed()
            states, inputs, jac, status = ops.rollout_forward_raw(model, pk, noise, x0, T, p, True, need_jac=True, gp_sharding=self.gp_sharding)
            self.last_status = status
            cf = self.cost_function
            if hasattr(cf, "_select"):
                cf._select(states, trial_index)
            if cf._packed is None or cf._packed.device != states.device:
                cf._packed = cf._pack(states)
            cost, std, g_states = ops.expected_cost_raw(cf._packed, states, cap["one"])
            g_ls, g_c, g_w, _, g_b = ops.rollout_backward_raw(model, pk, noise, states, inputs, jac, g_states, None, p)
            by_param = {id(pol_.log_lengthscales): g_ls, id(pol_.centers): g_c, id(pol_.f_linear.weight): g_w}
            if pol_.f_linear.bias is not None:
                by_param[id(pol_.f_linear.bias)] = g_b
            grads = (abi.dptr * len(ad["ps"]))(*[by_param[id(q)].data_ptr() for q in ad["ps"]])
            commit(cost, std, None, status, grads, rec_row)
            return states, inputs, cost, by_param

        def enqueue():
            """One attempt, start to finish, without a host sync."""
            snap = (self._rollout_calls, torch.cuda.get_rng_state(dev) if depth > 0 else None)
            slot = seq[0] % (depth + 2)
            gi = seq[0] & 1
            seq[0] += 1
            rec_src = None
            if cap["on"] and cap["eager"] >= 2:
                if cap["graphs"][gi] is None:
                    # record: the kernels are not run here; the replay below is this attempt
                    if self._call_dev is None:
                        self._call_dev = torch.zeros(1, dtype=torch.int64, device=dev)
                    self._call_dev.fill_(self._rollout_calls)
                    torch.cuda.synchronize(dev)
                    g = torch.cuda.CUDAGraph()
                    try:
                        with torch.cuda.graph(g):
                            cap["outs"][gi] = attempt_body_raw(cap["rec"][gi])
                        cap["graphs"][gi] = g
                    except Exception as e:  # noqa: BLE001  (a runtime that cannot record this sequence: the eager loop is the same loop)
                        print("\nreinforce_policy: recording an attempt into a graph failed (%r) -- continuing with eager launches" % (e,))
                        cap["on"] = False
                        self._rollout_calls = snap[0]
                        if snap[1] is not None:
                            torch.cuda.set_rng_state(snap[1], dev)
                        drop_graphs()
                else:
                    self._rollout_calls += 1  # (the host's mirror of the counter the replay advances)
                if cap["graphs"][gi] is not None:
                    cap["graphs"][gi].replay()
                    self.attempts_replayed += 1
                    states, inputs, cost, cap["last_flat"] = cap["outs"][gi]
                    rec_src = cap["rec"][gi]
            if rec_src is None:
                cap["eager"] += 1
                states, inputs, cost, cap["last_flat"] = attempt_body(rec_dev[slot])
                rec_src = rec_dev[slot]
            ring[slot].copy_(rec_src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return dict(states=states, inputs=inputs, rec=ring[slot], ev=ev, snap=snap, cost=cost)

        def read(h):
            h["ev"].synchronize()
            return h["rec"].tolist()

        def step_print(k, cost_now, rabs):
            print("\nOptimization step: ", k)
            print("cost: ", cost_now)
            print("cost improvement: ", hs["prev_cost"] - cost_now)
            print("p_dropout_applied: ", hs["p_drop"])
            print("current_min_diff_cost; ", hs["min_diff"])
            print("current_min_step: ", hs["min_step"])
            print("diff_cost_ratio: ", rabs)
            print("time elapsed: ", time.time() - hs["t_mark"])
            hs["prev_cost"] = cost_now
            hs["t_mark"] = time.time()

        def lr_or_exit(k):
            """The condition of MC_PILCO.py:540-547 held at step k.  True: leave the loop."""
            if hs["lr"] > lr_min:
                print("Optimization_step:", k)
                print("\nREDUCING THE LEARNING RATE:")
                hs["lr"] = max(hs["lr"] * lr_reduction_ratio, lr_min)
                print("lr: ", hs["lr"])
                hs["min_diff"] = max(hs["min_diff"] / 2, 0.01)
                hs["min_step"] = k + num_min_diff_cost
                print("\nREDUCING THE DROPOUT:")
                hs["p_drop"] = max(hs["p_drop"] - p_drop_reduction, 0.0)
                print("p_dropout_applied: ", hs["p_drop"])
                return False
            print("\nEXIT FROM OPTIMIZATION: diff_cost_ratio < min_diff_cost for num_min_diff_cost steps")
            return True

        def discard(queue):
            """The attempts enqueued while the device was waiting for the host: the device ignored them; the noise counters go back
            to where the first of them found them, so the run continues exactly as one that never enqueued them."""
            if queue:
                for h in queue:
                    r = read(h)
                    assert r[1] == 1.0 and r[0] == 0.0, "an attempt enqueued past a host decision was not void"
                self._rollout_calls = queue[0]["snap"][0]
                if self._call_dev is not None:
                    self._call_dev.fill_(self._rollout_calls)
                if queue[0]["snap"][1] is not None:
                    torch.cuda.set_rng_state(queue[0]["snap"][1], dev)
                queue.clear()

        hs["t_mark"] = time.time()
        queue, last, done, reinits, leave = [], None, 0, 0, False
        while not leave:
            queue.append(enqueue())
            while queue and (len(queue) > depth) and not leave:
                h = queue.pop(0)
                counted, void, k, failed, pending, cost_now, _std, rabs, nan, sync, nonpos, _tot = read(h)
                k = int(k)
                assert void == 0.0, "the oldest attempt in flight cannot be void"
                if counted == 0.0:
                    self._judge_attempt(nan, sync, nonpos)  # (raises on a non-positive variance; switches GP sharding off after a time-out)
                    print("\nCost is NaN: try sampling again")
                    last = h
                    if failed >= abi.OPT_MAX_ATTEMPTS:
                        # ten failed attempts in a row (MC_PILCO.py:573-607): the reference takes the step on the failed cost (its monitors
                        # and messages included) and restarts from a re-initialised policy
                        discard(queue)
                        if k % num_step_print == 0:
                            step_print(k, cost_now, float("nan"))
                        if k > hs["min_step"]:  # (its lr / exit test looks at the window BEFORE this step's ratio: it may still fire; only the
                            win = torch.abs(ratio[max(k + 1 - num_min_diff_cost, 0):k + 1])  # messages matter, everything is reset below)
                            if int(torch.sum(win < hs["min_diff"])) >= num_min_diff_cost and k + 1 >= num_min_diff_cost:
                                lr_or_exit(k)
                        reinits += 1
                        print("\nCost is NaN: re-initialize control policy [attempt #" + str(reinits) + "]")
                        self.control_policy.reinit(**policy_reinit_dict)
                        st[0:5].zero_()  # (ES2 and cost_tm1 are NOT reset by the reference: they keep the failed step's values)
                        for a in (cost_list, std_list, es1, ratio):
                            a.zero_()
                        hs.update(lr=lr_list[trial_index], p_drop=p_drop0, min_diff=min_diff_cost, min_step=min_step, prev_cost=0.0)
                        opt = make_opt(p=self.control_policy.parameters(), lr=hs["lr"])
                        adam = self._plain_adam(opt)
                        params = [q for q in self.control_policy.parameters()]
                        fresh_adam_state()
                        drop_graphs()  # (new moments / learning rate / dropout / parameters: what the graphs recorded is gone)
                        done = 0
                    continue
                # the attempt counted: step k was taken
                last, done = h, k + 1
                if adam is None:
                    opt.step()  # (depth 0: the host knows the attempt counted before it updates)
                if k % num_step_print == 0:
                    step_print(k, cost_now, rabs)
                if pending != 0.0:
                    discard(queue)
                    leave = lr_or_exit(k)
                    if not leave:
                        opt = make_opt(p=self.control_policy.parameters(), lr=hs["lr"])
                        adam_now = self._plain_adam(opt)
                        if (adam_now is None) != (adam is None):
                            raise RuntimeError("f_optimizer must build the same kind of optimizer on every call")
                        adam = adam_now
                        fresh_adam_state()
                        drop_graphs()  # (new moments / learning rate / dropout / parameters: what the graphs recorded is gone)
                    st[2:3].zero_()
                if done >= n_steps:
                    discard(queue)
                    leave = True
        discard(queue)
        states, inputs = last["states"], last["inputs"]
        if cap["last_flat"] is not None:  # (the last attempt was a replay: its gradients are where autograd would have left them)
            for q in params:
                g_ = cap["last_flat"].get(id(q))
                q.grad = None if g_ is None else g_.reshape(q.shape)
        drop_graphs()  # (rollouts after this call count by value again; the host's mirror of the counter is current)
        return (cost_list[0:done].detach().cpu().numpy(), std_list[0:done].detach().cpu().numpy(), states.detach().cpu().numpy(),
                inputs.detach().cpu().numpy())

    # ------------------------------------------------------------------------------------------------------------
    # trial loop
    # ------------------------------------------------------------------------------------------------------------
    def _draw_x0(self, initial_state, initial_state_var, random_initial_state, flg_uniform, low, up, flg_multi):
        if not random_initial_state:
            return initial_state
        if flg_uniform:
            return np.random.uniform(low, up)
        if flg_multi:
            k = np.random.randint(initial_state.shape[0])
            return np.random.normal(initial_state[k, :], np.sqrt(initial_state_var[k, :]))
        return np.random.normal(initial_state, np.sqrt(initial_state_var))

    def _save_log(self):
        if self.log_path is not None:
            print("Save log file...")
            pkl.dump(self.log_dict, open(self.log_path + "/log.pkl", "wb"))

    def reinforce(self, initial_state, initial_state_var, T_exploration, T_control, num_trials, model_optimization_opt_list,
                  policy_optimization_dict, num_explorations=1, flg_init_uniform=False, init_up_bound=None, init_low_bound=None,
                  flg_init_multi_gauss=False, random_initial_state=True, loaded_model=False):
        """Alternates model learning, policy optimisation on the learned model, and interaction with the system."""
        x0_args = (initial_state, initial_state_var, random_initial_state, flg_init_uniform, init_low_bound, init_up_bound, flg_init_multi_gauss)
        if not loaded_model:
            print("\n\n\n\n----------------- INITIAL EXPLORATIONS -----------------")
            for k in range(num_explorations):
                print("\nEXPLORATION # " + str(k))
                self.get_data_from_system(initial_state=self._draw_x0(*x0_args), T_exploration=T_exploration, flg_exploration=True, trial_index=k)
            costs, stds, params, pstates, pinputs = [], [], [], [], []
            first = num_explorations - 1
        else:
            costs, stds = self.log_dict["cost_trial_list"], self.log_dict["std_cost_trial_list"]
            params, pstates, pinputs = (self.log_dict["parameters_trial_list"], self.log_dict["particles_states_list"],
                                        self.log_dict["particles_inputs_list"])
            first = len(self.state_samples_history) - 1
        t_dev = lambda a: torch.tensor(a, dtype=self.dtype, device=self.device)
        for trial in range(first, first + num_trials):
            print("\n\n\n\n----------------- TRIAL " + str(trial) + " -----------------")
            print("\n\n----- REINFORCE THE MODEL -----")
            self.model_learning.reinforce_model(optimization_opt_list=model_optimization_opt_list)
            with torch.no_grad():
                if self.log_path is not None:
                    ml = self.model_learning
                    self.log_dict["parameters_gp_" + str(trial)] = [copy.deepcopy(ml.gp_list[k].state_dict()) for k in range(ml.num_gp)]
                    self.log_dict["gp_inputs_" + str(trial)] = ml.gp_inputs
                    self.log_dict["gp_output_list_" + str(trial)] = ml.gp_output_list
                    self.log_dict["state_samples_history"] = self.state_samples_history
                    self.log_dict["input_samples_history"] = self.input_samples_history
                    self.log_dict["noiseless_states_history"] = self.noiseless_states_history
                    self._save_log()
                print("\n\n----- CHECK THE ROLLOUT PERFORMANCE (after model update) -----")
                self.get_rollout_prediction_performance(data_collection_index=trial)
            print("\n\n----- REINFORCE THE POLICY -----")
            self.model_learning.set_eval_mode()
            cost_list, std_list, p_states, p_inputs = self.reinforce_policy(
                T_control=T_control, particles_initial_state_mean=t_dev(initial_state), particles_initial_state_var=t_dev(initial_state_var),
                flg_particles_init_uniform=flg_init_uniform, particles_init_up_bound=t_dev(init_up_bound) if flg_init_uniform else None,
                particles_init_low_bound=t_dev(init_low_bound) if flg_init_uniform else None,
                flg_particles_init_multi_gauss=flg_init_multi_gauss, trial_index=trial, **policy_optimization_dict)
            costs.append(cost_list)
            stds.append(std_list)
            pstates.append(p_states)
            pinputs.append(p_inputs)
            params.append(copy.deepcopy(self.control_policy.state_dict()))
            if self.log_path is not None:
                self.log_dict.update(cost_trial_list=costs, std_cost_trial_list=stds, parameters_trial_list=params, particles_states_list=pstates,
                                     particles_inputs_list=pinputs)
                self._save_log()
            self.model_learning.set_training_mode()
            print("\n\n----- APPLY THE CONTROL POLICY -----")
            self.get_data_from_system(initial_state=self._draw_x0(*x0_args), T_exploration=T_control, flg_exploration=False, trial_index=trial + 1)
            if self.log_path is not None:
                self.log_dict["state_samples_history"] = self.state_samples_history
                self.log_dict["input_samples_history"] = self.input_samples_history
                self.log_dict["noiseless_states_history"] = self.noiseless_states_history
                self._save_log()
        return costs, pstates, pinputs

    # ------------------------------------------------------------------------------------------------------------
    # interaction with the system, mean rollouts, logs
    # ------------------------------------------------------------------------------------------------------------
    def get_data_from_system(self, initial_state, T_exploration, trial_index, flg_exploration=False):
        policy = self.rand_exploration_policy if flg_exploration else self.control_policy
        noisy, inputs, clean = self.system.rollout(s0=initial_state, policy=policy.get_np_policy(), T=T_exploration, dt=self.T_sampling,
                                                   noise=self.std_meas_noise)
        self.state_samples_history.append(noisy)
        self.input_samples_history.append(inputs)
        self.noiseless_states_history.append(clean)
        self.num_data_collection += 1
        self.model_learning.add_data(new_state_samples=noisy, new_input_samples=inputs)

    def rollout(self, data_collection_index, T_rollout=None, particle_pred=False):
        """Open-loop prediction of one recorded trajectory with the learned model (mean prediction by default)."""
        xs = self.state_samples_history[data_collection_index]
        us = torch.tensor(self.input_samples_history[data_collection_index], dtype=self.dtype, device=self.device)
        n = xs.shape[0] if T_rollout is None else T_rollout
        traj = torch.zeros([n, self.state_dim], dtype=self.dtype, device=self.device)
        traj[0:1, :] = torch.tensor(xs[0:1, :], dtype=self.dtype, device=self.device)
        for t in range(1, n):
            traj[t:t + 1, :], _, _ = self.model_learning.get_next_state(current_state=traj[t - 1:t, :], current_input=us[t - 1:t, :],
                                                                         particle_pred=particle_pred)
        return traj.detach().cpu().numpy()

    def get_model_learning_performance(self, data_collection_index, flg_pretrain=False):
        """One-step GP predictions on the data of one interaction with the system (MC_PILCO.py:260-306): prints the MSE per GP,
        returns (gp_inputs, targets [numpy per GP], means [numpy per GP], variances [tensors, scaled by norm_list^2])."""
        ml = self.model_learning
        t_dev = lambda a: torch.tensor(a, dtype=self.dtype, device=self.device)
        gp_inputs, targets, means, variances = ml.get_gp_estimate_from_data(states=t_dev(self.state_samples_history[data_collection_index]),
                                                                           inputs=t_dev(self.input_samples_history[data_collection_index]),
                                                                           flg_pretrain=flg_pretrain)
        variances = [variances[i] * ml.norm_list[i] ** 2 for i in range(ml.num_gp)]
        targets = [targets[i].detach().cpu().numpy() for i in range(ml.num_gp)]
        means = [means[i].detach().cpu().numpy() for i in range(ml.num_gp)]
        for i in range(ml.num_gp):
            print("MSE gp" + str(i) + ": ", ((targets[i] - means[i]) ** 2).mean())
        return gp_inputs, targets, means, variances

    def get_rollout_prediction_performance(self, data_collection_index, T_rollout=None, add_name="", particle_pred=False):
        """Open-loop rollout of the learned model on the inputs of one interaction (MC_PILCO.py:308-345); ``particle_pred``:
        sampled instead of mean predictions, as in the reference."""
        pred = self.rollout(data_collection_index, T_rollout=T_rollout, particle_pred=particle_pred)
        obs = self.state_samples_history[data_collection_index][: pred.shape[0]]
        print("Rollout prediction MSE per state:", np.mean((pred - obs) ** 2, 0))
        return pred, obs, self.input_samples_history[data_collection_index]

    def load_policy_from_log(self, num_trial, folder="results_tmp/1/"):
        log = pkl.load(open(folder + "log.pkl", "rb"))
        self.control_policy.load_state_dict(log["parameters_trial_list"][num_trial - 1])

    def load_model_from_log(self, num_trial, folder="results_tmp/1/"):
        """Replays the logged data into the model, restores the GP hyper-parameters of trial ``num_trial-1`` and pretrains."""
        log = pkl.load(open(folder + "log.pkl", "rb"))
        self.log_dict = log
        for k in ("cost_trial_list", "parameters_trial_list", "particles_states_list", "particles_inputs_list"):
            self.log_dict[k] = self.log_dict[k][0:num_trial]
        for j in range(num_trial + 1):
            self.state_samples_history.append(log["state_samples_history"][j])
            self.input_samples_history.append(log["input_samples_history"][j])
            self.noiseless_states_history.append(log["noiseless_states_history"][j])
            self.num_data_collection += 1
            self.model_learning.add_data(new_state_samples=log["state_samples_history"][j], new_input_samples=log["input_samples_history"][j])
        t = num_trial - 1
        ml = self.model_learning
        ml.gp_inputs = log["gp_inputs_" + str(t)].to(self.device)
        ml.gp_output_list = [y.to(self.device) for y in log["gp_output_list_" + str(t)]]
        for k in range(ml.num_gp):
            ml.gp_list[k].load_state_dict(log["parameters_gp_" + str(t)][k])
        with torch.no_grad():
            for k in range(ml.num_gp):
                ml.pretrain_gp(k)


class MC_PILCO4PMS(MC_PILCO):
    """MC-PILCO for partially measurable systems -- drop-in for ``MC_PILCO4PMS`` (MC_PILCO.py:755-958).

    Particles evolve on their true states; the policy is evaluated on a simulated *measurement*: positions plus Gaussian
    noise, velocities by backward difference of the noisy positions, smoothed online by a first-order Butterworth filter
    (``filtering_dict["fc"]``).  With the speed-integration models and the RBF policies the rollout is the same single fused
    launch as ``MC_PILCO.apply_policy``: the kernels carry the filter's states per particle and the reverse sweep its adjoint
    recursion (``mcp_meas``).  ``fused = False`` (or any other model / policy object) runs step by step on the posterior and
    policy operators with the filter as torch device ops.
    """

    def __init__(self, T_sampling, state_dim, input_dim, f_sim, f_model_learning, model_learning_par, f_rand_exploration_policy,
                 rand_exploration_policy_par, f_control_policy, control_policy_par, f_cost_function, cost_function_par, pos_indeces,
                 vel_indeces, std_meas_noise=None, log_path=None, filtering_dict={}, std_meas_noise_sim=None, dtype=torch.float64,
                 device=torch.device("cuda")):
        super().__init__(T_sampling=T_sampling, state_dim=state_dim, input_dim=input_dim, f_sim=f_sim, f_model_learning=f_model_learning,
                         model_learning_par=model_learning_par, f_rand_exploration_policy=f_rand_exploration_policy,
                         rand_exploration_policy_par=rand_exploration_policy_par, f_control_policy=f_control_policy,
                         control_policy_par=control_policy_par, f_cost_function=f_cost_function, cost_function_par=cost_function_par,
                         std_meas_noise=std_meas_noise, log_path=log_path, dtype=dtype, device=device)
        self.system = _sim.PMS_Model(f_sim, filtering_dict)
        self.filtering_dict = filtering_dict
        self.pos_indeces = pos_indeces
        self.vel_indeces = vel_indeces
        # (the reference leaves the attribute unset when a value is passed, MC_PILCO.py:802-803; here it is always defined)
        self.std_meas_noise_sim = std_meas_noise if std_meas_noise_sim is None else std_meas_noise_sim
        self.fused = True  # False: step-by-step rollout on the posterior / policy operators (any model or policy object)

    def apply_policy(self, particles_initial_state_mean, particles_initial_state_var, flg_particles_init_uniform, particles_init_up_bound,
                     particles_init_low_bound, flg_particles_init_multi_gauss, num_particles, T_control, p_dropout=0.0):
        from scipy import signal

        world, rank = self._world()
        self._shard = sharding.shard_range(int(num_particles), world, rank)
        self._m_total = int(num_particles)
        M, T = self._shard[1], int(T_control)
        pol, ml = self.control_policy, self.model_learning
        ref = self.noise_mode == "reference"  # draw on the CPU with the reference's calls, in its order
        ndev = torch.device("cpu") if ref else self.device
        x = self._shard_slice(self.sample_initial_particles(particles_initial_state_mean, particles_initial_state_var, flg_particles_init_uniform,
                                                            particles_init_up_bound, particles_init_low_bound, flg_particles_init_multi_gauss,
                                                            self._m_total), 0)
        b, a = signal.butter(1, self.filtering_dict["fc"])
        pos, vel = list(self.pos_indeces), list(self.vel_indeces)
        if self.fused and isinstance(pol, _Policy.Sum_of_gaussians) and hasattr(ml, "vel_indeces"):
            # one fused launch: the kernels carry the measurement filter's states per particle (mcp_meas)
            p = float(p_dropout) if getattr(pol, "flg_drop", True) else 0.0
            G, B = ml.num_gp, pol.num_basis
            self._rollout_calls += 1
            if ref:  # the reference's draw order: mask_0; per step: eps_t, position noise, mask_t (whole swarm, then this rank's slice)
                Mt = self._m_total
                masks = [torch.empty(Mt, 1, B, dtype=self.dtype).bernoulli_(1 - p).reshape(Mt, B)] if p > 0 else None
                eps, pn = [], []
                for _ in range(1, T):
                    eps.append(torch.empty(Mt, G, dtype=self.dtype).normal_())
                    pn.append(torch.randn(Mt, len(pos), dtype=self.dtype))
                    if p > 0:
                        masks.append(torch.empty(Mt, 1, B, dtype=self.dtype).bernoulli_(1 - p).reshape(Mt, B))
                stack = lambda l, w: self._shard_slice(torch.stack(l) if l else torch.zeros(0, Mt, w, dtype=self.dtype), 1).to(self.device).contiguous()
                noise = ops.NoiseSpec(eps=stack(eps, G), masks=None if masks is None else
                                      self._shard_slice(torch.stack(masks).to(torch.uint8), 1).to(self.device).contiguous())
                pos_noise = stack(pn, len(pos))
            else:
                noise = self._philox_noise()
                pos_noise = None
            meas = ops.MeasSpec(pos=pos, vel=vel, std_pos=[float(v) for v in np.asarray(self.std_meas_noise_sim)[pos]], b=b, a=a,
                                pos_noise=pos_noise)
            states, inputs, status = ops.rollout(ml.packed(), pol.packed(), noise, x, T, p, meas=meas, gp_sharding=self.gp_sharding)
            self.last_status = status
            return states, inputs
        self.last_status = None  # (no fused launch)
        if world > 1:  # (per-LOCAL-particle torch draws: identically seeded ranks would simulate correlated shards)
            raise NotImplementedError("particle sharding needs the fused rollout (fused=True, Sum_of_gaussians policy, speed-integration model)")
        std_pos = torch.tensor(np.asarray(self.std_meas_noise_sim)[pos], dtype=self.dtype, device=self.device)
        saved_mode = getattr(pol, "noise_mode", None)
        if ref and saved_mode is not None:
            pol.noise_mode = "torch_cpu"
        try:
            xs = [x]
            noisy_prev, meas_prev = x, x
            us = [pol(x, t=0, p_dropout=p_dropout)]
            for t in range(1, T):
                _, _, mean_list, var_list = ml.get_one_step_gp_out(states=xs[-1], inputs=us[-1])
                var_list = [v * ml.norm_list[i] ** 2 for i, v in enumerate(var_list)]
                mean, var = torch.cat(mean_list, 1), torch.cat(var_list, 1)
                eps = torch.empty(M, ml.num_gp, dtype=self.dtype, device=ndev).normal_().to(self.device)
                delta = mean + torch.sqrt(var) * eps
                x, _, _ = ml.get_next_state_from_gp_output(current_state=xs[-1], current_input=us[-1],
                                                           gp_output_mean_list=[delta[:, g:g + 1] for g in range(ml.num_gp)],
                                                           gp_output_var_list=var_list, particle_pred=False)
                xs.append(x)
                n = torch.randn(M, len(pos), dtype=self.dtype, device=ndev).to(self.device)
                noisy = x.clone()
                noisy[:, pos] = noisy[:, pos] + std_pos * n
                noisy[:, vel] = (noisy[:, pos] - noisy_prev[:, pos]) / self.T_sampling
                meas = noisy.clone()
                meas[:, vel] = (b[0] * noisy[:, vel] + b[1] * noisy_prev[:, vel] - a[1] * meas_prev[:, vel]) / a[0]
                us.append(pol(meas, t=t, p_dropout=p_dropout))
                noisy_prev, meas_prev = noisy, meas
        finally:
            if saved_mode is not None:
                pol.noise_mode = saved_mode
        return torch.stack(xs), torch.stack(us)

    def get_data_from_system(self, initial_state, T_exploration, trial_index, flg_exploration=False):
        policy = self.rand_exploration_policy if flg_exploration else self.control_policy
        meas, inputs, clean, noisy = self.system.rollout(s0=initial_state, policy=policy.get_np_policy(), T=T_exploration, dt=self.T_sampling,
                                                         noise=self.std_meas_noise, vel_indeces=self.vel_indeces, pos_indeces=self.pos_indeces)
        states, meas, inputs, clean, noisy = self.get_velocities(meas, inputs, clean, noisy)
        self.state_samples_history.append(states)
        self.input_samples_history.append(inputs)
        self.noiseless_states_history.append(clean)
        self.num_data_collection += 1
        self.model_learning.add_data(new_state_samples=states, new_input_samples=inputs)

    def get_velocities(self, meas_states, input_samples, noiseless_samples, noisy_samples):
        """Offline filtering of the collected data for model learning (MC_PILCO.py:938-958): zero-phase second-order Butterworth
        on the positions, central-difference velocities, first and last sample dropped."""
        from scipy import signal

        states = np.zeros([noisy_samples.shape[0] - 2, noisy_samples.shape[1]])
        bb, aa = signal.butter(2, 0.5)
        for i in range(len(self.pos_indeces)):
            pos = signal.filtfilt(bb, aa, noisy_samples[:, self.pos_indeces[i]])
            states[:, self.pos_indeces[i]] = pos[1:-1]
            states[:, self.vel_indeces[i]] = (pos[2:] - pos[:-2]) / (2 * self.T_sampling)
        return states, meas_states[1:-1, :], input_samples[1:-1, :], noiseless_samples[1:-1, :], noisy_samples[1:-1, :]
