"""Host-side operators over the C ABI (hipabi): descriptor packing, workspaces and the
torch.autograd.Function wrappers that make the HIP kernels differentiable from Python.

Everything here is plumbing around ``libmcpilco_hip.so``; no arithmetic of the hot path is
done in PyTorch.  All tensors are float64 on the GPU.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import hipabi as abi

DT = torch.float64


def _dev(device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("mc_pilco_amd runs on the GPU only (device=%s requested); there is no CPU path" % device)
    return device


def _t(x, device):
    if isinstance(x, torch.Tensor):
        return x.detach().to(device=device, dtype=DT).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float64)).to(device).contiguous()


def pad16(n):
    return (int(n) + 15) // 16 * 16


# --------------------------------------------------------------------------------------
# kernel hyper-parameters -> mcp_kernel
# --------------------------------------------------------------------------------------
@dataclass
class KernelSpec:
    """SE (+ Volterra polynomial) covariance in the library's parametrisation.

    ``w1`` [D+1], ``w20``/``w21`` [D] are the squared diagonal weights the reference's
    ``MPK_GP.get_Sigma`` builds (Sparse_GP.py:613-623): for MPK_k, factor d has
    s_d = (k-d)*exp(par[d*n:(d+1)*n]) and weight s_d**2.
    """

    lengthscales: torch.Tensor  # [D]
    lam: float
    sigma_n2: float
    mean: float = 0.0
    w1: Optional[torch.Tensor] = None
    w20: Optional[torch.Tensor] = None
    w21: Optional[torch.Tensor] = None
    scal: Optional[torch.Tensor] = None  # device [lambda, sigma_n2, mean]: overrides the three floats (no host round trip to fill them)
    _keep: dict = field(default_factory=dict, repr=False, compare=False)

    @property
    def D(self):
        return int(self.lengthscales.numel())

    @property
    def poly_deg(self):
        return 0 if self.w1 is None else (1 if self.w20 is None else 2)

    def _device_operands(self, device):
        """Device copies of the hyper-parameter vectors, uploaded ONCE per device and kept for the life of the spec: every
        descriptor built from this spec (mcp_kernel inside mcp_gp inside mcp_model) points at the same tensors, so a later
        to_c() can never free memory an older descriptor still refers to."""
        key = (device.type, device.index)
        ops_ = self._keep.get(key)
        if ops_ is None:
            ops_ = {"inv_ls": (1.0 / _t(self.lengthscales, device)).contiguous()}
            for name in ("w1", "w20", "w21"):
                v = getattr(self, name)
                ops_[name] = None if v is None else _t(v, device)
            self._keep[key] = ops_
        return ops_

    def to_c(self, device):
        device = _dev(device)
        ops_ = self._device_operands(device)
        k = abi.Kernel()
        k.D = self.D
        k.poly_deg = self.poly_deg
        k.lam = float(self.lam)
        k.sigma_n2 = float(self.sigma_n2)
        k.mean = float(self.mean)
        k.inv_ls = ops_["inv_ls"].data_ptr()
        for name in ("w1", "w20", "w21"):
            setattr(k, name, None if ops_[name] is None else ops_[name].data_ptr())
        k.scal = None if self.scal is None else self.scal.data_ptr()
        return k


def mpk_weights(log_par, k):
    """Squared diagonal weights of MPK_k from its raw log parameters (list over the k factors)."""
    par = torch.as_tensor(log_par, dtype=DT).reshape(-1)
    n = par.numel() // k
    return [((k - d) * torch.exp(par[d * n:(d + 1) * n])) ** 2 for d in range(k)]


# --------------------------------------------------------------------------------------
# pretrain primitives
# --------------------------------------------------------------------------------------
def cov_build(spec: KernelSpec, X1, X2=None, noise=False):
    X1 = _t(X1, X1.device if isinstance(X1, torch.Tensor) else None)
    dev = X1.device
    X2t = X1 if X2 is None else _t(X2, dev)
    K = torch.empty(X1.shape[0], X2t.shape[0], dtype=DT, device=dev)
    kc = spec.to_c(dev)
    abi.check(abi.lib().mcp_cov_build(C.byref(kc), X1.shape[0], abi.ptr(X1), X2t.shape[0], abi.ptr(X2t), int(bool(noise)), abi.ptr(K),
                                      K.shape[1], abi.stream()), "mcp_cov_build")
    return K


def cov_diag(spec: KernelSpec, X, noise=False):
    X = _t(X, X.device)
    d = torch.empty(X.shape[0], dtype=DT, device=X.device)
    kc = spec.to_c(X.device)
    abi.check(abi.lib().mcp_cov_diag(C.byref(kc), X.shape[0], abi.ptr(X), int(bool(noise)), abi.ptr(d), abi.stream()), "mcp_cov_diag")
    return d


def chol_factor(K):
    """Returns (U upper with K=U^T U, logdet, status word tensor).  K is not modified."""
    U = K.detach().clone().contiguous()
    N = U.shape[0]
    logdet = torch.zeros(1, dtype=DT, device=U.device)
    status = torch.zeros(1, dtype=torch.int32, device=U.device)
    abi.check(abi.lib().mcp_chol_factor_ex(N, abi.ptr(U), U.shape[1], abi.ptr(logdet), abi.ptr(status), abi.stream(), C.byref(abi.DISPATCH)),
              "mcp_chol_factor")
    return U, logdet[0], status


def chol_inverse(U):
    N = U.shape[0]
    Ui = torch.zeros(N, N, dtype=DT, device=U.device)
    Kinv = torch.empty(N, N, dtype=DT, device=U.device)
    abi.check(abi.lib().mcp_chol_inverse_ex(N, abi.ptr(U), U.shape[1], abi.ptr(Ui), N, abi.ptr(Kinv), N, abi.stream(), C.byref(abi.DISPATCH)),
              "mcp_chol_inverse")
    return Ui, Kinv


def sym_sandwich(A, G):
    """A G A for a symmetric A (K^-1) and any G, on the library's MFMA GEMM (mcp_sym_sandwich)."""
    N = A.shape[0]
    A = A.detach().to(dtype=DT).contiguous()
    G = G.detach().to(device=A.device, dtype=DT).contiguous()
    out = torch.empty(N, N, dtype=DT, device=A.device)
    scratch = torch.empty(N, N, dtype=DT, device=A.device)
    abi.check(abi.lib().mcp_sym_sandwich(N, abi.ptr(A), N, abi.ptr(G), N, abi.ptr(out), N, abi.ptr(scratch), abi.stream()), "mcp_sym_sandwich")
    return out


def gp_alpha(Kinv, Y, mean=0.0):
    N = Kinv.shape[0]
    Y = _t(Y, Kinv.device).reshape(-1)
    alpha = torch.empty(N, dtype=DT, device=Kinv.device)
    abi.check(abi.lib().mcp_gp_alpha(N, abi.ptr(Kinv), Kinv.shape[1], abi.ptr(Y), float(mean), abi.ptr(alpha), abi.stream()), "mcp_gp_alpha")
    return alpha.reshape(-1, 1)


sod_fallbacks = 0  # calls of sod_select whose multi-workgroup launch reported "never met" and were repeated on one workgroup


def sod_select(spec: KernelSpec, X, threshold, one_workgroup=False) -> List[int]:
    """GP_prior.get_SOD (GP_prior.py:232-257) on the device.  From 256 candidates on the selection runs across workgroups (one per 64 candidates)
    when the workspace has room for their exchange -- `mcp_sod_workspace_bytes` says how much; ``one_workgroup`` passes the first part only (W and
    the running sums) and so keeps the one-workgroup kernel (tests, timing)."""
    X = _t(X, X.device)
    N = X.shape[0]
    dev = X.device
    nbytes = 8 * (N * N + 2 * N) if one_workgroup else abi.lib().mcp_sod_workspace_bytes(N)
    ws = torch.empty((nbytes + 7) // 8, dtype=DT, device=dev)
    idx = torch.zeros(N, dtype=torch.int32, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    kc = spec.to_c(dev)
    abi.check(abi.lib().mcp_sod_select(C.byref(kc), N, abi.ptr(X), float(threshold), abi.ptr(idx), abi.ptr(n), abi.ptr(ws), nbytes,
                                       abi.stream()), "mcp_sod_select")
    cnt = int(n.item())
    if cnt < 0:  # the workgroups never met (the device could not hold the grid, e.g. shared with another process): the one-workgroup kernel
        if one_workgroup:
            raise RuntimeError("mcp_sod_select reported %d kept rows" % cnt)
        global sod_fallbacks
        sod_fallbacks += 1
        if sod_fallbacks == 1:  # (every such call has spun to its limit first -- seconds: say so once, keep counting)
            import warnings

            warnings.warn("mcp_sod_select: the workgroups of the multi-workgroup selection never met (is the GPU shared with another process?); "
                          "this and later such calls are repeated on the one-workgroup kernel (ops.sod_fallbacks counts them)", RuntimeWarning)
        return sod_select(spec, X, threshold, one_workgroup=True)
    return [int(i) for i in idx[:cnt].tolist()]


# --------------------------------------------------------------------------------------
# packed GP / model descriptors
# --------------------------------------------------------------------------------------
class PackedGP:
    """Device-resident operands of one pretrained GP in the kernels' layout (mcp_gp)."""

    def __init__(self, spec: KernelSpec, X, alpha, Kinv):
        dev = _dev(Kinv.device)
        X = _t(X, dev)
        alpha = _t(alpha, dev).reshape(-1)
        Kinv = _t(Kinv, dev)
        N, D = X.shape
        self.N, self.D, self.Npad = N, D, pad16(N)
        self.spec = spec
        self.Xt = torch.empty(D, self.Npad, dtype=DT, device=dev)
        self.X = torch.empty(self.Npad, D, dtype=DT, device=dev)
        self.alpha = torch.empty(self.Npad, dtype=DT, device=dev)
        self.Kinv = torch.empty(self.Npad, self.Npad, dtype=DT, device=dev)
        self.aX = torch.empty(D, dtype=DT, device=dev)
        abi.check(abi.lib().mcp_gp_pack(N, D, abi.ptr(X), abi.ptr(alpha), abi.ptr(Kinv), Kinv.shape[1], self.Npad, abi.ptr(self.Xt),
                                        abi.ptr(self.X), abi.ptr(self.alpha), abi.ptr(self.Kinv), abi.ptr(self.aX), abi.stream()),
                  "mcp_gp_pack")
        self.device = dev

    def fill(self, g: abi.GP):
        g.kern = self.spec.to_c(self.device)
        g.N, g.Npad = self.N, self.Npad
        g.Xt, g.X, g.alpha, g.Kinv, g.aX = (self.Xt.data_ptr(), self.X.data_ptr(), self.alpha.data_ptr(), self.Kinv.data_ptr(),
                                            self.aX.data_ptr())

    def to_c(self):
        g = abi.GP()
        self.fill(g)
        return g


class PackedModel:
    """mcp_model: the speed-integration dynamics model with its G packed GPs."""

    def __init__(self, gps: Sequence[PackedGP], S, U, Ts, angle, not_angle, vel, not_vel, var_scale=None):
        self.gps = list(gps)
        m = abi.Model()
        m.S, m.U, m.G, m.D = int(S), int(U), len(self.gps), self.gps[0].D
        m.n_angle, m.n_not_angle = len(angle), len(not_angle)
        for i, v in enumerate(angle):
            m.angle[i] = int(v)
        for i, v in enumerate(not_angle):
            m.not_angle[i] = int(v)
        for i, v in enumerate(vel):
            m.vel[i] = int(v)
        for i, v in enumerate(not_vel):
            m.not_vel[i] = int(v)
        m.Ts = float(Ts)
        for g in range(abi.MAX_GP):
            m.var_scale[g] = 1.0 if var_scale is None or g >= len(var_scale) else float(var_scale[g])
        for g, pg in enumerate(self.gps):
            pg.fill(m.gp[g])
        self.c = m
        self.S, self.U, self.G, self.D = m.S, m.U, m.G, m.D
        self.device = self.gps[0].device


class PackedPolicy:
    """mcp_policy over the live parameter tensors (no copies: the optimizer's updates are seen)."""

    def __init__(self, kind, S, log_ls, centers, weight, u_max, squash=True, angle=(), non_angle=(), target_traj=None, bias=None):
        dev = _dev(centers.device)
        self.log_ls, self.centers, self.weight = log_ls, centers, weight
        self.bias = bias  # [U] f_linear.bias (flg_bias) or None
        B, P = centers.shape
        U = weight.shape[0]
        um = np.full(U, float(u_max)) if np.isscalar(u_max) else np.asarray(u_max, dtype=np.float64).reshape(-1)
        self.u_max = _t(um, dev)
        self.target_traj = None if target_traj is None else _t(target_traj, dev)
        p = abi.Policy()
        p.kind = {"plain": abi.POLICY_PLAIN, "angles": abi.POLICY_ANGLES, "traj": abi.POLICY_TRAJ}[kind]
        p.S, p.P, p.B, p.U, p.squash = int(S), int(P), int(B), int(U), int(bool(squash))
        p.n_angle, p.n_non_angle = len(angle), len(non_angle)
        for i, v in enumerate(angle):
            p.angle[i] = int(v)
        for i, v in enumerate(non_angle):
            p.non_angle[i] = int(v)
        p.traj_len = 0 if self.target_traj is None else int(self.target_traj.shape[0])
        p.u_max = self.u_max.data_ptr()
        p.target_traj = None if self.target_traj is None else self.target_traj.data_ptr()
        self.c = p
        self.kind, self.S, self.P, self.B, self.U = kind, int(S), int(P), int(B), int(U)
        self.device = dev
        self.grad_flat = None  # optional caller-owned flat fp64 buffer the adjoint sweep writes its gradients into (rollout_backward_raw)

    def grad_numel(self):
        """Doubles of the flat gradient [log_ls | centers | weight | bias]."""
        return self.P + self.B * self.P + self.U * self.B + (self.U if self.bias is not None else 0)

    def bind(self, p_drop):
        """Refreshes parameter pointers (they must be contiguous fp64 GPU tensors) and p_drop."""
        for t in (self.log_ls, self.centers, self.weight) + (() if self.bias is None else (self.bias,)):
            if t.dtype != DT or not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("policy parameters must be contiguous float64 GPU tensors")
        if self.bias is not None and self.bias.numel() != self.U:
            raise RuntimeError("policy bias must have one entry per input")
        self.c.bias = None if self.bias is None else self.bias.data_ptr()
        self.c.g_bias = None
        self.c.log_ls = self.log_ls.data_ptr()
        self.c.centers = self.centers.data_ptr()
        self.c.weight = self.weight.data_ptr()
        self.c.p_drop = float(p_drop)
        return self.c


@dataclass
class NoiseSpec:
    """Parity mode: eps [T-1,M,G] float64 and masks [T,M,B] uint8 on the GPU.  Performance mode:
    both None -> in-kernel Philox keyed by (seed, call) and the global particle id."""

    eps: Optional[torch.Tensor] = None
    masks: Optional[torch.Tensor] = None
    seed: int = 0
    call: int = 0
    particle_offset: int = 0
    call_dev: Optional[torch.Tensor] = None  # device int64 [1] added to ``call`` by the kernels (a rollout replayed from a HIP graph: the graph advances it)

    def to_c(self):
        n = abi.Noise()
        n.eps = None if self.eps is None else self.eps.data_ptr()
        n.masks = None if self.masks is None else self.masks.data_ptr()
        n.seed, n.call, n.particle_offset = int(self.seed) & (2**64 - 1), int(self.call) & (2**64 - 1), int(self.particle_offset)
        if self.call_dev is not None:
            if self.call_dev.dtype != torch.int64 or not self.call_dev.is_cuda or self.call_dev.numel() != 1:
                raise RuntimeError("call_dev must be a one-element int64 GPU tensor")
            n.call_dev = self.call_dev.data_ptr()
        return n


@dataclass
class MeasSpec:
    """Measurement model of ``MC_PILCO4PMS.apply_policy`` (MC_PILCO.py:808-906) for the fused rollout: the policy is fed
    noisy positions and backward-difference velocities smoothed by the first-order filter (b, a) = butter(1, fc).
    ``pos_noise`` [T-1,M,n] standard normals on the GPU (parity mode) or None (in-kernel Philox)."""

    pos: Sequence[int]
    vel: Sequence[int]
    std_pos: Sequence[float]
    b: Sequence[float]
    a: Sequence[float]
    pos_noise: Optional[torch.Tensor] = None

    def fill(self, m, T, M, meas_buf):
        n = len(self.pos)
        if len(self.vel) != n or len(self.std_pos) != n or n > abi.MAX_STATE:
            raise RuntimeError("pos / vel / std_pos must have the same length")
        m.n = n
        for i in range(n):
            m.pos[i], m.vel[i], m.std_pos[i] = int(self.pos[i]), int(self.vel[i]), float(self.std_pos[i])
        m.b0, m.b1, m.a0, m.a1 = float(self.b[0]), float(self.b[1]), float(self.a[0]), float(self.a[1])
        if self.pos_noise is not None:
            q = self.pos_noise
            if q.dtype != DT or not q.is_cuda or not q.is_contiguous() or tuple(q.shape) != (max(T - 1, 0), M, n):
                raise RuntimeError("pos_noise must be a contiguous float64 GPU tensor of shape [T-1,M,n]")
        m.pos_noise = None if self.pos_noise is None else self.pos_noise.data_ptr()
        m.meas = meas_buf.data_ptr()


def _set_meas(policy, meas, T, M, buf):
    """Points policy.c.meas at the measurement model for the next launch (n = 0: the policy sees the true state)."""
    if meas is None:
        policy.c.meas.n = 0
        policy.c.meas.meas = None
        policy.c.meas.pos_noise = None
    else:
        meas.fill(policy.c.meas, T, M, buf)


# --------------------------------------------------------------------------------------
# fused rollout (autograd)
# --------------------------------------------------------------------------------------
def _check_noise(noise, T, M, G, B, p_drop, particle_pred):
    if noise.eps is not None:
        e = noise.eps
        if e.dtype != DT or not e.is_cuda or not e.is_contiguous() or tuple(e.shape) != (max(T - 1, 0), M, G):
            raise RuntimeError("eps must be a contiguous float64 GPU tensor of shape [T-1,M,G]")
    if noise.masks is not None:
        k = noise.masks
        if k.dtype != torch.uint8 or not k.is_cuda or not k.is_contiguous() or tuple(k.shape) != (T, M, B):
            raise RuntimeError("masks must be a contiguous uint8 GPU tensor of shape [T,M,B]")


def _mc(model):
    return None if model is None else C.byref(model.c)


def _workspace(model, policy, pc, M, T, which):
    """(bytes, tensor) of the rollout workspace for this (model, policy shape, M, T), kept on the model object: the forward and the
    backward call each have their own (a step's backward runs after its forward on the same stream, and the next forward after that).
    One rollout at a time per model object: two host threads driving the SAME PackedModel on different streams would share these buffers
    (give each its own PackedModel; the operand tensors can be shared)."""
    if model is None:  # (policy-only evaluation: nothing to keep it on)
        nbytes = abi.lib().mcp_rollout_workspace_bytes(None, C.byref(pc), M, T)
        return nbytes, (torch.empty((nbytes + 7) // 8, dtype=DT, device=policy.device) if nbytes else None), None
    cache = model.__dict__.setdefault("_ws_cache", {})
    key = (which, int(M), int(T), policy.kind, policy.B, policy.P, policy.U, policy.c.meas.n)
    hit = cache.get(key)
    if hit is None:
        nbytes = abi.lib().mcp_rollout_workspace_bytes(_mc(model), C.byref(pc), M, T)
        if len(cache) >= 8:  # (a few shapes per model at most: the warm-up rollout, the optimisation, an evaluation)
            cache.clear()
        hit = cache[key] = (nbytes, torch.empty((nbytes + 7) // 8, dtype=DT, device=model.device) if nbytes else None, {"packed": 0})
    return hit


def rollout_forward_raw(model: Optional[PackedModel], policy: PackedPolicy, noise: NoiseSpec, x0, T, p_drop, particle_pred=True, need_jac=True,
                        meas: Optional[MeasSpec] = None, gp_sharding=True, status=None):
    """model None (only with T == 1) evaluates the policy alone.  gp_sharding False: the library never launches GP-sharded (the
    recovery path after MCP_STATUS_SYNC) -- by a flag, the workspace with the kernels' packed operand copies is still passed.
    ``status``: an int32[1] on the device that the kernels OR their flags INTO (a caller that only looks at the flags of many rollouts together
    passes one buffer to all of them: no zero-fill launch per rollout); None: a fresh zeroed word."""
    dev = policy.device if model is None else model.device
    x0 = x0.detach().to(device=dev, dtype=DT).contiguous()
    M = x0.shape[0]
    G, D = (0, policy.U) if model is None else (model.G, model.D)
    _check_noise(noise, T, M, G, policy.B, p_drop, particle_pred)
    states = torch.empty(T, M, policy.S, dtype=DT, device=dev)
    inputs = torch.empty(T, M, policy.U, dtype=DT, device=dev)
    jac = torch.empty(max(T - 1, 1), M, max(G, 1), D, dtype=DT, device=dev) if (need_jac and T > 1) else None
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    elif status.dtype != torch.int32 or status.numel() != 1 or status.device != x0.device or not status.is_contiguous():
        raise RuntimeError("status must be a one-element int32 tensor on the rollout's device")
    pc = policy.bind(p_drop)
    nz = noise.to_c()
    meas_buf = torch.empty(T, M, policy.S, dtype=DT, device=dev) if meas is not None else None
    _set_meas(policy, meas, T, M, meas_buf)
    # workspace: the hand-off granules of the GP-sharded launch (small swarms) and the kernels' packed operand copies; the library zeroes /
    # rebuilds what it uses on the stream, so ONE buffer per (model, shape) serves every step of an optimisation (no size query, no
    # allocation per step)
    nbytes, ws, wstate = _workspace(model, policy, pc, M, T, "fwd") if (model is not None and T > 1) else (0, None, None)
    # the packed operand copies in the workspace (the lean kernel's Kinv tiles, the wide classes' phase-J operands) are functions of the model's
    # arrays alone: built by the first rollout that needs them, then kept -- a PackedModel's tensors are never written again, and this buffer
    # belongs to it (MCP_FWD_KT_PACKED / MCP_FWD_XJ_PACKED)
    pflags = 0 if wstate is None else wstate["packed"]
    ev = fwd_events
    try:
        if ev is not None:
            ev[0].record()
        # (the `_ex` entry point = mcp_rollout_fwd + the process's dispatch request / report, all zero unless a test or tool set it)
        abi.check(abi.lib().mcp_rollout_fwd_ex(_mc(model), C.byref(pc), C.byref(nz), M, T,
                                               int(bool(particle_pred)) | (0 if gp_sharding else abi.FWD_NO_GP_SHARDING) | pflags, abi.ptr(x0),
                                               abi.ptr(states), abi.ptr(inputs), abi.ptr(jac), abi.ptr(status), abi.ptr(ws), nbytes, abi.stream(),
                                               C.byref(abi.DISPATCH)), "mcp_rollout_fwd")
        if ev is not None:
            ev[1].record()
        if wstate is not None:  # what this call left in the workspace (it reports which kernel family ran)
            if abi.DISPATCH.ran_fwd_lean:
                wstate["packed"] |= abi.FWD_KT_PACKED
            elif abi.DISPATCH.ran_particles == 16 and model.D + 1 > 16:
                wstate["packed"] |= abi.FWD_XJ_PACKED
    finally:
        _set_meas(policy, None, T, M, None)
    if meas is not None:
        return states, inputs, jac, status, meas_buf
    return states, inputs, jac, status


# measurement hook (bench.py): a pair of torch.cuda.Event recorded on the launch stream right around mcp_rollout_bwd -- the adjoint sweep
# runs inside autograd's backward, where the caller cannot bracket it.  None (the default) = nothing is recorded.  ``fwd_events``: the same
# around mcp_rollout_fwd (operand packing + hand-off buffer reset + the rollout kernel, without the host's tensor allocations).
bwd_events = None
fwd_events = None


def rollout_backward_raw(model: PackedModel, policy: PackedPolicy, noise: NoiseSpec, states, inputs, jac, g_states, g_inputs, p_drop,
                         want_gx0=False, meas: Optional[MeasSpec] = None, meas_buf=None):
    dev = policy.device
    T, M = states.shape[0], states.shape[1]
    pc = policy.bind(p_drop)
    nz = noise.to_c()
    nbytes, ws, _ = _workspace(model, policy, pc, M, T, "bwd")
    # ONE flat buffer [dJ/dlog_ls (P) | dJ/dcenters (B P) | dJ/dweight (U B) | dJ/dbias (U)]: the three (four) gradients are views of it --
    # autograd hands them to the parameters' .grad as they are, so a particle-sharded step all-reduces its message IN PLACE when the caller
    # supplies the head of that message as ``policy.grad_flat`` (sharding.StepMessage); otherwise one allocation per step instead of four
    P_, B_, U_ = policy.P, policy.B, policy.U
    n_ls, n_c, n_w = P_, B_ * P_, U_ * B_
    n_all = n_ls + n_c + n_w + (U_ if policy.bias is not None else 0)
    flat = getattr(policy, "grad_flat", None)
    if flat is None or flat.numel() < n_all or flat.dtype != DT or flat.device != dev or not flat.is_contiguous():
        flat = torch.empty(n_all, dtype=DT, device=dev)
    g_ls = flat[0:n_ls].view(1, P_)
    g_c = flat[n_ls:n_ls + n_c].view(B_, P_)
    g_w = flat[n_ls + n_c:n_ls + n_c + n_w].view(U_, B_)
    g_b = flat[n_ls + n_c + n_w:n_all] if policy.bias is not None else None
    g_x0 = torch.empty(M, policy.S, dtype=DT, device=dev) if want_gx0 else None
    pc.g_bias = None if g_b is None else g_b.data_ptr()
    gs = None if g_states is None else g_states.to(dtype=DT).contiguous()
    gi = None if g_inputs is None else g_inputs.to(dtype=DT).contiguous()
    _set_meas(policy, meas, T, M, meas_buf)
    ev = bwd_events
    try:
        if ev is not None:
            ev[0].record()
        abi.check(abi.lib().mcp_rollout_bwd_ex(_mc(model), C.byref(pc), C.byref(nz), M, T, abi.ptr(states), abi.ptr(inputs), abi.ptr(jac),
                                               abi.ptr(gs), abi.ptr(gi), abi.ptr(g_ls), abi.ptr(g_c), abi.ptr(g_w), abi.ptr(g_x0), abi.ptr(ws),
                                               nbytes, abi.stream(), C.byref(abi.DISPATCH)), "mcp_rollout_bwd")
        if ev is not None:
            ev[1].record()
    finally:
        _set_meas(policy, None, T, M, None)
        pc.g_bias = None
    return g_ls, g_c, g_w, g_x0, g_b


class RolloutFunction(torch.autograd.Function):
    """(x0, log_lengthscales, centers, weight) -> (states [T,M,S], inputs [T,M,U]) through the fused
    HIP rollout; backward is the fused reverse-time adjoint.  Gradients flow to the three policy
    parameters (and x0); the GP model is frozen, as after ``Model_learning.set_eval_mode``."""

    @staticmethod
    def forward(ctx, x0, log_ls, centers, weight, bias, model, policy, noise, T, p_drop, particle_pred, meas=None, gp_sharding=True, status_acc=None):
        need = any(ctx.needs_input_grad[:5])
        ctx.set_materialize_grads(False)  # (an output the cost does not use -- usually the inputs -- arrives as None, not as a zero-filled tensor)
        out = rollout_forward_raw(model, policy, noise, x0, T, p_drop, particle_pred, need_jac=need, meas=meas, gp_sharding=gp_sharding, status=status_acc)
        states, inputs, jac, status = out[:4]
        if status_acc is not None:
            status = status.view(1)  # (an output must not BE an input; a view launches nothing)
        ctx.model, ctx.policy, ctx.noise, ctx.p_drop = model, policy, noise, p_drop
        ctx.meas, ctx.meas_buf = meas, (out[4] if meas is not None else None)
        ctx.has_jac = jac is not None
        ctx.save_for_backward(states, inputs, jac if jac is not None else torch.empty(0, device=states.device))
        ctx.mark_non_differentiable(status)
        return states, inputs, status

    @staticmethod
    def backward(ctx, g_states, g_inputs, _g_status):
        states, inputs, jac = ctx.saved_tensors
        if not ctx.has_jac:
            jac = None
        if g_states is None and g_inputs is None:  # (nothing downstream depends on the rollout)
            g_states = torch.zeros_like(states)
        g_ls, g_c, g_w, g_x0, g_b = rollout_backward_raw(ctx.model, ctx.policy, ctx.noise, states, inputs, jac, g_states, g_inputs, ctx.p_drop,
                                                         want_gx0=ctx.needs_input_grad[0], meas=ctx.meas, meas_buf=ctx.meas_buf)
        if g_b is not None:
            g_b = g_b.reshape(ctx.policy.bias.shape)
        return g_x0, g_ls.reshape(ctx.policy.log_ls.shape), g_c, g_w, g_b, None, None, None, None, None, None, None, None, None


def rollout(model, policy, noise, x0, T, p_drop=0.0, particle_pred=True, meas: Optional[MeasSpec] = None, gp_sharding=True, status=None):
    """Differentiable fused rollout.  Returns (states, inputs, status).  ``meas``: measurement model between the particles and
    the policy (partially measurable systems); None = the policy sees the true state.  ``gp_sharding`` False forbids the
    GP-sharded launch forms (used to repeat a step whose hand-off reported MCP_STATUS_SYNC).  ``status``: see rollout_forward_raw."""
    return RolloutFunction.apply(x0, policy.log_ls, policy.centers, policy.weight, policy.bias, model, policy, noise, int(T), float(p_drop),
                                 bool(particle_pred), meas, bool(gp_sharding), status)


# --------------------------------------------------------------------------------------
# cost (autograd)
# --------------------------------------------------------------------------------------
class PackedCost:
    def __init__(self, kind, S, device, **kw):
        dev = _dev(device)
        c = abi.Cost()
        c.S = int(S)
        self._keep = []
        if kind == "cartpole":
            c.kind = abi.COST_CARTPOLE
            c.angle_index, c.pos_index = int(kw["angle_index"]), int(kw["pos_index"])
            ts = [float(v) for v in kw["target_state"]]
            ls = [float(v) for v in kw["lengthscales"]]
            c.target_angle, c.target_pos, c.ls_angle, c.ls_pos = ts[0], ts[1], ls[0], ls[1]
        elif kind == "traj":
            c.kind = abi.COST_TRAJ
            used = list(range(S)) if kw.get("used") is None else [int(u) for u in kw["used"]]
            c.n_used = len(used)
            for i, u in enumerate(used):
                c.used[i] = u
            tt = _t(kw["target_traj"], dev)
            ls = _t(kw["lengthscales"], dev).reshape(-1)
            if ls.numel() != len(used):
                raise ValueError("trajectory cost: one lengthscale per used state index is required")
            self._keep = [tt, ls]
            c.target_traj, c.lengthscales = tt.data_ptr(), ls.data_ptr()
            self.traj_len = int(tt.shape[0])
        else:
            raise ValueError(kind)
        self.kind, self.c, self.device = kind, c, dev


_COST_STATUS_SINK = {}  # per device: where mcp_cost_fwd ORs its flags when the caller does not ask for them (never read, never zeroed again)


def cost_moments(cost: PackedCost, states, status=None):
    """Per-time-step (mean, centred sum of squares) over this rank's particles -> [T,2], plus costs [T,M].  ``status``: a zeroed int32[1] to receive the
    kernel's flags (NaN in the costs); None: the flags go to a per-device sink -- the loops decide on the NaN of the cost itself, and a zero-fill
    launch per optimizer step is 4 us of the 0.8 ms a launch-script step takes."""
    T, M, _ = states.shape
    if cost.kind == "traj" and cost.traj_len != T:
        raise RuntimeError("target trajectory has %d rows but the rollout has %d steps" % (cost.traj_len, T))
    st = states.detach().contiguous()
    costs = torch.empty(T, M, dtype=DT, device=st.device)
    mom = torch.empty(T, 2, dtype=DT, device=st.device)
    if status is None:
        status = _COST_STATUS_SINK.get(st.device)
        if status is None:
            status = _COST_STATUS_SINK[st.device] = torch.zeros(1, dtype=torch.int32, device=st.device)
    abi.check(abi.lib().mcp_cost_fwd(C.byref(cost.c), T, M, abi.ptr(st), abi.ptr(costs), abi.ptr(mom), abi.ptr(status), abi.stream()),
              "mcp_cost_fwd")
    return mom, costs, status


def cost_finalize(moments_all, counts):
    """moments_all [R,T,2] (all ranks), counts: list of R ints -> tensor [2] = (cost, std)."""
    R, T = moments_all.shape[0], moments_all.shape[1]
    out = torch.empty(2, dtype=DT, device=moments_all.device)
    cnt = (C.c_int64 * R)(*[int(c) for c in counts])
    abi.check(abi.lib().mcp_cost_finalize(T, R, abi.ptr(moments_all.contiguous()), cnt, abi.ptr(out), abi.stream()), "mcp_cost_finalize")
    return out


class ExpectedCostFunction(torch.autograd.Function):
    """states [T,M,S] -> (sum_t mean_m c, sum_t std_m c) with the HIP cost kernels; with a
    torch.distributed ``group`` the mean/std pool all ranks' particles (one all-gather of the
    [T,2] moments).  ``counts``: particles per rank (default: every rank holds M)."""

    @staticmethod
    def forward(ctx, states, cost, group, counts):
        ctx.set_materialize_grads(False)
        mom, _, _ = cost_moments(cost, states)
        M = states.shape[1]
        if group is not None:
            from . import sharding

            mom_all = sharding.gather_moments(mom, group)
            counts = [M] * mom_all.shape[0] if counts is None else [int(c) for c in counts]
        else:
            counts = [M]
            mom_all = mom.unsqueeze(0)
        out = cost_finalize(mom_all, counts)
        ctx.cost, ctx.m_total = cost, sum(counts)
        ctx.save_for_backward(states.detach())
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cost, _g_std):
        (states,) = ctx.saved_tensors
        T, M, _ = states.shape
        if g_cost is None:  # (only the std output is used downstream: this operator carries no gradient through it, as before)
            return torch.zeros_like(states), None, None, None
        g = torch.empty_like(states)
        st = states.contiguous()
        gc = g_cost.detach().to(dtype=DT).reshape(1).contiguous()  # stays on the device: no host sync
        abi.check(abi.lib().mcp_cost_bwd(C.byref(ctx.cost.c), T, M, abi.ptr(st), abi.ptr(gc), 1.0 / float(ctx.m_total), abi.ptr(g), abi.stream()),
                  "mcp_cost_bwd")
        return g, None, None, None


def expected_cost_raw(cost: PackedCost, states, g_one):
    """(cost, std, dJ/dstates) of this process's particles without autograd: exactly the launches ExpectedCostFunction's forward and backward
    make (``g_one``: a device scalar 1.0, the upstream gradient ``cost.backward()`` starts from).  For callers that replay the step from a
    HIP graph (MC_PILCO.reinforce_policy): the autograd engine's own stream bookkeeping does not survive a stream capture."""
    mom, _, _ = cost_moments(cost, states)
    T, M, _ = states.shape
    out = cost_finalize(mom.unsqueeze(0), [M])
    st = states.detach().contiguous()
    g = torch.empty_like(st)
    abi.check(abi.lib().mcp_cost_bwd(C.byref(cost.c), T, M, abi.ptr(st), abi.ptr(g_one), 1.0 / float(M), abi.ptr(g), abi.stream()), "mcp_cost_bwd")
    return out[0], out[1], g


def expected_cost(cost: PackedCost, states, group=None, counts=None):
    return ExpectedCostFunction.apply(states, cost, group, counts)


class LocalCostFunction(torch.autograd.Function):
    """This rank's share of a particle-sharded expected cost, in the form ONE all-reduce(sum) can pool:
    states [T,M_local,S] -> (share = sum_t sum_m c_tm / m_total  [differentiable; its gradient is what this rank contributes to
    d(sum_t mean_m c)/dtheta],  sums [2T] = mcp_cost_sums: per time step sum_m (c - shift_t) and sum_m (c - shift_t)^2)."""

    @staticmethod
    def forward(ctx, states, cost, m_total, shift, sums_out=None):
        mom, _, _ = cost_moments(cost, states)
        T, M = states.shape[0], states.shape[1]
        # (sums_out: a 1-tuple holding the caller's slot of the step's all-reduce message, sharding.StepMessage.sums -- in a tuple so that
        #  autograd does not take the slot for an input of this function)
        sums = sums_out[0] if sums_out is not None else torch.empty(2 * T, dtype=DT, device=states.device)
        if sums.numel() != 2 * T or sums.dtype != DT or not sums.is_contiguous():
            raise RuntimeError("sums_out must be a contiguous float64 tensor of 2 T entries")
        abi.check(abi.lib().mcp_cost_sums(T, M, abi.ptr(mom), abi.ptr(shift), abi.ptr(sums), abi.stream()), "mcp_cost_sums")
        local = cost_finalize(mom.unsqueeze(0), [M])[0] * (float(M) / float(m_total))
        ctx.cost, ctx.m_total = cost, int(m_total)
        ctx.save_for_backward(states.detach())
        ctx.mark_non_differentiable(sums)
        return local, sums

    @staticmethod
    def backward(ctx, g_local, _g_sums):
        (states,) = ctx.saved_tensors
        T, M, _ = states.shape
        g = torch.empty_like(states)
        st = states.contiguous()
        gc = g_local.detach().to(dtype=DT).reshape(1).contiguous()
        abi.check(abi.lib().mcp_cost_bwd(C.byref(ctx.cost.c), T, M, abi.ptr(st), abi.ptr(gc), 1.0 / float(ctx.m_total), abi.ptr(g), abi.stream()),
                  "mcp_cost_bwd")
        return g, None, None, None, None


def local_cost(cost: PackedCost, states, m_total, shift=None, sums_out=None):
    return LocalCostFunction.apply(states, cost, int(m_total), shift, None if sums_out is None else (sums_out,))


def cost_from_sums(sums, n_total, shift=None, mean_out=None):
    """Pooled (sum_t mean_m c, sum_t unbiased std_m c) from the all-reduced sums of ``local_cost`` -> tensor [2]."""
    T = sums.numel() // 2
    out = torch.empty(2, dtype=DT, device=sums.device)
    abi.check(abi.lib().mcp_cost_finalize_sums(T, int(n_total), abi.ptr(sums.contiguous()), abi.ptr(shift), abi.ptr(out), abi.ptr(mean_out),
                                               abi.stream()), "mcp_cost_finalize_sums")
    return out


# --------------------------------------------------------------------------------------
# single-step posterior (autograd)
# --------------------------------------------------------------------------------------
class PosteriorFunction(torch.autograd.Function):
    """Z [M,D] -> (mu [M,1], var [M]) = GP_prior.get_estimate_from_alpha on the packed GP."""

    @staticmethod
    def forward(ctx, Z, gp: PackedGP):
        Zc = Z.detach().to(dtype=DT).contiguous()
        M, D = Zc.shape
        need = ctx.needs_input_grad[0]
        mu = torch.empty(M, dtype=DT, device=Zc.device)
        var = torch.empty(M, dtype=DT, device=Zc.device)
        Jm = torch.empty(M, D, dtype=DT, device=Zc.device) if need else None
        Jv = torch.empty(M, D, dtype=DT, device=Zc.device) if need else None
        status = torch.zeros(1, dtype=torch.int32, device=Zc.device)
        g = gp.to_c()
        abi.check(abi.lib().mcp_posterior_fwd_ex(C.byref(g), M, abi.ptr(Zc), abi.ptr(mu), abi.ptr(var), abi.ptr(Jm), abi.ptr(Jv), abi.ptr(status),
                                                 abi.stream(), C.byref(abi.DISPATCH)), "mcp_posterior_fwd")
        if need:
            ctx.save_for_backward(Jm, Jv)
        return mu.reshape(-1, 1), var

    @staticmethod
    def backward(ctx, g_mu, g_var):
        Jm, Jv = ctx.saved_tensors
        M, D = Jm.shape
        gz = torch.empty(M, D, dtype=DT, device=Jm.device)
        gm = (torch.zeros(M, dtype=DT, device=Jm.device) if g_mu is None else g_mu.reshape(-1)).contiguous()
        gv = (torch.zeros(M, dtype=DT, device=Jm.device) if g_var is None else g_var.reshape(-1)).contiguous()
        abi.check(abi.lib().mcp_posterior_bwd(M, D, abi.ptr(gm), abi.ptr(gv), abi.ptr(Jm), abi.ptr(Jv), abi.ptr(gz), abi.stream()),
                  "mcp_posterior_bwd")
        return gz, None


def posterior(gp: PackedGP, Z):
    return PosteriorFunction.apply(Z, gp)


def status_flags(status):
    """Decodes a device status word (synchronises)."""
    v = int(status.item())
    return {"nan": bool(v & abi.STATUS_NAN), "nonpos_var": bool(v & abi.STATUS_NONPOS_VAR), "not_spd": bool(v & abi.STATUS_NOT_SPD),
            "sync": bool(v & abi.STATUS_SYNC)}
