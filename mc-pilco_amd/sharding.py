"""Particle sharding over the GPUs of one node.

Particles are independent through the whole rollout and its adjoint; they meet only in the cost's mean over
particles (Cost_function.py:33) and in the parameter-gradient sum.  So every rank simulates its own contiguous slice
of the particles, and per optimizer step there is exactly ONE small exchange (latency-bound, 10-100 KB):

  StepReducer.reduce  all-reduce(sum) of the flat fp64 message
                      [dJ/dlog_lengthscales | dJ/dcenters | dJ/dweight | sum_m (c_t - shift_t) (T) | sum_m (c_t - shift_t)^2 (T) | flags]
                      (each rank's gradient is already scaled by 1/M_total; the rank runs its backward sweep BEFORE the
                      exchange -- d(sum_t mean_m c)/dx_tm = (1/M_total) dc/dx needs nothing from the other ranks), after
                      which every rank forms the pooled cost / std from the sums and applies the identical optimizer update.
                      The flags carry the kernels' status bits (NaN, hand-off time-out), so every rank takes the same
                      retry decision.  Transport: torch.distributed (backend "nccl" = RCCL on the GPUs, gloo in the CPU
                      tests) or the C ABI's own RCCL communicator (mcp_comm_init / mcp_allreduce_grad).

  gather_moments / allreduce_gradients   the two-exchange form of round 1 (all-gather of [T,2] moments, then the gradient
                      all-reduce); kept for callers that want the cost before they run backward.

Noise is counted by GLOBAL particle id (mcp_noise.particle_offset), so a sharded run draws what one GPU would.
This module is host-side plumbing only (no arithmetic of the hot path); it works with any backend
(nccl = RCCL on the GPUs, gloo in the CPU tests).
"""
from typing import List, Sequence, Tuple

import torch


def shard_range(m_total: int, world: int, rank: int) -> Tuple[int, int]:
    """(offset, count) of rank's particle slice; the first m_total % world ranks hold one particle more."""
    base, extra = divmod(int(m_total), int(world))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def shard_counts(m_total: int, world: int) -> List[int]:
    return [shard_range(m_total, world, r)[1] for r in range(world)]


def gather_moments(moments: torch.Tensor, group) -> torch.Tensor:
    """[T,2] per-rank cost moments -> [R,T,2] (rank order)."""
    import torch.distributed as dist

    R = dist.get_world_size(group)
    out = [torch.empty_like(moments) for _ in range(R)]
    dist.all_gather(out, moments.contiguous(), group=group)
    return torch.stack(out)


def pooled_cost_reference(moments_all: torch.Tensor, counts: Sequence[int]):
    """Plain-torch statement of what the HIP kernel ``mcp_cost_finalize`` computes (Chan et al. pooled variance):
    sum_t mean_t and sum_t unbiased std_t over all ranks' particles.  Used by the CPU tests of the sharding
    arithmetic; the product path calls the kernel."""
    n = torch.as_tensor(list(counts), dtype=moments_all.dtype, device=moments_all.device).reshape(-1, 1)
    n_tot = n.sum()
    mean = (n * moments_all[:, :, 0]).sum(0) / n_tot
    m2 = (moments_all[:, :, 1] + n * (moments_all[:, :, 0] - mean) ** 2).sum(0)
    return mean.sum(), torch.sqrt(m2 / (n_tot - 1)).sum()


def allreduce_gradients(params, group) -> None:
    """In-place all-reduce(sum) of the parameters' gradients as ONE flat message."""
    import torch.distributed as dist

    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, group=group)
    o = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].reshape(p.shape))
        o += n


class StepMessage:
    """The flat fp64 message of a sharded optimizer step, allocated ONCE per (policy shape, horizon) and all-reduced in place:
        [dJ/dlog_lengthscales | dJ/dcenters | dJ/dweight (| dJ/dbias) | sum_m (c_t - shift_t) (T) | sum_m (c_t - shift_t)^2 (T) | flags]
    ``grad`` is handed to the adjoint sweep (``PackedPolicy.grad_flat``: mcp_rollout_bwd writes the gradients there and the parameters'
    ``.grad`` become views of it), ``sums`` to the cost (``local_cost(..., sums_out=)``: mcp_cost_sums writes there), ``flags`` is filled by
    one small copy -- no torch.cat before the collective and no copy back after it (round 5: cat + 3 copies per step)."""

    def __init__(self, n_grad: int, T: int, n_flags: int, device, dtype=torch.float64):
        self.n_grad, self.T, self.n_flags = int(n_grad), int(T), int(n_flags)
        self.flat = torch.zeros(self.n_grad + 2 * self.T + self.n_flags, dtype=dtype, device=device)
        self.grad = self.flat[:self.n_grad]
        self.sums = self.flat[self.n_grad:self.n_grad + 2 * self.T]
        self.flags = self.flat[self.n_grad + 2 * self.T:]

    def fits(self, n_grad, T, n_flags, device):
        return (self.n_grad, self.T, self.n_flags) == (int(n_grad), int(T), int(n_flags)) and self.flat.device == torch.device(device)

    def holds(self, params) -> bool:
        """True when the gradients of ``params`` (in that order; those with a gradient) ARE consecutive pieces of ``grad``, from its start."""
        o, item = 0, self.flat.element_size()
        for p in params:
            g = p.grad
            if g is None:
                continue
            if (g.dtype != self.flat.dtype or g.device != self.flat.device or not g.is_contiguous()
                    or g.data_ptr() != self.flat.data_ptr() + o * item):
                return False
            o += g.numel()
        return o == self.n_grad


class StepReducer:
    """The single collective of a sharded optimizer step.  ``transport``: "torch" (torch.distributed on ``group``) or "abi"
    (libmcpilco_hip's RCCL communicator, created once per process; the unique id travels through torch.distributed's
    object broadcast, any backend)."""

    def __init__(self, group=None, transport="torch"):
        import torch.distributed as dist

        self.group = dist.group.WORLD if group is None else group
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.transport = transport
        self.last_in_place = None  # whether the last reduce_message found the gradients already inside the message
        if transport == "abi":
            import ctypes as C

            from . import hipabi as abi

            lib = abi.lib()
            if lib.mcp_comm_world() == 0:
                buf = C.create_string_buffer(abi.COMM_ID_BYTES)
                if self.rank == 0:
                    abi.check(lib.mcp_comm_unique_id(buf), "mcp_comm_unique_id")
                box = [bytes(buf.raw)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0), group=self.group)
                abi.check(lib.mcp_comm_init(self.world, self.rank, box[0]), "mcp_comm_init")
            elif lib.mcp_comm_world() != self.world:
                raise RuntimeError("libmcpilco_hip already holds a communicator of another size")
        elif transport != "torch":
            raise ValueError(transport)

    def allreduce_(self, flat: torch.Tensor) -> torch.Tensor:
        if self.transport == "abi":
            from . import hipabi as abi

            abi.check(abi.lib().mcp_allreduce_grad(abi.ptr(flat), flat.numel(), abi.stream()), "mcp_allreduce_grad")
        else:
            import torch.distributed as dist

            dist.all_reduce(flat, group=self.group)
        return flat

    def reduce_message(self, msg: "StepMessage", params, sums: torch.Tensor, flags: torch.Tensor):
        """The step's ONE all-reduce on the caller's persistent message.  When the gradients already live in ``msg.grad`` (the adjoint sweep
        wrote them there) and ``sums`` is ``msg.sums``, nothing is gathered or scattered: flags -> their slot (one small copy), all-reduce of
        ``msg.flat`` in place, and the parameters' ``.grad`` hold the reduced values.  Anything else (a gradient that autograd accumulated
        into a tensor of its own, an optimizer that replaced ``.grad``) falls back to copies in and out -- same result, same single message.
        Returns (sums, flags) views of the reduced message (valid until the next step overwrites it)."""
        ps = [p for p in params if p.grad is not None]
        in_place = msg.holds(ps) if ps else False
        if ps and not in_place:
            o = 0
            for p in ps:
                n = p.numel()
                msg.grad[o:o + n].copy_(p.grad.reshape(-1))
                o += n
            if o != msg.n_grad:
                raise RuntimeError("the step's message was sized for %d gradient entries, the parameters hold %d" % (msg.n_grad, o))
        if sums.data_ptr() != msg.sums.data_ptr():
            msg.sums.copy_(sums.reshape(-1))
        msg.flags.copy_(flags.reshape(-1).to(msg.flat.dtype))
        self.allreduce_(msg.flat)
        if ps and not in_place:
            o = 0
            for p in ps:
                n = p.numel()
                p.grad.copy_(msg.grad[o:o + n].reshape(p.shape))
                o += n
        self.last_in_place = bool(in_place)
        return msg.sums, msg.flags

    def reduce(self, params, sums: torch.Tensor, flags: torch.Tensor):
        """params: tensors whose ``.grad`` (this rank's share, None = no gradient wanted) are summed in place; sums [2T] and
        flags [k] (non-negative doubles) are summed too and returned."""
        ps = [p for p in params if p.grad is not None]
        parts = [p.grad.reshape(-1) for p in ps] + [sums.reshape(-1), flags.reshape(-1).to(sums.dtype)]
        flat = self.allreduce_(torch.cat(parts))
        o = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[o:o + n].reshape(p.shape))
            o += n
        ns = sums.numel()
        return flat[o:o + ns], flat[o + ns:]


def finish_step(cost_cls, reducer: "StepReducer", params, sums: torch.Tensor, flags: torch.Tensor, m_total: int, shift: torch.Tensor, msg=None):
    """What follows a rank's own backward sweep in a sharded optimizer step: the ONE all-reduce of [gradients | cost sums | flags],
    the pooled (cost, std) from the reduced sums, and the shift of the next step's sums.

    Returns (cost, std, flags, next_shift).  ``flags[0]`` additionally counts a non-finite POOLED cost -- computed from the reduced
    sums, hence identical on every rank: all ranks take the same retry decision (MC_PILCO.py:497) even when the NaN sits in
    another rank's particles.  The shift is only the numerical centre of the summable moments; a component that came out
    non-finite (a NaN rollout) keeps its previous value, so the steps after a NaN rollout are not poisoned by it.
    ``cost_cls`` provides ``from_sums(sums, n_total, shift, mean_out)`` (policy_learning.Cost_function.Expected_cost or its HIP
    subclasses).  ``msg``: the step's persistent StepMessage -- the all-reduce then runs in place on it (StepReducer.reduce_message)."""
    sums_all, fl = reducer.reduce(params, sums, flags) if msg is None else reducer.reduce_message(msg, params, sums, flags)
    new_shift = torch.empty_like(shift)
    cost, std = cost_cls.from_sums(sums_all, m_total, shift, new_shift)
    next_shift = torch.where(torch.isfinite(new_shift), new_shift, shift)
    fl = fl.clone()
    fl[0] = fl[0] + (~torch.isfinite(cost.detach())).to(fl.dtype)
    return cost, std, fl, next_shift
