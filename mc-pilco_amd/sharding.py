"""Particle sharding over the GPUs of one node.

Particles are independent through the whole rollout and its adjoint; they meet only in the cost's mean over
particles (Cost_function.py:33) and in the parameter-gradient sum.  So every rank simulates its own contiguous slice
of the particles, and per optimizer step there are exactly two small exchanges (RCCL through torch.distributed;
messages are a few KB, latency-bound):

  gather_moments      all-gather of the [T,2] per-time-step cost moments (mean, centred sum of squares)
  allreduce_gradients all-reduce(sum) of the flattened policy gradient (each rank's gradient is already scaled
                      by 1/M_total), after which every rank applies the identical optimizer update

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
