"""GPU, two ranks (both on cuda:0, gloo rendezvous): the sharded HIP path -- per-rank fused rollout with global
particle ids, this rank's cost share and adjoint sweep, then ONE all-reduce of [gradient | cost sums | flags]
(sharding.StepReducer, mcp_cost_sums / mcp_cost_finalize_sums) -- reproduces the single-process HIP result on the
same particles; the two-exchange form of round 1 (all-gathered moments + gradient all-reduce) is checked beside it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import conftest  # noqa: F401  (registers the package, also in spawned workers)

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out_q, fused=True):
    import torch.distributed as dist

    import mcp_boot  # noqa: F401
    from mc_pilco_amd import ops, sharding, workloads

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.group.WORLD if world > 1 else None
    m_total, Tn = 48, 10
    w = workloads.build("tiny", device=dev, M=m_total, T=Tn)
    off, cnt = sharding.shard_range(m_total, world, rank)
    g = torch.Generator(device="cpu")
    g.manual_seed(11)
    x0_all = (w.x0_mean.cpu() + w.x0_std.cpu() * torch.randn(m_total, 4, dtype=torch.float64, generator=g)).to(dev)
    nz = ops.NoiseSpec(seed=5, call=3, particle_offset=off)
    for p in w.params:
        p.grad = None
    st, inp, status = ops.rollout(w.model, w.policy, nz, x0_all[off:off + cnt], Tn, w.p_drop)
    if world > 1 and fused:
        shift = torch.full((Tn,), 0.4, dtype=torch.float64, device=dev)
        # round 6: the step's message is ONE persistent flat buffer -- mcp_rollout_bwd writes the gradients into its head (the parameters'
        # .grad are views of it), mcp_cost_sums into its middle -- and the all-reduce runs on it in place (sharding.StepMessage)
        msg = sharding.StepMessage(w.policy.grad_numel(), Tn, 1, dev)
        w.policy.grad_flat = msg.grad
        share, sums = ops.local_cost(w.cost, st, m_total, shift, sums_out=msg.sums)
        share.backward()
        w.policy.grad_flat = None
        red = sharding.StepReducer(group)
        assert msg.holds(w.params) and sums.data_ptr() == msg.sums.data_ptr()
        sums_all, fl = red.reduce_message(msg, w.params, sums, status.to(torch.float64))
        assert red.last_in_place and all(p.grad.untyped_storage().data_ptr() == msg.flat.untyped_storage().data_ptr() for p in w.params)
        cost, std = ops.cost_from_sums(sums_all, m_total, shift)
        assert float(fl.sum()) == 0.0
    else:
        cost, std = ops.expected_cost(w.cost, st, group, sharding.shard_counts(m_total, world) if world > 1 else None)
        cost.backward()
        if world > 1:
            sharding.allreduce_gradients(w.params, group)
    torch.cuda.synchronize()
    res = (float(cost), float(std), [p.grad.cpu().numpy().copy() for p in w.params], st.detach().cpu().numpy(), off, cnt)
    if rank == 0 or world == 1:
        out_q.put(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("fused", [True, False])
def test_two_rank_hip_sharding_matches_single_process(fused):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p1 = ctx.Process(target=_run, args=(0, 1, 0, q))
    p1.start()
    c1, s1, g1, st1, _, _ = q.get(timeout=300)
    p1.join(timeout=60)
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q, fused)) for r in range(2)]
    for p in procs:
        p.start()
    c2, s2, g2, st2, off, cnt = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(st2, st1[:, off:off + cnt])  # same noise per GLOBAL particle: bit-identical trajectories
    assert abs(c2 - c1) < 1e-13 * abs(c1)
    assert abs(s2 - s1) < 1e-10 * abs(s1)
    for a, b in zip(g2, g1):
        assert np.max(np.abs(a - b)) < 1e-12 * max(1.0, np.max(np.abs(b)))


def _abi_comm_single_rank(out_q):
    """mcp_comm_* / mcp_allreduce_grad on a one-rank communicator (all a one-GPU box can host): RCCL loads at run time, the
    communicator is created once, an all-reduce over one rank is the identity, a second init with the same geometry is a no-op."""
    import ctypes as C

    import mcp_boot  # noqa: F401
    from mc_pilco_amd import hipabi

    torch.cuda.set_device(0)
    lib = hipabi.lib()
    uid = C.create_string_buffer(hipabi.COMM_ID_BYTES)
    rc = [lib.mcp_comm_unique_id(uid), lib.mcp_comm_init(1, 0, uid.raw), lib.mcp_comm_world(), lib.mcp_comm_init(1, 0, uid.raw)]
    x = torch.arange(1506, dtype=torch.float64, device="cuda") * 0.5
    y = x.clone()
    rc.append(lib.mcp_allreduce_grad(hipabi.ptr(y), y.numel(), hipabi.stream()))
    torch.cuda.synchronize()
    same = bool(torch.equal(x, y))
    rc += [lib.mcp_comm_init(2, 0, uid.raw), lib.mcp_comm_destroy(), lib.mcp_comm_world()]
    out_q.put((rc, same))


def test_abi_collective_on_a_single_rank_communicator():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_abi_comm_single_rank, args=(q,))
    p.start()
    rc, same = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert rc == [0, 0, 1, 0, 0, -1, 0, 0] and same  # (a second init with another geometry is refused: one communicator per process)
