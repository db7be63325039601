"""N>1 path on CPU: two gloo ranks shard the particles, exchange cost moments and policy gradients through
mc_pilco_amd.sharding, and must reproduce the single-process result (cost, std, gradient) of the same particles.
The per-rank rollouts are computed by the CPU oracle here (the HIP path needs a GPU; its own 2-rank test is
tests/test_gpu_sharding.py); what is under test is the sharding arithmetic and the collectives."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden  # noqa: I001  (first: registers the package, also in spawned workers)
from helpers import T, oracle_cost_fn, oracle_model, oracle_policy


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m_total, out_q):
    import torch.distributed as dist

    import mcp_boot  # noqa: F401
    from mc_pilco_amd import sharding
    from oracle import mcpilco_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load_golden("rollout_se")
    m = oracle_model(fx, "se")
    pp = oracle_policy(fx, "se")
    cost_fn = oracle_cost_fn(fx, "se")
    off, cnt = sharding.shard_range(m_total, world, rank)
    sl = slice(off, off + cnt)
    x0 = T(fx["states"][0][sl])
    eps, masks = T(fx["eps"][:, sl]), T(fx["masks"][:, sl])
    prm = [pp.log_ls, pp.centers, pp.weight]
    for p in prm:
        p.requires_grad_(True)
    st, _ = orc.apply_policy(m, pp, x0, fx["states"].shape[0], float(fx["p_drop"]), eps, masks)
    c = cost_fn(st)  # [T, cnt]
    mean = c.mean(1)
    mom = torch.stack([mean.detach(), ((c.detach() - mean.detach()[:, None]) ** 2).sum(1)], 1)
    mom_all = sharding.gather_moments(mom, dist.group.WORLD)
    cost, std = sharding.pooled_cost_reference(mom_all, sharding.shard_counts(m_total, world))
    (c.sum() / m_total).backward()  # this rank's share of d(sum_t mean_m c)/dtheta
    sharding.allreduce_gradients(prm, dist.group.WORLD)
    if rank == 0:
        out_q.put((float(cost), float(std), [p.grad.numpy().copy() for p in prm]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("m_total", [24, 23])
def test_two_rank_sharding_reproduces_single_process(m_total):
    from oracle import mcpilco_oracle as orc

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, m_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    cost, std, grads = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process on the same particles
    fx = load_golden("rollout_se")
    m, pp, cost_fn = oracle_model(fx, "se"), oracle_policy(fx, "se"), oracle_cost_fn(fx, "se")
    sl = slice(0, m_total)
    c1, s1, g1, _, _ = orc.policy_grad_step(m, pp, T(fx["states"][0][sl]), fx["states"].shape[0], cost_fn, float(fx["p_drop"]), T(fx["eps"][:, sl]),
                                            T(fx["masks"][:, sl]))
    assert abs(cost - float(c1)) < 1e-12 * abs(float(c1))
    assert abs(std - float(s1)) < 1e-10 * abs(float(s1))
    for g, k in zip(grads, ["log_ls", "centers", "weight"]):
        assert np.max(np.abs(g - g1[k].numpy())) < 1e-12 * max(1.0, float(g1[k].abs().max()))


def _worker_fused(rank, world, port, m_total, use_shift, out_q):
    """The single-collective form (sharding.StepReducer): backward first, then ONE all-reduce of [grad | cost sums | flags]."""
    import torch.distributed as dist

    import mcp_boot  # noqa: F401
    from mc_pilco_amd import sharding
    from mc_pilco_amd.policy_learning.Cost_function import Expected_cost
    from oracle import mcpilco_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load_golden("rollout_se")
    m = oracle_model(fx, "se")
    pp = oracle_policy(fx, "se")
    cost_fn = oracle_cost_fn(fx, "se")
    off, cnt = sharding.shard_range(m_total, world, rank)
    sl = slice(off, off + cnt)
    Tn = fx["states"].shape[0]
    prm = [pp.log_ls, pp.centers, pp.weight]
    for p in prm:
        p.requires_grad_(True)
    st, _ = orc.apply_policy(m, pp, T(fx["states"][0][sl]), Tn, float(fx["p_drop"]), T(fx["eps"][:, sl]), T(fx["masks"][:, sl]))
    ec = Expected_cost(lambda x, u, k: cost_fn(x))
    shift = T(0.3 + 0.05 * np.arange(Tn)) if use_shift else None
    share, sums = ec.local_moments(st, None, 0, m_total, shift)
    share.backward()
    red = sharding.StepReducer(dist.group.WORLD)
    flags = torch.tensor([0.0, 1.0 if rank == world - 1 else 0.0], dtype=torch.float64)  # one rank raises a flag: every rank must see it
    sums_all, fl = red.reduce(prm, sums, flags)
    mean_out = torch.empty(Tn, dtype=torch.float64)
    cost, std = Expected_cost.from_sums(sums_all, m_total, shift, mean_out)
    out_q.put((rank, float(cost), float(std), [p.grad.numpy().copy() for p in prm], fl.tolist(), mean_out.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m_total,use_shift", [(2, 24, False), (4, 23, True), (4, 22, False)])
def test_single_allreduce_step_reproduces_single_process(world, m_total, use_shift):
    """2 and 4 gloo ranks, even and uneven shards: gradient, cost, std and the flags agree on every rank and with one process."""
    from oracle import mcpilco_oracle as orc

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fused, args=(r, world, port, m_total, use_shift, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fx = load_golden("rollout_se")
    m, pp, cost_fn = oracle_model(fx, "se"), oracle_policy(fx, "se"), oracle_cost_fn(fx, "se")
    sl = slice(0, m_total)
    c1, s1, g1, st1, _ = orc.policy_grad_step(m, pp, T(fx["states"][0][sl]), fx["states"].shape[0], cost_fn, float(fx["p_drop"]), T(fx["eps"][:, sl]),
                                              T(fx["masks"][:, sl]))
    mean1 = cost_fn(st1).mean(1).detach().numpy()
    for rank, cost, std, grads, fl, mean_t in res:
        assert abs(cost - float(c1)) < 1e-12 * abs(float(c1))
        assert abs(std - float(s1)) < 1e-9 * abs(float(s1))
        assert np.max(np.abs(mean_t - mean1)) < 1e-13
        assert fl[0] == 0.0 and fl[1] == 1.0
        for g, k in zip(grads, ["log_ls", "centers", "weight"]):
            assert np.max(np.abs(g - g1[k].numpy())) < 1e-12 * max(1.0, float(g1[k].abs().max()))
        for g, g0 in zip(grads, res[0][3]):
            assert np.array_equal(g, g0)  # identical on every rank: identical optimizer updates


def test_shard_ranges_partition_the_particles():
    from mc_pilco_amd import sharding

    for m_total in (1, 7, 8, 400, 401, 32000):
        for world in (1, 2, 3, 8):
            rs = [sharding.shard_range(m_total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and sum(c for _, c in rs) == m_total
            for (o0, c0), (o1, _) in zip(rs, rs[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in rs) - min(c for _, c in rs) <= 1


def _worker_nan_step(rank, world, port, m_total, out_q):
    """Three sharded steps on the same particles; in step 0 one particle of rank 1 has a NaN cost."""
    import torch.distributed as dist

    import mcp_boot  # noqa: F401
    from mc_pilco_amd import sharding
    from mc_pilco_amd.policy_learning.Cost_function import Expected_cost
    from oracle import mcpilco_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load_golden("rollout_se")
    m, pp, cost_fn = oracle_model(fx, "se"), oracle_policy(fx, "se"), oracle_cost_fn(fx, "se")
    off, cnt = sharding.shard_range(m_total, world, rank)
    sl = slice(off, off + cnt)
    Tn = fx["states"].shape[0]
    prm = [pp.log_ls, pp.centers, pp.weight]
    for p in prm:
        p.requires_grad_(True)
    red = sharding.StepReducer(dist.group.WORLD)
    shift = torch.zeros(Tn, dtype=torch.float64)
    out = []
    for step in range(3):
        for p in prm:
            p.grad = None
        st, _ = orc.apply_policy(m, pp, T(fx["states"][0][sl]), Tn, float(fx["p_drop"]), T(fx["eps"][:, sl]), T(fx["masks"][:, sl]))

        def poisoned(x, u, k, step=step):
            c = cost_fn(x)
            if step == 0 and rank == 1:
                c = c.clone()
                c[3, 0] = float("nan")
            return c

        ec = Expected_cost(poisoned)
        share, sums = ec.local_moments(st, None, 0, m_total, shift)
        share.backward()
        local_nan = torch.isnan(share.detach()).reshape(()).to(torch.float64)
        cost, std, fl, shift = sharding.finish_step(Expected_cost, red, prm, sums, torch.stack([local_nan, torch.zeros((), dtype=torch.float64)]),
                                                    m_total, shift)
        out.append((float(cost), float(std), fl.tolist(), bool(torch.isfinite(shift).all())))
    out_q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_a_nan_rollout_does_not_poison_the_following_sharded_steps():
    """after a NaN rollout the pooled per-step means must not become the next step's shift, and the rank whose
    particles are clean must still see the retry flag.  Step 0: NaN on rank 1 only -> cost NaN and flags[0] > 0 on BOTH ranks;
    steps 1, 2: finite, equal to the single-process cost of the same particles."""
    from oracle import mcpilco_oracle as orc

    world, m_total = 2, 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nan_step, args=(r, world, port, m_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fx = load_golden("rollout_se")
    m, pp, cost_fn = oracle_model(fx, "se"), oracle_policy(fx, "se"), oracle_cost_fn(fx, "se")
    sl = slice(0, m_total)
    c1, s1, _, _, _ = orc.policy_grad_step(m, pp, T(fx["states"][0][sl]), fx["states"].shape[0], cost_fn, float(fx["p_drop"]), T(fx["eps"][:, sl]),
                                           T(fx["masks"][:, sl]))
    for rank in range(world):
        steps = res[rank]
        assert np.isnan(steps[0][0]) and steps[0][2][0] > 0 and steps[0][3]  # NaN seen, flag raised on every rank, shift still finite
        for cost, std, fl, ok in steps[1:]:
            assert fl[0] == 0 and ok
            assert abs(cost - float(c1)) < 1e-12 * abs(float(c1))
            assert abs(std - float(s1)) < 1e-9 * abs(float(s1))


def test_eight_rank_shard_arithmetic_of_the_c4_swarm():
    """BASELINE.json configs[3]: M = 32 000 particles over 8 ranks -- counts, offsets, the 1/M_total scaling of the shares and the
    pooled cost / std from the summed moments (plain arithmetic, no process group: what 8 ranks would send and receive)."""
    from mc_pilco_amd import sharding
    from mc_pilco_amd.policy_learning.Cost_function import Expected_cost

    M, R, Tn = 32000, 8, 5
    rs = [sharding.shard_range(M, R, r) for r in range(R)]
    assert [c for _, c in rs] == [4000] * 8 and [o for o, _ in rs] == [4000 * r for r in range(R)]
    g = torch.Generator().manual_seed(4)
    costs = torch.rand(Tn, M, dtype=torch.float64, generator=g)
    shift = costs.mean(1) + 0.01
    ec = Expected_cost(lambda x, u, k: x)
    shares, sums = zip(*[ec.local_moments(costs[:, o:o + c], None, 0, M, shift) for o, c in rs])
    assert abs(float(sum(shares)) - float(costs.mean(1).sum())) < 1e-12
    cost, std = Expected_cost.from_sums(sum(sums), M, shift)
    assert abs(float(cost) - float(costs.mean(1).sum())) < 1e-12
    assert abs(float(std) - float(costs.std(1).sum())) < 1e-10


def _worker_message(rank, world, port, m_total, out_q):
    """Round 6: the step's persistent flat message (sharding.StepMessage) all-reduced IN PLACE -- gradients left there by the backward pass
    (views handed to autograd, as ops.rollout_backward_raw does with PackedPolicy.grad_flat), cost sums written there by local_moments --
    against the gather / scatter form (StepReducer.reduce) on the same data; then the fallback (gradients in tensors of their own)."""
    import torch.distributed as dist

    import mcp_boot  # noqa: F401
    from mc_pilco_amd import sharding
    from mc_pilco_amd.policy_learning.Cost_function import Expected_cost
    from oracle import mcpilco_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load_golden("rollout_se")
    m, pp, cost_fn = oracle_model(fx, "se"), oracle_policy(fx, "se"), oracle_cost_fn(fx, "se")
    off, cnt = sharding.shard_range(m_total, world, rank)
    sl = slice(off, off + cnt)
    Tn = fx["states"].shape[0]
    prm = [pp.log_ls, pp.centers, pp.weight]
    for p in prm:
        p.requires_grad_(True)
    red = sharding.StepReducer(dist.group.WORLD)
    ec = Expected_cost(lambda x, u, k: cost_fn(x))
    flags = torch.tensor([0.0, float(rank), 1.0], dtype=torch.float64)

    def local_step(sums_out=None):
        for p in prm:
            p.grad = None
        st, _ = orc.apply_policy(m, pp, T(fx["states"][0][sl]), Tn, float(fx["p_drop"]), T(fx["eps"][:, sl]), T(fx["masks"][:, sl]))
        share, sums = ec.local_moments(st, None, 0, m_total, None, sums_out=sums_out)
        share.backward()
        return sums

    # (1) the round-5 form: cat -> all-reduce -> copies back
    sums = local_step()
    s_ref, f_ref = red.reduce(prm, sums, flags)
    g_ref = [p.grad.clone() for p in prm]
    s_ref, f_ref = s_ref.clone(), f_ref.clone()
    # (2) in place: the gradients become views of the message (what the adjoint sweep's flat output gives autograd)
    n_grad = sum(p.numel() for p in prm)
    msg = sharding.StepMessage(n_grad, Tn, 3, "cpu")
    sums = local_step(sums_out=msg.sums)
    assert sums.data_ptr() == msg.sums.data_ptr()
    o = 0
    for p in prm:
        view = msg.grad[o:o + p.numel()].view(p.shape)
        view.copy_(p.grad)
        p.grad = view
        o += p.numel()
    assert msg.holds(prm)
    s_in, f_in = red.reduce_message(msg, prm, sums, flags)
    in_place = red.last_in_place and all(p.grad.untyped_storage().data_ptr() == msg.flat.untyped_storage().data_ptr() for p in prm)
    same = all(torch.equal(p.grad, g) for p, g in zip(prm, g_ref)) and torch.equal(s_in, s_ref) and torch.equal(f_in, f_ref)
    # (3) fallback: gradients in tensors of their own -> copied in and out, same numbers
    sums = local_step(sums_out=msg.sums)
    assert not msg.holds(prm)
    s_fb, f_fb = red.reduce_message(msg, prm, sums, flags)
    fallback_ok = (red.last_in_place is False) and all(torch.equal(p.grad, g) for p, g in zip(prm, g_ref)) and torch.equal(s_fb, s_ref)
    out_q.put((rank, bool(in_place), bool(same), bool(fallback_ok), f_in.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m_total", [(2, 24), (4, 23)])
def test_step_message_is_reduced_in_place_and_matches_the_gather_scatter_form(world, m_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_message, args=(r, world, port, m_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, in_place, same, fallback_ok, fl in res:
        assert in_place, "the message was not reduced in place"
        assert same, "in-place reduction differs from the gather / scatter form"
        assert fallback_ok
        assert fl == [0.0, float(sum(range(world))), float(world)]
