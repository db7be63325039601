#!/usr/bin/env python3
"""Benchmark of the MC-PILCO hot path on MI355X: one "step" = one policy-gradient step of
MC_PILCO.reinforce_policy (policy_learning/MC_PILCO.py:484-525) -- fused particle rollout,
expected cost, reverse-time adjoint, [all-reduce of the gradient], Adam update -- on synthetic
cart-pole-shaped data (BASELINE.json configs[1]: 4-D state, SE kernel, N=300, M=400 particles
per GPU, T=150).  Metric: particle-steps/s = M*T / step time, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c3|c5] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: particles are sharded (weak scaling: M per GPU fixed); the only exchange is one
all-gather of 2T cost moments and one all-reduce(sum) of the flattened policy gradient per step
(RCCL).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import mcp_boot  # noqa: E402,F401

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (= vector) peak, datasheet; see DESIGN.md


def cpu_baseline(problem, M, T, p_drop, budget_s=25.0):
    """The CPU oracle (a port of the reference's PyTorch path, validated against it in
    tests/test_oracle_golden.py) on the same synthetic problem, 1 thread like the reference's
    launch scripts (test_mcpilco_cartpole_rbf_ker.py:47-48).  Bounded sample."""
    import numpy as np

    from oracle import mcpilco_oracle as orc

    c = problem["cfg"]
    Tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        hyp = []
        for g in range(c["G"]):
            pw = None if problem["poly"] is None else [torch.log(Tt(w)) for w in problem["poly"][g]]
            hyp.append(orc.GPHyper(torch.log(Tt(c["lengthscales"])), torch.log(Tt([c["lam"]])), torch.log(Tt([c["sigma_n"]])), poly_log_par=pw))
        caches = [orc.pretrain_gp(hyp[g], Tt(problem["Z"]), Tt(problem["Ys"][g])) for g in range(c["G"])]
        m = orc.SpeedModel(hyp, caches, c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])
        pi = problem["policy"]
        pp = orc.PolicyPar(torch.log(Tt(pi["lengthscales"])).reshape(1, -1), Tt(pi["centers"]), Tt(pi["weight"]), c["u_max"],
                           problem["policy_kind"], target_traj=None if problem["target_traj"] is None else Tt(problem["target_traj"]),
                           **{k: v for k, v in problem["policy_extra"].items()})
        if problem["system"] == "cartpole":
            cost_fn = lambda st: orc.cart_pole_cost(st, Tt(c["cost_target"]), Tt(c["cost_ls"]), c["cost_angle_index"], c["cost_pos_index"])
        else:
            tt, ls = Tt(problem["target_traj"]), Tt(c["cost_ls"])
            cost_fn = lambda st: orc.traj_cost(st, tt, ls)
        torch.manual_seed(1)
        x0 = orc.sample_x0(Tt(c["x0_mean"]), Tt(c["x0_var"]), M)
        orc.policy_grad_step(m, pp, x0, min(T, 3), cost_fn, p_drop)  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            orc.policy_grad_step(m, pp, x0, T, cost_fn, p_drop)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 5:
                break
        return {"value": M * T * n / el, "unit": "particle-steps/s", "cores": 1, "kind": "port",
                "sample": "%d policy-gradient step(s) of the same workload (M=%d, T=%d, fwd+cost+bwd, no Adam), torch CPU fp64, 1 thread" % (n, M, T),
                "s_per_step": el / n}
    finally:
        torch.set_num_threads(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c1")
    ap.add_argument("--particles", type=int, default=0, help="particles per GPU (default: the workload's M)")
    ap.add_argument("--horizon", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--noise", default="philox", choices=["philox", "buffers"])
    ap.add_argument("--pms", action="store_true", help="partially measurable system: the policy sees noisy positions and filtered "
                    "finite-difference velocities (MC_PILCO4PMS.apply_policy); cart-pole workloads")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        group = dist.group.WORLD

    from mc_pilco_amd import ops, sharding, workloads

    w = workloads.build(args.workload, device=dev, M=args.particles or None, T=args.horizon or None)
    M, T = w.M, w.T
    meas = None
    if args.pms:
        from scipy import signal

        bb, aa = signal.butter(1, 0.5)  # test_mcpilco4pms_cartpole.py:155-157: pos [0,2], vel [1,3], fc 0.5
        meas = ops.MeasSpec(pos=[0, 2], vel=[1, 3], std_pos=[0.01, 0.01], b=bb, a=aa)
    opt = torch.optim.Adam(w.params, lr=0.01)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    eps = masks = None
    # one HIP event pair per timed step around the dominant kernel's launch (same stream the kernel
    # is launched on: torch's current stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i, ev):
        nonlocal eps, masks
        x0 = w.sample_x0(M, generator=gen)
        if args.noise == "buffers":
            eps = torch.randn(T - 1, M, w.model.G, dtype=torch.float64, device=dev, generator=gen)
            masks = (torch.rand(T, M, w.policy.B, device=dev, generator=gen) >= w.p_drop).to(torch.uint8)
        nz = ops.NoiseSpec(eps=eps, masks=masks, seed=2026, call=i + 1, particle_offset=rank * M)
        opt.zero_grad(set_to_none=True)
        for p in w.params:
            p.grad = None
        if ev is not None:
            ev[0].record()
        states, inputs, status = ops.rollout(w.model, w.policy, nz, x0, T, w.p_drop, meas=meas)
        if ev is not None:
            ev[1].record()
        cost, std = ops.expected_cost(w.cost, states, group)
        cost.backward()
        if world > 1:
            sharding.allreduce_gradients(w.params, group)
        opt.step()
        return cost, status

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier(group=group)
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, None)
    barrier()
    t0 = time.perf_counter()
    last_cost = None
    for i in range(args.steps):
        last_cost, status = step(args.warmup + i, evs[i])
    barrier()
    el = time.perf_counter() - t0
    durs = [a.elapsed_time(b) for (a, b) in evs]
    fwd_avg_ms = sum(durs) / len(durs)

    if world > 1:
        import torch.distributed as dist

        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
        el = float(tmax.item())
    ms_per_step = 1e3 * el / args.steps
    value = world * M * T / (el / args.steps)

    if rank == 0:
        from mc_pilco_amd import hipabi

        # which forward kernel the library dispatched (16 particles per workgroup = the MFMA tile kernel, M > 1024)
        fwd_kernel_name = "rollout_fwd_tile_kernel" if hipabi.lib().mcp_debug_last_particles_per_wg() == 16 else "rollout_fwd_kernel"
        F = workloads.flops_per_particle_step(w)
        achieved = F * M * T / (fwd_avg_ms * 1e-3) / 1e12
        out = {
            "metric": "particle-steps/s (M x T per policy-grad step), cart-pole GP",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (RK4 cart-pole rollouts, N=%d training points/GP, fixed trained-like hyper-parameters, on-device %s noise)"
                    % (w.model.gps[0].N, "Philox" if args.noise == "philox" else "torch.randn buffers"),
            "config": {"workload": "%s: %s, %d GPs, D=%d, N=%d, B=%d, M=%d particles/GPU, T=%d, p_dropout=%.2f; step = rollout fwd + cost + "
                                   "adjoint bwd%s + Adam%s" % (args.workload, w.problem["system"], w.model.G, w.model.D, w.model.gps[0].N, w.policy.B, M, T,
                                                            w.p_drop, " + RCCL all-reduce" if world > 1 else "",
                                                            "; policy on measured states (MC_PILCO4PMS)" if args.pms else ""),
                       "particles_per_gpu": M, "horizon": T, "parallelism": "particle-dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                         "traffic": None, "kernel": fwd_kernel_name, "kernel_ms": fwd_avg_ms,
                         "alg_flops_per_particle_step": F, "units_per_launch": M * T},
            "final_cost": float(last_cost),
        }
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                out["roofline"]["traffic"] = json.load(open(tf)).get(args.workload)
            except Exception:
                pass
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(w.problem, M, T, w.p_drop)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
