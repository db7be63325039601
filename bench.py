#!/usr/bin/env python3
"""Benchmark of the MC-PILCO hot path on MI355X: one "step" = one policy-gradient step of
MC_PILCO.reinforce_policy (policy_learning/MC_PILCO.py:484-525) -- fused particle rollout,
expected cost, reverse-time adjoint, [one all-reduce of gradient + cost sums], Adam update -- on
synthetic cart-pole-shaped data (BASELINE.json configs[1]: 4-D state, SE kernel, N=300, M=400
particles per GPU, T=150).  Metric: particle-steps/s = M*T / step time, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c3|c5] [--no-cpu] [--no-extra]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself as
child processes -- before this process makes any GPU call -- and relays rank 0's JSON line.

Multi-GPU: particles are sharded (weak scaling: M per GPU fixed); the ranks meet in ONE all-reduce
per step (RCCL): [gradient | per-time-step cost sums | status flags].  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (= vector) peak, datasheet; see DESIGN.md


def note(msg):
    """Progress line on stderr (stdout carries the one JSON line)."""
    print("[bench %.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c1")
    ap.add_argument("--particles", type=int, default=0, help="particles per GPU (default: the workload's M)")
    ap.add_argument("--horizon", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_workloads (c3, c5) of the default single-GPU run")
    ap.add_argument("--noise", default="philox", choices=["philox", "buffers"])
    ap.add_argument("--pms", action="store_true", help="partially measurable system: the policy sees noisy positions and filtered "
                    "finite-difference velocities (MC_PILCO4PMS.apply_policy); cart-pole workloads")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--transport", default="torch", choices=["torch", "abi"],
                    help="who carries the step's all-reduce: torch.distributed, or the C ABI's RCCL communicator (mcp_allreduce_grad)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def spawn_ranks(args):
    """`bench.py --gpus N` started bare: run the N ranks as children (one per GPU) and relay rank 0's output.  Nothing in this
    process has touched the GPU (torch is not even imported yet), and the children are fresh interpreters -- no exec of a
    process that holds the device."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = rc or p.wait()
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    raise SystemExit(rc)


def cpu_baseline(problem, M, T, p_drop, threads, budget_s=12.0):
    """The CPU oracle (a port of the reference's PyTorch path, validated against it in
    tests/test_oracle_golden.py) on the same synthetic problem.  threads = 1 is the reference's own
    launch setting (test_mcpilco_cartpole_rbf_ker.py:47-48).  Bounded sample."""
    import numpy as np
    import torch

    from oracle import mcpilco_oracle as orc

    c = problem["cfg"]
    Tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
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
        return {"value": M * T * n / el, "unit": "particle-steps/s", "cores": threads, "kind": "port",
                "sample": "%d policy-gradient step(s) of the same workload (M=%d, T=%d, fwd+cost+bwd, no Adam), torch CPU fp64, %d thread(s)"
                          % (n, M, T, threads),
                "s_per_step": el / n}
    finally:
        torch.set_num_threads(prev)


class Runner:
    """One workload on this rank's GPU: builds it, and runs / times policy-gradient steps."""

    def __init__(self, args, name, dev, rank, world, reducer, M=None, T=None):
        import torch

        from mc_pilco_amd import ops, workloads

        self.torch, self.ops = torch, ops
        self.args, self.name, self.dev, self.rank, self.world, self.reducer = args, name, dev, rank, world, reducer
        self.w = workloads.build(name, device=dev, M=M, T=T)
        self.M, self.T = self.w.M, self.w.T
        self.meas = None
        if args.pms:
            from scipy import signal

            bb, aa = signal.butter(1, 0.5)  # test_mcpilco4pms_cartpole.py:155-157: pos [0,2], vel [1,3], fc 0.5
            self.meas = ops.MeasSpec(pos=[0, 2], vel=[1, 3], std_pos=[0.01, 0.01], b=bb, a=aa)
        # (fused=True: one kernel for the three parameter tensors -- the same update as the reference's "torch.optim.Adam(p, lr)")
        self.opt = torch.optim.Adam(self.w.params, lr=0.01, fused=True)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(1234 + rank)
        self.status_or = torch.zeros(1, dtype=torch.int32, device=dev)
        self.shift = torch.zeros(self.T, dtype=torch.float64, device=dev)
        self.flops = workloads.flops_per_particle_step(self.w)

    def step(self, i, ev):
        torch, ops, w, args = self.torch, self.ops, self.w, self.args
        M, T = self.M, self.T
        x0 = w.sample_x0(M, generator=self.gen)
        eps = masks = None
        if args.noise == "buffers":
            eps = torch.randn(T - 1, M, w.model.G, dtype=torch.float64, device=self.dev, generator=self.gen)
            masks = (torch.rand(T, M, w.policy.B, device=self.dev, generator=self.gen) >= w.p_drop).to(torch.uint8)
        nz = ops.NoiseSpec(eps=eps, masks=masks, seed=2026, call=i + 1, particle_offset=self.rank * M)
        for p in w.params:
            p.grad = None
        if ev is not None:
            ev[0].record()
        states, inputs, status = ops.rollout(w.model, w.policy, nz, x0, T, w.p_drop, meas=self.meas)
        if ev is not None:
            ev[1].record()
        self.status_or |= status
        if self.world > 1:
            # this rank's share of the pooled cost -> its own adjoint sweep -> ONE all-reduce of [gradient | cost sums | flags]
            share, sums = ops.local_cost(w.cost, states, self.world * M, self.shift)
            share.backward()
            sums_all, _flags = self.reducer.reduce(w.params, sums, status.to(torch.float64))
            new_shift = torch.empty_like(self.shift)
            cost = ops.cost_from_sums(sums_all, self.world * M, self.shift, new_shift)[0]
            self.shift = new_shift
        else:
            cost, _std = ops.expected_cost(w.cost, states)
            cost.backward()
        self.opt.step()
        return cost

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist

            dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup):
        """warmup untimed steps, then exactly `steps` timed ones between barriers.  Returns (seconds [max over ranks],
        mean forward-kernel ms from HIP events on the launch stream, last cost)."""
        torch = self.torch
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for i in range(warmup):
            self.step(i, None)
        self.barrier()
        t0 = time.perf_counter()
        cost = None
        for i in range(steps):
            cost = self.step(warmup + i, evs[i])
        self.barrier()
        el = time.perf_counter() - t0
        fwd_ms = sum(a.elapsed_time(b) for (a, b) in evs) / len(evs)
        if self.world > 1:
            import torch.distributed as dist

            tmax = torch.tensor([el], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        st = int(self.status_or.item())
        if st != 0:
            raise SystemExit("bench: kernel status flags %s were raised during the run -- the measurement is invalid" % self.ops.status_flags(self.status_or))
        return el, fwd_ms, float(cost.detach())

    def kernel_name(self):
        from mc_pilco_amd import hipabi

        L = hipabi.lib()
        name = "rollout_fwd_tile_kernel" if L.mcp_debug_last_particles_per_wg() == 16 else "rollout_fwd_kernel"
        return name + (" (GP-sharded)" if L.mcp_debug_last_gp_sharded() else "")

    def roofline(self, fwd_ms):
        ach = self.flops * self.M * self.T / (fwd_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": None,
                "kernel": self.kernel_name(), "kernel_ms": fwd_ms, "alg_flops_per_particle_step": self.flops, "units_per_launch": self.M * self.T,
                "achieved_basis": "algorithmic fwd+bwd flops per particle-step (SURVEY 8d) x M x T / the FORWARD kernel's mean launch time "
                                  "(HIP events on the launch stream); the forward kernel also forms the GP Jacobians the adjoint sweep consumes"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch

    import mcp_boot  # noqa: F401

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    reducer = None
    if world > 1:
        import torch.distributed as dist

        from mc_pilco_amd import sharding

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        reducer = sharding.StepReducer(dist.group.WORLD, args.transport)

    if rank == 0:
        note("building workload %s" % args.workload)
    main_run = Runner(args, args.workload, dev, rank, world, reducer, M=args.particles or None, T=args.horizon or None)
    M, T, w = main_run.M, main_run.T, main_run.w
    el, fwd_ms, last_cost = main_run.run(args.steps, args.warmup)
    if rank == 0:
        note("%s: %d steps in %.3f s" % (args.workload, args.steps, el))
    ms_per_step = 1e3 * el / args.steps
    value = world * M * T / (el / args.steps)

    if rank == 0:
        regimes = {"rollout_fwd_kernel": "per-CU L2->L1 stream of Kinv (phase V) + T-sequential per-step latency; peak = fp64 MFMA/VALU rate",
                   "rollout_fwd_tile_kernel": "fp64 matrix pipe (MFMA 16x16x4) beside VALU exp / Philox phases"}
        roof = main_run.roofline(fwd_ms)
        roof["regime"] = regimes.get(roof["kernel"].split(" ")[0], "")
        out = {
            "metric": "particle-steps/s (M x T per policy-grad step), cart-pole GP",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (RK4 cart-pole rollouts, N=%d training points/GP, fixed trained-like hyper-parameters, on-device %s noise)"
                    % (w.model.gps[0].N, "Philox" if args.noise == "philox" else "torch.randn buffers"),
            "config": {"workload": "%s: %s, %d GPs, D=%d, N=%d, B=%d, M=%d particles/GPU, T=%d, p_dropout=%.2f; step = rollout fwd + cost + "
                                   "adjoint bwd%s + Adam%s" % (args.workload, w.problem["system"], w.model.G, w.model.D, w.model.gps[0].N, w.policy.B, M, T,
                                                            w.p_drop, " + 1 RCCL all-reduce [grad|cost sums|flags]" if world > 1 else "",
                                                            "; policy on measured states (MC_PILCO4PMS)" if args.pms else ""),
                       "particles_per_gpu": M, "horizon": T, "parallelism": "particle-dp%d" % world,
                       "collective": None if world == 1 else {"per_step": 1, "doubles": sum(p.numel() for p in w.params) + 2 * T + 1,
                                                              "transport": args.transport, "backend": args.backend}},
            "roofline": roof,
            "final_cost": last_cost,
        }
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        traffic = {}
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf))
            except Exception:
                traffic = {}
        out["roofline"]["traffic"] = traffic.get(args.workload)
        default_run = world == 1 and args.workload == "c1" and not args.particles and not args.horizon and not args.pms
        if default_run and not args.no_extra:
            # the other single-GPU configurations of BASELINE.json, measured in the same run: c3 (SE+poly(2), M=4000) and c5 (UR5, 6 GPs,
            # D=24, N=400, M=2000, T=300) -- same step definition, fewer steps
            extra = []
            for name in ("c3", "c5"):
                note("extra workload %s" % name)
                r = Runner(args, name, dev, rank, world, reducer)
                k = 5 if name == "c3" else 3
                e2, f2, c2 = r.run(k, 2)
                note("%s: %d steps in %.3f s" % (name, k, e2))
                rf = r.roofline(f2)
                rf.pop("achieved_basis")
                rf["traffic"] = traffic.get(name)
                extra.append({"workload": name, "particles": r.M, "horizon": r.T, "N": r.w.model.gps[0].N, "gps": r.w.model.G,
                              "value": r.M * r.T / (e2 / k), "unit": "particle-steps/s", "ms_per_step": 1e3 * e2 / k, "steps": k, "warmup": 2,
                              "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "frac": rf["frac"], "roofline": rf, "final_cost": c2})
                del r
                torch.cuda.empty_cache()
            out["extra_workloads"] = extra
        if world == 1 and not args.no_cpu:
            # the cores this process may actually run on (the box gives one GPU's share of the host, not os.cpu_count())
            ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            ncores = max(1, min(ncores, int(os.environ.get("MCP_BENCH_MAX_THREADS", "32"))))
            note("cpu_baseline: 1 thread")
            out["cpu_baseline"] = cpu_baseline(w.problem, M, T, w.p_drop, 1)
            if ncores > 1:
                note("cpu_baseline: %d threads" % ncores)
                out["cpu_baseline_all_cores"] = cpu_baseline(w.problem, M, T, w.p_drop, ncores)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        if args.transport == "abi":
            from mc_pilco_amd import hipabi

            hipabi.lib().mcp_comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
