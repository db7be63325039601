#!/usr/bin/env python3
"""Benchmark of the MC-PILCO hot path on MI355X: one "step" = one policy-gradient step of
MC_PILCO.reinforce_policy (policy_learning/MC_PILCO.py:484-525) -- fused particle rollout,
expected cost, reverse-time adjoint, [one all-reduce of gradient + cost sums], Adam update -- on
synthetic cart-pole-shaped data.  Metric: particle-steps/s = M*T / step time, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c1_script|c2_script|pms_script|pms_script_n450|c1_script_n360|c2_script_n360|ur5_script|c3|c4|c5] [--no-cpu] [--no-extra]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload.  One GPU: BASELINE.json configs[1] (c1: 4-D state, SE kernel, N=300, M=400 particles, T=150).  N > 1 GPUs:
configs[3] (c4: SE + polynomial(2), 4000 particles PER GPU -- 32 000 over 8 -- T=150), weak scaling; the JSON line of an
N > 1 run carries `one_gpu_same_workload` = the same per-GPU shard timed by rank 0 ALONE on one GPU before the ranks start, and
`scaling_efficiency` = value / (N x that); the N = 1 line carries `scale_base` = that workload's one-GPU value (its extra `c3`),
so a 1 -> N ratio is always formed on ONE workload.  The latency-bound c1 shard is an extra of the N > 1 line.

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself as child processes -- before
this process makes any GPU call (after building the library once, in the parent) -- relays rank 0's JSON line and WATCHES the
children: a rank that dies, or 180 s without a sign of life from ANY rank, ends the run with a non-zero exit code instead of a hang.

Multi-GPU: particles are sharded; the ranks meet in ONE all-reduce per step (RCCL): [gradient | per-time-step cost sums |
status flags].  Before the workload is built every rank runs a PRE-FLIGHT (one 8-byte all-reduce under a time-out) whose
failure names backend and transport.  Prints ONE JSON line on rank 0.

The timed region: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize; blocks are
repeated until >= 1 s has been measured and the MEDIAN block is reported (`blocks`, `block_ms` in the line).
"""
import argparse
import datetime
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (= vector) peak, datasheet; see DESIGN.md
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E, MI355X_MICROARCH.md


def note(msg):
    """Progress line on stderr (stdout carries the one JSON line)."""
    print("[bench %.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="c1 (default on one GPU), c1_script, c2_script, pms_script, ur5_script, c3, c4 (default on N > 1 GPUs), c5")
    ap.add_argument("--particles", type=int, default=0, help="particles per GPU (default: the workload's M)")
    ap.add_argument("--horizon", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_workloads of the default runs")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the K-step block until this much has been measured")
    ap.add_argument("--noise", default="philox", choices=["philox", "buffers"])
    ap.add_argument("--pms", action="store_true", help="partially measurable system: the policy sees noisy positions and filtered "
                    "finite-difference velocities (MC_PILCO4PMS.apply_policy); cart-pole workloads")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--transport", default="torch", choices=["torch", "abi"],
                    help="who carries the step's all-reduce: torch.distributed, or the C ABI's RCCL communicator (mcp_allreduce_grad)")
    ap.add_argument("--collective-smoke", action="store_true",
                    help="N > 1: additionally time the step's message through the OTHER transport (the C ABI's communicator when the run "
                         "uses torch.distributed and vice versa) and report both latencies")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--watchdog-seconds", type=float, default=180.0, help="spawned ranks: give up after this long without progress")
    return ap.parse_args()


def spawn_ranks(args):
    """`bench.py --gpus N` started bare: run the N ranks as children (one per GPU), relay rank 0's output and watch them.
    Nothing in this process has touched the GPU (torch is not even imported yet), and the children are fresh interpreters --
    no exec of a process that holds the device.  A child that exits non-zero, or `--watchdog-seconds` without any sign of
    life (no child finishing, no new stderr/stdout byte from rank 0), terminates the rest: exit code != 0, never a hang."""
    import selectors
    import socket

    # the library is built HERE, once, before any rank starts: a first-time hipcc build inside the children (minutes of silence)
    # would look like a hang to the watchdog, and N ranks would race for the same object files.  Compiling does not touch the GPU.
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "mc-pilco_amd", "build.py")], stdout=sys.stderr)
    except subprocess.CalledProcessError as e:
        note("building libmcpilco_hip.so failed (exit code %d)" % e.returncode)
        raise SystemExit(e.returncode or 1)
    # the rendezvous port: kept BOUND (SO_REUSEADDR / SO_REUSEPORT) until the children have been started, so that nobody else is
    # handed the same port in between; rank 0's store binds it with the same options
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    if hasattr(socket, "SO_REUSEPORT"):
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEPORT, 1)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE))
    s.close()
    # a sign of life = a byte on rank 0's stdout or on ANY rank's stderr (every rank writes progress notes there), or a child exiting
    sel = selectors.DefaultSelector()
    streams = [procs[0].stdout] + [p.stderr for p in procs]
    for f in streams:
        os.set_blocking(f.fileno(), False)
        sel.register(f, selectors.EVENT_READ)
    out = b""
    last_life = time.monotonic()
    open_streams = len(streams)
    rc = None
    while rc is None:
        for key, _ in sel.select(timeout=1.0):
            data = key.fileobj.read()
            if data:
                last_life = time.monotonic()
                if key.fileobj is procs[0].stdout:
                    out += data
                else:
                    sys.stderr.buffer.write(data)
                    sys.stderr.flush()
            elif data == b"":
                sel.unregister(key.fileobj)
                open_streams -= 1
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            note("rank %d exited with code %d -- terminating the other ranks" % bad[0])
            rc = bad[0][1] or 1
        elif all(c == 0 for c in codes) and open_streams == 0:
            rc = 0
        elif time.monotonic() - last_life > args.watchdog_seconds:
            note("watchdog: no sign of life from any rank for %.0f s -- terminating all ranks" % args.watchdog_seconds)
            rc = 124
    if rc != 0:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    raise SystemExit(rc)


def kernel_sources_sha16():
    """Identity of the kernel sources a measurement belongs to (profiles/traffic.json is only quoted when it matches)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mc-pilco_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "mcpilco_hip.h"), "rb").read())
    return h.hexdigest()[:16]


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

    STATUS_BITS = 4  # NaN, non-positive variance, not SPD, hand-off time-out (include/mcpilco_hip.h)

    def __init__(self, args, name, dev, rank, world, reducer, M=None, T=None):
        import torch

        from mc_pilco_amd import ops, sharding, workloads

        self.torch, self.ops, self.sharding = torch, ops, sharding
        self.args, self.name, self.dev, self.rank, self.world, self.reducer = args, name, dev, rank, world, reducer
        self.w = workloads.build(name, device=dev, M=M, T=T)
        self.M, self.T = self.w.M, self.w.T
        self.meas = self.w.meas  # (pms_script carries its own measurement model)
        if args.pms:
            from scipy import signal

            bb, aa = signal.butter(1, 0.5)  # test_mcpilco4pms_cartpole.py:155-157: pos [0,2], vel [1,3], fc 0.5
            self.meas = ops.MeasSpec(pos=[0, 2], vel=[1, 3], std_pos=[0.01, 0.01], b=bb, a=aa)
        # (fused=True: one kernel for the three parameter tensors -- the same update as the reference's "torch.optim.Adam(p, lr)")
        self.opt = torch.optim.Adam(self.w.params, lr=0.01, fused=True)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(1234 + rank)
        self.status_or = torch.zeros(1, dtype=torch.int32, device=dev)
        self.flags_sum = torch.zeros(1 + self.STATUS_BITS, dtype=torch.float64, device=dev)  # reduced flags of every step, on every rank alike
        self.shift = torch.zeros(self.T, dtype=torch.float64, device=dev)
        self.flops = workloads.flops_per_particle_step(self.w)
        self.flops_fwd = workloads.flops_per_particle_step(self.w, forward_only=True)
        self.bits = torch.tensor([1 << b for b in range(self.STATUS_BITS)], dtype=torch.int32, device=dev)
        # the step's flat message [gradient | cost sums | flags], allocated once: the adjoint sweep writes its gradients into the head (the
        # parameters' .grad are views of it), mcp_cost_sums into the middle; a sharded step all-reduces it in place (sharding.StepMessage)
        self.msg = sharding.StepMessage(self.w.policy.grad_numel(), self.T, 1 + self.STATUS_BITS, dev)
        self.w.policy.grad_flat = self.msg.grad

    def step(self, i, ev, sharded=None):
        torch, ops, w, args = self.torch, self.ops, self.w, self.args
        M, T = self.M, self.T
        sharded = (self.world > 1) if sharded is None else sharded
        x0 = w.sample_x0(M, generator=self.gen)
        eps = masks = None
        if args.noise == "buffers":
            eps = torch.randn(T - 1, M, w.model.G, dtype=torch.float64, device=self.dev, generator=self.gen)
            masks = (torch.rand(T, M, w.policy.B, device=self.dev, generator=self.gen) >= w.p_drop).to(torch.uint8)
        nz = ops.NoiseSpec(eps=eps, masks=masks, seed=2026, call=i + 1, particle_offset=self.rank * M)
        for p in w.params:
            p.grad = None
        # HIP events on the launch stream right around mcp_rollout_fwd / mcp_rollout_bwd (ops.fwd_events / ops.bwd_events)
        ops.fwd_events, ops.bwd_events = (None, None) if ev is None else ((ev[0], ev[1]), (ev[2], ev[3]))
        # (the kernels OR their flags into ONE word for the whole run -- checked after the timed blocks, check_flags -- instead of a fresh word and an
        #  OR launch per step: 8 us of launches that are the bench's own bookkeeping, not the path's)
        states, inputs, status = ops.rollout(w.model, w.policy, nz, x0, T, w.p_drop, meas=self.meas, status=self.status_or)
        if sharded:
            # this rank's share of the pooled cost -> its own adjoint sweep -> ONE all-reduce of [gradient | cost sums | flags]; the
            # flags are one 0/1 entry per status bit (summed doubles are counts, not an OR) + "local share is NaN"
            share, sums = ops.local_cost(w.cost, states, self.world * M, self.shift, sums_out=self.msg.sums)
            share.backward()
            fl = torch.cat([torch.isnan(share.detach()).reshape(1).to(torch.float64), ((status & self.bits) != 0).to(torch.float64)])
            cost, _std, fl_all, self.shift = self.sharding.finish_step(_SumsCost, self.reducer, w.params, sums, fl, self.world * M, self.shift,
                                                                       msg=self.msg)
            self.flags_sum += fl_all
        else:
            cost, _std = ops.expected_cost(w.cost, states)
            cost.backward()
        ops.fwd_events = ops.bwd_events = None
        self.opt.step()
        return cost

    def barrier(self, solo=False):
        if self.world > 1 and not solo:
            import torch.distributed as dist

            dist.barrier()
        self.torch.cuda.synchronize()

    def check_flags(self):
        """After the timed blocks: any status bit raised on ANY rank invalidates the measurement -- decided alike on every rank
        (the flags travelled with every step's all-reduce), so no rank is left waiting in a collective."""
        bad = int(self.status_or.item()) != 0
        if self.world > 1:
            bad = bad or bool((self.flags_sum > 0).any().item())
        if bad:
            raise SystemExit("bench: kernel status flags were raised during the run (local %s, reduced counts %s) -- the measurement is invalid"
                             % (self.ops.status_flags(self.status_or), self.flags_sum.tolist()))

    def run(self, steps, warmup, min_seconds=0.0, sharded=None, solo=False):
        """``solo``: this rank alone (no barrier / time exchange with the other ranks -- they wait elsewhere).
        warmup untimed steps, then blocks of exactly `steps` timed ones between barriers, repeated until `min_seconds` have been
        measured (the number of blocks follows from the first block's time, maximum over ranks, so every rank runs the same).
        Returns (median block seconds [max over ranks], mean forward-kernel ms from HIP events on the launch stream, last cost,
        all block seconds)."""
        torch = self.torch
        sharded = (self.world > 1) if sharded is None else sharded
        self.rank_block_s = []
        for i in range(warmup):
            self.step(i, None, sharded)
        blocks, fwd, bwd, cost, k, nblocks = [], [], [], None, warmup, 1
        while len(blocks) < nblocks:
            evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(steps)]
            self.barrier(solo)
            t0 = time.perf_counter()
            for i in range(steps):
                cost = self.step(k + i, evs[i], sharded)
            self.barrier(solo)
            el = time.perf_counter() - t0
            k += steps
            if self.world > 1 and not solo:  # (also without the step's collective: every rank must arrive at the same number of blocks)
                import torch.distributed as dist

                tmax = torch.tensor([el, -el], dtype=torch.float64, device=self.dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                self.rank_block_s.append((-float(tmax[1].item()), float(tmax[0].item())))  # (fastest rank, slowest rank) of this block
                el = float(tmax[0].item())
            blocks.append(el)
            fwd.append(sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs))
            bwd.append(sum(e[2].elapsed_time(e[3]) for e in evs) / len(evs))
            if len(blocks) == 1 and min_seconds > 0:
                nblocks = max(1, min(50, int(min_seconds / max(el, 1e-6)) + 1))
        self.check_flags()
        order = sorted(range(len(blocks)), key=lambda j: blocks[j])
        mid = order[len(order) // 2]
        self.rank_block_mid = self.rank_block_s[mid] if self.rank_block_s else None
        self.bwd_ms = bwd[mid]
        return blocks[mid], fwd[mid], float(cost.detach()), blocks

    def kernel_name(self):
        from mc_pilco_amd import hipabi

        L = hipabi.lib()
        if L.mcp_debug_last_fwd_lean():
            return "rollout_fwd_lat_kernel (GP-sharded)"
        name = "rollout_fwd_tile_kernel" if L.mcp_debug_last_particles_per_wg() == 16 else "rollout_fwd_kernel"
        rs = L.mcp_debug_last_row_split()
        if rs:
            return name + " (GP-sharded, %d workgroups per (tile, GP) on row parts of Kinv)" % rs
        return name + (" (GP-sharded)" if L.mcp_debug_last_gp_sharded() else "")

    def bwd_kernel_name(self):
        from mc_pilco_amd import hipabi

        return "rollout_bwd_lat_kernel" if hipabi.lib().mcp_debug_last_bwd_lean() else "rollout_bwd_kernel"

    def roofline(self, fwd_ms, step_s, traffic, profile=None):
        """The dominant kernel is the forward rollout.  `frac` = ITS OWN algorithmic flops (the forward terms of SURVEY 8d's figure)
        x M x T / its mean launch duration (HIP events on the launch stream right around mcp_rollout_fwd) / the fp64 peak -- the
        number that follows from profiles/*_kernel_stats.csv.  `kernels` lists forward and backward the same way; `frac_step`
        prices the whole step's flops (fwd + bwd) against ms_per_step (forward + cost + adjoint + Adam [+ all-reduce])."""
        name = self.kernel_name()
        small = not name.startswith("rollout_fwd_tile")
        units = self.M * self.T
        flops_bwd = self.flops - self.flops_fwd
        ach_fwd = self.flops_fwd * units / (fwd_ms * 1e-3) / 1e12
        ach_bwd = flops_bwd * units / (self.bwd_ms * 1e-3) / 1e12
        ach_step = self.flops * units / step_s / 1e12
        c = self.w.problem["cfg"]
        alg_bytes = 16 * (c["S"] + c["U"] + c["G"]) * units  # B_alg = 16 (S + U + G) per particle-step
        r = {"bound": "mfma", "bound_in_practice": "l2-stream+latency" if small else "fp64 units (mfma)", "achieved": ach_fwd,
             "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_fwd / FP64_PEAK_TFLOPS, "frac_step": ach_step / FP64_PEAK_TFLOPS,
             "traffic": traffic, "alg_bytes_per_launch": alg_bytes, "traffic_ratio": (traffic / alg_bytes) if traffic else None,
             "kernel": name, "kernel_ms": fwd_ms, "alg_flops_per_particle_step": self.flops_fwd, "units_per_launch": units,
             "alg_flops_fwd_plus_bwd_per_particle_step": self.flops,
             "kernels": [{"name": name, "avg_ms": fwd_ms, "alg_flops_per_unit": self.flops_fwd, "frac": ach_fwd / FP64_PEAK_TFLOPS},
                         {"name": self.bwd_kernel_name(), "avg_ms": self.bwd_ms, "alg_flops_per_unit": flops_bwd, "frac": ach_bwd / FP64_PEAK_TFLOPS,
                          "note": "mcp_rollout_bwd: the adjoint sweep(s) + the fixed-order gradient reduction"}],
             "achieved_basis": "the forward kernel's own algorithmic flops per particle-step (SURVEY 8d, forward terms) x M x T / its mean launch "
                               "duration (HIP events on the launch stream around mcp_rollout_fwd; profiles/r06_*_kernel_stats.csv is the rocprofv3 "
                               "average of the same kernel); frac_step: fwd+bwd flops / ms_per_step",
             "regime": ("T-sequential chain of 4 barrier-separated phases per step + the per-CU L2->CU stream of one Kinv per workgroup and "
                        "step; fp64 flop roof not reachable at M=400 (DESIGN.md 4.1)") if small else
                       "fp64 matrix pipe (MFMA 16x16x4) beside VALU exp / Philox phases"}
        if traffic:
            gbps = traffic / (fwd_ms * 1e-3) / 1e9
            r["hbm_gbps"], r["hbm_frac"] = gbps, gbps / HBM_PEAK_GBPS
        if name.startswith("rollout_fwd_lat"):
            r["l1_fill"] = self.l1_fill(fwd_ms, profile or {})
        return r

    def l1_fill(self, fwd_ms, profile):
        """The second roof of the lean small-swarm kernel: every workgroup pulls its GP's Kinv through its CU's L1 once per
        time step, less what stays in registers -- 8 waves x 3 resident register buffers x 6 KB --, and a CU's L1 delivers 64 B per clock.  `bytes_per_wg_step`
        / 64 = the cycles that stream needs per step; `cycles_per_step` = the forward kernel's measured time per step at 2.4 GHz; from
        profiles/traffic.json (tools/collect_profiles.sh, same kernel sources only) the phase's own cycles with both sides, with the MFMAs compiled out
        (stream only) and with the loads compiled out (MFMAs only), and the launch's L1 -> L2 read requests (counter TCP_TCC_READ_REQ_sum, 128-byte lines)."""
        npad = max(g.Npad for g in self.w.model.gps)
        kinv = npad * npad * 8
        resident = 8 * 3 * 6 * 64 * 16
        streamed = max(0, kinv - resident)
        out = {"bound": "per-CU L1 fill (64 B/clk) of one Kinv per workgroup and time step", "kinv_bytes": kinv, "register_resident_bytes": resident,
               "bytes_per_wg_step": streamed, "cycles_at_64B_clk": streamed / 64.0, "clock_ghz": 2.4,
               "cycles_per_step": fwd_ms * 1e-3 * 2.4e9 / max(1, self.T - 1),
               "stream_share_of_step": (streamed / 64.0) / (fwd_ms * 1e-3 * 2.4e9 / max(1, self.T - 1))}
        pv = profile.get("c1_phase_v_cycles") if self.name == "c1" else None
        if pv:
            out["measured_cycles"] = {"phase_v_plus_j": pv["both"]["v_plus_j"], "phase_v_slowest_wave": pv["both"]["slowest_wave_v"],
                                      "stream_only_slowest_wave": pv["stream_only"]["slowest_wave_v"], "mfma_only_slowest_wave": pv["mfma_only"]["slowest_wave_v"],
                                      "what": "tools/phase_stamps.py c1 on the kernel and on its RLX_NOFMA / RLX_NOLOAD experiment builds (profiles/r06_c1_*stamps.txt)"}
        if self.name == "c1" and profile.get("c1_tcp_tcc_read_req"):
            units = 200 * (self.T - 1)
            out["tcp_tcc_read_req_per_launch"] = profile["c1_tcp_tcc_read_req"]
            # (a request = one 128-byte line: 1.405e8 per launch / (200 workgroups x 149 steps) x 128 B = 603 KB, the streamed 592 KB + the start-up)
            out["l1_to_l2_bytes_per_wg_step_at_128B_per_request"] = profile["c1_tcp_tcc_read_req"] * 128.0 / units
        if self.name == "c1" and profile.get("c1_mfma_pipe_busy") is not None:
            out["mfma_pipe_busy"] = profile["c1_mfma_pipe_busy"]
        return out


class _SumsCost:
    """`from_sums` of sharding.finish_step through the C ABI (mcp_cost_finalize_sums)."""

    @staticmethod
    def from_sums(sums, n_total, shift=None, mean_out=None):
        from mc_pilco_amd import ops

        out = ops.cost_from_sums(sums, n_total, shift, mean_out)
        return out[0], out[1]


def preflight(dist, torch, dev, args, rank, world):
    """One 8-byte all-reduce before anything is built: a broken transport shows up here, with a message, inside the process
    group's time-out -- not as a hang in step 0."""
    t0 = time.perf_counter()
    try:
        x = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(x)
        torch.cuda.synchronize()
        seen = int(x.item())
        if seen != world:
            raise RuntimeError("all-reduce of ones returned %s on rank %d, expected %d" % (x.item(), rank, world))
    except Exception as e:  # noqa: BLE001
        raise SystemExit("bench: pre-flight all-reduce failed on rank %d (backend %s, transport %s, world %d, device %s): %r"
                         % (rank, args.backend, args.transport, world, dev, e))
    return time.perf_counter() - t0, seen


def time_collective(torch, reducer, n, dev, calls=50):
    """Mean latency of the step's message (n doubles) through `reducer`, HIP-event timed, after 5 warm-up calls."""
    buf = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(5):
        reducer.allreduce_(buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        reducer.allreduce_(buf)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / calls


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MCP_BENCH_TEST_HANG") == str(rank):  # test hook of the watchdog (tests/test_bench_cpu.py): this rank never shows up
        time.sleep(3600)
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    default_workload = args.workload is None
    if default_workload:
        args.workload = "c1" if world == 1 else "c4"

    import torch

    import mcp_boot  # noqa: F401

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    reducer = None
    collective = None
    if world > 1:
        import torch.distributed as dist

        from mc_pilco_amd import sharding

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=120)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=tmo)
        except Exception as e:  # noqa: BLE001
            raise SystemExit("bench: init_process_group(%s) failed on rank %d of %d: %r" % (args.backend, rank, world, e))
        pf, ranks_seen = preflight(dist, torch, dev, args, rank, world)
        if rank == 0:
            note("pre-flight all-reduce over %d ranks (%s): %.2f s" % (world, args.backend, pf))
        reducer = sharding.StepReducer(dist.group.WORLD, args.transport)

    note("rank %d: building workload %s" % (rank, args.workload))  # (every rank: its stderr is the launcher's sign of life)
    main_run = Runner(args, args.workload, dev, rank, world, reducer, M=args.particles or None, T=args.horizon or None)
    M, T, w = main_run.M, main_run.T, main_run.w
    nmsg = sum(p.numel() for p in w.params) + 2 * T + 1 + Runner.STATUS_BITS
    if world > 1:
        us = time_collective(torch, reducer, nmsg, dev)
        from mc_pilco_amd import hipabi

        # what the collective actually saw: the process group's size, the sum of ones of the pre-flight all-reduce, and (transport
        # "abi") the size of the C ABI's own RCCL communicator; a second all-reduce of ones through the step's OWN reducer, too
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        reducer.allreduce_(ones)
        collective = {"per_step": 1, "doubles": nmsg, "transport": args.transport, "backend": args.backend, "us_per_call": us,
                      "world": dist.get_world_size(), "ranks_seen": ranks_seen, "ranks_seen_by_step_reducer": int(ones.item()),
                      "transport_world": hipabi.lib().mcp_comm_world() if args.transport == "abi" else None,
                      "devices": "one GPU per rank" if not args.single_device else "ALL ranks on cuda:0 (rehearsal)"}
        if args.collective_smoke:
            other = "abi" if args.transport == "torch" else "torch"
            try:
                from mc_pilco_amd import sharding

                collective["us_per_call_" + other] = time_collective(torch, sharding.StepReducer(dist.group.WORLD, other), nmsg, dev)
            except Exception as e:  # noqa: BLE001  (a smoke: reported, not fatal)
                collective["error_" + other] = repr(e)
    one_gpu = None
    if world > 1 and not args.no_extra:
        # the SAME per-GPU shard on ONE GPU, alone: rank 0 times it (no collective, median of blocks) while the other ranks wait at the
        # barrier -- N x this value is perfect weak scaling of this workload
        if rank == 0:
            k1 = max(3, args.steps // 2)
            saved = [p.detach().clone() for p in w.params]  # (rank 0's solo steps must not leave it with other parameters than its peers)
            s1, f1, _, b1 = main_run.run(k1, 2, min(args.min_seconds, 1.0), sharded=False, solo=True)
            with torch.no_grad():
                for p, q in zip(w.params, saved):
                    p.copy_(q)
            main_run.opt = torch.optim.Adam(w.params, lr=0.01, fused=True)
            one_gpu = {"what": "the same %s shard (M=%d, T=%d) on ONE GPU without the collective, measured by rank 0 ALONE before the ranks start "
                               "(the others wait at a barrier), median of %d block(s) of %d steps" % (args.workload, M, T, len(b1), k1),
                       "workload": args.workload, "value": M * T / (s1 / k1), "unit": "particle-steps/s", "ms_per_step": 1e3 * s1 / k1, "kernel_ms": f1}
            note("one GPU alone, %s: %.3f ms per step" % (args.workload, 1e3 * s1 / k1))
        import torch.distributed as dist

        dist.barrier()
    block_s, fwd_ms, last_cost, blocks = main_run.run(args.steps, args.warmup, args.min_seconds)
    if rank != 0:
        note("rank %d: timed blocks done" % rank)
    if rank == 0:
        note("%s: %d block(s) of %d steps, median %.3f s" % (args.workload, len(blocks), args.steps, block_s))
    ms_per_step = 1e3 * block_s / args.steps
    value = world * M * T / (block_s / args.steps)

    traffic_all, sha = {}, kernel_sources_sha16()
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            traffic_all = json.load(open(tf))
        except Exception:  # noqa: BLE001
            traffic_all = {}
    traffic_ok = traffic_all.get("kernel_sources_sha16") == sha  # counters of OTHER kernel sources are not quoted
    tkey = {"c4": "c3", "c1_script": None}.get(args.workload, args.workload)
    traffic_of = lambda k: traffic_all.get(k) if (traffic_ok and k) else None

    # every rank takes part in the extras that involve all ranks (none do: the extras below are single-rank work on rank 0's
    # schedule, so the other ranks simply wait at the final barrier); only rank 0 prints
    out = None
    if rank == 0:
        roof = main_run.roofline(fwd_ms, block_s / args.steps, traffic_of(tkey), traffic_all if traffic_ok else None)
        if not traffic_ok:
            roof["traffic_note"] = "profiles/traffic.json was collected for other kernel sources (%s != %s): not quoted" % (
                traffic_all.get("kernel_sources_sha16"), sha)
        out = {
            "metric": "particle-steps/s (M x T per policy-grad step), cart-pole GP",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (RK4 cart-pole rollouts, N=%d training points/GP, fixed trained-like hyper-parameters, on-device %s noise)"
                    % (w.model.gps[0].N, "Philox" if args.noise == "philox" else "torch.randn buffers"),
            "config": {"workload": "%s: %s, %d GPs, D=%d, N=%d, B=%d, poly degree %d, M=%d particles/GPU, T=%d, p_dropout=%.2f; step = rollout fwd + cost + "
                                   "adjoint bwd%s + Adam%s" % (args.workload, w.problem["system"], w.model.G, w.model.D, w.model.gps[0].N, w.policy.B,
                                                            w.problem["deg"], M, T, w.p_drop,
                                                            (" + 1 %s all-reduce [grad|cost sums|flags]" % ("RCCL" if args.backend == "nccl" or args.transport == "abi"
                                                                                                                    else args.backend)) if world > 1 else "",
                                                            "; policy on measured states (MC_PILCO4PMS)" if args.pms else ""),
                       "particles_per_gpu": M, "particles_total": world * M, "horizon": T, "parallelism": "particle-dp%d" % world,
                       "collective": collective},
            "blocks": len(blocks), "block_ms": [1e3 * b for b in blocks],
            "rank_ms_per_step": None if not main_run.rank_block_mid else {"fastest_rank": 1e3 * main_run.rank_block_mid[0] / args.steps,
                                                                         "slowest_rank": 1e3 * main_run.rank_block_mid[1] / args.steps},
            "roofline": roof,
            "kernel_sources_sha16": sha,
            "final_cost": last_cost,
        }
    if world > 1 and not args.no_extra:
        # the latency-bound c1 shard (M = 400 per GPU), sharded, for comparison with the single-GPU headline (lockstep on every rank:
        # it contains collectives)
        c1 = Runner(args, "c1", dev, rank, world, reducer)
        c1_s, c1_f, _, _ = c1.run(args.steps, 2, 0.0)
        if rank == 0:
            out["one_gpu_same_workload"] = one_gpu
            out["scaling_efficiency"] = value / (world * one_gpu["value"])
            out["extra_workloads"] = [{"workload": "c1 (M=%d per GPU, sharded over %d GPUs: latency bound)" % (c1.M, world), "value": world * c1.M * c1.T / (c1_s / args.steps),
                                       "unit": "particle-steps/s", "ms_per_step": 1e3 * c1_s / args.steps, "kernel": c1.kernel_name(), "kernel_ms": c1_f}]
        del c1
    if rank == 0:
        default_run = world == 1 and default_workload and not args.particles and not args.horizon and not args.pms
        if default_run and not args.no_extra:
            # the other single-GPU configurations of BASELINE.json, measured in the same run: the launch script's own horizon (T=60,
            # SURVEY 8d), c3 (SE+poly(2), M=4000) and c5 (UR5, 6 GPs, D=24, N=400, M=2000, T=300) -- same step definition, fewer steps
            extra = []
            # (same rule as the headline: blocks of exactly k steps, repeated until >= min-seconds have been measured, median block)
            for name, k in (("c1_script", 20), ("c2_script", 20), ("pms_script", 20), ("pms_script_n450", 20), ("c1_script_n360", 20), ("c2_script_n360", 20),
                            ("ur5_script", 10), ("c3", 5), ("c5", 3)):
                note("extra workload %s" % name)
                r = Runner(args, name, dev, rank, world, reducer)
                e2, f2, c2, b2 = r.run(k, 2, args.min_seconds)
                note("%s: median of %d block(s) of %d steps: %.3f s" % (name, len(b2), k, e2))
                rf = r.roofline(f2, e2 / k, traffic_of(name if name in ("c3", "c5") else None))
                rf.pop("achieved_basis")
                extra.append({"workload": name, "particles": r.M, "horizon": r.T, "N": r.w.model.gps[0].N, "gps": r.w.model.G,
                              "poly_degree": r.w.problem["deg"], "measured_states": r.meas is not None,
                              "value": r.M * r.T / (e2 / k), "unit": "particle-steps/s", "ms_per_step": 1e3 * e2 / k,
                              "us_per_time_step": 1e3 * 1e3 * e2 / k / r.T, "steps": k, "warmup": 2, "blocks": len(b2),
                              "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "frac": rf["frac"], "frac_step": rf["frac_step"],
                              "roofline": rf, "final_cost": c2})
                del r
                torch.cuda.empty_cache()
            out["extra_workloads"] = extra
            c3x = [e for e in extra if e["workload"] == "c3"][0]
            out["scale_base"] = {"workload": "c3 (= c4's per-GPU shard: the workload `bench.py --gpus N>1` runs on every GPU)", "value": c3x["value"],
                                 "unit": "particle-steps/s", "ms_per_step": c3x["ms_per_step"],
                                 "what": "the one-GPU value a multi-GPU line of this bench must be divided by (its own `one_gpu_same_workload` "
                                         "re-measures it); the headline `value` above is c1 (M=400), a different workload"}
            # the drop-in class's own loop (monitors, NaN test, printing, Adam): MC_PILCO.reinforce_policy, 100 steps, same shape
            note("MC_PILCO.reinforce_policy on the drop-in classes (100 steps)")
            from mc_pilco_amd import workloads

            s_loop, c_first, c_last = workloads.time_reinforce_policy(dev, 300)
            out["loop_ms_per_step"] = 1e3 * s_loop
            out["loop"] = {"what": "MC_PILCO.reinforce_policy of the drop-in package, 300 optimizer steps at the c1 shape (monitors, NaN retry "
                                   "logic, lr / exit test from step 200 on, Adam, printing included; outcome of each attempt read one attempt late)",
                           "value": M * T / s_loop, "unit": "particle-steps/s", "cost_first": c_first, "cost_last": c_last,
                           "over_bench_step": 1e3 * s_loop / ms_per_step - 1.0}
            s_loop2, _, _ = workloads.time_reinforce_policy(dev, 300, T_control=3.0)
            out["loop_c1_script"] = {"what": "the same at the launch script's horizon (T = 60)", "loop_ms_per_step": 1e3 * s_loop2,
                                     "value": M * 60 / s_loop2, "unit": "particle-steps/s",
                                     "over_bench_step": 1e3 * s_loop2 / extra[0]["ms_per_step"] - 1.0}
            note("GP hyper-parameter training (fit_model), N=300, 2 GPs, 100 epochs each")
            s_ep, n_tr = workloads.time_fit_model(dev, 300, 100)
            out["fit_model"] = {"what": "Model_learning.reinforce_model on the drop-in package: Adam on the marginal likelihood, full batch, "
                                        "N=%d training points, D=6, 2 GPs, 100 epochs each (both trained epoch-synchronously: mcp_nll_epoch + one Adam launch per epoch)" % n_tr,
                                "ms_per_epoch_per_gp": 1e3 * s_ep, "epochs_per_s": 1.0 / s_ep}
            note("GP hyper-parameter training (fit_model), UR5 shape: N=400, D=24, 6 GPs, SE+poly(1), 200 epochs each")
            s_ep6, n_tr6 = workloads.time_fit_model_ur5(dev, 400, 200)
            out["fit_model_ur5"] = {"what": "the same for the UR5-shaped model: N=%d, D=24, SE + polynomial(1), 6 GPs, 200 epochs each (all six trained "
                                            "epoch-synchronously: mcp_nll_epoch)" % n_tr6,
                                    "ms_per_epoch_all_6_gps": 1e3 * s_ep6, "ms_per_epoch_per_gp": 1e3 * s_ep6 / 6}
        if default_run and not args.no_extra:
            note("pretrain_gp with SOD (cart-pole N=300, UR5 N=600)")
            from mc_pilco_amd import workloads

            pre = {}
            for shape in ("cartpole", "ur5"):
                pre[shape] = workloads.time_pretrain(dev, shape)
            out["pretrain"] = {"what": "Model_learning.pretrain_gp on the drop-in classes with the launch scripts' SOD settings (cart-pole: N=300, relative "
                                       "0.5; UR5 shape: N=600, D=24, SE+poly(1), absolute 0.001): seconds per GP for the whole call (subset selection, Gram, "
                                       "Cholesky, inverse, alpha, pack, posterior at all rows, the MSE print's host syncs) and the device time of each stage "
                                       "(HIP events, median of 5)", "reference_cpu_s_per_gp_n300": 0.56,
                               "reference_cpu_source": "BASELINE.md section 2 (survey container, 1 thread)", **pre}
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
        from mc_pilco_amd import hipabi

        if hipabi.lib().mcp_comm_world() != 0:
            hipabi.lib().mcp_comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
