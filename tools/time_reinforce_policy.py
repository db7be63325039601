"""Diagnostic: optimizer steps per second of MC_PILCO.reinforce_policy on the drop-in classes at the headline shape
(cart-pole, N=300, M=400, T=150, B=200) -- the same work as bench.py's step plus the loop's monitors and NaN check.
(bench.py reports the same figure as ``loop_ms_per_step``.)    python tools/time_reinforce_policy.py [steps] [pms]
``--capture-ab``: the loop with its attempts launched eagerly and replayed from HIP graphs (MC_PILCO.capture_attempts), at T = 150 and T = 60."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, torch
from mc_pilco_amd import workloads
dev = torch.device("cuda", 0)
if "--capture-ab" in sys.argv:
    for Tc, name in ((7.5, "c1 (T=150)"), (3.0, "c1_script (T=60)")):
        for cap in (False, True):
            obj, args = workloads.dropin_c1(dev, T_control=Tc)
            obj.capture_attempts = cap
            with contextlib.redirect_stdout(io.StringIO()):
                obj.reinforce_policy(opt_steps_list=[10], **args)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = obj.reinforce_policy(opt_steps_list=[300], **args)
                torch.cuda.synchronize(); el = time.perf_counter() - t0
            print("%-18s attempts %-22s %.4f ms per optimizer step  (replayed %d of 300; cost %.4f -> %.4f)"
                  % (name, "replayed from graphs" if cap else "launched eagerly", el / 300 * 1e3, obj.attempts_replayed, out[0][0], out[0][-1]))
    sys.exit(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pms = len(sys.argv) > 2 and sys.argv[2] == "pms"
s, c0, c1 = workloads.time_reinforce_policy(dev, steps, pms)
print("reinforce_policy%s: %d steps (+1 reference rollout) -> %.2f ms per step, %.3e particle-steps/s; cost %.4f -> %.4f"
      % (" [PMS]" if pms else "", steps, 1e3 * s, 400 * 150 / s, c0, c1))
