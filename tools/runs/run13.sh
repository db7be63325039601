set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_dropin.py -q -x -s -k "pipelined or another_optimizer or reinforce" > $O/r6_graph_tests.log 2>&1; echo "rc=$?" >> $O/r6_graph_tests.log
tail -30 $O/r6_graph_tests.log
python - > $O/r6_loop_times.txt 2>&1 <<'P'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import mcp_boot, torch
from mc_pilco_amd import workloads
from mc_pilco_amd.policy_learning import MC_PILCO
dev = torch.device("cuda", 0)
for Tc, name in ((7.5, "c1 (T=150)"), (3.0, "c1_script (T=60)")):
    for cap in (False, True):
        MC_PILCO.MC_PILCO.capture_default = cap
        import contextlib, io
        obj, args = workloads.dropin_c1(dev, T_control=Tc)
        obj.capture_attempts = cap
        with contextlib.redirect_stdout(io.StringIO()):
            obj.reinforce_policy(opt_steps_list=[10], **args)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = obj.reinforce_policy(opt_steps_list=[300], **args)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        print("%-18s capture %-5s  %.4f ms per optimizer step  (replayed %d of 300, cost %.6f -> %.6f)" % (name, cap, el / 300 * 1e3, obj.attempts_replayed, out[0][0], out[0][-1]))
P
cat $O/r6_loop_times.txt
