"""bench.py's multi-rank launcher on a machine without a GPU: it must FAIL, quickly and with a non-zero exit code -- a rank that
dies takes the others with it, and a rank that never shows up is caught by the watchdog ("make the multi-GPU run
unfailable": no hang, whatever goes wrong)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags, timeout=120):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu", "--no-extra"] + list(flags),
                       env=env, capture_output=True, text=True, timeout=timeout)
    return p, time.time() - t0


def test_a_dying_rank_ends_the_run_with_a_nonzero_code():
    p, el = _run({})
    assert p.returncode != 0 and el < 90
    assert "needs a GPU" in p.stderr or "exited with code" in p.stderr


def test_a_rank_that_never_arrives_is_caught_by_the_watchdog():
    p, el = _run({"MCP_BENCH_TEST_HANG": "0"}, "--watchdog-seconds", "4")  # rank 0 silent for ever; rank 1 would fail on its own...
    assert p.returncode != 0 and el < 90
    p, el = _run({"MCP_BENCH_TEST_HANG": "0", "CUDA_VISIBLE_DEVICES": ""}, "--watchdog-seconds", "4")
    assert p.returncode != 0 and el < 90
