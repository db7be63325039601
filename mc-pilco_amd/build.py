"""Builds libmcpilco_hip.so (gfx950) in-tree with hipcc.  No GPU is needed to compile.

    python mc-pilco_amd/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gp_pretrain.hip", "rollout_fwd.hip", "rollout_fwd_lean.hip", "rollout_fwd_tile.hip", "rollout_bwd.hip", "cost.hip", "policy_opt.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "mcp_device.h"), os.path.join(CSRC, "rollout_common.h"), os.path.join(CSRC, "rollout_fwd_shared.h"),
           os.path.join(os.path.dirname(HERE), "include", "mcpilco_hip.h"),
           os.path.join(os.path.dirname(HERE), "include", "mcpilco_hip_debug.h")]  # (mcp_dispatch: every translation unit must see the same layout)
LIB = os.path.join(HERE, "libmcpilco_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -disable-machine-licm: the machine-level loop-invariant hoisting pulls the ~60 double-precision literals of exp / sincos / log /
# tanh out of the time-step loop into VGPRs, which then spill to scratch and are reloaded (global-latency) in every phase
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-mllvm", "-disable-machine-licm"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(tag, defines, verbose=True, only=None):
    """Experiment helper: a second library libmcpilco_hip_<tag>.so with extra -D defines (select it with MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=<path>).
    ``only``: the sources the defines concern (the others are linked from the main build's objects)."""
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if only is not None and src not in only:
            objs.append(os.path.join(CSRC, src.replace(".hip", ".o")))
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".%s.o" % tag))
        if _stale(o, [s] + HEADERS):
            cmd = [HIPCC] + FLAGS + ["-D" + d for d in defines] + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    lib = os.path.join(HERE, "libmcpilco_hip_%s.so" % tag)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", lib])
    return lib


def _compile_all(jobs, verbose):
    """Runs the hipcc commands of the stale sources side by side (one process each, at most the CPU count)."""
    from concurrent.futures import ThreadPoolExecutor

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1, int(os.environ.get("MCP_BUILD_JOBS", "8"))))) as ex:
            list(ex.map(run, jobs))


def _sources_digest():
    """Identity of what the library is built from: every source, header and the flags."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


STAMP = LIB + ".stamp"


def build(force=False, verbose=True):
    """Compiles what is stale and links.  A library whose stamp (libmcpilco_hip.so.stamp: a digest of sources, headers and flags) matches is
    up to date whatever the file times say and whether or not the objects are there -- a snapshot of the tree (gpurun, the driver's GPU box)
    carries the library and its stamp, not the objects (.gpurunignore), and must not spend minutes rebuilding them."""
    digest = _sources_digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == digest:
        return LIB
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])
        objs.append(o)
    _compile_all(jobs, verbose)
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python build.py --variant TAG DEF1 DEF2 ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    elif "--variant-only" in sys.argv:  # python build.py --variant-only a.hip,b.hip TAG DEF1 DEF2 ...: recompiling the named sources only
        i = sys.argv.index("--variant-only")
        print(build_variant(sys.argv[i + 2], sys.argv[i + 3:], only=sys.argv[i + 1].split(",")))
    elif "--variant-fwd" in sys.argv:  # the same, recompiling rollout_fwd.hip only
        i = sys.argv.index("--variant-fwd")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], only=["rollout_fwd.hip"]))
    elif "--variant-lean" in sys.argv:  # the same, recompiling rollout_fwd_lean.hip only
        i = sys.argv.index("--variant-lean")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], only=["rollout_fwd_lean.hip"]))
    elif "--variant-gp" in sys.argv:  # the same, recompiling gp_pretrain.hip only
        i = sys.argv.index("--variant-gp")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], only=["gp_pretrain.hip"]))
    elif "--variant-bwd" in sys.argv:  # the same, recompiling rollout_bwd.hip only
        i = sys.argv.index("--variant-bwd")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], only=["rollout_bwd.hip"]))
    else:
        build(force="--force" in sys.argv)
