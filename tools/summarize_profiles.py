"""Turn the raw rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the files cited from
profiles/: per-config kernel statistics, the counter rows of the rollout kernels, and traffic.json (HBM bytes per forward launch
= (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md; steady-state launches).

    python tools/summarize_profiles.py <tag of gpurun_out/prof_<tag>> <round prefix, e.g. r02>
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
SHAPES = {"c1": (400, 150, 112), "c3": (4000, 150, 112), "c5": (2000, 300, 384)}  # M, T, algorithmic HBM bytes per particle-step

for cfg in list(SHAPES) + ["c1_script", "c2_script", "pms_script", "pms_script_n450", "c2_script_n360", "ur5_script"]:
    f = os.path.join(src, cfg + "_stats", cfg + "_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, cfg)))
    f = os.path.join(src, cfg + "_stamps.txt")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s_stamps.txt" % (rnd, cfg)))

for cfg in ("fit_c1", "fit_ur5"):  # GP training epochs (round 4)
    f = os.path.join(src, cfg + "_stats", cfg + "_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, cfg)))
for name in ("chol_times.txt", "chol_stamps_n300.txt", "chol_stamps_n400.txt", "pretrain_times.txt", "vsym_bench.txt", "ur5_script_half1_stamps.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (rnd, name)))

out = {"_note": "forward rollout kernel, HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (gfx950 "
                "FETCH_SIZE correction x2, MI355X_MICROARCH.md), mean over the steady-state launches; *_alg = algorithmic bytes per launch "
                "(16*(S+U+G) B per particle-step x M x T, SURVEY 8d)"}
for cfg, (M, T, balg) in SHAPES.items():
    vals = {}
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = os.path.join(src, "%s_%s" % (cfg, name), cfg + "_counter_collection.csv.rollout")
        if not os.path.exists(f):
            continue
        shutil.copy(f, os.path.join(dst, "%s_%s_pmc_%s_size.csv" % (rnd, cfg, name)))
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "rollout_fwd" in r["Kernel_Name"]]
        v = [float(r["Counter_Value"]) for r in rows]
        vals[name] = sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]  # skip the first (cold L2) launch
        print(cfg, counter, "launches", len(v), "steady-state mean KB", vals[name], "first", v[0])
    if len(vals) == 2:
        out[cfg] = (2.0 * vals["fetch"] + vals["write"]) * 1024.0
        out[cfg + "_fetch_size_kb"] = vals["fetch"]
        out[cfg + "_write_size_kb"] = vals["write"]
        out[cfg + "_alg"] = balg * M * T
    f = os.path.join(src, cfg + "_mfma", cfg + "_counter_collection.csv.rollout")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s_pmc_mfma_busy.csv" % (rnd, cfg)))
        rows = [r for r in csv.DictReader(open(f)) if "rollout_fwd_tile" in r["Kernel_Name"]]
        busy = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
        act = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
        if busy and act:  # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs in total
            util = busy[-1] / (act[-1] / 8 * 1024)
            out[cfg + "_mfma_pipe_busy"] = util
            print("tile kernel (%s): MFMA busy %.3e cycles per launch = %.3e v_mfma_f64_16x16x4 at 64 cycles, matrix-pipe utilisation %.1f %%"
                  % (cfg, busy[-1], busy[-1] / 64, 100 * util))
# identity of what was measured: bench.py quotes these numbers only while the kernel sources are the ones profiled
sys.path.insert(0, ROOT)
import datetime
import subprocess

import bench

out["kernel_sources_sha16"] = bench.kernel_sources_sha16()
out["date"] = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ")
try:
    out["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:  # noqa: BLE001
    out["commit"] = None
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
for cfg in SHAPES:
    f = os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, cfg))
    if os.path.exists(f):
        print("==", cfg)
        for r in list(csv.DictReader(open(f)))[:6]:
            print("  %-70s calls %4s  avg %10.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
