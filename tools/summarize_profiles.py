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
for name in ("chol_times.txt", "chol_stamps_n300.txt", "chol_stamps_n400.txt", "pretrain_times.txt", "vsym_bench.txt", "ur5_script_half1_stamps.txt",
             "c1_nofma_stamps.txt", "c1_noload_stamps.txt", "vissue_bench.txt", "ur5_row_split_forms.txt", "mfma4x4_probe.txt", "stream_mfma_overlap.txt", "c3_bwd_wave_stamps.txt", "c5_bwd_wave_stamps.txt", "loop_times.txt"):
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
# round 6: the lean (headline) kernel -- matrix-pipe busy share, L1 -> L2 read requests per launch, and phase V with one side compiled out
import re


def _lean_rows(path, counter):
    if not os.path.exists(path):
        return []
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "rollout_fwd_lat" in r["Kernel_Name"]]


f = os.path.join(src, "c1_mfma", "c1_counter_collection.csv.rollout")
if os.path.exists(f):
    shutil.copy(f, os.path.join(dst, "%s_c1_pmc_mfma_busy.csv" % rnd))
    busy, act = _lean_rows(f, "SQ_VALU_MFMA_BUSY_CYCLES"), _lean_rows(f, "GRBM_GUI_ACTIVE")
    if busy and act:
        out["c1_mfma_pipe_busy"] = busy[-1] / (act[-1] / 8 * 1024)
        print("lean kernel (c1): MFMA busy %.3e cycles per launch, matrix-pipe utilisation %.1f %% of all 1024 SIMDs (200 of 256 CUs hold a workgroup)"
              % (busy[-1], 100 * out["c1_mfma_pipe_busy"]))
f = os.path.join(src, "c1_tcp", "c1_counter_collection.csv.rollout")
if os.path.exists(f):
    shutil.copy(f, os.path.join(dst, "%s_c1_pmc_tcp_tcc_read_req.csv" % rnd))
    req = _lean_rows(f, "TCP_TCC_READ_REQ_sum")
    if req:
        out["c1_tcp_tcc_read_req"] = req[-1]
        print("lean kernel (c1): TCP_TCC_READ_REQ_sum %.4e per launch" % req[-1])


def _phase_v(path):
    """(V + J interval, slowest wave's own phase V) in cycles per step from a tools/phase_stamps.py output."""
    if not os.path.exists(path):
        return None
    t = open(path).read()
    m1 = re.search(r"V \+ J \(to the barrier\)\s+\d+\s+[0-9.]+%\s+(\d+) cyc/step", t)
    m2 = re.search(r"phase V per wave \(own time, before the barrier\): ([0-9 ]+)", t)
    if not (m1 and m2):
        return None
    return {"v_plus_j": int(m1.group(1)), "slowest_wave_v": max(int(x) for x in m2.group(1).split())}


pv = {k: _phase_v(os.path.join(src, n)) for k, n in (("both", "c1_stamps.txt"), ("stream_only", "c1_nofma_stamps.txt"), ("mfma_only", "c1_noload_stamps.txt"))}
if all(pv.values()):
    out["c1_phase_v_cycles"] = pv
    print("lean kernel (c1), phase V per step:", pv)
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
