"""Turn the raw rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the files cited from
profiles/: per-config kernel statistics, the counter rows of the rollout kernels, and traffic.json (HBM bytes per
forward launch = (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md; steady-state launches)."""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01f"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
for cfg in ("c1", "c3", "c5"):
    f = os.path.join(src, cfg + "_stats", cfg + "_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, cfg)))
vals = {}
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = os.path.join(src, "c1_" + name, "c1_counter_collection.csv.rollout")
    shutil.copy(f, os.path.join(dst, "%s_c1_pmc_%s_size.csv" % (rnd, name)))
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "rollout_fwd" in r["Kernel_Name"]]
    v = [float(r["Counter_Value"]) for r in rows]
    vals[name] = sum(v[1:]) / max(1, len(v) - 1)  # skip the first (cold L2) launch
    print(counter, "launches", len(v), "steady-state mean KB", vals[name], "first", v[0])
f = os.path.join(src, "c3_mfma", "c3_counter_collection.csv.rollout")
if os.path.exists(f):
    shutil.copy(f, os.path.join(dst, "%s_c3_pmc_mfma_busy.csv" % rnd))
    rows = [r for r in csv.DictReader(open(f)) if "rollout_fwd_tile" in r["Kernel_Name"]]
    busy = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
    act = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    if busy and act:  # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs in total
        print("tile kernel (c3): MFMA busy %.3e cycles per launch, %.3e MFMAs at 64 cycles, pipe utilisation %.1f %%"
              % (busy[-1], busy[-1] / 64, 100 * busy[-1] / (act[-1] / 8 * 1024)))
M, T = 400, 150
out = {"c1": (2.0 * vals["fetch"] + vals["write"]) * 1024.0,
       "_note": "rollout_fwd_kernel, bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (gfx950 FETCH_SIZE "
                "correction x2), steady-state launches; algorithmic bytes per launch = 112 B * M*T = %d" % (112 * M * T),
       "fetch_size_kb": vals["fetch"], "write_size_kb": vals["write"]}
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
for cfg in ("c1", "c3", "c5"):
    f = os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, cfg))
    if os.path.exists(f):
        print("==", cfg)
        for r in list(csv.DictReader(open(f)))[:6]:
            print("  %-70s calls %4s  avg %10.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
