#!/usr/bin/env python3
"""Prints the figures of a bench.py JSON line one per row.   python tools/show_bench.py gpurun_out/<file>.txt"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %-10s value %.3e  %.3f ms/step  kernel %s %.3f ms  frac %.3f (its own flops / its own time)  step %.3f  bwd %.3f ms" % (
    d["config"]["workload"].split(":")[0], d["value"], d["ms_per_step"], r["kernel"], r["kernel_ms"], r["frac"], r["frac_step"], r["kernels"][1]["avg_ms"]))
for e in d.get("extra_workloads", []):
    print("%-16s N=%-4s M=%-5s T=%-4s %.3f ms/step  %6.2f us/time-step  %s %.3f ms  frac %.3f  step %.3f  blocks %s" % (
        e["workload"][:16], e.get("N"), e.get("particles"), e.get("horizon"), e["ms_per_step"], e.get("us_per_time_step", 0), e["kernel"], e["kernel_ms"],
        e.get("frac", 0), e.get("frac_step", 0), e.get("blocks")))
for k in ("loop", "loop_c1_script"):
    if k in d:
        print(k, "%.3f ms/step" % (d.get("loop_ms_per_step") if k == "loop" else d[k]["loop_ms_per_step"]), "over bench step %+.1f %%" % (100 * d[k]["over_bench_step"]))
for k in ("fit_model", "fit_model_ur5"):
    if k in d:
        print(k, {a: b for a, b in d[k].items() if a != "what"})
if "pretrain" in d:
    print("pretrain", {k: (round(v["s_per_gp"] * 1e3, 2), "ms per GP", v["rows_kept"]) for k, v in d["pretrain"].items() if isinstance(v, dict)})
if "scale_base" in d:
    print("scale_base %.3e (%s)" % (d["scale_base"]["value"], d["scale_base"]["workload"][:2]))
for k in ("cpu_baseline", "cpu_baseline_all_cores"):
    if k in d:
        print(k, "%.3e" % d[k]["value"], d[k]["cores"], "core(s)")
