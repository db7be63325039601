#!/usr/bin/env python3
"""Register / scratch / LDS table of every gfx950 kernel in the library, from hipcc's -Rpass-analysis=kernel-resource-usage
(device-only recompile, no GPU needed).   python tools/kernel_resources.py [file.hip ...] [--md]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mc-pilco_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm", "--cuda-device-only",
         "-Rpass-analysis=kernel-resource-usage"]


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def table(src):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", out)[1:]:
        name = re.sub(r"\(.*", "", demangle(b.split("\n")[0].strip().split(" ")[0]))
        name = name.replace("void ", "")

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]")))
    return rows


if __name__ == "__main__":
    md = "--md" in sys.argv
    files = [a for a in sys.argv[1:] if not a.startswith("--")] or [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != "comm.hip"]
    for f in files:
        src = f if os.path.exists(f) else os.path.join(CSRC, f)
        print(("\n**%s**\n\n| kernel | VGPR | AGPR | SGPR | SGPR spill | VGPR spill | scratch B/lane | waves/SIMD |\n|---|---|---|---|---|---|---|---|" if md else "== %s") % os.path.basename(src))
        for r in table(src):
            print(("| `%s` | %d | %d | %d | %d | %d | %d | %d |" if md else "%-64s vgpr %3d agpr %3d sgpr %3d spillS %3d spillV %3d scratch %4d occ %d") % ((r[0][-64:],) + r[1:]))
