#!/bin/bash
# Diagnostic: shader-core counters of the forward rollout kernel of one workload (separate --pmc passes, never combined
# with trace domains).  Usage (GPU box, through gpurun): tools/diag_pmc.sh <tag> <workload> "<counters pass 1>" "<counters pass 2>" ...
# Rows of the rollout kernels end up in gpurun_out/diag_<tag>/pass<i>.csv.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; W=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/diag_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || true
i=0
for pass in "$@"; do
  i=$((i+1))
  echo "[$(date +%T)] pass $i: $pass"
  timeout -k 10 240 rocprofv3 --output-format csv --pmc $pass -d "$OUT/p$i" -o d -- python3 $R/bench.py --no-cpu --no-extra --workload $W --steps 2 --warmup 1 > "$OUT/p$i.log" 2>&1
  rc=$?
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > "$OUT/pass$i.csv"; grep "rollout_fwd" "$f" >> "$OUT/pass$i.csv"; fi
  rm -rf "$OUT/p$i"
  echo "   rc=$rc rows=$(wc -l < "$OUT/pass$i.csv" 2>/dev/null)"
  if [ $rc -ge 124 ]; then echo "pass timed out: stopping"; exit 1; fi
done
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/pass*.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-32s launches %2d  last %.4e" % (k, len(v), v[-1]))
EOF
