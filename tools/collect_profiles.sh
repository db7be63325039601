#!/bin/bash
# Collect the rocprofv3 evidence cited in DESIGN.md / bench.py (run on the GPU box through gpurun):
#   kernel-trace statistics of the default bench command, separate --pmc passes for the HBM traffic of the
#   forward kernel (never combined with trace domains), and kernel statistics of the large-swarm configs.
# Outputs land under gpurun_out/; tools/summarize_profiles.py turns them into the files under profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c1_stats" -o c1 -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/c1_stats.log" 2>&1 || exit 1
echo "c1 stats done"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/c1_fetch" -o c1 -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu > "$OUT/c1_fetch.log" 2>&1 || exit 1
echo "c1 fetch done"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/c1_write" -o c1 -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu > "$OUT/c1_write.log" 2>&1 || exit 1
echo "c1 write done"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c3_stats" -o c3 -- python3 "$R/bench.py" --workload c3 --steps 5 --warmup 2 --no-cpu > "$OUT/c3_stats.log" 2>&1 || exit 1
echo "c3 stats done"
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/c3_mfma" -o c3 -- python3 "$R/bench.py" --workload c3 --steps 3 --warmup 1 --no-cpu > "$OUT/c3_mfma.log" 2>&1 || exit 1
echo "c3 mfma counters done"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c5_stats" -o c5 -- python3 "$R/bench.py" --workload c5 --horizon 100 --steps 3 --warmup 1 --no-cpu > "$OUT/c5_stats.log" 2>&1 || exit 1
echo "c5 stats done"
# keep what is cited: the per-kernel statistics, and of the counter passes only the rollout kernels' rows
for f in $(find "$OUT" -name "*counter_collection.csv"); do
  head -1 "$f" > "$f.rollout" && grep "rollout_" "$f" >> "$f.rollout"; rm -f "$f"
done
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
find "$OUT" -type f | head -40; du -sh "$OUT"
