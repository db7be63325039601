#!/bin/bash
# Collect the rocprofv3 evidence cited in DESIGN.md / bench.py (run on the GPU box through gpurun):
#   kernel-trace statistics of the bench command per workload (c1 = the default line's workload, c3, c5 at its stated T = 300),
#   and separate --pmc passes (never combined with trace domains) for the HBM traffic of the forward kernels and the
#   matrix-pipe busy cycles of the 16-particle kernel.
# Outputs land under gpurun_out/prof_<tag>/; tools/summarize_profiles.py turns them into the files under profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
B="python3 $R/bench.py --no-cpu --no-extra --min-seconds 0"
run() { echo "[$(date +%T)] $1"; shift; "$@" || exit 1; }
run "c1 stats" rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c1_stats" -o c1 -- $B --steps 20 --warmup 3 > "$OUT/c1_stats.log" 2>&1
run "c3 stats" rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c3_stats" -o c3 -- $B --workload c3 --steps 5 --warmup 2 > "$OUT/c3_stats.log" 2>&1
run "c5 stats" rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c5_stats" -o c5 -- $B --workload c5 --steps 3 --warmup 1 > "$OUT/c5_stats.log" 2>&1
# the reference's own small-swarm launch scripts' shapes (round 4)
for w in c1_script c2_script pms_script pms_script_n450 c2_script_n360 ur5_script; do
  run "$w stats" rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/${w}_stats" -o $w -- $B --workload $w --steps 20 --warmup 3 > "$OUT/${w}_stats.log" 2>&1
done
# phase stamps (tools/phase_stamps.py) of the same shapes
for w in c1 c1_script c2_script pms_script pms_script_n450 ur5_script c3 c5; do
  run "$w stamps" python3 $R/tools/phase_stamps.py $w > "$OUT/${w}_stamps.txt" 2>&1
done
# the finishing part of a row-split cluster (round 6, three row parts dealt row part major: block 208 = item 26 = tile 0, GP 0, the last rows; block 0
# above is the first sender).  Then the forward time of every form of the cluster (tools/row_split_soak.py) and the 4x4x4 MFMA probes.
echo "[$(date +%T)] ur5_script stamps, finishing part"; MCP_STAMP_BLOCK=208 python3 $R/tools/phase_stamps.py ur5_script > "$OUT/ur5_script_half1_stamps.txt" 2>&1 || exit 1
run "row split forms" python3 $R/tools/row_split_soak.py 20 > "$OUT/ur5_row_split_forms.txt" 2>&1
( hipcc --offload-arch=gfx950 -O3 -w -o /tmp/smo $R/tools/stream_mfma_overlap.hip && /tmp/smo ) > "$OUT/stream_mfma_overlap.txt" 2>&1 || exit 1
( hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma4x4_probe $R/tools/mfma4x4_probe.hip && /tmp/mfma4x4_probe && hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_bank_probe $R/tools/mfma_bank_probe.hip && /tmp/mfma_bank_probe ) > "$OUT/mfma4x4_probe.txt" 2>&1 || exit 1
for w in c1 c3 c5; do
  st=4; [ $w = c5 ] && st=2
  run "$w fetch" rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/${w}_fetch" -o $w -- $B --workload $w --steps $st --warmup 1 > "$OUT/${w}_fetch.log" 2>&1
  run "$w write" rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/${w}_write" -o $w -- $B --workload $w --steps $st --warmup 1 > "$OUT/${w}_write.log" 2>&1
done
# round 6: the headline (lean) kernel's own counter passes -- matrix-pipe busy cycles, and the L1 -> L2 read requests of the launch (the "one Kinv per
# workgroup and step through the CU's L1" claim as a counter)
run "c1 mfma" rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/c1_mfma" -o c1 -- $B --steps 4 --warmup 1 > "$OUT/c1_mfma.log" 2>&1
run "c1 tcp" rocprofv3 --output-format csv --pmc TCP_TCC_READ_REQ_sum -d "$OUT/c1_tcp" -o c1 -- $B --steps 4 --warmup 1 > "$OUT/c1_tcp.log" 2>&1
run "c3 mfma" rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/c3_mfma" -o c3 -- $B --workload c3 --steps 3 --warmup 1 > "$OUT/c3_mfma.log" 2>&1
run "c5 mfma" rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/c5_mfma" -o c5 -- $B --workload c5 --steps 2 --warmup 1 > "$OUT/c5_mfma.log" 2>&1
# GP training epoch (round 4): per-kernel statistics at the cart-pole and the UR5 shape, the factorisation kernels alone, and -- when the
# CLX_STAMPS experiment build is there (python mc-pilco_amd/build.py --variant-gp stamps CLX_STAMPS) -- the Cholesky's cycles per block row
run "fit c1 stats" rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/fit_c1_stats" -o fit_c1 -- python3 $R/tools/time_fit_model.py 300 100 > "$OUT/fit_c1_stats.log" 2>&1
run "fit ur5 stats" rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/fit_ur5_stats" -o fit_ur5 -- python3 $R/tools/time_fit_ur5.py 100 > "$OUT/fit_ur5_stats.log" 2>&1
run "chol times" python3 $R/tools/time_chol.py 300 400 500 600 1000 1153 2048 4096 > "$OUT/chol_times.txt" 2>&1
run "pretrain times" python3 $R/tools/time_pretrain.py > "$OUT/pretrain_times.txt" 2>&1
# round 6: what bounds phase V of the lean kernel -- the kernel itself with one side compiled out (experiment builds, when they are there:
#   python mc-pilco_amd/build.py --variant-lean nofma RLX_NOFMA;  ... --variant-lean noload RLX_NOLOAD), and the issue microbenchmark
for t in nofma noload; do
  if [ -f $R/mc-pilco_amd/libmcpilco_hip_$t.so ]; then
    MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_$t.so python3 $R/tools/phase_stamps.py c1 > "$OUT/c1_${t}_stamps.txt" 2>&1 || exit 1
  fi
done
( hipcc --offload-arch=gfx950 -O3 -w -o /tmp/vissue_bench $R/tools/vissue_bench.hip && /tmp/vissue_bench ) > "$OUT/vissue_bench.txt" 2>&1 || exit 1
if [ -f $R/mc-pilco_amd/libmcpilco_hip_bws.so ]; then
  for w in c3 c5; do MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_bws.so python3 $R/tools/bwd_wave_stamps.py $w > "$OUT/${w}_bwd_wave_stamps.txt" 2>&1 || exit 1; done
fi
run "loop times" python3 $R/tools/time_reinforce_policy.py --capture-ab > "$OUT/loop_times.txt" 2>&1
# the symmetric phase-V experiment of round 5 (measured and dropped) beside the full stream it would replace
( hipcc --offload-arch=gfx950 -O3 -w -o /tmp/vsym_bench $R/tools/vsym_bench.hip && /tmp/vsym_bench && hipcc --offload-arch=gfx950 -O3 -w -DNRES=0 -o /tmp/v4_bench $R/tools/v4_bench.hip && /tmp/v4_bench ) > "$OUT/vsym_bench.txt" 2>&1 || exit 1
if [ -f $R/mc-pilco_amd/libmcpilco_hip_stamps.so ]; then
  for n in 300 400; do
    MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_stamps.so python3 $R/tools/chol_stamps.py $n > "$OUT/chol_stamps_n$n.txt" 2>&1 || exit 1
  done
fi
# keep what is cited: the per-kernel statistics, and of the counter passes only the rollout kernels' rows
for f in $(find "$OUT" -name "*counter_collection.csv"); do
  head -1 "$f" > "$f.rollout" && grep "rollout_" "$f" >> "$f.rollout"; rm -f "$f"
done
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
find "$OUT" -type f | head -60; du -sh "$OUT"
