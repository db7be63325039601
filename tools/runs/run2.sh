set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
tools/ab_bench.sh "main sc1 vd3 sc1vd3" "c3 c5" 4 > $O/r6_ab1.txt 2>&1
tools/ab_bench.sh "main sc1" "ur5_script" 10 >> $O/r6_ab1.txt 2>&1
MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_bws.so python3 tools/bwd_wave_stamps.py c5 > $O/r6_bws_c5.txt 2>&1
MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_bws.so python3 tools/bwd_wave_stamps.py c3 > $O/r6_bws_c3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu --no-extra --min-seconds 0"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/r6_c5_fetch_main -o c5 -- $B --workload c5 --steps 2 --warmup 1 > $O/r6_c5_fetch_main.log 2>&1
MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_sc1.so rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/r6_c5_fetch_sc1 -o c5 -- $B --workload c5 --steps 2 --warmup 1 > $O/r6_c5_fetch_sc1.log 2>&1
for f in $(find $O/r6_c5_fetch_main $O/r6_c5_fetch_sc1 -name "*counter_collection.csv"); do head -1 "$f" > "$f.rollout"; grep "rollout_fwd" "$f" >> "$f.rollout"; rm -f "$f"; done
find $O/r6_c5_fetch_main $O/r6_c5_fetch_sc1 -name "*.db" -delete
cd $R
python -m pytest tests -m gpu -q > $O/r6_t2.log 2>&1; echo "tests rc=$?" >> $O/r6_t2.log
python bench.py --no-cpu > $O/r6_bench0.txt 2> $O/r6_bench0.err
cat $O/r6_ab1.txt; tail -3 $O/r6_t2.log
