set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
export MCPILCO_HIP_EXPERIMENT=1
MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_v2.so python -m pytest tests/test_gpu_realsize.py -q -x -k "(alone and (201 or 202 or 204)) or beyond_npad_384 or long_horizon" > $O/r6_v2_tests.log 2>&1; echo "rc=$?" >> $O/r6_v2_tests.log
tools/ab_bench.sh "main v2 v2e early" "c1" 20 > $O/r6_ab3.txt 2>&1
tools/ab_bench.sh "main v2 v2e" "pms_script_n450 c2_script" 20 >> $O/r6_ab3.txt 2>&1
for t in v2 v2e v2nl; do
  export MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_$t.so
  echo "=== $t" >> $O/r6_stamps4.txt; python3 tools/phase_stamps.py c1 2>&1 | grep -v amdgpu.ids | grep -v "^bwd" | grep "phase V per wave\|cyc/step\|per step" >> $O/r6_stamps4.txt
done
tail -3 $O/r6_v2_tests.log; cat $O/r6_ab3.txt $O/r6_stamps4.txt
