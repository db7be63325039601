set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_realsize.py tests/test_gpu_realsize_r5.py tests/test_gpu_pins_r6.py -q -x -k "lean or alone or beyond_npad_384 or long_horizon or measurement_model or headline or dropout_bits" > $O/r6_deal_tests.log 2>&1; echo "rc=$?" >> $O/r6_deal_tests.log
tools/ab_bench.sh "noasm main" "c1 c1_script c2_script pms_script c2_script_n360" 20 > $O/r6_ab5.txt 2>&1
for t in main noasm; do
  if [ "$t" = main ]; then unset MCPILCO_HIP_EXPERIMENT MCPILCO_HIP_LIB; else export MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_$t.so; fi
  echo "=== $t" >> $O/r6_stamps6.txt; python3 tools/phase_stamps.py c1 2>&1 | grep -v amdgpu.ids | grep -v "^bwd" | grep "phase V per wave\|cyc/step\|per step" >> $O/r6_stamps6.txt
done
tail -3 $O/r6_deal_tests.log; cat $O/r6_ab5.txt $O/r6_stamps6.txt
