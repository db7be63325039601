set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
tools/ab_bench.sh "main pre1 pre2 early" "c1" 20 > $O/r6_ab2.txt 2>&1
for t in main pre2 early; do
  if [ "$t" = main ]; then unset MCPILCO_HIP_EXPERIMENT MCPILCO_HIP_LIB; else export MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_$t.so; fi
  echo "=== $t" >> $O/r6_stamps2.txt; python3 tools/phase_stamps.py c1 2>&1 | grep -v amdgpu.ids | grep -v "^bwd" >> $O/r6_stamps2.txt
done
unset MCPILCO_HIP_EXPERIMENT MCPILCO_HIP_LIB
echo "=== main M=100" >> $O/r6_stamps2.txt; python3 tools/phase_stamps.py c1 0 100 2>&1 | grep -v amdgpu.ids | grep -v "^bwd" >> $O/r6_stamps2.txt
echo "=== main M=4 (one cluster)" >> $O/r6_stamps2.txt; python3 tools/phase_stamps.py c1 0 4 2>&1 | grep -v amdgpu.ids | grep -v "^bwd" >> $O/r6_stamps2.txt
cat $O/r6_ab2.txt $O/r6_stamps2.txt
