set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for t in nofma noload; do
  export MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_$t.so
  echo "=== $t" >> $O/r6_stamps3.txt; python3 tools/phase_stamps.py c1 2>&1 | grep -v amdgpu.ids | grep -v "^bwd" | grep "phase V per wave\|V + J\|per step" >> $O/r6_stamps3.txt
done
cat $O/r6_stamps3.txt
