#!/bin/bash
# round 6: phase shares on the wall clock (100 MHz counter; experiment build MCPX_WALL_STAMPS) beside the cycle shares
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for w in c3 c5 c1 ur5_script; do
  echo "== $w, shader cycles"; python3 tools/phase_stamps.py $w 2>&1 | grep -v "amdgpu.ids" | tail -14
  echo "== $w, 10 ns ticks"; MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_wall.so python3 tools/phase_stamps.py $w 2>&1 | grep -v "amdgpu.ids" | tail -14
done > gpurun_out/r6_wall_stamps.txt 2>&1
grep -A11 "== c3" gpurun_out/r6_wall_stamps.txt | grep -v "per wave\|detail\|J finish"
