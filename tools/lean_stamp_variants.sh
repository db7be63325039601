#!/bin/bash
# Diagnostic: phase stamps of the lean forward kernel for a list of experiment libraries (mc-pilco_amd/build.py --variant-fwd TAG ...).
#   tools/lean_stamp_variants.sh TAG1 TAG2 ...   -> gpurun_out/lean_<TAG>.txt   ("main" = the product library)
mkdir -p gpurun_out
for tag in "$@"; do
  lib=$PWD/mc-pilco_amd/libmcpilco_hip_$tag.so
  [ "$tag" = main ] && lib=$PWD/mc-pilco_amd/libmcpilco_hip.so
  MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$lib timeout -k 10 120 python tools/phase_stamps.py c1 > gpurun_out/lean_$tag.txt 2>&1 || exit 1
  echo "== $tag"; grep -E "per step|lean kernel|cyc/step" gpurun_out/lean_$tag.txt | grep -v "^-  \|tile kernel\|J finish"
done
