#!/bin/bash
# A/B of experiment builds (mc-pilco_amd/build.py --variant*): bench.py's step time and forward / backward kernel times of the named workloads on
# the main library ("main") and on each libmcpilco_hip_<tag>.so.      tools/ab_bench.sh "main sc1 vd3" "c3 c5" [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAGS=${1:-main}; WLS=${2:-c3}; STEPS=${3:-5}
for w in $WLS; do
  for t in $TAGS; do
    if [ "$t" = main ]; then unset MCPILCO_HIP_EXPERIMENT MCPILCO_HIP_LIB; else export MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=$R/mc-pilco_amd/libmcpilco_hip_$t.so; fi
    python3 $R/bench.py --workload $w --no-cpu --no-extra --steps $STEPS --warmup 2 --min-seconds 1.0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('%-10s %-8s step %8.3f ms   fwd %8.3f ms   bwd %7.3f ms   frac %.3f  frac_step %.3f' % ('$w','$t',d['ms_per_step'],k[0]['avg_ms'],k[1]['avg_ms'],d['roofline']['frac'],d['roofline']['frac_step']))" || echo "$w $t FAILED"
  done
done
