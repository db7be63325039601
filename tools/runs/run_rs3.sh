#!/bin/bash
# round 6: three row parts / row-part-major deal of the GP-sharded 16-particle kernel: parity tests, then the forward times of the forms
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_realsize.py -x -q -m gpu -k "row_split" > gpurun_out/rs3_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/rs3_tests.txt
[ $rc = 0 ] || exit 1
timeout -k 10 200 python tools/row_split_soak.py 20 > gpurun_out/rs3_soak.txt 2>&1 || { tail -20 gpurun_out/rs3_soak.txt; exit 1; }
cat gpurun_out/rs3_soak.txt
