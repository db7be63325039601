#!/bin/bash
# round 6: the pipelined backward sweep (one particle per workgroup, wide class): parity, the wide-class backward tests, then the times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pins_r6.py tests/test_gpu_realsize.py -x -q -m gpu -k "pipelined or eight_particles or row_split_cluster_of or ur5" -s > gpurun_out/pipe_tests.txt 2>&1; rc=$?
grep "pipelined vs" gpurun_out/pipe_tests.txt; tail -3 gpurun_out/pipe_tests.txt
[ $rc = 0 ] || exit 1
python tools/time_bwd.py ur5_script > gpurun_out/pipe_time_bwd.txt 2>&1; tail -5 gpurun_out/pipe_time_bwd.txt
python bench.py --workload ur5_script --no-cpu --no-extra --steps 20 --warmup 3 --min-seconds 1 2>/dev/null | python tools/show_bench.py /dev/stdin 2>/dev/null | head -3
python tools/phase_stamps.py ur5_script 2>&1 | grep "bwd per step"
