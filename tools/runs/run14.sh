set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/r6_t3.log 2>&1; echo "tests rc=$?" >> $O/r6_t3.log; tail -4 $O/r6_t3.log
bash tools/runs/run13.sh > /dev/null 2>&1; grep "per optimizer step" $O/r6_loop_times.txt
tools/ab_bench.sh "main" "c1 c1_script c2_script pms_script ur5_script c3" 20 > $O/r6_ab6.txt 2>&1; cat $O/r6_ab6.txt
