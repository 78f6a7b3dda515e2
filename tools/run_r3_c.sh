set -o pipefail
python -m pytest tests -q -x -m gpu > gpurun_out/r3_pytest4.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest4.log; tail -3 gpurun_out/r3_pytest4.log
python tools/ab_tune.py CONV_BALANCE 0 -1 --rounds 5 > gpurun_out/r3_ab_bal3.txt 2>&1; cat gpurun_out/r3_ab_bal3.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_tl; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-layers > $O/kt.log 2>&1
cd $R
python3 tools/prof_steps_csv.py $O/kt/kt_kernel_trace.csv 8 > $O/per_step.txt; head -5 $O/per_step.txt
N=$(head -1 $O/per_step.txt | sed 's/.*launches\/step \([0-9]*\).*/\1/')
python3 tools/prof_timeline.py $O/kt/kt_kernel_trace.csv $N 2 4 full > $O/timeline.txt; head -40 $O/timeline.txt
