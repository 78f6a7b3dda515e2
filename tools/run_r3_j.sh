set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_eval16; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --mode eval --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-layers > $O/kt.log 2>&1
cd $R
python3 tools/prof_sequence.py $O/kt/kt_kernel_trace.csv 20 1 > $O/sequence.txt; cat $O/sequence.txt
