set -o pipefail
python bench.py > gpurun_out/r3_bench_train.json 2> gpurun_out/r3_bench_train.err; echo rc=$?; tail -c 600 gpurun_out/r3_bench_train.json
python bench.py --mode eval --batch 16 > gpurun_out/r3_bench_eval.json 2> gpurun_out/r3_bench_eval.err; echo rc=$?; tail -c 300 gpurun_out/r3_bench_eval.json
