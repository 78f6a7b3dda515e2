set -o pipefail
python -m pytest tests -q -m gpu > gpurun_out/r3_pytest_final.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest_final.log; tail -3 gpurun_out/r3_pytest_final.log
python bench.py > gpurun_out/r3_bench_train.json 2> gpurun_out/r3_bench_train.err; echo rc=$?
python bench.py --mode eval --batch 16 > gpurun_out/r3_bench_eval.json 2> gpurun_out/r3_bench_eval.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['conv_roofline_frac'], d['eval_b16'], d['full_objective'])
PY
