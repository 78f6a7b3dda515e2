set -o pipefail
python -m pytest tests -q -m gpu > gpurun_out/r3_pytest16.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest16.log; tail -4 gpurun_out/r3_pytest16.log
python tools/signal_bench.py --seconds 60 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('mrstft_fwd_bwd_ms','mrstft_value_only_ms','specific_istft_ms')})"
