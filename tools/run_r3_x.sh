set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_pytest_x.log 2>&1; echo pytest_rc=$? ; tail -3 gpurun_out/r3_pytest_x.log
python tools/signal_bench.py 2>&1 | tail -25
python tools/ab_tune.py BN_INLINE 0 -1 --rounds 3 2>&1 | tail -2
