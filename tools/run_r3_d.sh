set -o pipefail
python tools/ab_c1_tiled.py 64 2>&1 | tail -3
python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -x -m gpu > gpurun_out/r3_pytest5.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest5.log; tail -3 gpurun_out/r3_pytest5.log
python tools/ab_tune.py CONV_C1_TILED 0 -1 --rounds 4 2>&1 | tail -2
