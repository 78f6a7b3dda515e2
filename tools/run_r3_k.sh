set -o pipefail
python tools/ab_tune.py CONV_PF 1 2 --eval --batch 16 --rounds 4 2>&1 | tail -2
python tools/ab_tune.py CONV_PF 1 2 --eval --batch 1 --rounds 3 2>&1 | tail -2
python tools/ab_tune.py CONV_PF 1 2 --eval --batch 4 --rounds 3 2>&1 | tail -2
python tools/ab_tune.py CONV_PF 1 2 --eval --batch 64 --rounds 3 2>&1 | tail -2
python tools/ab_tune.py CONV_PF 1 2 --batch 64 --rounds 4 2>&1 | tail -2
python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -x -m gpu > gpurun_out/r3_pytest9.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest9.log; tail -3 gpurun_out/r3_pytest9.log
