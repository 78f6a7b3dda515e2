set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu > gpurun_out/r3_pytest_w.log 2>&1; echo pytest_rc=$? ; tail -5 gpurun_out/r3_pytest_w.log
python tools/ab_tune.py BN_INLINE 0 -1 --rounds 3 2>&1 | tail -4
python tools/ab_tune.py BN_INLINE 128 2048 --rounds 3 2>&1 | tail -4
python tools/ab_tune.py BN_BLOCKS 256 1024 --rounds 3 2>&1 | tail -4
