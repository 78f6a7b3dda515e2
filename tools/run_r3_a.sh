set -o pipefail
python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -x -m gpu > gpurun_out/r3_pytest3.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest3.log; tail -3 gpurun_out/r3_pytest3.log
python tools/gemm_sweep.py --batch 64 --no-wgrad --cfgs 99 --ks 1 --only conv4.fwd,conv5,conv6,deconv1,deconv2.fwd --ab --ab-env CONV_BALANCE --ab-vals 0,1,512,640,768,896,1024,1152,1280,1536 > gpurun_out/r3_sweep_bal.txt 2>&1
grep -E "default|A/B" gpurun_out/r3_sweep_bal.txt
python tools/ab_tune.py CONV_BALANCE 0 -1 --rounds 5 > gpurun_out/r3_ab_bal.txt 2>&1; cat gpurun_out/r3_ab_bal.txt
