set -o pipefail
for bal in 0 768 1024 1280; do
  SVS_CONV_BALANCE=$bal python tools/gemm_sweep.py --batch 64 --no-wgrad --cfgs 5,6 --ks 4 --only conv4,conv5,conv6,deconv1,deconv2,deconv3 > gpurun_out/r3_sweep_cfg_bal$bal.txt 2>&1
  echo "== balance $bal"; grep -E "default|cfg" gpurun_out/r3_sweep_cfg_bal$bal.txt
done
python tools/ab_tune.py CONV_BALANCE 0 -1 --rounds 5 > gpurun_out/r3_ab_bal2.txt 2>&1; cat gpurun_out/r3_ab_bal2.txt
