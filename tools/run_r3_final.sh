set -o pipefail
python -m pytest tests -q -m gpu > gpurun_out/r3_pytest_final.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest_final.log; tail -4 gpurun_out/r3_pytest_final.log
bash tools/final_profiles.sh gpurun_out/r3_final > gpurun_out/r3_final.log 2>&1; tail -15 gpurun_out/r3_final.log
