set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mrstft or istft or stft" 2>&1 | tail -2
timeout -k 10 300 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "mr or objective" 2>&1 | tail -2
python tools/signal_bench.py 2>&1 | grep -o '"mrstft_fwd_bwd_ms": [0-9.]*, "mrstft_value_only_ms": [0-9.]*'
python tools/signal_bench.py 2>&1 | grep -o '"mrstft_fwd_bwd_ms": [0-9.]*'
