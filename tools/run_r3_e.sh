set -o pipefail
python -m pytest tests -q -m gpu > gpurun_out/r3_pytest6.log 2>&1; echo pytest_rc=$? >> gpurun_out/r3_pytest6.log; tail -4 gpurun_out/r3_pytest6.log
python tools/ab_tune.py CONV_C1_TILED 0 -1 --rounds 4 2>&1 | tail -2
for kb in 1 2 4; do SVS_BF16_KB=$kb python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from svs_unet_pytorch_amd import synth
from svs_unet_pytorch_amd.model import UNet
model = UNet(); model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state().items()}); model.to("cuda").eval()
for B in (216, 16):
    x = torch.rand((B, 1, 512, 128), device="cuda")
    with torch.no_grad():
        model.eval_precision = "fp32"; ref = model(x)
        model.eval_precision = "bf16"; got = model(x)
        for _ in range(5): model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): model(x)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"BF16_KB={os.environ['SVS_BF16_KB']} B={B}: {ms:.4f} ms  {B / ms * 1e3:.0f} tiles/s  mean|d| {(got - ref).abs().mean().item():.2e} max {(got - ref).abs().max().item():.2e}")
PY
done
