import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_unet_pytorch_amd import synth
from svs_unet_pytorch_amd.model import UNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 216
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
model = UNet()
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state().items()})
model.to("cuda").eval()
model.eval_precision = prec
x = torch.rand((B, 1, 512, 128), device="cuda")
with torch.no_grad():
    for _ in range(8):
        y = model(x)
torch.cuda.synchronize()
