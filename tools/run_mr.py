import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svs_unet_pytorch_amd import _lib
L = _lib.lib()
B, Lw = 64, 768 * 127
x = (torch.rand((B, Lw), device="cuda") - 0.5) * 0.4
y = x * 0.7 + (torch.rand((B, Lw), device="cuda") - 0.5) * 0.2
ws = torch.empty(int(L.svs_mrstft_workspace_bytes(B, Lw)), dtype=torch.uint8, device="cuda")
loss, dx = torch.zeros(1, device="cuda"), torch.empty_like(x)
for _ in range(6):
    L.svs_mrstft_loss_fwd_bwd(x.data_ptr(), y.data_ptr(), B, Lw, 1.0, loss.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
torch.cuda.synchronize()
