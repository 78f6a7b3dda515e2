import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_unet_pytorch_amd import _lib, synth
from svs_unet_pytorch_amd.data import stft_to_tiles, istft_from_tiles
n = int(240 * 44100)
y = torch.from_numpy(np.stack([synth.audio(n, 20), synth.audio(n, 21)])).cuda()
for _ in range(5):
    tiles, phase, peak, T = stft_to_tiles(y)
    mask = torch.rand_like(tiles)
    out = istft_from_tiles(tiles, mask, phase, T)
torch.cuda.synchronize()
