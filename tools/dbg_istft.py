import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import stft_oracle as so
from svs_unet_pytorch_amd import _lib, synth
from svs_unet_pytorch_amd.data import istft, stft_magphase
n = 20000
y = synth.audio(n)
d = so.stft(y)
mag_o, ph_o = so.magphase(d)
want = so.istft(mag_o * ph_o)
m = torch.from_numpy(mag_o).cuda(); p = torch.from_numpy(ph_o).cuda()
got = istft(m, p).cpu().numpy()
T = mag_o.shape[1]
print("T", T, "n_out", got.shape, want.shape)
err = np.abs(got - want)
for h in range(0, len(want), 768):
    seg = slice(h, h + 768)
    print(h // 768, "err max %.3e  got rms %.3e  want rms %.3e  corr %.3f" % (err[seg].max(), np.sqrt((got[seg] ** 2).mean()), np.sqrt((want[seg] ** 2).mean()),
          float((got[seg] * want[seg]).sum() / (np.linalg.norm(got[seg]) * np.linalg.norm(want[seg]) + 1e-30))))
