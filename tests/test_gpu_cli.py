"""End-to-end CLI parity (-m gpu): wav -> data.py to_spec -> train.py (a few steps) -> inference.py ->
data.py to_wave on synthetic audio, with every intermediate file checked against the oracle
(oracle/stft_oracle.py, oracle/tiling_oracle.py, oracle/unet_oracle.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import stft_oracle as so
from oracle import tiling_oracle as to
from oracle import unet_oracle as uo
from svs_unet_pytorch_amd import data as svs_data
from svs_unet_pytorch_amd import inference as svs_inference
from svs_unet_pytorch_amd import synth
from svs_unet_pytorch_amd import train as svs_train

pytestmark = pytest.mark.gpu
SR = 8192


def _write_song(folder, idx, n):
    from scipy.io import wavfile
    os.makedirs(folder, exist_ok=True)
    voc = synth.audio(n - 1000, 2 * idx) * 0.3
    acc = synth.audio(n, 2 * idx + 1) * 0.5
    mix = acc.copy()
    mix[: voc.size] += voc
    wavfile.write(os.path.join(folder, "mixture.wav"), SR, mix.astype(np.float32))
    wavfile.write(os.path.join(folder, "vocals.wav"), SR, voc.astype(np.float32))       # shorter: exercises data.py:97-98
    return mix.astype(np.float32), voc.astype(np.float32)


def test_pipeline_end_to_end(tmp_path, report, monkeypatch):
    monkeypatch.chdir(tmp_path)
    src = tmp_path / "wav"
    songs = {name: _write_song(str(src / name), i, n) for i, (name, n) in enumerate((("songA", 110000), ("songB", 99000)))}
    spec_dir = tmp_path / "spec"
    svs_data.main(["--src", str(src), "--tar", str(spec_dir), "--direction", "to_spec"])
    names = sorted(os.listdir(spec_dir / "mixture"))
    assert names == ["0000_songA_phase.npy", "0000_songA_spec.npy", "0001_songB_phase.npy", "0001_songB_spec.npy"]   # data.py:107-109
    for i, (name, (mix, voc)) in enumerate(songs.items()):
        for track, y in (("mixture", mix), ("vocal", voc)):
            spec = np.load(spec_dir / track / f"{i:04d}_{name}_spec.npy")
            phase = np.load(spec_dir / track / f"{i:04d}_{name}_phase.npy")
            want_s, want_p = so.to_spec(mix, y)
            assert spec.dtype == np.float32 and phase.dtype == np.complex64 and spec.shape == want_s.shape == (513, 1 + mix.size // 768)
            assert report(f"cli to_spec {name}/{track} magnitude", np.abs(spec - want_s).max(), 2e-6)
            strong = want_s > 1e-3                      # the phase of a near-zero bin is noise in any fp32 FFT
            assert report(f"cli to_spec {name}/{track} phase", np.abs(phase - want_p)[strong].max(), 2e-3)
        assert abs(np.load(spec_dir / "mixture" / f"{i:04d}_{name}_spec.npy").max() - 1.0) <= 1e-6

    # a short training run through the CLI: files and keys of train.py:169-171,369-382
    svs_train.main(["--train_folder", str(spec_dir), "--valid_folder", str(spec_dir), "--label", "t", "--epoch", "2", "--batch_size", "8",
                    "--val_interval", "1", "--load_path", "none.pth"])
    ck = torch.load(tmp_path / "CKPT" / "svs_t.pth", map_location="cpu")
    assert ck["epoch"] == 2 and len(ck["model_state_dict"]) == 79 and "optim" in ck and "scheduler" in ck
    assert os.path.exists(tmp_path / "CKPT" / "svs_best_t.pth")
    lines = open(tmp_path / "LOG" / "log_t.txt").read().split()
    assert len([x for x in lines if x == "Val"]) == 2 and all(np.isfinite(float(x)) for x in lines if x != "Val")

    # inference through the CLI with the trained checkpoint, checked against the oracle on the same weights
    pred_dir = tmp_path / "pred"
    svs_inference.main(["--model_path", str(tmp_path / "CKPT" / "svs_t.pth"), "--mixture_folder", str(spec_dir / "mixture"),
                        "--tar", str(pred_dir), "--vocal_solo", "1"])
    st = {k: v.clone() for k, v in ck["model_state_dict"].items()}
    for i, name in enumerate(songs):
        fname = f"{i:04d}_{name}_spec.npy"
        got = np.load(pred_dir / fname)
        mixspec = np.load(spec_dir / "mixture" / fname)
        with torch.no_grad():
            want = to.separate(mixspec, lambda t: uo.forward(st, torch.from_numpy(t)).numpy())
        assert got.shape == mixspec.shape and np.all(got[0] == 0)
        assert report(f"cli inference {name}", np.abs(got - want).max(), 1e-4)

    # back to audio
    wav_dir = tmp_path / "out_wav"
    svs_data.main(["--src", str(pred_dir), "--phase", str(spec_dir / "mixture"), "--tar", str(wav_dir), "--direction", "to_wave"])
    from scipy.io import wavfile
    for i, name in enumerate(songs):
        rate, y = wavfile.read(wav_dir / f"{i:04d}_{name}.wav")
        mag = np.load(pred_dir / f"{i:04d}_{name}_spec.npy")
        ph = np.load(spec_dir / "mixture" / f"{i:04d}_{name}_phase.npy")
        want = so.to_wave(mag, ph)
        assert rate == SR and y.shape == want.shape
        assert abs(np.abs(y).max() - 0.9) <= 1e-5                                     # data.py:162-164
        assert report(f"cli to_wave {name} (interior)", np.abs(y - want)[1024:-1024].max(), 5e-5)
    with pytest.raises(Exception):
        svs_data.main(["--src", str(pred_dir), "--tar", str(wav_dir), "--direction", "to_wave"])    # data.py:118


def test_streaming_end_to_end(report):
    """BASELINE configs[4] in fp32: waveform -> STFT -> U-Net -> mask -> iSTFT without leaving the GPU, against the
    same chain built from the three oracles; 44.1 kHz-length stereo input (the sample rate only labels the file)."""
    from svs_unet_pytorch_amd.model import UNet
    from svs_unet_pytorch_amd.streaming import separate_waveform
    state = synth.closed_form_state()
    model = UNet()
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    model.to("cuda").eval()
    n = 44100 * 3
    y = np.stack([synth.audio(n, 10), synth.audio(n, 11)])
    got = separate_waveform(model, torch.from_numpy(y).to("cuda")).cpu().numpy()
    T = 1 + n // 768
    assert got.shape == (2, 768 * (T - 1))
    st = uo.to_torch_state(state)
    for ch in range(2):
        spec, phase = so.to_spec(y[ch], y[ch])
        with torch.no_grad():
            pred = to.separate(spec, lambda t: uo.forward(st, torch.from_numpy(t)).numpy())
        want = so.to_wave(pred, phase)
        assert abs(np.abs(got[ch]).max() - 0.9) <= 1e-5
        assert report(f"streaming separation channel {ch} (interior)", np.abs(got[ch] - want)[1024:-1024].max(), 1e-4)


def test_resident_training_set(tmp_path, report):
    """Tiles cut on the GPU from the HBM-resident training set against the reference's __getitem__ rule (oracle), bit for
    bit: long / exactly-128 / short songs, every start incl. the last valid one; then one epoch of batches."""
    rng = np.random.default_rng(11)
    root = tmp_path / "spec"
    os.makedirs(root / "mixture"), os.makedirs(root / "vocal")
    files = {}
    for i, T in enumerate((300, 128, 50, 129)):
        name = f"{i:04d}_song{i}_spec.npy"
        mix, voc = rng.random((513, T), dtype=np.float32), rng.random((513, T), dtype=np.float32)
        np.save(root / "mixture" / name, mix), np.save(root / "vocal" / name, voc)
        files[i] = (mix, voc, T)
    ds = svs_train.SpectrogramDataset(str(root), samples_per_song=3)
    res = svs_train.ResidentSpectrograms(ds, torch.device("cuda"))
    assert len(res) == 12
    songs = [0, 0, 0, 1, 2, 3, 3]
    starts = [0, 57, 172, 0, 0, 0, 1]
    mix, voc = res.crop(songs, starts)
    for b, (sidx, st) in enumerate(zip(songs, starts)):
        wm, wv = to.crop_item(files[sidx][0], files[sidx][1], st)
        assert np.array_equal(mix[b].cpu().numpy(), wm) and np.array_equal(voc[b].cpu().numpy(), wv), (sidx, st)
    report("resident crop vs __getitem__ rule (bit-exact)", 0.0, 0.0)
    s2, st2 = res.draw(range(12))
    assert s2 == [i % 4 for i in range(12)] and all(0 <= st <= max(files[s][2] - 128, 0) for s, st in zip(s2, st2))
    seen = 0
    for m, v in res.batches(5, shuffle=True):
        assert m.shape[1:] == (1, 512, 128) and m.shape == v.shape and m.is_cuda
        seen += m.shape[0]
    assert seen == 12 and res.num_batches(5) == 3



def test_resident_set_matches_reference_dataset_items(tmp_path, golden, report):
    """The HBM-resident training set against the reference's OWN SpectrogramDataset (tests/golden/dataset_items.npz,
    captured by running /root/reference/train.py as a script): magnitudes bit for bit (sha256), angles (np.angle done on
    the device with atan2f) to 1e-6 rad, with the start drawn exactly as train.py:121 draws it."""
    import hashlib
    import random
    g = golden("dataset_items.npz")
    lengths = [int(t) for t in g["lengths"]]
    root = tmp_path / "spec"
    os.makedirs(root / "mixture"), os.makedirs(root / "vocal")
    songs = {}
    for n, T in enumerate(lengths):
        mix, voc, pm, pv = synth.song(n, T)
        songs[n] = (mix, voc, pm, pv)
        base = f"{n:04d}_len{T}"
        np.save(root / "mixture" / f"{base}_spec.npy", mix), np.save(root / "vocal" / f"{base}_spec.npy", voc)
        np.save(root / "mixture" / f"{base}_phase.npy", pm), np.save(root / "vocal" / f"{base}_phase.npy", pv)
    ds = svs_train.SpectrogramDataset(str(root))
    assert len(ds) == int(g["len"])
    res = svs_train.ResidentSpectrograms(ds, torch.device("cuda"))
    assert res.has_phase
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
    worst = 0.0
    for seed in g["seeds"]:
        items = [0, 1, 2, 3, 5, 6]
        starts = []
        for idx in items:                                  # the reference seeds per item in the generator; one draw per item
            random.seed(int(seed) * 1000 + idx)
            s_, st_ = res.draw([idx], rng=random)
            assert s_ == [idx % len(lengths)]
            starts.append(st_[0])
            assert st_[0] == int(g[f"s{int(seed)}.i{idx}.start"])
        mix, voc, mph, vph = res.crop([i % len(lengths) for i in items], starts, with_phase=True)
        for b, idx in enumerate(items):
            p = f"s{int(seed)}.i{idx}."
            assert np.array_equal(sha(mix[b].cpu().numpy()), g[p + "mix_sha"]), p
            assert np.array_equal(sha(voc[b].cpu().numpy()), g[p + "voc_sha"]), p
            n = idx % len(lengths)
            for got, z, key in ((mph[b], songs[n][2], "mix_phase_sha"), (vph[b], songs[n][3], "voc_phase_sha")):
                want = to.crop_phase(z, starts[b])
                assert np.array_equal(sha(want), g[p + key])           # the oracle IS the reference here (bit-exact)
                d = np.abs(got.cpu().numpy() - want)
                d = np.minimum(d, 2 * np.pi - d)                        # +pi and -pi are the same angle
                worst = max(worst, float(d.max()))
    assert report("resident phase tiles vs reference SpectrogramDataset (rad)", worst, 1e-6)
    report("resident magnitude tiles vs reference SpectrogramDataset (sha256)", 0.0, 0.0)


def test_bench_contract_line():
    """bench.py as the driver runs it (one rank, child process, few steps): exactly one JSON line on stdout with the contract's
    keys, BASELINE's metric / config, and the `roofline` and `cpu_baseline` objects."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "tiles/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "batch 64" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.05 < rf["frac"] < 1.0 and rf["kernel"].startswith(("conv_gemm_kernel", "wgrad_", "parity_window"))
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "tiles/s" and cb["sample"]
