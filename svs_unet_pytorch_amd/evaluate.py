"""Separation-quality evaluation -- same command line and CSV as the reference's evaluate.py
(/root/reference/evaluate.py:87-182): vocal SDR / SIR / SAR from BSS-eval on the 2-source problem
(vocal, mixture - vocal) and NSDR = SDR(estimate) - SDR(mixture taken as the estimate) (evaluate.py:26-84).

    python -m svs_unet_pytorch_amd.evaluate --est out_wav --mix test/mixture_wav --ref test/vocal_wav [--out_csv r.csv]

This is host code outside the accelerated path: the metric runs once per song on the CPU in the
reference too.  The reference delegates to `mir_eval.separation.bss_eval_sources` (mir_eval 0.8.2, uv.lock:925-926) and
`librosa.load`, neither of which is installable here -- PARITY UNPINNED.  `bss_eval_sources` below restates the
published BSS-eval v3 algorithm that function implements (Vincent, Gribonval, Fevotte 2006; 512-tap time-invariant
distortion filters): the estimate is projected by least squares onto the span of the reference source(s) delayed by
0..511 samples (Gram matrix from FFT cross-correlations, block-Toeplitz), which splits it into target + spatial
distortion, interference and artifacts; SDR / SIR / SAR are the energy ratios of those parts; the source permutation
is the one with the best mean SIR.  tests/test_host.py checks the defining properties (a filtered copy of the
reference scores > 100 dB, a known interference mix scores its mixing ratio, additive noise scores its SNR).
wav files are read with scipy.io.wavfile (mono downmix, native sample rate: evaluate.py:15-23 passes sr=None).
"""
from __future__ import annotations

import argparse
import csv
import glob
import itertools
import os

import numpy as np

FILTER_LEN = 512


def load_mono_audio(path):
    """wav -> (float64 mono waveform, sample rate); the file's own rate is kept (evaluate.py:15-23)."""
    from scipy.io import wavfile
    if not os.path.exists(path):
        raise FileNotFoundError(f"File not found: {path}")
    sr, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float64) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float64) - 128.0) / 128.0
    else:
        data = data.astype(np.float64)
    if data.ndim == 2:
        data = data.mean(axis=1)
    return data, sr


def _project(reference_sources, estimated_source, flen):
    """Least-squares projection of the estimate onto the references delayed by 0 .. flen-1 samples."""
    from scipy.linalg import toeplitz
    from scipy.signal import fftconvolve
    nsrc, nsampl = reference_sources.shape
    refs = np.hstack((reference_sources, np.zeros((nsrc, flen - 1))))
    est = np.hstack((estimated_source, np.zeros(flen - 1)))
    n_fft = int(2 ** np.ceil(np.log2(nsampl + flen - 1.0)))
    sf = np.fft.fft(refs, n=n_fft, axis=1)
    sef = np.fft.fft(est, n=n_fft)
    G = np.zeros((nsrc * flen, nsrc * flen))
    for i in range(nsrc):
        for j in range(i + 1):
            ssf = np.real(np.fft.ifft(sf[i] * np.conj(sf[j])))
            ss = toeplitz(np.hstack((ssf[0], ssf[-1:-flen:-1])), r=ssf[:flen])
            G[i * flen:(i + 1) * flen, j * flen:(j + 1) * flen] = ss
            G[j * flen:(j + 1) * flen, i * flen:(i + 1) * flen] = ss.T
    D = np.zeros(nsrc * flen)
    for i in range(nsrc):
        ssef = np.real(np.fft.ifft(sf[i] * np.conj(sef)))
        D[i * flen:(i + 1) * flen] = np.hstack((ssef[0], ssef[-1:-flen:-1]))
    try:
        C = np.linalg.solve(G, D).reshape(flen, nsrc, order="F")
    except np.linalg.LinAlgError:
        C = np.linalg.lstsq(G, D, rcond=None)[0].reshape(flen, nsrc, order="F")
    sproj = np.zeros(nsampl + flen - 1)
    for i in range(nsrc):
        sproj += fftconvolve(C[:, i], refs[i])[:nsampl + flen - 1]
    return sproj


def _decompose(reference_sources, estimated_source, j, flen):
    """estimate = s_true + e_spat + e_interf + e_artif with respect to reference source j."""
    nsampl = estimated_source.size
    s_true = np.hstack((reference_sources[j], np.zeros(flen - 1)))
    e_spat = _project(reference_sources[j, np.newaxis, :], estimated_source, flen) - s_true
    e_interf = _project(reference_sources, estimated_source, flen) - s_true - e_spat
    e_artif = -s_true - e_spat - e_interf
    e_artif[:nsampl] += estimated_source
    return s_true, e_spat, e_interf, e_artif


def _safe_db(num, den):
    return np.inf if den == 0 else 10.0 * np.log10(num / den)


def _criteria(s_true, e_spat, e_interf, e_artif):
    s_filt = s_true + e_spat
    sdr = _safe_db(np.sum(s_filt ** 2), np.sum((e_interf + e_artif) ** 2))
    sir = _safe_db(np.sum(s_filt ** 2), np.sum(e_interf ** 2))
    sar = _safe_db(np.sum((s_filt + e_interf) ** 2), np.sum(e_artif ** 2))
    return sdr, sir, sar


def bss_eval_sources(reference_sources, estimated_sources, compute_permutation=True, flen=FILTER_LEN):
    """(sdr, sir, sar, perm) per reference source, as mir_eval.separation.bss_eval_sources returns them
    (evaluate.py:58,74): perm[i] is the index of the estimate matched to reference i."""
    ref = np.atleast_2d(np.asarray(reference_sources, dtype=np.float64))
    est = np.atleast_2d(np.asarray(estimated_sources, dtype=np.float64))
    if ref.shape != est.shape:
        raise ValueError(f"reference {ref.shape} and estimate {est.shape} must have the same shape")
    nsrc = ref.shape[0]
    if not compute_permutation:
        out = np.array([_criteria(*_decompose(ref, est[j], j, flen)) for j in range(nsrc)])
        return out[:, 0], out[:, 1], out[:, 2], np.arange(nsrc)
    sdr, sir, sar = (np.empty((nsrc, nsrc)) for _ in range(3))
    for jest in range(nsrc):
        for jtrue in range(nsrc):
            sdr[jest, jtrue], sir[jest, jtrue], sar[jest, jtrue] = _criteria(*_decompose(ref, est[jest], jtrue, flen))
    perms = list(itertools.permutations(range(nsrc)))
    idx = np.arange(nsrc)
    mean_sir = [np.mean(sir[list(p), idx]) for p in perms]
    popt = list(perms[int(np.argmax(mean_sir))])
    return sdr[popt, idx], sir[popt, idx], sar[popt, idx], np.asarray(popt)


def compute_metrics_for_track(mix_path, vocal_ref_path, vocal_est_path):
    """evaluate.py:26-84: vocal SDR / SIR / SAR on (vocal, mixture - vocal) and NSDR against the mixture."""
    mix, sr_mix = load_mono_audio(mix_path)
    vocal_ref, sr_ref = load_mono_audio(vocal_ref_path)
    vocal_est, sr_est = load_mono_audio(vocal_est_path)
    if not (sr_mix == sr_ref == sr_est):
        raise ValueError(f"Sample rate mismatch: mix={sr_mix}, ref={sr_ref}, est={sr_est}")
    n = min(len(mix), len(vocal_ref), len(vocal_est))
    mix, vocal_ref, vocal_est = mix[:n], vocal_ref[:n], vocal_est[:n]
    return metrics_from_waveforms(mix, vocal_ref, vocal_est)


def metrics_from_waveforms(mix, vocal_ref, vocal_est):
    sources_ref = np.stack([vocal_ref, mix - vocal_ref], axis=0)
    sources_est = np.stack([vocal_est, mix - vocal_est], axis=0)
    sdr, sir, sar, perm = bss_eval_sources(sources_ref, sources_est)
    v = int(perm[0])                                   # estimate matched to the vocal reference (evaluate.py:62)
    sdr_mix, _, _, _ = bss_eval_sources(vocal_ref[None, :], mix[None, :])
    return {"SDR": float(sdr[v]), "SIR": float(sir[v]), "SAR": float(sar[v]), "NSDR": float(sdr[v]) - float(sdr_mix[0])}


def main(argv=None):
    parser = argparse.ArgumentParser(description="Evaluate SVS results with SDR / SIR / SAR / NSDR (vocal only).")
    parser.add_argument("--est", type=str, required=True, help="folder of predicted vocal wav files")
    parser.add_argument("--mix", type=str, required=True, help="folder of mixture wav files")
    parser.add_argument("--ref", type=str, required=True, help="folder of reference vocal wav files")
    parser.add_argument("--ext", type=str, default="wav")
    parser.add_argument("--out_csv", type=str, default=None)
    args = parser.parse_args(argv)
    pred_files = sorted(glob.glob(os.path.join(args.est, f"*.{args.ext}")))
    if not pred_files:
        print(f"[Error] No *.{args.ext} files found in {args.est}")
        return
    print("=== Start Evaluation ===")
    print(f"#tracks = {len(pred_files)}\n")
    results = []
    for pred_path in pred_files:
        base = os.path.basename(pred_path)
        mix_path, ref_path = os.path.join(args.mix, base), os.path.join(args.ref, base)
        if not os.path.exists(mix_path):
            print(f"[Warning] Mixture file not found, skip: {mix_path}")
            continue
        if not os.path.exists(ref_path):
            print(f"[Warning] Vocal ref file not found, skip: {ref_path}")
            continue
        try:
            m = compute_metrics_for_track(mix_path, ref_path, pred_path)
        except Exception as e:                          # evaluate.py:127-131
            print(f"[Error] Failed on {base}: {e}")
            continue
        name = os.path.splitext(base)[0]
        print(f"{name[:20]}:\tSDR={m['SDR']:.3f} dB,\tSIR={m['SIR']:.3f} dB,\tSAR={m['SAR']:.3f} dB,\tNSDR={m['NSDR']:.3f} dB")
        results.append({"track": name, **m})
    if not results:
        print("\n[Error] No valid tracks evaluated.")
        return
    print("\n=== Overall Mean Metrics (vocal) ===")
    for k in ("SDR", "SIR", "SAR", "NSDR"):
        print(f"Mean {k:4s}: {float(np.mean([r[k] for r in results])):.3f} dB")
    if args.out_csv is not None:
        with open(args.out_csv, "w", newline="", encoding="utf-8") as f:
            w = csv.DictWriter(f, fieldnames=["track", "SDR", "SIR", "SAR", "NSDR"])
            w.writeheader()
            w.writerows(results)
        print(f"\n[Info] Results saved to {args.out_csv}")
    return results


if __name__ == "__main__":
    main()
