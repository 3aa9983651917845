"""numpy (float64) restatement of the log-mel frontend chain — TEST INFRASTRUCTURE ONLY.

Follows SURVEY.md Appendix A.4 / training/caiman_asr_train/data/dali/pipeline.py:260-315,439-462:
DALI ops PreemphasisFilter(border=clamp) -> Spectrogram(nfft, window_length, window_step,
center_windows=False, power=2, default Hann window) -> MelFilterBank(slaney, normalize=True) ->
ToDecibels(multiplier=ln10, reference=1, cutoff_db=ln 1e-20) == ln(max(x, 1e-20)) -> Normalize.
DALI itself is not in this image and the reference's golden tensor needs a FLAC decoder, so the
operator defaults restated here are NOT pinned against DALI ("parity unpinned").
Written independently of caiman_asr_amd/data/frontend.py (explicit per-bin loops, np.fft).
"""
import numpy as np


def _hann_dali(n):
    return np.array([0.5 * (1.0 - np.cos(2.0 * np.pi * (t + 0.5) / n)) for t in range(n)])


def _slaney_hz_to_mel(f):
    return f / (200.0 / 3.0) if f < 1000.0 else 15.0 + np.log(f / 1000.0) / (np.log(6.4) / 27.0)


def _slaney_mel_to_hz(m):
    return m * (200.0 / 3.0) if m < 15.0 else 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))


def mel_weights(sr, nfft, nmel):
    lo, hi = _slaney_hz_to_mel(0.0), _slaney_hz_to_mel(sr / 2.0)
    pts = [_slaney_mel_to_hz(lo + (hi - lo) * i / (nmel + 1)) for i in range(nmel + 2)]
    w = np.zeros((nmel, nfft // 2 + 1))
    for i in range(nmel):
        f0, f1, f2 = pts[i], pts[i + 1], pts[i + 2]
        for k in range(nfft // 2 + 1):
            f = k * sr / nfft
            if f0 < f <= f1:
                w[i, k] = (f - f0) / (f1 - f0)
            elif f1 < f < f2:
                w[i, k] = (f2 - f) / (f2 - f1)
        w[i] *= 2.0 / (f2 - f0)
    return w


def logmel(samples, sr=16000, win=400, hop=160, nfft=512, nmel=80, preemph=0.97, initial_pad=240, noise=None,
           dither=0.0):
    """One utterance: samples [n] -> [nmel, n_frames]."""
    x = np.concatenate([np.zeros(initial_pad), np.asarray(samples, dtype=np.float64)])
    if noise is not None:
        x = x + dither * noise
    y = np.empty_like(x)
    y[0] = x[0] - preemph * x[0]  # clamp border
    y[1:] = x[1:] - preemph * x[:-1]
    nfr = (len(x) - win) // hop + 1
    w = _hann_dali(win)
    mw = mel_weights(sr, nfft, nmel)
    out = np.zeros((nmel, max(nfr, 0)))
    for f in range(max(nfr, 0)):
        seg = np.zeros(nfft)
        seg[:win] = y[f * hop:f * hop + win] * w
        p = np.abs(np.fft.rfft(seg)) ** 2
        out[:, f] = np.log(np.maximum(mw @ p, 1e-20))
    return out


def normalize(feats, n, ds_mean=None, ds_std=None, ratio=0.0):
    """feats [nmel, T] with n valid frames -> blended normalisation, zeros past n."""
    out = np.zeros_like(feats)
    v = feats[:, :n]
    mu, sd = v.mean(1, keepdims=True), v.std(1, keepdims=True)  # ddof = 0
    o = (1.0 - ratio) * (v - mu) / sd if ratio < 1.0 else 0.0
    if ratio > 0.0:
        o = o + ratio * (v - ds_mean[:, None]) / ds_std[:, None]
    out[:, :n] = o
    return out
