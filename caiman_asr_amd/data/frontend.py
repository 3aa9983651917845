"""Log-mel frontend + feature normalisation on the GPU (csrc/frontend.hip).

Host-side mirror of the DALI graph tail the reference builds
(training/caiman_asr_train/data/dali/pipeline.py:260-315,439-462) and of `MelFeatNormalizer`
(training/caiman_asr_train/data/dali/mel_normalization.py:38-141): same parameter names as the
YAML `filterbank_features` block.  The tables (window, twiddles, mel weights) are computed here once
in double precision and kept on the device.
"""
import math
from enum import Enum
from typing import Optional

import numpy as np
import torch

from caiman_asr_amd import _lib


def hann_window(n: int) -> np.ndarray:
    """DALI's default Spectrogram window: 0.5*(1 - cos(2*pi*(i + 0.5)/n))."""
    i = np.arange(n, dtype=np.float64)
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * (i + 0.5) / n))


def _hz_to_mel(f):  # Slaney scale (DALI MelFilterBank default mel_formula='slaney')
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3.0))


def mel_filterbank(sample_rate: int, nfft: int, nmel: int, fmin: float = 0.0, fmax: Optional[float] = None,
                   normalize: bool = True) -> np.ndarray:
    """[nmel, nfft/2+1] triangular filters, mel-spaced corners, weights linear in Hz, area-normalised."""
    fmax = sample_rate / 2.0 if fmax is None else fmax
    corners = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), nmel + 2))
    freqs = np.arange(nfft // 2 + 1, dtype=np.float64) * sample_rate / nfft
    w = np.zeros((nmel, nfft // 2 + 1))
    for i in range(nmel):
        f0, f1, f2 = corners[i], corners[i + 1], corners[i + 2]
        up = (freqs - f0) / (f1 - f0)
        down = (f2 - freqs) / (f2 - f1)
        w[i] = np.maximum(0.0, np.minimum(up, down))
        if normalize:
            w[i] *= 2.0 / (f2 - f0)
    return w


class LogMelFrontend(torch.nn.Module):
    """audio [B, max_samples] f32 + lengths -> log-mel [B, nmel, max_frames] f32 + frame counts."""

    def __init__(self, sample_rate=16000, window_size=0.025, window_stride=0.01, n_fft=512, n_filt=80,
                 dither=1e-5, preemph_coeff=0.97, turn_off_initial_padding=False, device="cuda"):
        super().__init__()
        self.sample_rate = sample_rate
        self.win_len = int(window_size * sample_rate)
        self.hop = int(window_stride * sample_rate)
        self.n_fft, self.n_filt = n_fft, n_filt
        self.dither, self.preemph = float(dither), float(preemph_coeff)
        # the ASR server pads the start with sr*(win - stride) zeros; train/val do the same (pipeline.py:260-268)
        self.initial_pad = 0 if turn_off_initial_padding else int(sample_rate * (window_size - window_stride))
        w = mel_filterbank(sample_rate, n_fft, n_filt)
        lo = np.array([np.nonzero(r)[0][0] if r.any() else 0 for r in w], dtype=np.int32)
        hi = np.array([np.nonzero(r)[0][-1] + 1 if r.any() else 0 for r in w], dtype=np.int32)
        k = np.arange(n_fft // 2, dtype=np.float64)
        dev = torch.device(device)
        self.register_buffer("window", torch.tensor(hann_window(self.win_len), dtype=torch.float32, device=dev))
        self.register_buffer("tw_cos", torch.tensor(np.cos(2 * np.pi * k / n_fft), dtype=torch.float32, device=dev))
        self.register_buffer("tw_sin", torch.tensor(np.sin(2 * np.pi * k / n_fft), dtype=torch.float32, device=dev))
        self.register_buffer("mel_w", torch.tensor(w, dtype=torch.float32, device=dev).contiguous())
        self.register_buffer("mel_lo", torch.tensor(lo, device=dev))
        self.register_buffer("mel_hi", torch.tensor(hi, device=dev))

    def n_frames(self, n_samples):
        return (n_samples + self.initial_pad - self.win_len) // self.hop + 1

    @torch.no_grad()
    def forward(self, audio: torch.Tensor, audio_lens: torch.Tensor, seed: Optional[int] = None):
        _lib.check_input(audio, "audio")
        if audio.dtype != torch.float32 or audio.dim() != 2:
            raise RuntimeError("audio must be a [B, max_samples] float32 tensor")
        B, S = audio.shape
        lens = audio_lens.to(device=audio.device, dtype=torch.int32).contiguous()
        max_frames = max(1, int(self.n_frames(S)))
        out = torch.empty((B, self.n_filt, max_frames), dtype=torch.float32, device=audio.device)
        out_len = torch.empty((B,), dtype=torch.int32, device=audio.device)
        dither = self.dither if self.training or seed is not None else self.dither
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if dither != 0.0 else 0
        with _lib.timed("logmel"):
            _lib.check(_lib.lib().caiman_logmel_forward(
                _lib.ptr(audio), _lib.ptr(lens), B, S, self.win_len, self.hop, self.n_fft, self.n_filt,
                self.initial_pad, self.preemph, dither, seed, 1e-20, _lib.ptr(self.window), _lib.ptr(self.tw_cos),
                _lib.ptr(self.tw_sin), _lib.ptr(self.mel_w), _lib.ptr(self.mel_lo), _lib.ptr(self.mel_hi),
                _lib.ptr(out), _lib.ptr(out_len), max_frames, _lib.stream()))
        return out, out_len


class NormType(Enum):
    DATASET_STATS = 0
    UTTERANCE_STATS = 1
    BLENDED_STATS = 2


class MelFeatNormalizer:
    """Blend of dataset-statistics and per-utterance normalisation on a step schedule
    (mel_normalization.py:38-141; ramp defaults training/caiman_asr_train/setup/mel_normalization.py:50-69)."""

    def __init__(self, mel_means: Optional[torch.Tensor], mel_stddevs: Optional[torch.Tensor],
                 ramp_start_step: Optional[int], ramp_end_step: Optional[int], starting_ratio: float,
                 norm_type: NormType = NormType.BLENDED_STATS):
        if mel_means is None:
            assert norm_type == NormType.UTTERANCE_STATS
        if ramp_start_step is None or ramp_end_step is None:
            assert norm_type != NormType.BLENDED_STATS, "Ramp params are required when using blended stats"
        self.means, self.stddevs = mel_means, mel_stddevs
        self.ramp_start_step, self.ramp_end_step = ramp_start_step, ramp_end_step
        self.starting_ratio = starting_ratio
        self.type = norm_type
        self._step = 0

    @property
    def dataset_to_utt_ratio(self) -> float:
        return self._calc_ratio(self._step)

    def _calc_ratio(self, step: int) -> float:
        if self.type == NormType.DATASET_STATS:
            return 1.0
        if self.type == NormType.UTTERANCE_STATS:
            return 0.0
        if step <= self.ramp_start_step:
            return self.starting_ratio
        if step >= self.ramp_end_step:
            return 1.0
        return self.starting_ratio + (step - self.ramp_start_step) / (
            self.ramp_end_step - self.ramp_start_step) * (1 - self.starting_ratio)

    def step(self, step: int) -> None:
        self._step = step

    @torch.no_grad()
    def __call__(self, feats: torch.Tensor, feat_lens: torch.Tensor) -> torch.Tensor:
        """feats [B, nmel, T] f32 (modified in place and returned)."""
        _lib.check_input(feats, "feats")
        B, M, T = feats.shape
        lens = feat_lens.to(device=feats.device, dtype=torch.int32).contiguous()
        r = self.dataset_to_utt_ratio
        mean = self.means.to(feats.device, torch.float32).contiguous() if r > 0 else None
        std = self.stddevs.to(feats.device, torch.float32).contiguous() if r > 0 else None
        with _lib.timed("mel_norm"):
            _lib.check(_lib.lib().caiman_mel_normalize(
                _lib.ptr(feats), _lib.ptr(lens), B, M, T, _lib.ptr(mean) if mean is not None else None,
                _lib.ptr(std) if std is not None else None, float(r), _lib.stream()))
        return feats


def norm_ramp_params(norm_type: NormType, warmup_steps, hold_steps, half_life_steps, ramp_start=None, ramp_end=None,
                     default_ramp: int = 5000):
    if norm_type != NormType.BLENDED_STATS:
        return None, None
    start = ramp_start if ramp_start is not None else warmup_steps + hold_steps + half_life_steps
    end = ramp_end if ramp_end is not None else start + default_ramp
    return start, end


class StreamingFrontend:
    """The same features for N live audio streams, chunk by chunk (BASELINE configs[4]: "real-time 16 kHz streams").

    `step(audio [N, n])` takes the next n samples of every stream and returns the spliced feature frames they
    complete, [frames, N, n_filt * stacking] -- what `RNNT.encode` eats.  Carried per stream: the last
    `win_len - hop + 1` samples (window overlap + the pre-emphasis neighbour; the zeros an utterance starts with are
    exactly the reference's initial padding, pipeline.py:260-268) and the log-mel frames not yet consumed by the frame
    splicing.  Normalisation uses dataset statistics, as inference does (the hardware checkpoint ships melmeans /
    melvars with melalpha = 0, hardware_ckpt.py:150-156); per-utterance statistics do not exist for a live stream.
    Pre-emphasis runs as one elementwise op on the carried buffer, then the log-mel kernel is called with its own
    pre-emphasis and padding off.  Chunked output == offline output on the concatenated audio (tests)."""

    def __init__(self, frontend: LogMelFrontend, n_streams: int, mel_means: Optional[torch.Tensor] = None,
                 mel_stddevs: Optional[torch.Tensor] = None, frame_stacking: int = 3, frame_subsampling: int = 3):
        assert frontend.initial_pad in (0, frontend.win_len - frontend.hop), "streaming assumes the standard initial padding"
        self.fe, self.N = frontend, n_streams
        self.stack, self.sub = frame_stacking, frame_subsampling
        dev = frontend.window.device
        self.keep = frontend.win_len - frontend.hop + 1
        self.tail = torch.zeros(n_streams, self.keep, device=dev)      # zeros = initial padding (+ a zero neighbour)
        if frontend.initial_pad == 0:
            self.tail = None                                             # first chunk starts the signal itself
        self.mean = None if mel_means is None else mel_means.to(dev, torch.float32).view(1, -1, 1)
        self.istd = None if mel_stddevs is None else (1.0 / mel_stddevs.to(dev, torch.float32)).view(1, -1, 1)
        self.frames = None        # log-mel frames waiting for the splicer, [N, n_filt, k]
        self.samples_seen = 0

    @torch.no_grad()
    def step(self, audio: torch.Tensor) -> Optional[torch.Tensor]:
        fe = self.fe
        assert audio.dim() == 2 and audio.shape[0] == self.N and audio.dtype == torch.float32
        if self.tail is None:    # no initial padding: the very first sample is its own pre-emphasis neighbour
            buf = torch.cat([audio[:, :1], audio], 1)
        else:
            buf = torch.cat([self.tail, audio], 1)
        self.samples_seen += audio.shape[1]
        L = buf.shape[1] - 1
        if L < fe.win_len:
            self.tail = buf
            return None
        n_new = (L - fe.win_len) // fe.hop + 1
        used = (n_new - 1) * fe.hop + fe.win_len                 # samples of y covered by the new frames
        y = (buf[:, 1:used + 1] - fe.preemph * buf[:, :used]).contiguous()
        self.tail = buf[:, n_new * fe.hop:].contiguous()         # overlap for the next frame + the neighbour sample
        lens = torch.full((self.N,), used, dtype=torch.int32, device=buf.device)
        out = torch.empty((self.N, fe.n_filt, n_new), dtype=torch.float32, device=buf.device)
        out_len = torch.empty((self.N,), dtype=torch.int32, device=buf.device)
        seed = self.samples_seen if fe.dither != 0.0 else 0
        _lib.check(_lib.lib().caiman_logmel_forward(
            _lib.ptr(y), _lib.ptr(lens), self.N, used, fe.win_len, fe.hop, fe.n_fft, fe.n_filt, 0, 0.0, fe.dither, seed,
            1e-20, _lib.ptr(fe.window), _lib.ptr(fe.tw_cos), _lib.ptr(fe.tw_sin), _lib.ptr(fe.mel_w), _lib.ptr(fe.mel_lo),
            _lib.ptr(fe.mel_hi), _lib.ptr(out), _lib.ptr(out_len), n_new, _lib.stream()))
        if self.mean is not None:
            out = (out - self.mean) * self.istd
        fr = out if self.frames is None else torch.cat([self.frames, out], 2)
        # splice: output frame j = frames [sub*j, sub*j + stack) stacked on the feature axis
        n_out = (fr.shape[2] - self.stack) // self.sub + 1 if fr.shape[2] >= self.stack else 0
        if n_out <= 0:
            self.frames = fr
            return None
        idx = (torch.arange(n_out, device=fr.device) * self.sub).view(-1, 1) + torch.arange(self.stack, device=fr.device)
        sp = fr[:, :, idx]                                        # [N, F, n_out, stack]
        sp = sp.permute(2, 0, 3, 1).reshape(n_out, self.N, self.stack * fe.n_filt)   # [stack][F] order = cat over frames
        self.frames = fr[:, :, n_out * self.sub:]
        return sp.contiguous()
