"""On-device feature processors.  Interface mirror of training/caiman_asr_train/data/features.py
(`SpecAugment` :34-115, `stack_subsample_frames` :118-139, `FrameSplicing` :142-157,
`PermuteAudio` :160-162).  SpecAugment is restated without the reference's per-mask `.item()`
host round trips: all mask geometry is drawn on the device in one shot."""
import torch
import torch.nn as nn


class BaseFeatures(nn.Module):
    @torch.no_grad()
    def calculate_features(self, audio, audio_lens):
        return audio, audio_lens

    def __call__(self, x):
        audio, audio_lens = x
        return self.calculate_features(audio, audio_lens)


class SpecAugment(BaseFeatures):
    """Zero `freq_masks` frequency bands of width U[min_freq, max_freq] and `time_masks` time
    spans of width U[min_time, max_time]; values in (0,1) are adaptive fractions of the
    utterance length (features.py:88-101).  x: [B, F, T]."""

    def __init__(self, freq_masks=0, min_freq=0, max_freq=10, time_masks=0, min_time=0, max_time=10,
                 noise_magnitude=0):
        super().__init__()
        assert 0 <= min_freq <= max_freq
        assert 0 <= min_time <= max_time
        if noise_magnitude:
            raise NotImplementedError("noise_magnitude > 0 is unused by the shipped configs")
        self.freq_masks, self.min_freq, self.max_freq = freq_masks, min_freq, max_freq
        self.time_masks, self.min_time, self.max_time = time_masks, min_time, max_time
        self.noise_magnitude = noise_magnitude

    @torch.no_grad()
    def make_mask(self, shape, x_lens, device, generator=None):
        B, F, T = shape
        lens = x_lens.to(device=device, dtype=torch.float32)
        mask = torch.zeros((B, F, T), dtype=torch.bool, device=device)

        def rnd(*size):
            return torch.rand(*size, device=device, generator=generator)

        if self.freq_masks > 0:
            w = torch.floor(rnd(B, self.freq_masks) * (self.max_freq - self.min_freq + 1)) + self.min_freq
            f0 = torch.floor(rnd(B, self.freq_masks) * torch.clamp(F - w + 1, min=1))
            idx = torch.arange(F, device=device).view(1, 1, F)
            fm = ((idx >= f0.unsqueeze(-1)) & (idx < (f0 + w).unsqueeze(-1))).any(1)
            mask |= fm.unsqueeze(-1)
        # adaptive count / width per utterance
        if 0 < self.time_masks < 1.0:
            n_masks = torch.round(lens * self.time_masks)
            max_n = int(round(T * self.time_masks)) + 1
        else:
            n_masks = torch.full((B,), float(self.time_masks), device=device)
            max_n = int(self.time_masks)
        if max_n > 0:
            if 0 < self.max_time < 1.0:
                max_t = torch.round(lens * self.max_time)
            else:
                max_t = torch.full((B,), float(self.max_time), device=device)
            w = torch.floor(rnd(B, max_n) * (max_t.unsqueeze(1) - self.min_time + 1)) + self.min_time
            t0 = torch.floor(rnd(B, max_n) * torch.clamp(T - w + 1, min=1))
            live = torch.arange(max_n, device=device).view(1, max_n) < n_masks.view(B, 1)
            w = torch.where(live, w, torch.zeros_like(w))
            idx = torch.arange(T, device=device).view(1, 1, T)
            tm = ((idx >= t0.unsqueeze(-1)) & (idx < (t0 + w).unsqueeze(-1))).any(1)
            mask |= tm.unsqueeze(1)
        return mask

    @torch.no_grad()
    def calculate_features(self, x, x_lens):
        mask = self.make_mask(x.shape, x_lens, x.device)
        return x.masked_fill(mask, 0), x_lens


def stack_subsample_frames(x, x_lens, stacking=1, subsampling=1):
    """[B, F, T] -> [B, F*stacking, ceil(T/subsampling)]: frame t carries frames t..t+stacking-1
    (zeros past the end), then every `subsampling`-th frame is kept (features.py:118-139)."""
    B, F, T = x.shape
    if stacking > 1:
        xp = torch.cat([x, x.new_zeros(B, F, stacking - 1)], 2)
        x = torch.cat([xp[:, :, n:n + T] for n in range(stacking)], 1)
    x = x[:, :, ::subsampling]
    if subsampling > 1:
        x_lens = torch.ceil(x_lens.float() / subsampling).int()
        max_len = int(x_lens.max().item())
        if x.size(2) > max_len:
            assert abs(x.size(2) - max_len) <= 1
            x = x[:, :, :max_len]
    return x, x_lens


class FrameSplicing(BaseFeatures):
    def __init__(self, frame_stacking=1, frame_subsampling=1):
        super().__init__()
        self.frame_stacking = frame_stacking
        self.frame_subsampling = frame_subsampling

    def calculate_features(self, x, x_lens):
        if self.frame_stacking > 1 or self.frame_subsampling > 1:
            x, x_lens = stack_subsample_frames(x, x_lens, self.frame_stacking, self.frame_subsampling)
        return x, x_lens


class PermuteAudio(nn.Module):
    def forward(self, x):
        return (x[0].permute(2, 0, 1), *x[1:])
