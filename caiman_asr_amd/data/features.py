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

    def _time_slots(self, T):
        """mask slots per utterance: the count itself, or room for the longest utterance's adaptive count"""
        return int(round(T * self.time_masks)) + 1 if 0 < self.time_masks < 1.0 else int(self.time_masks)

    def geometry_from_draws(self, r, lens, F, T):
        """The reference's mask arithmetic (features.py:60-101 there) on uniform draws r [B, 2 freq_masks + 2 slots]
        (columns: fw | f0 | tw | t0 draws), as torch operations: what the CPU path runs and what the kernel
        (csrc/frontend.hip::specaug_geometry_kernel) is tested against."""
        B, nf, nt = r.shape[0], self.freq_masks, self._time_slots(T)
        f0 = fw = t0 = tw = None
        if nf > 0:
            fw = torch.floor(r[:, :nf] * (self.max_freq - self.min_freq + 1)) + self.min_freq
            f0 = torch.floor(r[:, nf:2 * nf] * torch.clamp(F - fw + 1, min=1))
        if nt > 0:
            # adaptive count / width per utterance
            if 0 < self.time_masks < 1.0:
                n_masks = torch.round(lens * self.time_masks)
            else:
                n_masks = torch.full((B,), float(self.time_masks), device=r.device)
            if 0 < self.max_time < 1.0:
                max_t = torch.round(lens * self.max_time)
            else:
                max_t = torch.full((B,), float(self.max_time), device=r.device)
            w = torch.floor(r[:, 2 * nf:2 * nf + nt] * (max_t.unsqueeze(1) - self.min_time + 1)) + self.min_time
            t0 = torch.floor(r[:, 2 * nf + nt:] * torch.clamp(T - w + 1, min=1))
            live = torch.arange(nt, device=r.device).view(1, nt) < n_masks.view(B, 1)
            tw = torch.where(live, w, torch.zeros_like(w))
        return f0, fw, t0, tw

    @torch.no_grad()
    def mask_geometry(self, shape, x_lens, device, generator=None):
        """-> (f0, fw, t0, tw): start and width of every frequency mask [B, freq_masks] and time mask [B, slots] (float
        tensors holding integers; width 0 = no mask; None where there are no masks of that kind).  One draw of uniforms
        and, on the GPU, one kernel (caiman_specaug_geometry) -- no host round trip, no chain of small torch kernels."""
        B, F, T = shape
        device = torch.device(device)
        nf, nt = self.freq_masks, self._time_slots(T)
        if nf + nt == 0:
            return None, None, None, None
        r = torch.rand(B, 2 * nf + 2 * nt, device=device, generator=generator)
        if device.type != "cuda":
            return self.geometry_from_draws(r, x_lens.to(device=device, dtype=torch.float32), F, T)
        from caiman_asr_amd import _lib

        lens = x_lens.to(device)
        kind = {torch.int32: 0, torch.int64: 1, torch.float32: 2}.get(lens.dtype)
        if kind is None:
            lens, kind = lens.float(), 2
        lens = lens.contiguous()
        out = torch.empty(B * (2 * nf + 2 * nt), dtype=torch.float32, device=device)
        _lib.check(_lib.lib().caiman_specaug_geometry(
            _lib.ptr(r), _lib.ptr(lens), kind, B, F, T, nf, float(self.min_freq), float(self.max_freq - self.min_freq + 1),
            float(self.time_masks), nt, float(self.min_time), float(self.max_time), _lib.ptr(out), _lib.stream()))
        fw, f0, tw, t0 = torch.split(out, [B * nf, B * nf, B * nt, B * nt])
        two = lambda t, n: t.view(B, n) if n > 0 else None
        return two(f0, nf), two(fw, nf), two(t0, nt), two(tw, nt)

    @torch.no_grad()
    def make_mask(self, shape, x_lens, device, generator=None):
        B, F, T = shape
        f0, fw, t0, tw = self.mask_geometry(shape, x_lens, device, generator)
        mask = torch.zeros((B, F, T), dtype=torch.bool, device=device)
        if f0 is not None:
            idx = torch.arange(F, device=device).view(1, 1, F)
            fm = ((idx >= f0.unsqueeze(-1)) & (idx < (f0 + fw).unsqueeze(-1))).any(1)
            mask |= fm.unsqueeze(-1)
        if t0 is not None:
            idx = torch.arange(T, device=device).view(1, 1, T)
            tm = ((idx >= t0.unsqueeze(-1)) & (idx < (t0 + tw).unsqueeze(-1))).any(1)
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


@torch.no_grad()
def augment_splice_permute(spec, splice, feats, feat_lens, feat_lens_host, generator=None):
    """`PermuteAudio(FrameSplicing(SpecAugment(feats)))` in one kernel (include/caiman_rnnt.h caiman_specaug_splice): feats
    [B, F, T] f32 on the GPU -> (x [T_out, B, F * stacking] contiguous, lens on the host).  The masks are the ones
    `spec.make_mask` would build from the same generator state (`mask_geometry` draws them; the kernel applies them while it
    stacks, subsamples and transposes).  `spec` may be None (no augmentation: evaluation).  The separate modules stay for
    callers that want them one at a time, and are what the tests compare this with."""
    from caiman_asr_amd import _lib

    assert feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 3
    feats = feats.contiguous()
    B, F, T = feats.shape
    st, sub = splice.frame_stacking, splice.frame_subsampling
    lens_out = torch.ceil(feat_lens_host.float() / sub).int() if sub > 1 else feat_lens_host
    t_all = (T + sub - 1) // sub
    t_out = t_all
    if sub > 1:
        max_len = int(lens_out.max().item())
        if t_all > max_len:
            assert t_all - max_len <= 1
            t_out = max_len
    f0 = fw = t0 = tw = None
    if spec is not None:
        f0, fw, t0, tw = spec.mask_geometry(feats.shape, feat_lens, feats.device, generator)
    cf = lambda t: None if t is None else t.float().contiguous()
    f0, fw, t0, tw = cf(f0), cf(fw), cf(t0), cf(tw)
    out = torch.empty((t_out, B, F * st), dtype=torch.float32, device=feats.device)
    _lib.check(_lib.lib().caiman_specaug_splice(
        _lib.ptr(feats), B, F, T, _lib.ptr(f0) if f0 is not None else None, _lib.ptr(fw) if fw is not None else None,
        0 if f0 is None else f0.shape[1], _lib.ptr(t0) if t0 is not None else None, _lib.ptr(tw) if tw is not None else None,
        0 if t0 is None else t0.shape[1], st, sub, t_out, _lib.ptr(out), _lib.stream()))
    return out, lens_out
