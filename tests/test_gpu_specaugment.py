"""GPU test of the on-device SpecAugment (reference: a Python loop with one `.item()` host sync
per mask, training/caiman_asr_train/data/features.py:78-115).  The device version draws every mask of the batch in a
few vectorised ops without a sync; checked here on the GPU it runs on in the timed step: mask semantics (zero fill,
rectangular, inside the utterance), the reference's width distributions, determinism under a seed, no host sync."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _runs(mask_1d):
    """lengths of the zero runs of a boolean vector (True = kept)"""
    z = (~mask_1d).astype(np.int8)
    d = np.diff(np.concatenate([[0], z, [0]]))
    return (np.where(d == -1)[0] - np.where(d == 1)[0]).tolist()


def test_spec_augment_on_device_masks():
    from caiman_asr_amd.data.features import SpecAugment

    B, F, T = 64, 80, 900
    torch.manual_seed(0)
    lens = torch.randint(300, T + 1, (B,), device=DEV)
    x = torch.rand(B, F, T, device=DEV) + 1.0          # strictly positive: zeros are masks
    spec = SpecAugment(freq_masks=2, min_freq=0, max_freq=20, time_masks=10, min_time=0, max_time=0.03)
    torch.manual_seed(123)
    y, out_lens = spec((x.clone(), lens))
    torch.manual_seed(123)
    y2, _ = spec((x.clone(), lens))
    assert torch.equal(y, y2) and torch.equal(out_lens, lens)
    assert y.shape == x.shape and y.is_cuda
    kept = (y != 0)
    assert torch.equal(torch.where(kept, y, x), x)     # values are either untouched or zero
    k = kept.cpu().numpy()
    ln = lens.cpu().numpy()
    f_widths, t_widths = [], []
    for b in range(B):
        valid = k[b, :, :ln[b]]
        # a mask is a full band: a frequency row is either masked over the whole utterance or follows the time masks
        row_all_zero = ~valid.any(1)
        col_all_zero = ~valid.any(0)
        assert np.array_equal(valid, ~(row_all_zero[:, None] | col_all_zero[None, :]))   # rectangular structure
        f_runs, t_runs = _runs(~row_all_zero), _runs(~col_all_zero)
        assert len(f_runs) <= 2 and len(t_runs) <= 10                  # masks may overlap or be empty
        assert sum(f_runs) <= 2 * 20 and all(w <= 10 * round(0.03 * ln[b]) for w in t_runs)
        f_widths += f_runs
        t_widths += [w / ln[b] for w in t_runs]
    # widths are U[0, 20] bins / U[0, 3 %] of the utterance: the merged-run means sit near the single-mask means
    assert 6.0 < np.mean(f_widths) < 16.0
    assert 0.008 < np.mean(t_widths) < 0.03


def test_spec_augment_issues_no_host_sync():
    from caiman_asr_amd.data.features import SpecAugment

    spec = SpecAugment(freq_masks=2, min_freq=0, max_freq=20, time_masks=10, min_time=0, max_time=0.03)
    x = torch.rand(8, 80, 500, device=DEV)
    lens = torch.full((8,), 500, device=DEV)
    spec((x, lens))
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        spec((x, lens))
    finally:
        torch.cuda.set_sync_debug_mode("default")
