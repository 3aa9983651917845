"""GPU parity: log-mel frontend + normalisation kernels vs the numpy oracle (oracle/frontend.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _audio(lens, seed=0):
    rng = np.random.default_rng(seed)
    S = max(lens)
    a = np.zeros((len(lens), S), dtype=np.float32)
    for i, n in enumerate(lens):
        t = np.arange(n) / 16000.0
        a[i, :n] = (0.1 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * (200 + 150 * i) * t)).astype(np.float32)
    return a


@pytest.mark.parametrize("cfg", [dict(), dict(window_size=0.02, turn_off_initial_padding=True)])
def test_logmel_matches_oracle(cfg):
    from caiman_asr_amd.data.frontend import LogMelFrontend
    from oracle import frontend as of

    lens = [16000, 12345, 400, 161]
    a = _audio(lens)
    fe = LogMelFrontend(dither=0.0, device=DEV, **cfg)
    out, out_len = fe(torch.tensor(a, device=DEV), torch.tensor(lens))
    out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
    for i, n in enumerate(lens):
        ref = of.logmel(a[i, :n], win=fe.win_len, hop=fe.hop, initial_pad=fe.initial_pad)
        assert out_len[i] == ref.shape[1] == fe.n_frames(n)
        # fp32 FFT of a 512-point frame vs float64: 2e-4 abs on log-energies (the reference's own DALI
        # equivalence test uses atol 2e-4, training/tests/data/dali/test_data_loader.py:255-258)
        assert np.allclose(out[i, :, :out_len[i]], ref, atol=2e-4, rtol=1e-5), i
        assert np.all(out[i, :, out_len[i]:] == 0)


def test_logmel_of_silence_hits_the_log_floor_and_dither_lifts_it():
    from caiman_asr_amd.data.frontend import LogMelFrontend

    z = torch.zeros(1, 8000, device=DEV)
    fe = LogMelFrontend(dither=0.0, device=DEV)
    out, n = fe(z, torch.tensor([8000]))
    assert torch.allclose(out[0, :, : int(n[0])], torch.full_like(out[0, :, : int(n[0])], float(np.log(1e-20))))
    fd = LogMelFrontend(dither=1e-5, device=DEV)
    o1, _ = fd(z, torch.tensor([8000]), seed=1)
    o2, _ = fd(z, torch.tensor([8000]), seed=1)
    o3, _ = fd(z, torch.tensor([8000]), seed=2)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    # white noise of std 1e-5 through the chain: finite, far above the floor
    assert o1[0, :, :40].min() > -40 and torch.isfinite(o1).all()


@pytest.mark.parametrize("ratio", [0.0, 0.25, 1.0])
def test_normalisation_matches_oracle(ratio):
    from caiman_asr_amd.data.frontend import MelFeatNormalizer, NormType
    from oracle import frontend as of

    rng = np.random.default_rng(3)
    B, M, T = 3, 80, 300
    x = rng.standard_normal((B, M, T)).astype(np.float32) * 3 - 7
    lens = np.array([300, 123, 1])
    mean = rng.standard_normal(M).astype(np.float32) - 7
    std = (rng.random(M).astype(np.float32) + 0.5) * 3
    if ratio == 0.0:
        nz = MelFeatNormalizer(None, None, None, None, 0.0, NormType.UTTERANCE_STATS)
    elif ratio == 1.0:
        nz = MelFeatNormalizer(torch.tensor(mean), torch.tensor(std), None, None, 0.0, NormType.DATASET_STATS)
    else:
        nz = MelFeatNormalizer(torch.tensor(mean), torch.tensor(std), 100, 200, 0.25, NormType.BLENDED_STATS)
    out = nz(torch.tensor(x, device=DEV), torch.tensor(lens)).cpu().numpy()
    for b in range(B):
        if lens[b] == 1 and ratio < 1.0:
            continue  # std of a single frame is 0: undefined in the reference as well
        ref = of.normalize(x[b].astype(np.float64), int(lens[b]), mean.astype(np.float64), std.astype(np.float64), ratio)
        assert np.allclose(out[b], ref, atol=2e-4), (b, ratio)
    if ratio == 0.0:  # recoverable by hand: mean 0 / std 1 over valid frames (test_data_loader.py:334-383)
        v = out[1, :, :123]
        assert np.allclose(v.mean(1), 0, atol=1e-5) and np.allclose(v.std(1), 1, atol=1e-4)


def test_hip_frontend_reproduces_the_reference_golden_logmel():
    """HIP log-mel + utterance normalisation on the reference's test recording vs the reference's golden
    tensor (tests/golden/frontend_ref.npz <- training/tests/test_data/audio_tensor_batch.pt, atol 2e-4 there)."""
    import os

    from caiman_asr_amd.data.frontend import LogMelFrontend, MelFeatNormalizer, NormType

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_ref.npz"))
    x = torch.tensor(g["pcm"].astype(np.float32) / 32768.0, device=DEV).unsqueeze(0)
    fe = LogMelFrontend(window_size=0.02, dither=0.0, turn_off_initial_padding=True, device=DEV)
    out, n = fe(x, torch.tensor([x.shape[1]]))
    assert int(n[0]) == 888
    out = MelFeatNormalizer(None, None, None, None, 0.0, NormType.UTTERANCE_STATS)(out, n)
    err = (out[0].cpu().numpy() - g["logmel_norm"])
    assert np.abs(err).max() < 4e-4 and np.abs(err).mean() < 1e-5


@pytest.mark.parametrize("chunk", [960, 400, 1777])
def test_streaming_frontend_equals_offline(chunk):
    """Audio fed in chunks with carried state gives the frames of the offline chain (log-mel -> dataset-stats
    normalisation -> stack 3 / keep every 3rd) on the whole signal, up to the frames the tail has not completed."""
    from caiman_asr_amd.data.features import stack_subsample_frames
    from caiman_asr_amd.data.frontend import LogMelFrontend, StreamingFrontend

    n = 16000
    a = torch.tensor(_audio([n, n, n], seed=4), device=DEV)
    mean = torch.linspace(-12, -8, 80)
    std = torch.linspace(1.5, 3.0, 80)
    fe = LogMelFrontend(dither=0.0, device=DEV)
    full, full_len = fe(a, torch.tensor([n, n, n]))
    full = (full - mean.to(DEV).view(1, -1, 1)) / std.to(DEV).view(1, -1, 1)
    ref, ref_len = stack_subsample_frames(full, full_len, 3, 3)          # [B, 240, T3]
    sf = StreamingFrontend(fe, 3, mean, std)
    got = []
    for t0 in range(0, n, chunk):
        out = sf.step(a[:, t0:t0 + chunk].contiguous())
        if out is not None:
            got.append(out)
    got = torch.cat(got, 0)                                                 # [frames, B, 240]
    k = got.shape[0]
    n_complete = (int(full_len[0]) - 3) // 3 + 1                            # spliced frames whose 3 inputs all exist
    assert k == n_complete and k >= int(ref_len[0]) - 1
    assert torch.allclose(got.permute(1, 2, 0), ref[:, :, :k], atol=2e-4, rtol=1e-5)
