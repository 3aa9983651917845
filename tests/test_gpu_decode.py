"""GPU parity of the greedy decoders: token ids and frame indices bit-exact vs the REFERENCE's own
RNNTBatchedGreedyDecoder (tests/golden, oracle/gen_golden.py), confidences to 1e-4; streaming == offline."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def build(tag):
    from caiman_asr_amd.rnnt.model import RNNT

    g = np.load(os.path.join(GOLD, f"rnnt_{tag}.npz"))
    sd = {k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd.")}
    cfg = dict(json.loads(str(g["cfg"])), custom_lstm=True)
    m = RNNT(n_classes=int(g["n_classes"]), **cfg)
    m.load_state_dict(sd)
    return g, m.to(DEV).eval()


@pytest.mark.parametrize("tag", ["tiny", "mfma"])
@pytest.mark.parametrize("sync_every", [1, 8])
def test_greedy_matches_reference_decoder(tag, sync_every):
    from caiman_asr_amd.rnnt.decoder import RNNTBatchedGreedyDecoder, flatten_responses

    g, m = build(tag)
    V = int(g["n_classes"])
    dec = RNNTBatchedGreedyDecoder(m, blank_idx=V - 1, eos_strategy=None, max_inputs_per_batch=int(1e7),
                                   tokenizer=None, max_symbols_per_step=3, sync_every=sync_every)
    res = dec.decode(torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV))
    toks, frames, confs = flatten_responses(res)
    ref = json.loads(str(g["greedy"]))
    assert toks == ref["tokens"]
    assert frames == ref["frames"]
    for a, b in zip(confs, ref["confidence"]):
        assert np.allclose(a, b, atol=1e-4)
    # response objects carry the reference's fields
    first = res[0][min(res[0])]
    assert first.partials is None and first.final.duration_frames == 1 and not first.final.is_provisional


def test_greedy_chunked_encoder_equals_full():
    # encode_lower_batch_size == encode (training/tests/rnnt/test_unbatch_encoder.py:8-21)
    from caiman_asr_amd.rnnt.decoder import encode_lower_batch_size

    g, m = build("tiny")
    x, xl = torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV)
    with torch.no_grad():
        f1, l1 = encode_lower_batch_size(m, x, xl, int(1e7))
        f2, l2 = encode_lower_batch_size(m, x, xl, x.shape[0] * x.shape[2])  # one utterance at a time
    assert torch.equal(l1, l2) and torch.allclose(f1, f2, atol=1e-6)


def test_eos_strategies():
    from caiman_asr_amd.rnnt.decoder import RNNTCommonDecoder
    from caiman_asr_amd.rnnt.eos_strategy import EOSBlank, EOSIgnore, EOSPredict

    _, m = build("tiny")
    lp = torch.log_softmax(torch.randn(4, 30, device=DEV), -1)
    d = RNNTCommonDecoder(m, 29, EOSIgnore(3), None, 30)
    assert torch.isinf(d._eos_prob_correction(lp.clone())[:, 3]).all()
    d = RNNTCommonDecoder(m, 29, EOSBlank(3), None, 30)
    out = d._eos_prob_correction(lp.clone())
    assert torch.allclose(out[:, 29], torch.logaddexp(lp[:, 29], lp[:, 3])) and torch.isinf(out[:, 3]).all()
    d = RNNTCommonDecoder(m, 29, EOSPredict(3, 2.0, 0.5), None, 30)
    out = d._eos_prob_correction(lp.clone())
    exp = torch.where(lp[:, 3] * 2 > np.log(0.5), lp[:, 3] * 2, torch.full_like(lp[:, 3], -float("inf")))
    assert torch.equal(out[:, 3], exp) and d.eos_index == 3


def test_streaming_equals_offline():
    """Feeding the same features 2 frames (= 60 ms at the base config) at a time with carried encoder /
    prediction state reproduces the offline greedy transcript (state-passing equivalence,
    training/tests/rnnt/test_model.py:107-296, applied to decode)."""
    from caiman_asr_amd.rnnt.decoder import RNNTBatchedGreedyDecoder, StreamingGreedyDecoder, flatten_responses

    g, m = build("mfma")
    V = int(g["n_classes"])
    T = 22
    x = torch.tensor(g["x"][:T], device=DEV)
    B = x.shape[1]
    lens = torch.full((B,), T, device=DEV)
    off = RNNTBatchedGreedyDecoder(m, V - 1, None, int(1e7), None, max_symbols_per_step=30)
    toks, frames, _ = flatten_responses(off.decode(x, lens))
    sd = StreamingGreedyDecoder(m, V - 1, n_streams=B, max_symbols_per_step=30)
    got = [[] for _ in range(B)]
    got_frames = [[] for _ in range(B)]
    enc_frame = 0
    for s in range(0, T, 2):
        for tk, n in sd.step(x[s:s + 2].contiguous()):
            tk, n = tk.cpu(), n.cpu()
            for b in range(B):
                got[b] += tk[b, : int(n[b])].tolist()
                got_frames[b] += [enc_frame] * int(n[b])
            enc_frame += 1
    assert got == toks and got_frames == frames


@pytest.mark.parametrize("tag", ["default", "capped", "wide", "partials", "forced_finals", "keywords", "eos_blank",
                                 "eos_terminal", "sample_cap"])
def test_beam_matches_reference_decoder(tag, tmp_path):
    """Beam search with the prediction / joint steps on the HIP kernels: the finals (token ids, frames, the frames
    they were shipped on) and every partial equal the REFERENCE decoder's on the golden mini model."""
    from tests.test_beam_host import BEAM, build_decoder_from_case, check_against_reference

    g, m = build("mfma")
    with torch.no_grad():
        m.joint_fc.bias[0] = BEAM["unk_bias"]
    case = BEAM["results"][tag]
    dec = build_decoder_from_case(m, int(g["n_classes"]), case, tmp_path)
    res = dec.decode(torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV))
    check_against_reference(res, case, conf_atol=1e-4)


def test_beam_expander_batches_hypotheses():
    """One device step for N hypotheses == N single steps (scores, survivors, states)."""
    from caiman_asr_amd.rnnt.beam import BeamExpander

    g, m = build("mfma")
    V = int(g["n_classes"])
    ex = BeamExpander(m, V - 1)
    f, _, _ = m.encode(torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV))
    fr = f[:, 2:3]
    first = ex.expand(fr, None, None)
    y = torch.tensor([[3], [7], [11]], device=DEV)
    state = (torch.cat([e.pred_state[0] for e in first], 1), torch.cat([e.pred_state[1] for e in first], 1))
    both = ex.expand(fr, y, state)
    for i in range(3):
        one = ex.expand(fr[i:i + 1], y[i:i + 1], (state[0][:, i:i + 1].contiguous(), state[1][:, i:i + 1].contiguous()))[0]
        assert one.tokens.tolist() == both[i].tokens.tolist()
        assert torch.allclose(one.scores, both[i].scores, atol=1e-5) and abs(one.blank_logp - both[i].blank_logp) < 1e-5
        assert torch.allclose(one.pred_state[0], both[i].pred_state[0], atol=1e-6)
