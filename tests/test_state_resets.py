"""State-reset segmenter (caiman_asr_amd/evaluate/state_resets.py) against the reference's own functions
(tests/golden/state_resets.json, oracle/gen_golden_state_resets.py) and the known answers the reference documents
(training/caiman_asr_train/evaluate/state_resets/overlap_processing.py:22-32,72-93; tests/evaluate/state_resets/)."""
import json
import os

import pytest
import torch

from caiman_asr_amd.evaluate import state_resets as sr

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_resets.json")))


def test_window_count_and_seconds_to_frames_match_the_reference():
    for all_frames, overlap, window, n, pad in GOLD["plan"]:
        assert sr.window_count(all_frames, window, overlap) == (n, pad)
    for seg, ov, w, o in GOLD["stats"]:
        assert sr.window_frames(seg, ov, GOLD["cfg"]) == (w, o)
    with pytest.raises(ValueError):
        sr.window_frames(0.0, 0.0, GOLD["cfg"])
    with pytest.raises(ValueError):
        sr.window_frames(1.0, 1.0, GOLD["cfg"])
    with pytest.raises(ValueError):
        sr.window_frames(1.0, -0.1, GOLD["cfg"])


@pytest.mark.parametrize("case", range(len(GOLD["reshape"])))
def test_split_batch_matches_the_reference(case):
    c = GOLD["reshape"][case]
    feats = torch.tensor(c["feats"])
    out, lens, plans = sr.split_batch(feats, torch.tensor(c["lens"], dtype=torch.int32), c["sr_segment"], c["sr_overlap"], GOLD["cfg"])
    assert torch.equal(out, torch.tensor(c["out_feats"]))
    assert lens.tolist() == c["out_lens"]
    assert [[p.n_windows, p.window, p.overlap] for p in plans] == c["meta"]


def test_windows_known_answer():
    # [[A B C D E F G H]] with window 4, overlap 2 -> [A B C D] [C D E F] [E F G H]  (the reference's own example shape)
    feats = torch.arange(8.0).view(8, 1, 1)
    w, lens = sr.split_utterance(feats, torch.tensor([8]), 4, 2)
    assert w[:, :, 0].t().tolist() == [[0, 1, 2, 3], [2, 3, 4, 5], [4, 5, 6, 7]] and lens.tolist() == [4, 4, 4]
    w, lens = sr.split_utterance(torch.arange(9.0).view(9, 1, 1), torch.tensor([9]), 4, 1)   # padded tail
    assert w[:, :, 0].t().tolist() == [[0, 1, 2, 3], [3, 4, 5, 6], [6, 7, 8, 0]]
    short = torch.arange(3.0).view(3, 1, 1)
    w, lens = sr.split_utterance(short, torch.tensor([3]), 4, 1)
    assert w is short and lens.tolist() == [3]
    with pytest.raises(AssertionError):
        sr.split_utterance(torch.zeros(8, 2, 1), torch.tensor([8]), 4, 2)


@pytest.mark.parametrize("case", range(len(GOLD["merge"])))
def test_merge_batch_matches_the_reference(case):
    c = GOLD["merge"][case]
    stamps = [[sr.FullStamp(*t) for t in row] for row in c["timestamps"]] if c["full"] else [list(r) for r in c["timestamps"]]
    probs = [list(p) for p in c["probs"]] if c["with_probs"] else [[] for _ in c["pred"]]
    plans = [sr.WindowPlan(*m) for m in c["meta"]]
    t, s, p = sr.merge_batch([list(x) for x in c["pred"]], stamps, probs, c["enc_time_reduction"], plans, c["eos_idx"])
    assert t == c["out_pred"]
    got = [[[x.model, x.user_perceived] if c["full"] else x for x in row] for row in s]
    assert got == c["out_timestamps"]
    assert p == c["out_probs"]


def test_documented_known_answers():
    # overlap_processing.py:27-30
    # (the documented call is the time shift alone: hop (26 - 6) / 2 = 10 encoder frames; same hop here without an overlap to drop)
    t, s, _ = sr.merge_windows([[0] * 5, [0] * 5, [0] * 2], [[1, 3, 5, 6, 10], [2, 3, 5, 7, 8], [3, 4]], [], 2, 20, 0, lookahead=0)
    assert s == [[1, 3, 5, 6, 10, 12, 13, 15, 17, 18, 23, 24]]
    # overlap_processing.py:72-93: overlap 2 frames, no time reduction
    t, s, _ = sr.merge_windows([[7, 2, 3, 6, 5], [2, 6, 5, 9, 7]], [[1, 2, 3, 4, 6], [1, 3, 4, 5, 6]], [], 1, 10, 2)
    assert t == [[7, 2, 3, 6, 5, 9, 7]]
    assert sr.shift(sr.FullStamp(3, 4), 5) == sr.FullStamp(8, 9) and sr.model_time(sr.FullStamp(3, 4)) == 3
    assert sr.user_perceived_time(sr.FullStamp(3, 4)) == 4 and sr.user_perceived_time(7) == 7


def test_evaluate_with_state_resets_stitches_windows_back():
    """evaluate(sr_segment=...) cuts a long utterance into windows, decodes them as batch rows and merges: with a decoder
    that "transcribes" every non-zero input frame as its value at its encoder frame, the stitched hypothesis is the whole
    utterance once, in order, with timestamps on the original time axis."""
    import caiman_asr_amd.evaluate.core as core

    cfg = GOLD["cfg"]           # 30 ms frames

    class EchoDecoder:
        seen = []

        def decode(self, feats, lens):
            self.seen.append(tuple(feats.shape))
            # per utterance: [(token = the frame's value, encoder frame = input frame // 2)]
            return [[(int(feats[t, b, 0]), t // 2) for t in range(0, int(lens[b]), 2) if int(feats[t, b, 0])]
                    for b in range(feats.shape[1])]

    dec = EchoDecoder()
    T = 46
    feats = torch.zeros(T, 2, 1)
    feats[:, 0, 0] = torch.arange(1, T + 1)
    feats[1::2, 0, 0] = 0               # a token on every even input frame: values 1, 3, 5 ...
    feats[:10, 1, 0] = torch.tensor([5, 0, 6, 0, 7, 0, 8, 0, 9, 0])
    lens = torch.tensor([T, 10], dtype=torch.int32)
    real_flatten = core.flatten_responses   # the decoders' response objects are not the point here
    core.flatten_responses = lambda out: ([[v for v, _ in p] for p in out], [[t for _, t in p] for p in out],
                                          [[1.0 for _ in p] for p in out])
    try:
        res = core.evaluate([(feats, lens, torch.tensor([[1, 2], [3, 4]]), torch.tensor([2, 2]))], dec,
                            lambda ids: " ".join(map(str, ids)), autocast_dtype=None, standardize=False, sr_segment=0.6,
                            sr_overlap=0.12, model_config=cfg, enc_time_reduction=2)
    finally:
        core.flatten_responses = real_flatten
    # 0.6 s = 20 frames per window, 0.12 s = 4 frames overlap: utterance 0 (46 frames) -> 3 windows, utterance 1 stays whole
    assert dec.seen == [(20, 4, 1)]
    assert res["hypotheses"][0] == " ".join(str(v) for v in range(1, T + 1, 2))
    assert res["timestamps"][0] == list(range(0, T // 2))
    assert res["hypotheses"][1] == "5 6 7 8 9" and res["timestamps"][1] == [0, 1, 2, 3, 4]
