"""Host side of beam search on CPU.

Pins: (1) the reference's own keyword-trie known-answer table (training/tests/keywords/test_trie.py:14-35),
(2) the reference's MockModel known-answer cases for the beam decoder (training/tests/rnnt/test_decoders.py:
13-50, 88-89 -- the two cases that need no n-gram file), (3) finals produced by the reference's RNNTBeamDecoder
on the golden mini model (tests/golden/beam_mfma.json, oracle/gen_golden.py) with the network evaluated by the
CPU oracle, so that only the search logic is under test here.
"""
import json
import os

import numpy as np
import pytest
import torch

from caiman_asr_amd.keywords.process import load_keywords
from caiman_asr_amd.keywords.trie import Keywords
from caiman_asr_amd.rnnt.beam import RNNTBeamDecoder
from caiman_asr_amd.rnnt.decoder import flatten_responses
from caiman_asr_amd.rnnt.hypothesis import Hypothesis, init_sos_hyp, roll_hash
from caiman_asr_amd.rnnt.serialise_responses import ResponseSerializer
from tests.helpers import OracleBeamStep, OracleRNNT

GOLD = os.path.join(os.path.dirname(__file__), "golden")
BEAM = json.load(open(os.path.join(GOLD, "beam_mfma.json")))
PIECES = BEAM["pieces"]


# ---- keyword automaton ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("keywords, seq, expected", [
    ([], "", []),
    ([("ab", 1.0)], "cabc", [0, 1, 1, 0]),
    ([("abc", 1.0)], "cabx", [0, 1, 1, -2]),
    ([("ab", 1.0), ("abc", 1.0)], "abx", [2, 2, -2]),
    ([("ab", 1.0), ("abcd", 1.0)], "abcx", [2, 2, 1, -3]),
    ([("ab", -1.0)], "cabc", [0, -1, -1, 0]),
    ([("ab", 1.0), ("bcd", -1.0)], "abcx", [1, 0, -1, 2]),
    ([("ab", 1.0), ("abc", -1.0)], "abc", [0, 0, -1]),
    ([("ab", 2.0), ("abc", 1.0)], "abx", [3, 3, -2]),
    ([(" ab", 1.0)], "a ab", [0, 1, 1, 1]),
])
def test_keyword_deltas(keywords, seq, expected):
    kw = Keywords(keywords)
    state, got = Keywords.init(), []
    for ch in seq:
        d, state = kw.step(ch, state)
        got.append(d)
    assert np.allclose(got, expected, atol=1e-6)
    total, _ = kw.steps(seq, Keywords.init())
    assert abs(total - sum(expected)) < 1e-6


def test_keyword_duplicates_and_loader(tmp_path):
    with pytest.raises(AssertionError):
        Keywords([("ab", 1.0), ("ab", 2.0)])
    p = tmp_path / "kw.json"
    p.write_text(json.dumps({"keywords": {"new york": 2.0, "zz": -1}}))
    kw = load_keywords(str(p))
    total, _ = kw.steps("▁new▁york", Keywords.init())
    assert abs(total - 2.0 * len("new▁york")) < 1e-9
    p.write_text(json.dumps({"keywords": {"a": "x"}}))
    with pytest.raises(ValueError):
        load_keywords(str(p))
    p.write_text(json.dumps({"words": {}}))
    with pytest.raises(ValueError):
        load_keywords(str(p))


# ---- hypothesis record ------------------------------------------------------------------------------------------
def test_hypothesis_hash_truncate_clone():
    h = init_sos_hyp(-1)
    assert (h.score, h.y_seq, h.timesteps, h.s_seq, h.hashval, h.y_length_tot) == (0.0, [-1], [-1], ["▁"], 0, 1)
    a, b = h.clone(), h.clone()
    a.update_hash("▁he"); a.update_hash("llo")
    b.update_hash("▁hel"); b.update_hash("lo")
    assert a.hashval == b.hashval == roll_hash(0, "▁hello") != 0     # text, not tokenisation, is hashed
    assert roll_hash(0, "a") == ord("a") and roll_hash(5, "a") == (5 * 0x10FFFF + ord("a")) % 1_000_000_000_039
    for tok, s, t in [(3, "▁a", 0), (4, "b", 2), (5, "▁c", 5)]:
        a.y_seq.append(tok); a.s_seq.append(s); a.timesteps.append(t); a.p_seq.append(0.5)
    tot = a.y_length_tot
    c = a.clone()
    a.truncate(3)      # ships tokens 1..2, token 2 stays as the sentinel
    assert a.y_seq == [4, 5] and a.s_seq == ["b", "▁c"] and a.timesteps == [2, 5] and a.y_length_tot == tot
    assert c.y_seq == [-1, 3, 4, 5] and c.transcript == "ab c"
    a.check(); c.check()


def _hyp(tokens, frames, score=-1.0, pieces=PIECES):
    h = init_sos_hyp(-1)
    for tok, t in zip(tokens, frames):
        h.y_seq.append(tok); h.timesteps.append(t); h.s_seq.append(pieces[tok]); h.p_seq.append(0.9)
        h.update_hash(pieces[tok])
    h.score = score
    return h


def test_serialiser_common_prefix_final():
    ser = ResponseSerializer(lambda hs: sorted(hs, key=lambda h: h.score / h.y_length_tot, reverse=True))
    a, b, c = _hyp([4, 5, 6], [0, 3, 4], -1.0), _hyp([4, 5, 7], [1, 2, 6], -2.0), _hyp([4, 5], [0, 2], -3.0)
    kept = {h.hashval: h for h in (a, b, c)}
    fr, kept2 = ser.frame_responses(kept, 6, partials=True)
    assert fr.final is not None and not fr.final.is_provisional
    alt = fr.final.alternatives[0]
    assert alt.y_seq == [4, 5] and alt.timesteps == [0, 2] and alt.token_seq == [PIECES[4], PIECES[5]]
    assert (fr.final.start_frame_idx, fr.final.duration_frames) == (0, 3)
    assert a.y_seq == [5, 6] and b.y_seq == [5, 7] and c.y_seq == [5]          # prefix removed, sentinel kept
    assert fr.partials.is_provisional and [x.y_seq for x in fr.partials.alternatives] == [[6], [7]]
    assert (fr.partials.start_frame_idx, fr.partials.duration_frames) == (4, 3)
    fr2, _ = ser.frame_responses(kept2, 7, partials=False)                        # nothing shared any more
    assert fr2.final is None and fr2.partials is None
    last = ser.last_frame_response(kept2)
    assert last.final.alternatives[0].y_seq == [6] and last.partials is None
    assert ser.last_frame_response({c.hashval: c}).final is None


# ---- the reference's MockModel known answers ----------------------------------------------------------------------
class MockModel:
    """Only `joint` matters: odd calls say blank, even calls say tokens 1, 2, 3, ... in turn."""

    def __init__(self, vocab_size):
        self.calls, self.blank_idx, self.vocab_size, self.training = 0, 0, vocab_size, False

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def encode(self, x, x_lens, enc_state=None):
        return x.transpose(0, 1), x_lens, None

    def predict(self, y, pred_state=None, add_sos=True, special_sos=None):
        z = torch.zeros(1, 1, 1)
        return z, (z, z), None

    def joint(self, f, g, *a, **k):
        logits = torch.full((1, self.vocab_size + 1), -100.0)
        idx = self.blank_idx if self.calls % 2 == 1 else 1 + (self.calls // 2) % self.vocab_size
        logits[0, idx] = 10.0
        self.calls += 1
        return logits.view(1, 1, 1, -1)


@pytest.mark.parametrize("thresholds, tokens", [((0.4, 1.5), [2, 3, 4]), ((-1, -1), [5, 2, 3])])
def test_beam_mock_model_known_answers(thresholds, tokens):
    dec = RNNTBeamDecoder(MockModel(6), blank_idx=0, eos_strategy=None, sentpiece_model=PIECES, beam_width=4,
                          temperature=1.5, max_symbols_per_step=8, beam_prune_score_thresh=thresholds[0],
                          beam_prune_topk_thresh=thresholds[1], return_partials=True)
    res = dec.decode(torch.randn(4, 1, 1), torch.tensor([4]))
    tk, ts, pr = flatten_responses(res)
    assert tk == [tokens] and ts == [[1, 2, 3]] and pr == [[1.0, 1.0, 1.0]]
    assert all(r.partials is not None for t, r in res[0].items() if t < 4) and res[0][4].partials is None


# ---- the reference's decoder on the golden mini model ---------------------------------------------------------------
def _oracle_model():
    g = np.load(os.path.join(GOLD, "rnnt_mfma.npz"))
    sd = {k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd.")}
    sd["joint_net.2.bias"][0] = BEAM["unk_bias"]
    return g, OracleRNNT(sd, json.loads(str(g["cfg"])))


def build_decoder_from_case(model, V, case, tmp_path, native=False, oracle_step=False):
    """kwargs as stored by oracle/gen_golden.py: `eos` = [kind, idx, (alpha, beta)], `keywords` = {phrase: weight}."""
    from caiman_asr_amd.rnnt.eos_strategy import EOSBlank, EOSIgnore, EOSPredict

    args = dict(case["kwargs"])
    eos = args.pop("eos", None)
    strategy = None if eos is None else {"predict": EOSPredict, "blank": EOSBlank, "ignore": EOSIgnore}[eos[0]](*eos[1:])
    if "keywords" in args:
        kp = tmp_path / "kw.json"
        kp.write_text(json.dumps({"keywords": args.pop("keywords")}))
        args["keyword_boost_path"] = str(kp)
    if not native:
        return RNNTBeamDecoder(model, blank_idx=V - 1, eos_strategy=strategy, sentpiece_model=PIECES, **args)
    from caiman_asr_amd.rnnt.beam_native import RNNTBeamDecoderNative

    dec = RNNTBeamDecoderNative(model, blank_idx=V - 1, eos_strategy=strategy, sentpiece_model=PIECES,
                                device_step=(lambda *a: None) if oracle_step else None, **args)
    if oracle_step:
        dec.step = OracleBeamStep(dec, model)
    return dec


def check_against_reference(res, case, conf_atol=1e-5):
    assert len(res) == len(case["utts"])
    for r, u in zip(res, case["utts"]):
        tk, ts, cf, frames = [], [], [], []
        for t in sorted(r):
            if r[t].final is not None:
                a = r[t].final.alternatives[0]
                tk += a.y_seq; ts += a.timesteps; cf += a.confidence; frames.append(t)
                assert not r[t].final.is_provisional
        assert tk == u["tokens"] and ts == u["timesteps"] and frames == u["final_frames"]
        assert max(r) == u["last_key"]
        assert np.allclose(cf, u["confidence"], atol=conf_atol)
        got_parts = {str(t): dict(start=fr.partials.start_frame_idx, dur=fr.partials.duration_frames,
                                  alts=[[h.y_seq, h.timesteps] for h in fr.partials.alternatives])
                     for t, fr in r.items() if fr.partials is not None}
        assert got_parts == u["partials"]


@pytest.mark.parametrize("tag", sorted(BEAM["results"]))
def test_beam_matches_reference_finals(tag, tmp_path):
    g, m = _oracle_model()
    V = int(g["n_classes"])
    case = BEAM["results"][tag]
    dec = build_decoder_from_case(m, V, case, tmp_path)
    check_against_reference(dec.decode(torch.tensor(g["x"]), torch.tensor(g["x_lens"])), case)


@pytest.mark.parametrize("tag", sorted(BEAM["results"]))
def test_native_search_matches_reference_finals(tag, tmp_path):
    """The C++ search object (include/caiman_beam.h) driven through its C-ABI, network evaluated by the oracle."""
    g, m = _oracle_model()
    case = BEAM["results"][tag]
    dec = build_decoder_from_case(m, int(g["n_classes"]), case, tmp_path, native=True, oracle_step=True)
    check_against_reference(dec.decode(torch.tensor(g["x"]), torch.tensor(g["x_lens"])), case)


@pytest.mark.parametrize("thresholds, tokens", [((0.4, 1.5), [2, 3, 4]), ((-1, -1), [5, 2, 3])])
def test_native_search_mock_model_known_answers(thresholds, tokens):
    from caiman_asr_amd.rnnt.beam_native import RNNTBeamDecoderNative

    m = MockModel(6)
    dec = RNNTBeamDecoderNative(m, blank_idx=0, eos_strategy=None, sentpiece_model=PIECES, beam_width=4,
                                temperature=1.5, max_symbols_per_step=8, beam_prune_score_thresh=thresholds[0],
                                beam_prune_topk_thresh=thresholds[1], return_partials=True, device_step=lambda *a: None)
    dec.step = OracleBeamStep(dec, m)
    res = dec.decode(torch.randn(4, 1, 1), torch.tensor([4]))
    tk, ts, pr = flatten_responses(res)
    assert tk == [tokens] and ts == [[1, 2, 3]] and pr == [[1.0, 1.0, 1.0]]
    assert all(r.partials is not None for t, r in res[0].items() if t < 4) and res[0][4].partials is None


def test_native_search_abi_errors():
    from caiman_asr_amd.rnnt.beam_native import NativeBeamSearch

    with pytest.raises(RuntimeError, match="prune threshold"):
        NativeBeamSearch(2, PIECES, blank_idx=28, beam_prune_topk_thresh=0.0)
    with pytest.raises(RuntimeError, match="frame_width"):
        NativeBeamSearch(2, PIECES, blank_idx=28, eos_vad_threshold=1.0)
    with pytest.raises(RuntimeError, match="duplicated"):
        NativeBeamSearch(2, PIECES, blank_idx=28, keywords=["ab", "ab"], keyword_weights=[1.0, 2.0])
    s = NativeBeamSearch(2, PIECES, blank_idx=28)
    s.push_frame(np.array([0, 1]))
    s.push_frame(np.array([0]))                        # queued behind the frame stream 0 is busy with
    assert (s.backlog(0), s.backlog(1), s.backlog()) == (2, 1, 2)
    with pytest.raises(RuntimeError, match="frames to expand"):
        s.close_stream(0)
    stream, frame, y, s_in, s_out = s.requests()
    assert stream.tolist() == [0, 1] and frame.tolist() == [0, 0] and y.tolist() == [-1, -1] and s_in.tolist() == [-1, -1]
    assert s_out.tolist() == [0, 1] and s.state_slots() == 2
    with pytest.raises(RuntimeError, match="not been fed"):
        s.requests()
    with pytest.raises(RuntimeError, match="answers for"):
        s.feed(np.zeros((1, 4), np.float32), np.zeros((1, 4), np.int32), np.zeros(1, np.float32))
    sc = np.log(np.array([[0.6, 0.2, 0.1, 0.05]] * 2, np.float32))
    tk = np.array([[28, 3, 4, 5], [3, 28, 4, 5]], np.int32)
    s.feed(sc, tk, np.log(np.array([0.6, 0.2], np.float32)))
    with pytest.raises(RuntimeError, match="<unk>"):
        stream, *_ = s.requests()
        s.feed(np.log(np.full((len(stream), 4), 0.25, np.float32)), np.zeros((len(stream), 4), np.int32),
               np.full(len(stream), -3.0, np.float32))


def _drive_synthetic(search, n_streams, n_frames, k=4, blank=28, queue_all=False):
    """Answers are a pure function of the request, so two searches fed by it must agree whatever their threading
    -- and whether frames are pushed one at a time or queued ahead of the search."""
    out = [dict() for _ in range(n_streams)]
    asked = np.zeros(n_streams, np.int64)   # slot numbers depend on the threading, request counts do not
    if queue_all:
        for t in range(n_frames):
            search.push_frame(np.arange(n_streams))
        assert search.backlog() == n_frames
    for t in range(1 if queue_all else n_frames):
        if not queue_all:
            search.push_frame(np.arange(n_streams))
        while True:
            stream, frame, y, s_in, s_out = search.requests()
            if len(stream) == 0:
                break
            sc = np.empty((len(stream), k), np.float32)
            tk = np.empty((len(stream), k), np.int32)
            bl = np.empty(len(stream), np.float32)
            for i in range(len(stream)):
                asked[stream[i]] += 1
                rng = np.random.default_rng([int(stream[i]), int(frame[i]), int(y[i]) + 1, int(asked[stream[i]])])
                p = rng.dirichlet(np.full(6, 0.3)) * 0.98 + 0.0033
                ids = np.concatenate([[blank], 1 + rng.choice(27, 5, replace=False)])
                order = np.argsort(-p)[:k]
                sc[i], tk[i], bl[i] = np.log(p[order]), ids[order], np.log(p[0])
            search.feed(sc, tk, bl)
        for b, r in enumerate(search.take_responses()):
            out[b].update(r)
    for b in range(n_streams):
        search.close_stream(b)
    for b, r in enumerate(search.take_responses()):
        out[b].update(r)
    return out


def test_native_search_threads_agree(monkeypatch):
    from caiman_asr_amd.rnnt.beam_native import NativeBeamSearch

    n, T = 160, 6
    results = []
    for threads in ("1", "5"):
        monkeypatch.setenv("CAIMAN_BEAM_THREADS", threads)
        s = NativeBeamSearch(n, PIECES, blank_idx=28, return_partials=True, final_emission_thresh=0.12, frame_width=0.06)
        results.append(_drive_synthetic(s, n, T))
        assert s.state_slots() < n * 40
    assert results[0] == results[1]
    s = NativeBeamSearch(n, PIECES, blank_idx=28, return_partials=True, final_emission_thresh=0.12, frame_width=0.06)
    assert _drive_synthetic(s, n, T, queue_all=True) == results[0] and s.backlog() == 0
    assert sum(len(r.final.alternatives[0].y_seq) for per in results[0] for r in per.values() if r.final) > n


def test_native_search_expansion_cap():
    """max_expansions_per_frame (a serving safeguard the reference lacks) bounds the rounds a frame can take and is
    inert when it is not reached."""
    from caiman_asr_amd.rnnt.beam_native import NativeBeamSearch

    n, T = 24, 5
    free = NativeBeamSearch(n, PIECES, blank_idx=28)
    ref = _drive_synthetic(free, n, T)
    assert free.capped_frames() == 0
    loose = NativeBeamSearch(n, PIECES, blank_idx=28, max_expansions_per_frame=10_000)
    assert _drive_synthetic(loose, n, T) == ref and loose.capped_frames() == 0
    tight = NativeBeamSearch(n, PIECES, blank_idx=28, max_expansions_per_frame=2)
    rounds = 0
    for t in range(T):
        tight.push_frame(np.arange(n))
        while True:
            stream, frame, y, s_in, s_out = tight.requests()
            if len(stream) == 0:
                break
            rounds += 1
            sc = np.log(np.tile(np.array([[0.3, 0.28, 0.22, 0.2]], np.float32), (len(stream), 1)))
            tk = np.tile(np.array([[3, 4, 5, 28]], np.int32), (len(stream), 1))
            tight.feed(sc, tk, np.log(np.full(len(stream), 0.2, np.float32)))
    assert rounds == 2 * T and tight.capped_frames() == n * T and tight.backlog() == 0
    for b in range(n):
        tight.close_stream(b)
    assert all(len(r) > 0 for r in tight.take_responses())


def test_beam_limits_and_errors():
    g, m = _oracle_model()
    V = int(g["n_classes"])
    x, xl = torch.tensor(g["x"]), torch.tensor(g["x_lens"])
    # width 1 + wide-open pruning explores one path per frame; symbol cap per utterance stops the search early
    dec = RNNTBeamDecoder(m, V - 1, None, PIECES, beam_width=2, max_symbol_per_sample=3)
    tk, _, _ = flatten_responses(dec.decode(x, xl))
    assert all(len(t) <= 4 for t in tk)
    with pytest.raises(AssertionError):
        RNNTBeamDecoder(m, V - 1, None, PIECES, beam_prune_topk_thresh=0)
    with pytest.raises(AssertionError):
        RNNTBeamDecoder(m, V - 1, None, PIECES, eos_vad_threshold=1.0)           # needs frame_width
    with pytest.raises(NotImplementedError):
        RNNTBeamDecoder(m, V - 1, None, PIECES, fuzzy_topk_logits=True)
    # silence rule (beam.py:266-283): frames since the latest token of ANY live hypothesis, SOS (frame -1) excluded
    dec = RNNTBeamDecoder(m, V - 1, None, PIECES, eos_vad_threshold=0.12, frame_width=0.06)
    quiet, talking = _hyp([4], [3]), _hyp([4, 5], [3, 6])
    assert dec._silence_terminate({1: quiet}, 5) and not dec._silence_terminate({1: quiet}, 4)
    assert not dec._silence_terminate({1: quiet, 2: talking}, 7) and dec._silence_terminate({1: quiet, 2: talking}, 8)
    assert not dec._silence_terminate({1: init_sos_hyp(-1)}, 50)
    # unmodified weights emit <unk> (id 0), which the search refuses as the reference does (beam.py:621)
    g2 = np.load(os.path.join(GOLD, "rnnt_mfma.npz"))
    sd = {k[3:]: torch.tensor(g2[k]) for k in g2.files if k.startswith("sd.")}
    sd["joint_net.2.bias"][0] = 30.0
    with pytest.raises(AssertionError, match="<unk>"):
        RNNTBeamDecoder(OracleRNNT(sd, json.loads(str(g2["cfg"]))), V - 1, None, PIECES).decode(x, xl)
