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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("eos", [None, ("ignore", 5), ("blank", 5), ("predict", 5, 0.7, 0.0), ("predict", 5, 1.0, 0.02)])
def test_beam_topk_kernel(dtype, eos):
    """caiman_beam_topk == log_softmax(logits / T) -> EOS correction -> topk / blank column (decoder.py:139-172,
    beam.py:535-546): token ids exact, scores to 1e-5 (fp32 math in both)."""
    from caiman_asr_amd import _lib
    from caiman_asr_amd.rnnt.decoder import RNNTCommonDecoder
    from caiman_asr_amd.rnnt.eos_strategy import EOSBlank, EOSIgnore, EOSPredict

    torch.manual_seed(3)
    n, V, k, T, blank = 37, 8704, 4, 1.4, 8703
    logits = (torch.randn(n, V, device=DEV) * 3).to(dtype)
    logits[0, 100:104] = logits[0].max() + 1       # exact ties: the lower id wins
    logits[1, 5] += 12                              # EOS is the best entry of this row
    logits[2, blank] += 12
    strat = None if eos is None else {"ignore": EOSIgnore, "blank": EOSBlank, "predict": EOSPredict}[eos[0]](*eos[1:])
    ref_dec = RNNTCommonDecoder.__new__(RNNTCommonDecoder)
    ref_dec.eos_strategy, ref_dec.blank_idx = strat, blank
    lp = ref_dec._eos_prob_correction(torch.log_softmax(logits.float() / T, dim=-1))
    ref_s, ref_i = lp.topk(k, dim=1)
    sc = torch.empty(n, k, device=DEV)
    tk = torch.empty(n, k, dtype=torch.int32, device=DEV)
    bl = torch.empty(n, device=DEV)
    mode = (0, 0, 1.0, 0.0) if eos is None else ({"ignore": 1, "blank": 2, "predict": 3}[eos[0]], eos[1],
                                                 *(eos[2:] if len(eos) > 2 else (1.0, 0.0)))
    _lib.check(_lib.lib().caiman_beam_topk(_lib.ptr(logits), n, V, logits.stride(0), _lib.dtype_tag(dtype), T, blank,
                                           mode[0], mode[1], mode[2], mode[3], k, _lib.ptr(sc), _lib.ptr(tk),
                                           _lib.ptr(bl), _lib.stream()))
    assert torch.allclose(sc, ref_s, atol=1e-5)
    assert torch.allclose(bl, lp[:, blank], atol=1e-5)
    assert tk[0].tolist() == [100, 101, 102, 103]
    distinct = (ref_s[:, :-1] - ref_s[:, 1:]).min(dim=1).values > 1e-4          # rows without near-ties
    assert distinct.sum() > n // 2 and torch.equal(tk[distinct].long(), ref_i[distinct])
    # strided rows
    wide = torch.zeros(n, V + 64, device=DEV, dtype=dtype)
    wide[:, :V] = logits
    _lib.check(_lib.lib().caiman_beam_topk(_lib.ptr(wide), n, V, wide.stride(0), _lib.dtype_tag(dtype), T, blank,
                                           mode[0], mode[1], mode[2], mode[3], k, _lib.ptr(sc), _lib.ptr(tk),
                                           _lib.ptr(bl), _lib.stream()))
    assert torch.allclose(sc, ref_s, atol=1e-5)


@pytest.mark.parametrize("tag", ["default", "capped", "wide", "partials", "forced_finals", "vad", "sample_cap", "keywords",
                                 "eos_terminal", "eos_blank", "eos_ignore", "eos_predict_beta"])
def test_native_beam_matches_reference_decoder(tag, tmp_path):
    """The serving path (C++ search + HIP round with states in the slot pool) equals the REFERENCE decoder."""
    from tests.test_beam_host import BEAM, build_decoder_from_case, check_against_reference

    g, m = build("mfma")
    with torch.no_grad():
        m.joint_fc.bias[0] = BEAM["unk_bias"]
    case = BEAM["results"][tag]
    dec = build_decoder_from_case(m, int(g["n_classes"]), case, tmp_path, native=True)
    res = dec.decode(torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV))
    check_against_reference(res, case, conf_atol=1e-4)


def test_streaming_beam_equals_offline():
    """Chunked audio with carried encoder / beam state gives the finals of the offline decode, per stream."""
    from caiman_asr_amd.rnnt.beam_native import RNNTBeamDecoderNative, StreamingBeamDecoder
    from caiman_asr_amd.rnnt.decoder import flatten_responses
    from tests.test_beam_host import BEAM, PIECES

    g, m = build("mfma")
    with torch.no_grad():
        m.joint_fc.bias[0] = BEAM["unk_bias"]
    V = int(g["n_classes"])
    torch.manual_seed(5)
    T, B = 24, 6
    x = torch.randn(T, B, g["x"].shape[2], device=DEV)
    off = RNNTBeamDecoderNative(m, V - 1, None, PIECES).decode(x, torch.full((B,), T, device=DEV))
    dec = StreamingBeamDecoder(m, V - 1, B, PIECES)
    merged = [dict() for _ in range(B)]
    for t0 in range(0, T, 4):
        for b, r in enumerate(dec.step(x[t0:t0 + 4])):
            merged[b].update(r)
    for b, r in enumerate(dec.close()):
        merged[b].update(r)
    (tk_s, ts_s, cf_s), (tk_o, ts_o, cf_o) = flatten_responses(merged), flatten_responses(off)
    assert tk_s == tk_o and ts_s == ts_o and sum(map(len, tk_o)) > 10
    for a, b in zip(cf_s, cf_o):                 # batch shapes differ between the two runs: GEMM rounding only
        assert np.allclose(a, b, atol=1e-4)
    assert [sorted(r) for r in merged] == [sorted(r) for r in off]


def test_captured_round_equals_eager_round(monkeypatch):
    """The expansion round as ONE captured graph launch per bucket of row counts (rows padded up to the bucket: the padding
    starts from the zero state and writes into a spare pool row) against the nine eager launches: same tokens, timestamps and
    response keys, scores to GEMM rounding (the products run at the bucket's row count); row counts on both sides of a bucket
    edge, a pool that grows between ticks (the captured rounds are dropped and captured again)."""
    from caiman_asr_amd.rnnt import beam_native
    from caiman_asr_amd.rnnt.decoder import flatten_responses
    from tests.test_beam_host import BEAM, PIECES

    g, m = build("mfma")
    with torch.no_grad():
        m.joint_fc.bias[0] = BEAM["unk_bias"]
    V = int(g["n_classes"])
    torch.manual_seed(11)
    T, B = 24, 70                  # first rounds serve 70 requests (bucket 128), later ones fewer than 64
    x = torch.randn(T, B, g["x"].shape[2], device=DEV)
    results, captured = [], 0
    for graph in (False, True):
        monkeypatch.setattr(beam_native, "ROUND_GRAPH", graph)
        dec = beam_native.StreamingBeamDecoder(m, V - 1, B, PIECES)
        merged = [dict() for _ in range(B)]
        for t0 in range(0, T, 2):
            for b, r in enumerate(dec.step(x[t0:t0 + 2])):
                merged[b].update(r)
        for b, r in enumerate(dec.close()):
            merged[b].update(r)
        captured += len(dec.dec.step.graphs)
        results.append((flatten_responses(merged), [sorted(r) for r in merged]))
    assert captured >= 2, "fewer than two buckets were captured: the test does not cross a bucket edge"
    (tk0, ts0, cf0), keys0 = results[0]
    (tk1, ts1, cf1), keys1 = results[1]
    assert tk0 == tk1 and ts0 == ts1 and keys0 == keys1 and sum(map(len, tk0)) > 50
    for a, b in zip(cf0, cf1):
        assert np.allclose(a, b, atol=1e-4)


def test_streaming_beam_stragglers_catch_up():
    """Ending ticks early (slow streams keep their frame and queue the new ones) changes when responses appear,
    not what they are."""
    from caiman_asr_amd.rnnt.beam_native import StreamingBeamDecoder
    from caiman_asr_amd.rnnt.decoder import flatten_responses
    from tests.test_beam_host import BEAM, PIECES

    g, m = build("mfma")
    with torch.no_grad():
        m.joint_fc.bias[0] = BEAM["unk_bias"]
    V = int(g["n_classes"])
    torch.manual_seed(7)
    T, B = 32, 12
    x = torch.randn(T, B, g["x"].shape[2], device=DEV)
    results, lagged = [], 0
    for cutoff in (0, 6):
        dec = StreamingBeamDecoder(m, V - 1, B, PIECES, straggler_cutoff=cutoff, ring=8)
        merged = [dict() for _ in range(B)]
        for t0 in range(0, T, 2):
            for b, r in enumerate(dec.step(x[t0:t0 + 2])):
                merged[b].update(r)
            lagged += dec.backlog() if cutoff else 0
        for b, r in enumerate(dec.close()):
            merged[b].update(r)
        assert dec.backlog() == 0
        results.append((flatten_responses(merged), [sorted(r) for r in merged]))
    (tk0, ts0, cf0), keys0 = results[0]
    (tk1, ts1, cf1), keys1 = results[1]
    assert lagged > 0, "the cutoff never left a straggler: the test does not exercise the queue"
    assert tk0 == tk1 and ts0 == ts1 and keys0 == keys1
    for a, b in zip(cf0, cf1):
        assert np.allclose(a, b, atol=1e-4)


def test_large_batch_streaming_encoder_equals_module_path():
    """The GEMM + cell-kernel encoder step used for thousands of streams gives the frames of the step-kernel path."""
    from caiman_asr_amd.rnnt.decoder import StreamingEncoder

    g, m = build("mfma")
    torch.manual_seed(11)
    T, B = 12, 7
    x = torch.randn(T, B, g["x"].shape[2], device=DEV)
    a, b = StreamingEncoder(m, B, large_batch=False), StreamingEncoder(m, B, large_batch=True)
    fa, fb = [], []
    for t0, n in ((0, 2), (2, 3), (5, 1), (6, 4), (10, 2)):       # odd chunk sizes exercise the StackTime carry
        ya, yb = a.advance(x[t0:t0 + n]), b.advance(x[t0:t0 + n])
        assert (ya is None) == (yb is None)
        if ya is not None:
            fa.append(ya)
            fb.append(yb)
    fa, fb = torch.cat(fa, 1), torch.cat(fb, 1)
    assert fa.shape == fb.shape == (B, T // 2, fa.shape[2])
    assert torch.allclose(fa, fb, atol=2e-5, rtol=1e-5)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a, b = StreamingEncoder(m, B, large_batch=False), StreamingEncoder(m, B, large_batch=True)
        ya, yb = a.advance(x[:4]), b.advance(x[:4])
    assert torch.allclose(ya.float(), yb.float(), atol=3e-2, rtol=3e-2)


def test_greedy_large_batch_predictor_equals_module_path(monkeypatch):
    """The GEMM + cell-kernel prediction step (used from 256 rows on) decodes the same tokens as the module path."""
    from caiman_asr_amd.rnnt.decoder import RNNTBatchedGreedyDecoder, flatten_responses

    g, m = build("mfma")
    V = int(g["n_classes"])
    torch.manual_seed(2)
    T, B = 20, 9
    x = torch.randn(T, B, g["x"].shape[2], device=DEV)
    lens = torch.tensor([20, 18, 20, 7, 20, 13, 20, 20, 2], device=DEV)
    out = []
    for thresh in (10 ** 9, 1):
        monkeypatch.setattr(RNNTBatchedGreedyDecoder, "LARGE_BATCH", thresh)
        dec = RNNTBatchedGreedyDecoder(m, V - 1, None, int(1e7), None, max_symbols_per_step=3)
        out.append(flatten_responses(dec.decode(x, lens)))
    (tk0, ts0, cf0), (tk1, ts1, cf1) = out
    assert tk0 == tk1 and ts0 == ts1 and sum(map(len, tk0)) > 10
    for a, b in zip(cf0, cf1):
        assert np.allclose(a, b, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,I,H,last", [(300, 240, 128, False), (2000, 512, 512, True), (129, 1024, 1024, False), (7, 128, 256, True)])
def test_fused_lstm_step_gemm_matches_fp32_reference_and_the_two_launch_form(dtype, n, I, H, last):
    """caiman_lstm_step_gemm (csrc/proj_gemm.hip, CELL epilogue): one timestep of one LSTM layer for n rows with state
    pools addressed through slot indices, against (a) a plain fp32 torch evaluation of the same step
    (training/lib/csrc/lstm.cu:99-123, gate order i,f,g,o) and (b) library GEMM + caiman_beam_lstm_cell, the form it
    replaces.  Rows past a multiple of 128, a padded input width (I = 240 -> 256) and the last layer (no next pool)."""
    from caiman_asr_amd import _lib
    from caiman_asr_amd.rnnt import streaming_lstm as sl

    torch.manual_seed(n + I + H)
    lstm = torch.nn.LSTM(I, H, 1).to(DEV)
    with torch.no_grad():
        for p in lstm.parameters():
            p.mul_(2.0)
    W, b, Ip = sl.fused_layer_weights(lstm, 0, dtype)
    assert Ip % 128 == 0 and W.shape == (4 * H, Ip + H)
    slots = 2 * n + 3
    h_pool = (torch.randn(slots + 1, H, device=DEV) * 0.5).to(dtype)
    c_pool = torch.randn(slots + 1, H, device=DEV) * 0.5
    h_next = (torch.randn(slots + 1, H, device=DEV) * 0.5).to(dtype)
    perm = torch.randperm(slots, device=DEV)
    s_in = perm[:n].to(torch.int32).contiguous()
    s_in[::5] = -1                                            # pool row 0: the zero start state
    h_pool[0], c_pool[0] = 0, 0
    s_out = perm[n:2 * n].to(torch.int32).contiguous()        # fresh slots, as the search hands them out
    x = torch.randn(n, I, device=DEV).to(dtype)
    X = torch.zeros(n, Ip + H, device=DEV, dtype=dtype)
    X[:, :I] = x
    X[:, Ip:] = h_pool[(s_in + 1).long()]
    ldx = H if last else 2 * H
    tag, st = _lib.dtype_tag(dtype), _lib.stream()

    def run(fused):
        hp, cp = h_pool.clone(), c_pool.clone()
        nxt = torch.zeros(n, ldx, device=DEV, dtype=dtype)
        if fused:
            sl.lstm_step_gemm(X, W, b, n, H, cp, hp, None if last else h_next, _lib.ptr(s_in), _lib.ptr(s_out), nxt, tag, st)
        else:
            Wcat = torch.cat([lstm.weight_ih_l0, lstm.weight_hh_l0], 1).detach().to(dtype)
            bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().to(dtype)
            gates = torch.addmm(bias, torch.cat([x, X[:, Ip:]], 1), Wcat.t())
            _lib.check(_lib.lib().caiman_beam_lstm_cell(_lib.ptr(gates), H, _lib.ptr(cp), _lib.ptr(hp),
                                                        None if last else _lib.ptr(h_next), _lib.ptr(s_in), _lib.ptr(s_out), n,
                                                        _lib.ptr(nxt), ldx, tag, st))
        torch.cuda.synchronize()
        return hp, cp, nxt

    hp, cp, nxt = run(True)
    # (a) fp32 reference from the SAME 16-bit operands
    Wf = torch.cat([lstm.weight_ih_l0, lstm.weight_hh_l0], 1).detach().to(dtype).float()
    bf = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().to(dtype).float()
    z = torch.cat([x, X[:, Ip:]], 1).float() @ Wf.t() + bf
    zi, zf, zg, zo = z.split(H, 1)
    c_ref = torch.sigmoid(zi) * torch.tanh(zg) + torch.sigmoid(zf) * c_pool[(s_in + 1).long()]
    h_ref = torch.sigmoid(zo) * torch.tanh(c_ref)
    out_rows = (s_out + 1).long()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert torch.allclose(cp[out_rows], c_ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(hp[out_rows].float(), h_ref, atol=eps, rtol=eps)
    assert torch.equal(nxt[:, :H], hp[out_rows])
    if not last:
        assert torch.equal(nxt[:, H:], h_next[(s_in + 1).long()])
    untouched = torch.ones(slots + 1, dtype=torch.bool, device=DEV)
    untouched[out_rows] = False
    assert torch.equal(hp[untouched], h_pool[untouched]) and torch.equal(cp[untouched], c_pool[untouched])
    # (b) the two-launch form rounds the pre-activations to 16 bits before the cell: agreement to that resolution
    hp2, cp2, nxt2 = run(False)
    assert torch.allclose(cp, cp2, atol=16 * eps, rtol=16 * eps) and torch.allclose(hp.float(), hp2.float(), atol=16 * eps, rtol=0)


def test_large_batch_streaming_with_the_fused_step_equals_the_two_launch_form(monkeypatch):
    """Streaming encoder (LargeBatchLSTM) at a hidden size the fused step takes (the golden mini model's 64 is not):
    frames with CAIMAN_DECODE_FUSED_LSTM on and off agree to the storage resolution, for odd chunk sizes."""
    from caiman_asr_amd.rnnt import streaming_lstm as sl
    from caiman_asr_amd.rnnt.rnn import rnn

    torch.manual_seed(4)
    stack = rnn(input_size=240, hidden_size=256, num_layers=3, batch_norm=False, rw_dropout=0.0, dropout=0.0,
                tensor_name="pre_rnn", forget_gate_bias=1.0, custom_lstm=True, quantize=False, hidden_hidden_bias_scale=0.0,
                weights_init_scale=1.0, gpu_unavailable=False).to(DEV)
    T, B = 9, 300
    x = torch.randn(T, B, 240, device=DEV)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(sl, "FUSED_STEP", fused)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lb = sl.LargeBatchLSTM(stack, B)
            ys = [lb.forward(x[t0:t0 + n]) for t0, n in ((0, 2), (2, 3), (5, 4))]
        assert (lb.gates is None) == fused
        outs.append((torch.cat(ys, 0).float(), lb.state()[1].clone()))
    (y1, c1), (y0, c0) = outs
    assert y1.shape == (T, B, 256)
    assert torch.allclose(y1, y0, atol=3e-2, rtol=0) and torch.allclose(c1, c0, atol=5e-2, rtol=0)
