"""Beam search for thousands of concurrent streams: C++ search object + one HIP kernel per round
(include/caiman_beam.h; SURVEY §8 a22, f3, BASELINE configs[4]).

`rnnt/beam.py` mirrors the reference's Python search line of behaviour by line of behaviour and is the
readable statement of the algorithm; this module is the serving path.  The search of every stream lives in
one native object (`NativeBeamSearch`), prediction-network states stay in an HBM slot pool and are referred to
by index, and each expansion round is: gather states -> prediction step -> joint -> `caiman_beam_topk` -> one
fixed-shape device->host copy -> `caiman_beam_feed`.

* `RNNTBeamDecoderNative` -- same constructor and `decode()` result as `RNNTBeamDecoder` (offline batches).
* `StreamingBeamDecoder`  -- `step(feats)` per 60 ms tick for N live streams with carried encoder state.
"""
import ctypes
import json
import time
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from caiman_asr_amd import _lib
from caiman_asr_amd.rnnt.decoder import RNNTCommonDecoder, StreamingEncoder
from caiman_asr_amd.rnnt.eos_strategy import EOSBlank, EOSIgnore, EOSPredict
from caiman_asr_amd.rnnt.response import DecodingResponse, FrameResponses, HypothesisResponse

INF = float("inf")
_I32P = ctypes.POINTER(ctypes.c_int32)


def _pieces_of(sentpiece_model: Union[str, Sequence[str]], n: Optional[int] = None) -> List[str]:
    if isinstance(sentpiece_model, str):
        from sentencepiece import SentencePieceProcessor

        sp = SentencePieceProcessor(model_file=sentpiece_model)
        return [sp.id_to_piece(i) for i in range(sp.get_piece_size())]
    return list(sentpiece_model)


def _load_keyword_table(path: Optional[str]):
    if path is None:
        return [], []
    from caiman_asr_amd.keywords.process import SPACE_MARK, load_keywords

    load_keywords(path)  # schema errors surface as in the Python decoder
    kw = json.load(open(path))["keywords"]
    return [k.replace(" ", SPACE_MARK) for k in kw], [float(v) for v in kw.values()]


class NativeBeamSearch:
    """ctypes face of the `caiman_beam_*` object: the beams of `n_streams` streams."""

    def __init__(self, n_streams: int, pieces: Sequence[str], blank_idx: int, beam_width: int = 4,
                 max_symbols_per_step: Optional[int] = 8, max_symbol_per_sample: Optional[int] = None,
                 beam_prune_score_thresh: float = 0.4, beam_prune_topk_thresh: float = 1.5,
                 eos_vad_threshold: float = INF, final_emission_thresh: float = INF,
                 frame_width: Optional[float] = None, eos_terminal_idx: Optional[int] = None,
                 return_partials: bool = False, keywords: Sequence[str] = (), keyword_weights: Sequence[float] = ()):
        self.L = _lib.lib()
        self.n_streams, self.k = n_streams, beam_width
        cfg = _lib.BeamConfig(
            blank_idx=blank_idx, beam_width=beam_width, max_symbols_per_step=max_symbols_per_step or 0,
            max_symbol_per_sample=-1 if max_symbol_per_sample is None else max_symbol_per_sample,
            beam_prune_score_thresh=beam_prune_score_thresh, beam_prune_topk_thresh=beam_prune_topk_thresh,
            eos_vad_threshold=-1.0 if eos_vad_threshold == INF else eos_vad_threshold,
            final_emission_thresh=-1.0 if final_emission_thresh == INF else final_emission_thresh,
            frame_width=0.0 if frame_width is None else frame_width,
            eos_terminal_idx=-1 if eos_terminal_idx is None else eos_terminal_idx, return_partials=int(return_partials))
        pc = (ctypes.c_char_p * len(pieces))(*[p.encode("utf-8") for p in pieces])
        kw = (ctypes.c_char_p * max(len(keywords), 1))(*[k.encode("utf-8") for k in keywords])
        kww = (ctypes.c_double * max(len(keywords), 1))(*keyword_weights)
        self.h = self.L.caiman_beam_create(ctypes.byref(cfg), n_streams, pc, len(pieces), kw, kww, len(keywords))
        if not self.h:
            raise RuntimeError(self.L.caiman_last_error().decode())
        self.pieces = list(pieces)
        self._req = np.zeros((5, n_streams), dtype=np.int32)  # stream, frame, y_last, state_in, state_out

    def __del__(self):
        if getattr(self, "h", None):
            self.L.caiman_beam_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.caiman_last_error().decode())

    def reset_stream(self, i: int):
        self._check(self.L.caiman_beam_reset_stream(self.h, i))

    def push_frame(self, streams: np.ndarray):
        streams = np.ascontiguousarray(streams, dtype=np.int32)
        self._check(self.L.caiman_beam_push_frame(self.h, streams.ctypes.data, len(streams)))

    def requests(self):
        """-> (stream, frame, y_last, state_in, state_out) int32 arrays of the pending expansions (views that
        stay valid until the next call)."""
        r = self._req
        n = self.L.caiman_beam_requests(self.h, r[0].ctypes.data, r[1].ctypes.data, r[2].ctypes.data,
                                        r[3].ctypes.data, r[4].ctypes.data, r.shape[1])
        if n < 0:
            raise RuntimeError(self.L.caiman_last_error().decode())
        return tuple(r[i, :n] for i in range(5))

    def feed(self, scores: np.ndarray, tokens: np.ndarray, blank: np.ndarray):
        assert scores.dtype == np.float32 and tokens.dtype == np.int32 and blank.dtype == np.float32
        assert scores.flags.c_contiguous and tokens.flags.c_contiguous and blank.flags.c_contiguous
        n, k = scores.shape
        assert tokens.shape == (n, k) and blank.shape == (n,)
        self._check(self.L.caiman_beam_feed(self.h, n, k, scores.ctypes.data, tokens.ctypes.data, blank.ctypes.data))

    def close_stream(self, i: int):
        self._check(self.L.caiman_beam_close_stream(self.h, i))

    def done(self, i: int) -> bool:
        return bool(self.L.caiman_beam_stream_done(self.h, i))

    def backlog(self, stream: int = -1) -> int:
        """Frames pushed but not finished on `stream` (-1: the maximum over all streams)."""
        return int(self.L.caiman_beam_backlog(self.h, stream))

    def state_slots(self) -> int:
        return int(self.L.caiman_beam_state_slots(self.h))

    def take_responses(self) -> List[Dict[int, FrameResponses]]:
        """Responses since the last call -> per stream {frame key: FrameResponses}."""
        ip, fp = _I32P(), ctypes.POINTER(ctypes.c_float)()
        ni, nf = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.L.caiman_beam_responses(self.h, ctypes.byref(ip), ctypes.byref(ni), ctypes.byref(fp),
                                                 ctypes.byref(nf)))
        ints = np.ctypeslib.as_array(ip, (ni.value,)).tolist() if ni.value else []
        flts = np.ctypeslib.as_array(fp, (nf.value,)).tolist() if nf.value else []
        self.L.caiman_beam_clear_responses(self.h)
        out: List[Dict[int, FrameResponses]] = [dict() for _ in range(self.n_streams)]
        i = f = 0
        while i < len(ints):
            stream, key, kind, start, dur, n_alt = ints[i:i + 6]
            i += 6
            alts = []
            for _ in range(n_alt):
                n = ints[i]
                y, ts = ints[i + 1:i + 1 + n], ints[i + 1 + n:i + 1 + 2 * n]
                i += 1 + 2 * n
                alts.append(HypothesisResponse(y_seq=y, timesteps=ts, token_seq=[self.pieces[t] for t in y],
                                               confidence=flts[f:f + n]))
                f += n
            fr = out[stream].setdefault(key, FrameResponses(partials=None, final=None))
            if kind == 0:
                fr.final = DecodingResponse(start, dur, False, alts)
            elif kind == 1:
                fr.partials = DecodingResponse(start, dur, True, alts)
        # worker threads write records in their own buffers: restore frame order per stream
        return [dict(sorted(d.items())) if len(d) > 1 else d for d in out]


class HipBeamStep:
    """One expansion round on the device.  Prediction states of all live hypotheses sit in a slot pool
    [layers, 1 + slots, hidden] (row 0 = the all-zero start state); a round gathers by slot, runs the prediction
    LSTM for one step and the joint, and `caiman_beam_topk` leaves [n, k] scores / tokens and [n] blank
    log-probs in one buffer that is copied to pinned host memory."""

    def __init__(self, model, blank_idx: int, beam_width: int, temperature: float, eos_strategy=None):
        self.model = getattr(model, "module", model)
        self.blank_idx, self.k, self.temperature = blank_idx, beam_width, temperature
        self.eos = (0, 0, 1.0, 0.0)
        if isinstance(eos_strategy, EOSIgnore):
            self.eos = (1, eos_strategy.eos_idx, 1.0, 0.0)
        elif isinstance(eos_strategy, EOSBlank):
            self.eos = (2, eos_strategy.eos_idx, 1.0, 0.0)
        elif isinstance(eos_strategy, EOSPredict):
            self.eos = (3, eos_strategy.eos_idx, float(eos_strategy.alpha), float(eos_strategy.beta))
        self.h_pool = self.c_pool = None
        self.host = None
        self.stats = None

    def _ensure(self, n_slots: int, n: int, dev, dtype):
        m = self.model
        if self.h_pool is None or self.h_pool.shape[1] < n_slots + 1:
            cap = max(2 * n_slots, 64) + 1
            L, H = m.prediction["dec_rnn"].num_layers, m.pred_n_hid
            h = torch.zeros(L, cap, H, device=dev, dtype=dtype)
            c = torch.zeros(L, cap, H, device=dev, dtype=dtype)
            if self.h_pool is not None:
                h[:, :self.h_pool.shape[1]] = self.h_pool
                c[:, :self.c_pool.shape[1]] = self.c_pool
            self.h_pool, self.c_pool = h, c
        if self.host is None or self.host.shape[0] < n:
            cap = max(2 * n, 256)
            self.host = torch.empty(cap, 2 * self.k + 1, dtype=torch.float32).pin_memory()
            self.dev_out = torch.empty(cap, 2 * self.k + 1, dtype=torch.float32, device=dev)

    @torch.no_grad()
    def __call__(self, f: torch.Tensor, y_last: np.ndarray, state_in: np.ndarray, state_out: np.ndarray,
                 n_slots: int):
        """f [n, 1, Hj] encoder frames; y_last / state_in (-1 = start of sequence) / state_out [n] ->
        (scores [n, k] f32, tokens [n, k] i32, blank [n] f32) numpy views of the pinned buffer."""
        m, dev, n, k = self.model, f.device, f.shape[0], self.k
        w = m.joint_enc.weight
        self._ensure(n_slots, n, dev, w.dtype)
        idx = torch.from_numpy(np.stack([y_last, state_in + 1, state_out + 1]).astype(np.int64)).to(dev, non_blocking=True)
        y, s_in, s_out = idx[0], idx[1], idx[2]
        emb = m.prediction["embed"](y.clamp(min=0)) * (y >= 0).unsqueeze(1).to(w.dtype)   # SOS: zero vector
        h0, c0 = self.h_pool.index_select(1, s_in), self.c_pool.index_select(1, s_in)
        g, (h1, c1), _ = m.prediction["dec_rnn"](emb.unsqueeze(0), (h0, c0))
        self.h_pool.index_copy_(1, s_out, h1.to(self.h_pool.dtype))
        self.c_pool.index_copy_(1, s_out, c1.to(self.c_pool.dtype))
        g = m.joint_pred(g.transpose(0, 1))
        logits = m.joint(f, g)[:, 0, 0, :]
        if not logits.is_contiguous():
            logits = logits.contiguous()
        # tokens are written as int32 bit patterns into the float buffer; the three regions are separate
        # contiguous blocks of one allocation so that a single copy brings them back
        flat = self.dev_out.view(-1)
        sc = flat[: n * k]
        tk = flat[n * k: 2 * n * k].view(torch.int32)
        bl = flat[2 * n * k: 2 * n * k + n]
        L = _lib.lib()
        _lib.check(L.caiman_beam_topk(_lib.ptr(logits), n, logits.shape[1], logits.stride(0), _lib.dtype_tag(logits.dtype),
                                      self.temperature, self.blank_idx, self.eos[0], self.eos[1], self.eos[2],
                                      self.eos[3], k, _lib.ptr(sc), _lib.ptr(tk), _lib.ptr(bl), _lib.stream()))
        hflat = self.host.view(-1)
        hflat[: 2 * n * k + n].copy_(flat[: 2 * n * k + n], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        hn = hflat.numpy()
        if self.stats is not None:   # [sum of top-1 probabilities, rows]: how peaked the workload is
            self.stats[0] += float(np.exp(hn[: n * k: k]).sum())
            self.stats[1] += n
        return (hn[: n * k].reshape(n, k), hn[n * k: 2 * n * k].view(np.int32).reshape(n, k), hn[2 * n * k: 2 * n * k + n])


class RNNTBeamDecoderNative(RNNTCommonDecoder):
    """`RNNTBeamDecoder` (beam.py) with the search in native code.  Same constructor keywords; n-gram rescoring is
    only available in the Python decoder.  `device_step` replaces the HIP round (tests inject a CPU-oracle one)."""

    def __init__(self, model, blank_idx: int, eos_strategy, sentpiece_model: Union[str, Sequence[str]],
                 beam_width: int = 4, max_inputs_per_batch: int = int(1e7), max_symbols_per_step: Optional[int] = 8,
                 max_symbol_per_sample: Optional[int] = None, temperature: float = 1.4,
                 beam_prune_score_thresh: Union[int, float] = 0.4, beam_prune_topk_thresh: Union[int, float] = 1.5,
                 ngram_info=None, fuzzy_topk_logits: bool = False, return_partials: bool = False,
                 user_tokens: Optional[List[int]] = None, eos_is_terminal: bool = False,
                 eos_vad_threshold: float = INF, final_emission_thresh: float = INF,
                 frame_width: Optional[float] = None, keyword_boost_path: Optional[str] = None, device_step=None):
        super().__init__(model=model, blank_idx=blank_idx, eos_strategy=eos_strategy,
                         max_inputs_per_batch=max_inputs_per_batch, max_symbol_per_sample=max_symbol_per_sample,
                         max_symbols_per_step=max_symbols_per_step, temperature=temperature)
        if ngram_info is not None:
            raise NotImplementedError("n-gram rescoring is available in caiman_asr_amd.rnnt.beam.RNNTBeamDecoder")
        if fuzzy_topk_logits:
            raise NotImplementedError("fuzzy_topk_logits emulates the FPGA's packetised argmax; it is outside this path")
        assert beam_width > 0
        assert beam_prune_topk_thresh < 0 or beam_prune_topk_thresh > 1e-9, "use the greedy decoder instead"
        assert beam_prune_score_thresh < 0 or beam_prune_score_thresh > 1e-9, "use the greedy decoder instead"
        if final_emission_thresh < 0:
            final_emission_thresh = INF
        if eos_vad_threshold != INF or final_emission_thresh != INF:
            assert frame_width is not None and frame_width > 0.0
        kw, kww = _load_keyword_table(keyword_boost_path)
        self.beam_width = beam_width
        self.search_args = dict(
            pieces=_pieces_of(sentpiece_model), blank_idx=blank_idx, beam_width=beam_width,
            max_symbols_per_step=max_symbols_per_step, max_symbol_per_sample=max_symbol_per_sample,
            beam_prune_score_thresh=beam_prune_score_thresh, beam_prune_topk_thresh=beam_prune_topk_thresh,
            eos_vad_threshold=eos_vad_threshold, final_emission_thresh=final_emission_thresh, frame_width=frame_width,
            eos_terminal_idx=(eos_strategy.eos_idx if eos_is_terminal and isinstance(eos_strategy, EOSPredict) else None),
            return_partials=return_partials, keywords=kw, keyword_weights=kww)
        self.step = device_step or HipBeamStep(self.model, blank_idx, beam_width, temperature, eos_strategy)
        self.profile: Optional[Dict[str, float]] = None   # set to a defaultdict(float) to collect host timings

    def _rounds(self, search: NativeBeamSearch, frame_of, stop_below: int = 0):
        """Expansion rounds until no stream has a request left -- or, with `stop_below`, until a round served no
        more than that many streams (the stragglers carry on in the next call).  `frame_of(streams, frames)` ->
        f [n, 1, Hj]."""
        n_rounds = 0
        prof = self.profile
        while True:
            t0 = time.perf_counter()
            stream, frame, y_last, s_in, s_out = search.requests()
            if len(stream) == 0:
                return n_rounds
            n_rounds += 1
            t1 = time.perf_counter()
            scores, tokens, blank = self.step(frame_of(stream, frame), y_last, s_in, s_out, search.state_slots())
            t2 = time.perf_counter()
            search.feed(scores, tokens, blank)
            if prof is not None:
                t3 = time.perf_counter()
                prof["requests"] += t1 - t0
                prof["device_round"] += t2 - t1
                prof["feed"] += t3 - t2
                prof["rounds"] += 1
                prof["expansions"] += len(stream)
            if len(stream) <= stop_below:
                return n_rounds

    @torch.no_grad()
    def _inner_decode(self, encs: torch.Tensor, encs_len: torch.Tensor) -> List[Dict[int, FrameResponses]]:
        was_training = self.model.training
        self.model.eval()
        lens = np.asarray(encs_len.tolist(), dtype=np.int64)
        B = len(lens)
        search = NativeBeamSearch(B, **self.search_args)

        def frame_of(stream, frame):
            rows = torch.from_numpy(stream.astype(np.int64)).to(encs.device)
            cols = torch.from_numpy(frame.astype(np.int64)).to(encs.device)
            return encs[rows, cols].unsqueeze(1)

        for t in range(int(lens.max()) if B else 0):
            search.push_frame(np.nonzero(lens > t)[0])
            self._rounds(search, frame_of)
        for b in range(B):
            search.close_stream(b)
        self.model.train(was_training)
        return search.take_responses()


class StreamingBeamDecoder:
    """N live streams: `step(feats)` takes the next chunk of spliced features for all streams ([frames, N, in_feats];
    2 frames = 60 ms at the base config), advances the encoder with its carried state and runs the beam search over
    the encoder frames the chunk completes.  Returns the responses produced during the call.

    A frame on which a stream's beam is slow to settle can need a hundred expansions while the typical one needs
    a handful; waiting for it would make every stream late.  With `straggler_cutoff` > 0 a tick ends once a round
    served no more than that many streams: those streams keep their open frame, queue the frames that arrive
    meanwhile (`ring` encoder frames are kept) and catch up inside later ticks.  `backlog()` reports the lag."""

    def __init__(self, model, blank_idx: int, n_streams: int, sentpiece_model: Union[str, Sequence[str]],
                 straggler_cutoff: int = 0, ring: int = 32, **kwargs):
        self.dec = RNNTBeamDecoderNative(model, blank_idx, kwargs.pop("eos_strategy", None), sentpiece_model, **kwargs)
        self.model = self.dec.model
        self.B = n_streams
        self.encoder = StreamingEncoder(self.model, n_streams)
        self.search = NativeBeamSearch(n_streams, **self.dec.search_args)
        self.all_streams = np.arange(n_streams, dtype=np.int32)
        self.straggler_cutoff, self.ring = straggler_cutoff, ring
        self.frames = None        # [ring, N, Hj] the most recent encoder frames
        self.n_frames = 0
        self.rounds = 0

    def backlog(self) -> int:
        return self.search.backlog(-1)

    def _frame_of(self, stream, frame):
        if len(stream) == self.B and int(frame[0]) == int(frame[-1]) == self.n_frames - 1:
            return self.frames[(self.n_frames - 1) % self.ring].unsqueeze(1)   # everyone on the newest frame, in order
        dev = self.frames.device
        idx = torch.from_numpy(np.stack([frame % self.ring, stream]).astype(np.int64)).to(dev, non_blocking=True)
        return self.frames[idx[0], idx[1]].unsqueeze(1)

    @torch.no_grad()
    def step(self, feats: torch.Tensor) -> List[Dict[int, FrameResponses]]:
        self.model.eval()
        prof = self.dec.profile
        t0 = time.perf_counter()
        f_all = self.encoder.advance(feats)
        if prof is not None:
            torch.cuda.synchronize()
            prof["encoder"] += time.perf_counter() - t0
        if f_all is None:
            return [dict() for _ in range(self.B)]
        if self.frames is None:
            self.frames = torch.zeros(self.ring, self.B, f_all.shape[-1], device=f_all.device, dtype=f_all.dtype)
        for j in range(f_all.shape[1]):
            # the slot about to be overwritten must not be needed any more
            cutoff = self.straggler_cutoff if self.search.backlog(-1) < self.ring - 2 else 0
            self.frames[self.n_frames % self.ring] = f_all[:, j]
            self.n_frames += 1
            self.search.push_frame(self.all_streams)
            self.rounds += self.dec._rounds(self.search, self._frame_of, stop_below=cutoff)
        t0 = time.perf_counter()
        out = self.search.take_responses()
        if prof is not None:
            prof["responses"] += time.perf_counter() - t0
        return out

    def close(self) -> List[Dict[int, FrameResponses]]:
        self.rounds += self.dec._rounds(self.search, self._frame_of)   # let the stragglers finish
        for b in range(self.B):
            self.search.close_stream(b)
        return self.search.take_responses()
