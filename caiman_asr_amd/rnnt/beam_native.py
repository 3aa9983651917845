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
from caiman_asr_amd.rnnt import streaming_lstm

_SPIN_WAIT = __import__("os").environ.get("CAIMAN_BEAM_SPIN", "0") != "0"   # polling the event instead of stream-synchronise: no gain measured (p50 10.9-11.6 ms either way)
# One expansion round = index upload + gather + two cell GEMMs + joint_pred + relu-add + joint_fc + top-k + result download.
# The host waits for every round's result, so the ~10 us each of those nine commands takes to issue sits on the tick's
# critical path 20-30 times per tick.  CAIMAN_BEAM_GRAPH=1: the round is captured once per bucket of row counts as a
# hipGraph and replayed with one launch; rows are padded up to the bucket (the padding rows read the zero state and encoder
# row 0 and write into a spare pool row).  Default since round 4 (2 000 streams: p50 of the tick 8.1-8.9 -> 6.9-7.2 ms, same
# tokens); 0: nine eager launches per round.
ROUND_GRAPH = __import__("os").environ.get("CAIMAN_BEAM_GRAPH", "1") != "0"


def _rows_bucket(n: int) -> int:
    """rows of a captured round: 64, 128, 256, 512, then multiples of 256 (at most 25 % of padding above 1024 rows)"""
    b = 64
    while b < n and b < 512:
        b *= 2
    return b if n <= b else (n + 255) // 256 * 256


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
                 return_partials: bool = False, keywords: Sequence[str] = (), keyword_weights: Sequence[float] = (),
                 max_expansions_per_frame: int = 0):
        self.L = _lib.lib()
        self.n_streams, self.k = n_streams, beam_width
        cfg = _lib.BeamConfig(
            blank_idx=blank_idx, beam_width=beam_width, max_symbols_per_step=max_symbols_per_step or 0,
            max_symbol_per_sample=-1 if max_symbol_per_sample is None else max_symbol_per_sample,
            beam_prune_score_thresh=beam_prune_score_thresh, beam_prune_topk_thresh=beam_prune_topk_thresh,
            eos_vad_threshold=-1.0 if eos_vad_threshold == INF else eos_vad_threshold,
            final_emission_thresh=-1.0 if final_emission_thresh == INF else final_emission_thresh,
            frame_width=0.0 if frame_width is None else frame_width,
            eos_terminal_idx=-1 if eos_terminal_idx is None else eos_terminal_idx, return_partials=int(return_partials),
            max_expansions_per_frame=max_expansions_per_frame)
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

    def capped_frames(self) -> int:
        """Frames settled early by `max_expansions_per_frame`."""
        return int(self.L.caiman_beam_capped_frames(self.h))

    def state_slots(self) -> int:
        return int(self.L.caiman_beam_state_slots(self.h))

    def take_raw_responses(self):
        """Responses since the last call as the library's flat records (include/caiman_beam.h): (ints i32, floats f32)
        numpy copies.  What a server forwards without building Python objects; `count_final_tokens` reads them."""
        ip, fp = _I32P(), ctypes.POINTER(ctypes.c_float)()
        ni, nf = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.L.caiman_beam_responses(self.h, ctypes.byref(ip), ctypes.byref(ni), ctypes.byref(fp),
                                                 ctypes.byref(nf)))
        ints = np.ctypeslib.as_array(ip, (ni.value,)).copy() if ni.value else np.zeros(0, np.int32)
        flts = np.ctypeslib.as_array(fp, (nf.value,)).copy() if nf.value else np.zeros(0, np.float32)
        self.L.caiman_beam_clear_responses(self.h)
        return ints, flts

    @staticmethod
    def count_final_tokens(ints: np.ndarray, n_floats: int, has_partials: bool = False) -> int:
        """Tokens in the final records of a raw response block.  Without partials every confidence belongs to a
        final, so the count is the length of the float array; with partials the record headers are walked."""
        if not has_partials:
            return int(n_floats)
        i, total, n = 0, 0, len(ints)
        while i < n:
            kind, n_alt = int(ints[i + 2]), int(ints[i + 5])
            i += 6
            for _ in range(n_alt):
                k = int(ints[i])
                if kind == 0:
                    total += k
                i += 1 + 2 * k
        return total

    def take_responses(self) -> List[Dict[int, FrameResponses]]:
        """Responses since the last call -> per stream {frame key: FrameResponses}."""
        ints, flts = self.take_raw_responses()
        ints, flts = ints.tolist(), flts.tolist()
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
    """One expansion round on the device, ~11 launches: index upload, input gather, per LSTM layer one library GEMM
    + the cell kernel (new states scattered straight into the slot pool), joint_pred GEMM, relu(f + g), joint_fc
    GEMM, `caiman_beam_topk`, one copy of the fixed-shape result to pinned host memory.

    Prediction states of all live hypotheses sit in pools [layers, 1 + slots, hidden] (row 0 = the all-zero start
    state; h in the compute dtype, c in f32).  Weights are re-laid once per compute dtype: [W_ih | W_hh] per layer,
    b_ih + b_hh, everything cast to the autocast dtype when autocast is on."""

    def __init__(self, model, blank_idx: int, beam_width: int, temperature: float, eos_strategy=None):
        self.model = m = getattr(model, "module", model)
        self.blank_idx, self.k, self.temperature = blank_idx, beam_width, temperature
        self.eos = (0, 0, 1.0, 0.0)
        if isinstance(eos_strategy, EOSIgnore):
            self.eos = (1, eos_strategy.eos_idx, 1.0, 0.0)
        elif isinstance(eos_strategy, EOSBlank):
            self.eos = (2, eos_strategy.eos_idx, 1.0, 0.0)
        elif isinstance(eos_strategy, EOSPredict):
            self.eos = (3, eos_strategy.eos_idx, float(eos_strategy.alpha), float(eos_strategy.beta))
        rnn = m.prediction["dec_rnn"]
        if rnn.batch_norm or getattr(rnn.lstm, "hard", False):
            raise NotImplementedError("the fused beam round covers the plain (soft-activation, no batch-norm) "
                                      "prediction LSTM of the shipped configs")
        self.L, self.H = rnn.num_layers, m.pred_n_hid
        self.w = {}          # compute dtype -> re-laid weights
        self._done = None
        self.h_pool = self.c_pool = None
        self.cap = 0
        self.stats = None
        self.graphs = {}     # (rows bucket, operand addresses) -> captured round (hipGraph): one launch instead of nine
        self._graph_off = False

    def _weights(self, cd):
        w = self.w.get(cd)
        if w is None:
            m, lstm = self.model, self.model.prediction["dec_rnn"].lstm
            with torch.no_grad():
                w = dict(embed=m.prediction["embed"].weight.detach().to(cd).contiguous(),
                         Wp=m.joint_pred.weight.detach().to(cd).contiguous(), bp=m.joint_pred.bias.detach().to(cd),
                         Wf=m.joint_fc.weight.detach().to(cd).contiguous(), bf=m.joint_fc.bias.detach().to(cd))
                w["fused"] = streaming_lstm.fused_step_ok(self.H, cd) and w["embed"].shape[1] % 128 == 0
                for l in range(self.L):
                    if w["fused"]:   # one launch per layer-step: GEMM with the cell update as its epilogue (rounds up to FUSED_MAX_ROWS)
                        w[f"Wc{l}"], w[f"bc{l}"], _ = streaming_lstm.fused_layer_weights(lstm, l, cd)
                    w[f"W{l}"] = torch.cat([getattr(lstm, f"weight_ih_l{l}"), getattr(lstm, f"weight_hh_l{l}")], 1) \
                        .detach().to(cd).contiguous()
                    w[f"b{l}"] = (getattr(lstm, f"bias_ih_l{l}") + getattr(lstm, f"bias_hh_l{l}")).detach().to(cd)
            self.w[cd] = w
        return w

    def refresh_weights(self):
        """Call after the model's parameters changed."""
        self.w = {}

    def _ensure(self, n_slots: int, n: int, dev, cd):
        L, H, k = self.L, self.H, self.k
        if self.h_pool is None or self.h_pool.shape[1] < n_slots + 2 or self.h_pool.dtype != cd:
            rows = max(2 * n_slots, 64) + 2       # row 0: the zero start state; last row: where a captured round's padding rows write
            self.graphs.clear()                   # captured rounds hold the old pools' addresses
            h = torch.zeros(L, rows, H, device=dev, dtype=cd)
            c = torch.zeros(L, rows, H, device=dev, dtype=torch.float32)
            if self.h_pool is not None:
                h[:, :self.h_pool.shape[1]] = self.h_pool.to(cd)
                c[:, :self.c_pool.shape[1]] = self.c_pool
            self.h_pool, self.c_pool = h, c
        if self.cap < n or self.X[0].dtype != cd:
            cap = self.cap = max(2 * n, 256)
            self.graphs.clear()
            m = self.model
            E, Hj, V = m.prediction["embed"].weight.shape[1], m.joint_pred.weight.shape[0], m.joint_fc.weight.shape[0]
            self.idx_host = torch.empty(6, cap, dtype=torch.int32).pin_memory()    # y, in, out, row (int64 = 2 rows), pad
            self.idx_dev = torch.empty(6, cap, dtype=torch.int32, device=dev)
            self.X = [torch.empty(cap, (E if l == 0 else H) + H, device=dev, dtype=cd) for l in range(L)]
            self.G_in = torch.empty(cap, H, device=dev, dtype=cd)
            self.gates = torch.empty(cap, 4 * H, device=dev, dtype=cd)
            self.g = torch.empty(cap, Hj, device=dev, dtype=cd)
            self.A = torch.empty(cap, Hj, device=dev, dtype=cd)
            self.logits = torch.empty(cap, V, device=dev, dtype=cd)
            self.out_dev = torch.empty(cap * (2 * k + 1), dtype=torch.float32, device=dev)
            self.out_host = torch.empty(cap * (2 * k + 1), dtype=torch.float32).pin_memory()

    @torch.no_grad()
    def __call__(self, frames2d: torch.Tensor, rows: np.ndarray, y_last: np.ndarray, state_in: np.ndarray,
                 state_out: np.ndarray, n_slots: int):
        """frames2d [R, Hj] encoder frames and, per request, the row to use; y_last / state_in (-1 = start of
        sequence) / state_out [n] -> (scores [n, k] f32, tokens [n, k] i32, blank [n] f32): numpy views of the
        pinned result buffer, valid until the next call."""
        dev, n, k, L, H = frames2d.device, len(rows), self.k, self.L, self.H
        cd = torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else self.model.joint_fc.weight.dtype
        if frames2d.dtype != cd:
            frames2d = frames2d.to(cd)
        assert frames2d.is_contiguous()
        w = self._weights(cd)
        self._ensure(n_slots, n, dev, cd)
        ih = self.idx_host.numpy()
        ih[0, :n], ih[1, :n], ih[2, :n] = y_last, state_in, state_out
        ih[3:5].reshape(-1).view(np.int64)[:n] = rows
        nb = n
        if ROUND_GRAPH and not self._graph_off:
            # the round as ONE graph launch, captured per bucket of row counts; the padding rows start from the zero state,
            # read encoder row 0 and write their state into the pools' spare last row
            nb = _rows_bucket(n)
            self._ensure(n_slots, nb, dev, cd)
            ih = self.idx_host.numpy()
            if nb > n:
                ih[0, n:nb], ih[1, n:nb], ih[2, n:nb] = 0, -1, self.h_pool.shape[1] - 2
                ih[3:5].reshape(-1).view(np.int64)[n:nb] = 0
            env = (frames2d.data_ptr(), tuple(frames2d.shape), cd, id(w))
            g = self.graphs.get((nb,) + env)
            if g is None:
                # Capture EVERY bucket the buffers can hold now (a tick that meets a new row count later must not pay
                # for a capture: 5-10 ms each).  Before the first one the round runs once eagerly, so that the library's
                # handles and workspaces exist (it reads its states from slots it does not write: running it twice leaves
                # the same result).
                self._round(frames2d, nb, w, cd)
                torch.cuda.current_stream().synchronize()
                b = 64
                try:
                    while b <= self.cap:
                        if (b,) + env not in self.graphs:
                            gb = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(gb, capture_error_mode="thread_local"):   # other threads (a data feed) may use the GPU meanwhile
                                self._round(frames2d, b, w, cd)
                            self.graphs[(b,) + env] = gb
                        b = b * 2 if b < 512 else b + 256
                except Exception as e:      # a capture the runtime refuses: the same kernels as eager launches, said once
                    import warnings

                    warnings.warn(f"beam round: graph capture failed ({e!r}); continuing with eager launches")
                    self._graph_off = True
                    self.graphs.clear()
                    torch.cuda.synchronize()
                g = self.graphs.get((nb,) + env)
            if g is not None:
                g.replay()         # (the eager warm-up above has already run this round when the capture failed)
        else:
            self._round(frames2d, n, w, cd)
        k = self.k
        if _SPIN_WAIT:   # a round is ~0.2 ms of device work on the tick's critical path, 26 times per tick: poll the event
            ev = self._done if self._done is not None else torch.cuda.Event()   # instead of sleeping in stream synchronise
            self._done = ev
            ev.record()
            while not ev.query():
                pass
        else:
            torch.cuda.current_stream().synchronize()
        hn = self.out_host.numpy()      # laid out for nb rows (nb = n without the graph): scores | tokens | blank
        if self.stats is not None:   # [sum of top-1 probabilities, rows]: how peaked the workload is
            self.stats[0] += float(np.exp(hn[: n * k: k]).sum())
            self.stats[1] += n
        return (hn[: nb * k].reshape(nb, k)[:n], hn[nb * k: 2 * nb * k].view(np.int32).reshape(nb, k)[:n],
                hn[2 * nb * k: 2 * nb * k + nb][:n])

    def _round(self, frames2d, n, w, cd):
        """the device work of one expansion round for n rows of the index buffers, on the current stream (eager or captured)"""
        dev, k, L, H = frames2d.device, self.k, self.L, self.H
        self.idx_dev.copy_(self.idx_host, non_blocking=True)
        y, s_in, s_out = (_lib.ptr(self.idx_dev[i]) for i in range(3))
        row = _lib.ptr(self.idx_dev[3])
        lib, tag, st = _lib.lib(), _lib.dtype_tag(cd), _lib.stream()
        E = w["embed"].shape[1]
        _lib.check(lib.caiman_beam_gather_inputs(_lib.ptr(w["embed"]), E, _lib.ptr(self.h_pool[0]), H, y, s_in, n,
                                                 _lib.ptr(self.X[0]), self.X[0].shape[1], tag, st))
        for l in range(L):
            nxt = self.X[l + 1] if l + 1 < L else self.G_in
            if w["fused"] and n <= streaming_lstm.FUSED_MAX_ROWS:
                streaming_lstm.lstm_step_gemm(self.X[l], w[f"Wc{l}"], w[f"bc{l}"], n, H, self.c_pool[l], self.h_pool[l],
                                              self.h_pool[l + 1] if l + 1 < L else None, s_in, s_out, nxt, tag, st)
                continue
            torch.addmm(w[f"b{l}"], self.X[l][:n], w[f"W{l}"].t(), out=self.gates[:n])
            _lib.check(lib.caiman_beam_lstm_cell(_lib.ptr(self.gates), H, _lib.ptr(self.c_pool[l]), _lib.ptr(self.h_pool[l]),
                                                 _lib.ptr(self.h_pool[l + 1]) if l + 1 < L else None, s_in, s_out, n,
                                                 _lib.ptr(nxt), nxt.shape[1], tag, st))
        torch.addmm(w["bp"], self.G_in[:n], w["Wp"].t(), out=self.g[:n])
        Hj = self.g.shape[1]
        _lib.check(lib.caiman_beam_joint_act(_lib.ptr(frames2d), row, _lib.ptr(self.g), n, Hj, _lib.ptr(self.A), tag, st))
        torch.addmm(w["bf"], self.A[:n], w["Wf"].t(), out=self.logits[:n])
        flat = self.out_dev
        sc, tk, bl = flat[: n * k], flat[n * k: 2 * n * k].view(torch.int32), flat[2 * n * k: 2 * n * k + n]
        V = self.logits.shape[1]
        _lib.check(lib.caiman_beam_topk(_lib.ptr(self.logits), n, V, V, tag, self.temperature, self.blank_idx,
                                        self.eos[0], self.eos[1], self.eos[2], self.eos[3], k, _lib.ptr(sc),
                                        _lib.ptr(tk), _lib.ptr(bl), st))
        self.out_host[: 2 * n * k + n].copy_(flat[: 2 * n * k + n], non_blocking=True)


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
                 frame_width: Optional[float] = None, keyword_boost_path: Optional[str] = None, device_step=None,
                 max_expansions_per_frame: int = 0):
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
            return_partials=return_partials, keywords=kw, keyword_weights=kww,
            max_expansions_per_frame=max_expansions_per_frame)
        self.step = device_step or HipBeamStep(self.model, blank_idx, beam_width, temperature, eos_strategy)
        self.profile: Optional[Dict[str, float]] = None   # set to a defaultdict(float) to collect host timings

    def _rounds(self, search: NativeBeamSearch, frames2d: torch.Tensor, rows_of, stop_below: int = 0,
                deadline: Optional[float] = None):
        """Expansion rounds until no stream has a request left -- or, with `stop_below`, until a round served no
        more than that many streams (the stragglers carry on in the next call; with a `deadline`, a perf_counter
        time, they are still served until then).  `rows_of(streams, frames)` -> the row of `frames2d` [R, Hj] holding
        each request's encoder frame."""
        n_rounds = 0
        prof = self.profile
        while True:
            t0 = time.perf_counter()
            stream, frame, y_last, s_in, s_out = search.requests()
            if len(stream) == 0:
                return n_rounds
            n_rounds += 1
            t1 = time.perf_counter()
            scores, tokens, blank = self.step(frames2d, rows_of(stream, frame), y_last, s_in, s_out, search.state_slots())
            t2 = time.perf_counter()
            search.feed(scores, tokens, blank)
            if prof is not None:
                t3 = time.perf_counter()
                prof["requests"] += t1 - t0
                prof["device_round"] += t2 - t1
                prof["feed"] += t3 - t2
                prof["rounds"] += 1
                prof["expansions"] += len(stream)
            if len(stream) <= stop_below and (deadline is None or time.perf_counter() >= deadline):
                return n_rounds

    @torch.no_grad()
    def _inner_decode(self, encs: torch.Tensor, encs_len: torch.Tensor) -> List[Dict[int, FrameResponses]]:
        was_training = self.model.training
        self.model.eval()
        lens = np.asarray(encs_len.tolist(), dtype=np.int64)
        B = len(lens)
        search = NativeBeamSearch(B, **self.search_args)

        T = encs.shape[1]
        frames2d = encs.contiguous().view(B * T, -1)

        def rows_of(stream, frame):
            return stream.astype(np.int64) * T + frame

        for t in range(int(lens.max()) if B else 0):
            search.push_frame(np.nonzero(lens > t)[0])
            self._rounds(search, frames2d, rows_of)
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
    served no more than that many streams (and `tick_budget_s`, if given, has been used up): those streams keep
    their open frame, queue the frames that arrive meanwhile (`ring` encoder frames are kept) and catch up inside
    later ticks.  `backlog()` reports the lag."""

    def __init__(self, model, blank_idx: int, n_streams: int, sentpiece_model: Union[str, Sequence[str]],
                 straggler_cutoff: int = 0, ring: int = 32, tick_budget_s: Optional[float] = None,
                 raw_responses: bool = False, **kwargs):
        self.dec = RNNTBeamDecoderNative(model, blank_idx, kwargs.pop("eos_strategy", None), sentpiece_model, **kwargs)
        self.model = self.dec.model
        self.B = n_streams
        self.encoder = StreamingEncoder(self.model, n_streams)
        self.search = NativeBeamSearch(n_streams, **self.dec.search_args)
        self.all_streams = np.arange(n_streams, dtype=np.int32)
        self.straggler_cutoff, self.ring, self.tick_budget_s = straggler_cutoff, ring, tick_budget_s
        self.raw_responses = raw_responses   # step() returns the flat records instead of FrameResponses objects
        self.frames = None        # [ring, N, Hj] the most recent encoder frames
        self.n_frames = 0
        self.rounds = 0

    def backlog(self) -> int:
        return self.search.backlog(-1)

    def _rows_of(self, stream, frame):
        return (frame % self.ring).astype(np.int64) * self.B + stream

    @torch.no_grad()
    def step(self, feats: torch.Tensor) -> List[Dict[int, FrameResponses]]:
        self.model.eval()
        prof = self.dec.profile
        t0 = time.perf_counter()
        deadline = None if self.tick_budget_s is None else t0 + self.tick_budget_s
        f_all = self.encoder.advance(feats)
        if prof is not None:
            torch.cuda.synchronize()
            prof["encoder"] += time.perf_counter() - t0
        if f_all is None:
            return (np.zeros(0, np.int32), np.zeros(0, np.float32)) if self.raw_responses else [dict() for _ in range(self.B)]
        if self.frames is None:
            self.frames = torch.zeros(self.ring, self.B, f_all.shape[-1], device=f_all.device, dtype=f_all.dtype)
        for j in range(f_all.shape[1]):
            # the slot about to be overwritten must not be needed any more
            cutoff = self.straggler_cutoff if self.search.backlog(-1) < self.ring - 2 else 0
            self.frames[self.n_frames % self.ring] = f_all[:, j]
            self.n_frames += 1
            self.search.push_frame(self.all_streams)
            self.rounds += self.dec._rounds(self.search, self.frames.view(self.ring * self.B, -1), self._rows_of,
                                            stop_below=cutoff, deadline=deadline)
        t0 = time.perf_counter()
        out = self.search.take_raw_responses() if self.raw_responses else self.search.take_responses()
        if prof is not None:
            prof["responses"] += time.perf_counter() - t0
        return out

    def close(self) -> List[Dict[int, FrameResponses]]:
        if self.frames is not None:   # let the stragglers finish
            self.rounds += self.dec._rounds(self.search, self.frames.view(self.ring * self.B, -1), self._rows_of)
        for b in range(self.B):
            self.search.close_stream(b)
        return self.search.take_raw_responses() if self.raw_responses else self.search.take_responses()
