"""Batched greedy transducer decoding, offline and streaming.

Interface mirror of training/caiman_asr_train/rnnt/decoder.py:22-173 (`RNNTDecoder`,
`RNNTCommonDecoder`) and training/caiman_asr_train/rnnt/batched_greedy.py:22-331
(`RNNTBatchedGreedyDecoder`): same constructor keywords, `decode(feats, feat_lens)` ->
`List[Dict[int, FrameResponses]]`, same stop rules and emission bookkeeping.

Differences in HOW (not what): the reference leaves the device every iteration (`nonzero` to pick the
rows that emitted, batched_greedy.py:141-144).  Here the prediction network is stepped for every row
and merged with a mask, all loop state lives on the device, and the host looks at `done` only every
`sync_every` iterations, so thousands of streams advance without per-token host round trips.
`StreamingGreedyDecoder` keeps encoder / prediction state per stream between audio chunks
(`EncoderState` / `PredNetState`, training/caiman_asr_train/rnnt/state.py:13-38).
"""
from itertools import count
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from caiman_asr_amd import _lib
from caiman_asr_amd.rnnt.eos_strategy import EOSBlank, EOSIgnore, EOSPredict, EOSStrategy
from caiman_asr_amd.rnnt.response import DecodingResponse, FrameResponses, HypothesisResponse
from caiman_asr_amd.rnnt.state import EncoderState


def encode_lower_batch_size(model, feats, feat_lens, max_inputs_per_batch):
    """Encode in batch chunks so that T*F*b <= max_inputs_per_batch
    (training/caiman_asr_train/rnnt/unbatch_encoder.py:14-47)."""
    T, B, Fdim = feats.shape
    b = max(1, int(max_inputs_per_batch // max(1, T * Fdim)))
    if b >= B:
        f, lens, _ = model.encode(feats, feat_lens)
        return f, lens
    outs, lens_out = [], []
    for s in range(0, B, b):
        f, lens, _ = model.encode(feats[:, s:s + b].contiguous(), feat_lens[s:s + b])
        outs.append(f)
        lens_out.append(lens)
    Tm = max(o.shape[1] for o in outs)
    outs = [F.pad(o, (0, 0, 0, Tm - o.shape[1])) for o in outs]
    return torch.cat(outs, 0), torch.cat(lens_out, 0)


class RNNTCommonDecoder:
    def __init__(self, model, blank_idx: int, eos_strategy: EOSStrategy, max_symbol_per_sample: Optional[int],
                 max_symbols_per_step: Optional[int], max_inputs_per_batch: int = int(1e7), temperature: float = 1.0):
        assert max_symbols_per_step is None or max_symbols_per_step > 0
        assert max_symbol_per_sample is None or max_symbol_per_sample > 0
        self.model = getattr(model, "module", model)
        self.max_inputs_per_batch = max_inputs_per_batch
        self.eos_strategy = eos_strategy
        self.blank_idx = blank_idx
        self.max_symbols = max_symbols_per_step
        self.max_symbol_per_sample = max_symbol_per_sample
        self._SOS = -1
        self.temperature = temperature

    @property
    def eos_index(self) -> Optional[int]:
        return self.eos_strategy.eos_idx if isinstance(self.eos_strategy, EOSPredict) else None

    @torch.no_grad()
    def decode(self, feats: torch.Tensor, feat_lens: torch.Tensor) -> List[Dict[int, FrameResponses]]:
        """feats [T, B, F] log-mels, feat_lens [B] -> per utterance {frame index: FrameResponses}."""
        encs, enc_lens = encode_lower_batch_size(self.model, feats, feat_lens, self.max_inputs_per_batch)
        return self._inner_decode(encs, enc_lens)

    def _eos_prob_correction(self, logprobs):
        s = self.eos_strategy
        if s is None:
            return logprobs
        idx = s.eos_idx
        if isinstance(s, EOSIgnore):
            logprobs[:, idx] = -float("inf")
        elif isinstance(s, EOSBlank):
            logprobs[:, self.blank_idx] = torch.logaddexp(logprobs[:, self.blank_idx], logprobs[:, idx])
            logprobs[:, idx] = -float("inf")
        elif isinstance(s, EOSPredict):
            logprobs[:, idx] = logprobs[:, idx] * s.alpha
            if s.beta > 0:
                logprobs[:, idx] = torch.where(logprobs[:, idx] > np.log(s.beta), logprobs[:, idx],
                                               torch.full_like(logprobs[:, idx], -float("inf")))
        return logprobs

    def _joint_step(self, enc, pred):
        logits = self.model.joint(enc, pred)[:, 0, 0, :]
        # normalise BEFORE the EOS strategy so that beta is a probability threshold (decoder.py:161-172)
        return self._eos_prob_correction(F.log_softmax(logits.float() / self.temperature, dim=-1))

    def _joint_best(self, enc, pred):
        """-> (log-prob, token) of the most probable entry per row: `_joint_step(...).max(-1)` in one pass over the
        logits (`caiman_beam_topk`, k = 1: temperature, log-softmax, EOS correction and arg-max fused; ties go to the
        lower id, as torch's max does)."""
        logits = self.model.joint(enc, pred)[:, 0, 0, :]
        if not logits.is_cuda or logits.dtype == torch.float64:
            lp = self._eos_prob_correction(F.log_softmax(logits.float() / self.temperature, dim=-1))
            return lp.max(-1)
        if not logits.is_contiguous():
            logits = logits.contiguous()
        n, V = logits.shape
        s = self.eos_strategy
        mode = (0, 0, 1.0, 0.0)
        if isinstance(s, EOSIgnore):
            mode = (1, s.eos_idx, 1.0, 0.0)
        elif isinstance(s, EOSBlank):
            mode = (2, s.eos_idx, 1.0, 0.0)
        elif isinstance(s, EOSPredict):
            mode = (3, s.eos_idx, float(s.alpha), float(s.beta))
        score = torch.empty(n, dtype=torch.float32, device=logits.device)
        token = torch.empty(n, dtype=torch.int32, device=logits.device)
        blank = torch.empty(n, dtype=torch.float32, device=logits.device)
        _lib.check(_lib.lib().caiman_beam_topk(_lib.ptr(logits), n, V, logits.stride(0), _lib.dtype_tag(logits.dtype),
                                               float(self.temperature), self.blank_idx, mode[0], mode[1], mode[2], mode[3],
                                               1, _lib.ptr(score), _lib.ptr(token), _lib.ptr(blank), _lib.stream()))
        return score, token.long()


class RNNTBatchedGreedyDecoder(RNNTCommonDecoder):
    def __init__(self, model, blank_idx: int, eos_strategy: EOSStrategy, max_inputs_per_batch: int, tokenizer,
                 max_symbols_per_step: Optional[int] = 30, max_symbol_per_sample: Optional[int] = None,
                 sync_every: int = 8):
        super().__init__(model=model, blank_idx=blank_idx, eos_strategy=eos_strategy,
                         max_inputs_per_batch=max_inputs_per_batch, max_symbol_per_sample=max_symbol_per_sample,
                         max_symbols_per_step=max_symbols_per_step)
        self.detokenize = tokenizer.sentpiece.id_to_piece if tokenizer is not None else (lambda i: str(i))
        self.sync_every = sync_every

    # ---- one greedy iteration for every stream, fully on device ------------------------------------
    def _iterate(self, encs, st, active=None):
        """Advance every stream by one joint evaluation.  `active` (streaming): rows that are False
        keep all their state and report a blank.  Returns ((label, frame, prob), advanced_mask)."""
        B, _, jH = encs.shape
        blank = self.blank_idx
        idx = st["off"].clamp(max=encs.shape[1] - 1)
        f = torch.gather(encs, 1, idx.view(B, 1, 1).expand(-1, -1, jH))
        lp, k = self._joint_best(f, st["g"])  # first maximum wins (torch semantics), as in the reference
        at_end = st["off"] == st["max_off"]
        is_blank = k == blank
        # stop rules (batched_greedy.py:168-199): evaluated BEFORE this iteration's emission is counted
        done = st["done"] | (at_end & is_blank)
        if self.max_symbols is not None:
            done = done | (at_end & (st["per_step"] >= self.max_symbols))
        if self.max_symbol_per_sample is not None:
            done = done | (st["total"] >= self.max_symbol_per_sample)
        if active is not None:
            done = torch.where(active, done, st["done"])
        label = torch.where(done, torch.full_like(k, blank), k)
        if active is not None:
            label = torch.where(active, label, torch.full_like(k, blank))
        record = (label, st["off"].clone(), lp.exp())
        emitted = label != blank
        nonblank_k = ~is_blank
        total = st["total"] + nonblank_k.long() if self.max_symbol_per_sample is not None else st["total"]
        advance = is_blank
        per_step = st["per_step"]
        if self.max_symbols is not None:
            per_step = per_step + nonblank_k.long()
            advance = advance | (per_step >= self.max_symbols)
            # the counter is cleared only by a forced advance, never by a blank (batched_greedy.py:127-137)
            per_step = per_step * ((per_step < self.max_symbols) | at_end).long()
        off = torch.minimum(st["off"] + advance.long(), st["max_off"])
        if active is not None:
            keep = lambda new, old: torch.where(active, new, old)
            total, per_step, off, advance = keep(total, st["total"]), keep(per_step, st["per_step"]), \
                keep(off, st["off"]), advance & active
        st["done"], st["total"], st["per_step"], st["off"] = done, total, per_step, off
        # prediction network: step every row, keep the result where a symbol was emitted
        if st.get("big") is not None:   # thousands of rows: GEMM + cell kernel, states updated in place where emitted
            G = st["big"].step(torch.where(emitted, label, torch.zeros_like(label)), emitted)
            st["g"] = torch.where(emitted.view(B, 1, 1), G.view(B, 1, -1).to(st["g"].dtype), st["g"])
            return record, advance
        y = torch.where(emitted, label, torch.zeros_like(label)).unsqueeze(1)
        G, (HH, CC), _ = self.model.predict(y, (st["h"], st["c"]), add_sos=False)
        m = emitted.view(1, B, 1)
        st["h"] = torch.where(m, HH, st["h"])
        st["c"] = torch.where(m, CC, st["c"])
        st["g"] = torch.where(emitted.view(B, 1, 1), G.to(st["g"].dtype), st["g"])
        return record, advance

    LARGE_BATCH = 256   # from this many rows on the prediction step is a library GEMM + the cell kernel

    def _initial_state(self, B, device, enc_lens):
        z = lambda: torch.zeros(B, dtype=torch.long, device=device)
        rnn = self.model.prediction["dec_rnn"]
        if (B >= self.LARGE_BATCH and torch.device(device).type == "cuda" and hasattr(rnn, "lstm")
                and not rnn.batch_norm and not getattr(rnn.lstm, "hard", False)):
            from caiman_asr_amd.rnnt.streaming_lstm import LargeBatchPredictor

            big = LargeBatchPredictor(self.model, B)
            g = big.step(None, None, dev=device).view(B, 1, -1).clone()   # start of sequence: zero embedding, zero state
            return {"g": g, "h": None, "c": None, "big": big, "off": z(), "per_step": z(), "total": z(),
                    "done": torch.zeros(B, dtype=torch.bool, device=device),
                    "max_off": enc_lens.to(device=device, dtype=torch.long) - 1}
        g, (h, c), _ = self.model.predict(None, None, add_sos=False)  # zero embedding, zero state
        return {"g": g.expand(B, -1, -1).contiguous(), "h": h.expand(-1, B, -1).contiguous(),
                "c": c.expand(-1, B, -1).contiguous(), "off": z(), "per_step": z(), "total": z(),
                "done": torch.zeros(B, dtype=torch.bool, device=device),
                "max_off": enc_lens.to(device=device, dtype=torch.long) - 1}

    @torch.no_grad()
    def _inner_decode(self, encs: torch.Tensor, enc_lens: torch.Tensor) -> List[Dict[int, FrameResponses]]:
        training_state = self.model.training
        self.model.eval()
        B = encs.shape[0]
        st = self._initial_state(B, encs.device, enc_lens)
        labels, timestamps, probs = [], [], []
        for it in count():
            (lab, ts, pr), _ = self._iterate(encs, st)
            labels.append(lab)
            timestamps.append(ts)
            probs.append(pr)
            if (it + 1) % self.sync_every == 0 and bool(st["done"].all()):
                break
        self.model.train(training_state)
        return self._build_return_objs(enc_lens, labels, timestamps, probs)

    # ---- host-side assembly ------------------------------------------------------------------------------
    def _transpose_and_strip(self, labels, timestamps, probs) -> Tuple[List[List[int]], List[List[int]], List[List[float]]]:
        lab = torch.stack(labels, 1).cpu()
        ts = torch.stack(timestamps, 1).cpu()
        pr = torch.stack(probs, 1).cpu()
        keep = [(row != self.blank_idx).nonzero(as_tuple=True) for row in lab.unbind()]
        strip = lambda t: [x[ix].tolist() for x, ix in zip(t.unbind(), keep)]
        return strip(lab), strip(ts), strip(pr)

    def _build_return_objs(self, enc_lens, labels, timestamps, probs) -> List[Dict[int, FrameResponses]]:
        ys, tss, pss = self._transpose_and_strip(labels, timestamps, probs)
        out: List[Dict[int, FrameResponses]] = [{} for _ in range(len(enc_lens))]
        for i, (yy, tt, pp) in enumerate(zip(ys, tss, pss)):
            for y, t, p in zip(yy, tt, pp):
                if t not in out[i]:
                    out[i][t] = FrameResponses(None, final=DecodingResponse(
                        start_frame_idx=t, duration_frames=1, is_provisional=False,
                        alternatives=[HypothesisResponse(y_seq=[y], timesteps=[t], token_seq=[self.detokenize(y)],
                                                         confidence=[p])]))
                else:
                    hyp = out[i][t].final.alternatives[0]
                    hyp.y_seq.append(y)
                    hyp.timesteps.append(t)
                    hyp.token_seq.append(self.detokenize(y))
                    hyp.confidence.append(p)
        return out


def flatten_responses(responses: List[Dict[int, FrameResponses]]):
    """-> (tokens, frames, confidences) per utterance, in emission order."""
    toks, frames, confs = [], [], []
    for per_utt in responses:
        tk, fr, cf = [], [], []
        for t in sorted(per_utt):
            if per_utt[t].final is None:  # beam search: frames without a newly shared prefix
                continue
            hyp = per_utt[t].final.alternatives[0]
            tk += hyp.y_seq
            fr += hyp.timesteps
            cf += hyp.confidence
        toks.append(tk)
        frames.append(fr)
        confs.append(cf)
    return toks, frames, confs


class StreamingEncoder:
    """Encoder of B concurrent streams advanced chunk by chunk: pre-rnn with carried (h, c), StackTime over a
    carried remainder of pre-rnn frames, post-rnn with carried (h, c), joint_enc projection."""

    LARGE_BATCH = 256   # from this many streams on, a timestep of a layer is a library GEMM + the cell kernel

    def __init__(self, model, n_streams: int, large_batch: Optional[bool] = None):
        self.model = getattr(model, "module", model)
        self.B = n_streams
        self.enc_state: Optional[EncoderState] = None
        self.carry = None  # pre-rnn output frames not yet consumed by StackTime
        self.big = None
        if large_batch if large_batch is not None else n_streams >= self.LARGE_BATCH:
            from caiman_asr_amd.rnnt.streaming_lstm import LargeBatchLSTM

            self.big = (LargeBatchLSTM(self.model.encoder["pre_rnn"], n_streams),
                        LargeBatchLSTM(self.model.encoder["post_rnn"], n_streams))

    @torch.no_grad()
    def advance(self, feats: torch.Tensor) -> Optional[torch.Tensor]:
        """feats [n, B, in_feats] -> encoder frames completed by this chunk, [B, n_enc, Hj], or None."""
        m, B = self.model, self.B
        factor = m.enc_stack_time_factor
        if self.big is not None:      # states live inside the two LargeBatchLSTM objects
            x = self.big[0].forward(feats)
            if self.carry is not None:
                x = torch.cat([self.carry, x], 0)
            n_full = (x.shape[0] // factor) * factor
            self.carry = x[n_full:] if n_full < x.shape[0] else None
            if n_full == 0:
                return None
            xs = x[:n_full].view(n_full // factor, factor, B, -1).transpose(1, 2).reshape(n_full // factor, B, -1)
            y = self.big[1].forward(xs)
            return m.joint_enc(y.transpose(0, 1))
        x, pre_state, _ = m.encoder["pre_rnn"](feats, self.enc_state.pre_rnn if self.enc_state else None)
        if self.carry is not None:
            x = torch.cat([self.carry, x], 0)
        n_full = (x.shape[0] // factor) * factor
        self.carry = x[n_full:] if n_full < x.shape[0] else None
        post_prev = self.enc_state.post_rnn if self.enc_state else None
        if n_full == 0:
            self.enc_state = EncoderState(pre_rnn=pre_state, post_rnn=post_prev)
            return None
        xs = x[:n_full].view(n_full // factor, factor, B, -1).transpose(1, 2).reshape(n_full // factor, B, -1)
        y, post_state, _ = m.encoder["post_rnn"](xs, post_prev)
        self.enc_state = EncoderState(pre_rnn=pre_state, post_rnn=post_state)
        return m.joint_enc(y.transpose(0, 1))


class StreamingGreedyDecoder:
    """Thousands of concurrent real-time streams on one GPU: every `step()` takes the next chunk of
    spliced features for ALL streams ([frames, B, in_feats]; 2 frames = 60 ms at the base config),
    advances the encoder with its carried LSTM state, and runs the greedy loop over the new encoder
    frames with the carried prediction state.  Per-stream persistent state = encoder (h,c) for the
    pre/post stacks + a pending odd pre-rnn frame for StackTime, prediction (h,c), g, emission counters."""

    def __init__(self, model, blank_idx: int, n_streams: int, max_symbols_per_step: Optional[int] = 30,
                 eos_strategy: EOSStrategy = None):
        self.dec = RNNTBatchedGreedyDecoder(model, blank_idx, eos_strategy, int(1e12), None,
                                            max_symbols_per_step=max_symbols_per_step, sync_every=1)
        self.model = self.dec.model
        self.B = n_streams
        self.encoder = StreamingEncoder(self.model, n_streams)
        self.pred = None
        self.frames_seen = 0

    @torch.no_grad()
    def step(self, feats: torch.Tensor):
        """feats [n, B, in_feats] -> list (per new encoder frame) of (tokens [B, <=max_symbols], counts [B])."""
        m = self.model
        m.eval()
        dev = feats.device
        B = self.B
        f_all = self.encoder.advance(feats)  # [B, n_enc, Hj]
        if f_all is None:
            return []
        if self.pred is None:
            self.pred = self.dec._initial_state(B, dev, torch.ones(B, dtype=torch.long, device=dev))
            self.pred["max_off"].fill_(1 << 40)  # a live stream is never "at the last frame"
        out = []
        cap = self.dec.max_symbols or 30
        st = self.pred
        for j in range(f_all.shape[1]):
            # non-final-frame rule of the batched loop: stay on the frame until a blank or until the
            # per-frame symbol cap forces an advance (the cap counter is carried, exactly as offline)
            st["off"].zero_()
            toks = torch.full((B, cap), self.dec.blank_idx, dtype=torch.long, device=dev)
            n_emit = torch.zeros(B, dtype=torch.long, device=dev)
            active = torch.ones(B, dtype=torch.bool, device=dev)
            fj = f_all[:, j:j + 1].contiguous()
            for _ in range(cap + 1):
                (lab, _, _), advanced = self.dec._iterate(fj, st, active=active)
                emitted = lab != self.dec.blank_idx
                pos = n_emit.clamp(max=cap - 1).unsqueeze(1)
                toks.scatter_(1, pos, torch.where(emitted.unsqueeze(1), lab.unsqueeze(1), toks.gather(1, pos)))
                n_emit += emitted.long()
                active = active & ~advanced
                if not bool(active.any()):
                    break
            out.append((toks, n_emit))
            self.frames_seen += 1
        return out
