"""Beam-search decoder (SURVEY §8 a22 device step + f3 host search).

Behaviour of training/caiman_asr_train/rnnt/beam.py:77-687 with the reference's defaults (beam width 4, topk
prune 1.5 nats, score prune 0.4 nats per token, temperature 1.4, beam.py:121-127):

* per frame, the best open hypothesis is expanded with the surviving top-k tokens; a blank moves it to the
  closed set (merging by text hash with log-add), a token re-opens it (merging likewise, keeping the more
  probable token path); the frame ends once `beam_width` closed hypotheses beat every open one
  (beam.py:356-415, 449-516);
* the closed set is cut to the beam width and to those within `beam_prune_score_thresh` of the best
  length-normalised score (beam.py:661-683), then the shared text prefix is shipped as a final
  (serialise_responses.py);
* all utterances of a batch advance together: every round, the pending expansion of each utterance is stacked
  into one prediction + joint step on the device and ONE packed device->host copy brings back the survivors
  (beam.py:211-263, 518-612).

The reference drives this with one Python generator per utterance; here each utterance is an explicit
`_Search` state machine (`request()` / `feed()`), which keeps the device batching in one plain loop.
"""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from caiman_asr_amd.keywords.process import load_keywords
from caiman_asr_amd.keywords.trie import Keywords
from caiman_asr_amd.rnnt.decoder import RNNTCommonDecoder
from caiman_asr_amd.rnnt.eos_strategy import EOSPredict
from caiman_asr_amd.rnnt.hypothesis import SPU_UNICODE, Hypothesis, init_sos_hyp
from caiman_asr_amd.rnnt.response import FrameResponses
from caiman_asr_amd.rnnt.serialise_responses import ResponseSerializer

INF = float("inf")


@dataclass
class Expansion:
    """Result for one hypothesis: surviving (log-prob, token) pairs, the blank log-prob, new pred state."""
    scores: torch.Tensor       # CPU [<= beam_width]
    tokens: torch.Tensor       # CPU [<= beam_width]
    blank_logp: float
    pred_state: Tuple[torch.Tensor, torch.Tensor]  # device (h, c) each [L, 1, H]


class BeamExpander:
    """Device side: batched prediction + joint step for N hypotheses, temperature-scaled log-softmax, top-k and
    adaptive pruning, one packed device->host copy (beam.py:518-612)."""

    def __init__(self, model, blank_idx: int, beam_width: int = 4, temperature: float = 1.4,
                 beam_prune_topk_thresh: float = 1.5, logprob_correction=None):
        self.model = getattr(model, "module", model)
        self.blank_idx = blank_idx
        self.beam_width = beam_width
        self.temperature = temperature
        self.thresh = beam_prune_topk_thresh
        self.correct = logprob_correction

    @torch.no_grad()
    def log_probs(self, f: torch.Tensor, y_last: Optional[torch.Tensor], state):
        """f [N,1,Hj] encoder frames of the N hypotheses; y_last [N,1] last tokens (None: SOS step with a
        zero embedding); state (h, c) [L,N,H] or None -> (log_p [N,V] f32, (h, c))."""
        g, (h, c), _ = self.model.predict(y_last, state, add_sos=False)
        if y_last is None and f.shape[0] != g.shape[0]:
            g = g.expand(f.shape[0], -1, -1)
            h = h.expand(-1, f.shape[0], -1).contiguous()
            c = c.expand(-1, f.shape[0], -1).contiguous()
        logits = self.model.joint(f, g)[:, 0, 0, :]
        log_p = F.log_softmax(logits.float() / self.temperature, dim=-1)
        if self.correct is not None:
            log_p = self.correct(log_p)
        return log_p, (h, c)

    @torch.no_grad()
    def expand(self, f: torch.Tensor, y_last: Optional[torch.Tensor], state) -> List[Expansion]:
        log_p, (h, c) = self.log_probs(f, y_last, state)
        n = log_p.shape[0]
        top_s, top_i = log_p.topk(min(self.beam_width, log_p.shape[1]), dim=1)
        keep = top_s >= top_s.max(dim=1, keepdim=True).values - self.thresh
        counts = keep.sum(1)
        # one packed transfer: [kept scores | kept tokens | blank log-probs | counts]
        packed = torch.cat([top_s[keep], top_i[keep].to(top_s.dtype), log_p[:, self.blank_idx],
                            counts.to(top_s.dtype)]).cpu()
        total = int(packed.numel() - 2 * n) // 2
        scores, tokens = packed[:total], packed[total:2 * total].long()
        blank = packed[2 * total:2 * total + n]
        cnt = packed[2 * total + n:].long().tolist()
        out, lo = [], 0
        for i in range(n):
            j = i if h.shape[1] == n else 0
            out.append(Expansion(scores[lo:lo + cnt[i]], tokens[lo:lo + cnt[i]], float(blank[i]),
                                 (h[:, j:j + 1], c[:, j:j + 1])))
            lo += cnt[i]
        return out


class _Search:
    """Beam search over one utterance as a state machine: `request()` names the hypothesis to expand next (or
    None once the utterance is finished), `feed()` applies the device's answer."""

    def __init__(self, dec: "RNNTBeamDecoder", n_frames: int):
        self.dec = dec
        self.n_frames = n_frames
        sos = init_sos_hyp(dec._SOS, dec.ngram_lm)
        self.kept: Dict[int, Hypothesis] = {sos.hashval: sos}
        self.open: Dict[int, Hypothesis] = {}
        self.closed: Dict[int, Hypothesis] = {}
        self.cur: Optional[Hypothesis] = None
        self.responses: Dict[int, FrameResponses] = {}
        self.t = 0
        self.last_final_idx = 0
        self.done = False
        self._open_frame()

    # ---- frame boundaries ---------------------------------------------------------------------
    def _finish(self, key: int) -> None:
        self.responses[key] = self.dec.serialiser.last_frame_response(self.kept)
        self.done = True

    def _open_frame(self) -> None:
        d = self.dec
        if self.t >= self.n_frames:
            return self._finish(self.t)
        if d.max_symbol_per_sample is not None:
            best = max(self.kept.values(), key=lambda h: h.score)
            if best.y_length_tot > d.max_symbol_per_sample:
                return self._finish(self.t + 1)   # the reference's loop variable still names this frame (:302-307,353)
        for h in self.kept.values():
            h.y_len_t = 0
        self.open, self.closed = self.kept, {}

    def _close_frame(self) -> None:
        d, t = self.dec, self.t
        self.kept = d._prune_beam(self.closed)
        if max(self.kept.values(), key=lambda h: h.score).is_terminal:
            self.responses[t] = d.serialiser.last_frame_response(self.kept)
            self.done = True
            return
        since_final = (t - self.last_final_idx) * d.frame_width
        while True:
            self.responses[t], self.kept = d.serialiser.frame_responses(self.kept, t, d.return_partials)
            if len(self.kept) <= 1:
                self.last_final_idx = t
                break
            if self.responses[t].final is not None:
                self.last_final_idx = min(h.timesteps[0] for h in self.kept.values())
                break
            if since_final <= d.final_emission_thresh:
                break
            # overdue: drop the weakest hypothesis until the rest agree on a prefix (:345-348)
            self.kept.pop(min(self.kept.values(), key=d.normalised_score).hashval)
        if d._silence_terminate(self.kept, t):
            return self._finish(t + 1)
        self.t += 1
        self._open_frame()

    # ---- expansion ------------------------------------------------------------------------------
    def request(self) -> Optional[Tuple[Hypothesis, int]]:
        if self.done:
            return None
        best = max(self.open.values(), key=lambda h: h.score)
        self.cur = self.open.pop(best.hashval)
        return self.cur, self.t

    def feed(self, ex: Expansion) -> None:
        d, cur = self.dec, self.cur
        if d.add_ys(cur):
            steps = list(zip(ex.scores.tolist(), ex.tokens.tolist()))
            if all(tok != d.blank_idx for _, tok in steps):
                steps.append((ex.blank_logp, d.blank_idx))   # blank is always an option (:432-439)
        else:
            steps = [(ex.blank_logp, d.blank_idx)]           # symbol budget of this frame is spent
        for logp, tok in steps:
            d._update_hyps(logp, tok, cur, self.closed, self.open, self.t, ex.pred_state)
        if self.open:
            bar = max(h.score for h in self.open.values())
            ahead = {k: h for k, h in self.closed.items() if h.score > bar}
            if len(ahead) < d.beam_width:
                return                                        # keep expanding this frame
            self.closed = d._best_beam_width(ahead)
        else:
            self.closed = d._best_beam_width(self.closed)
        self._close_frame()


class RNNTBeamDecoder(RNNTCommonDecoder):
    """Constructor keywords follow beam.py:115-137.  `sentpiece_model` is a sentencepiece .model path or a
    ready list of piece strings indexed by token id (what `id_to_piece` would return)."""

    def __init__(self, model, blank_idx: int, eos_strategy, sentpiece_model: Union[str, Sequence[str]],
                 beam_width: int = 4, max_inputs_per_batch: int = int(1e7), max_symbols_per_step: Optional[int] = 8,
                 max_symbol_per_sample: Optional[int] = None, temperature: float = 1.4,
                 beam_prune_score_thresh: Union[int, float] = 0.4, beam_prune_topk_thresh: Union[int, float] = 1.5,
                 ngram_info=None, fuzzy_topk_logits: bool = False, return_partials: bool = False,
                 user_tokens: Optional[List[int]] = None, eos_is_terminal: bool = False,
                 eos_vad_threshold: float = INF, final_emission_thresh: float = INF,
                 frame_width: Optional[float] = None, keyword_boost_path: Optional[str] = None):
        super().__init__(model=model, blank_idx=blank_idx, eos_strategy=eos_strategy,
                         max_inputs_per_batch=max_inputs_per_batch, max_symbol_per_sample=max_symbol_per_sample,
                         max_symbols_per_step=max_symbols_per_step, temperature=temperature)
        assert beam_width > 0
        self.beam_width = beam_width
        if final_emission_thresh < 0:
            final_emission_thresh = INF
        if eos_vad_threshold != INF or final_emission_thresh != INF:
            assert frame_width is not None and frame_width > 0.0
        self.eos_vad_threshold = eos_vad_threshold
        self.final_emission_thresh = final_emission_thresh
        self.frame_width = 0.0 if frame_width is None else frame_width

        if isinstance(sentpiece_model, str):
            from sentencepiece import SentencePieceProcessor

            self.detokenize = SentencePieceProcessor(model_file=sentpiece_model).id_to_piece
        else:
            self.detokenize = list(sentpiece_model).__getitem__
        self.user_tokens = [] if user_tokens is None else user_tokens
        self.eos_is_terminal = eos_is_terminal
        self.keywords = Keywords([]) if keyword_boost_path is None else load_keywords(keyword_boost_path)

        if ngram_info:
            # kenlm is a third-party C++ dependency of the reference (lm/kenlm_ngram.py:4); it is bound here
            # only when importable -- the search itself needs nothing but begin_state()/score_ngram().
            from caiman_asr_amd.lm.kenlm_ngram import KenLmModel

            assert os.path.isfile(ngram_info.path), f"N-gram LM path {ngram_info.path} does not exist."
            assert ngram_info.scale_factor >= 0.0, f"N-gram scale factor is negative, {ngram_info.scale_factor}"
            self.ngram_lm, self.ngram_alpha = KenLmModel(ngram_info.path), ngram_info.scale_factor
        else:
            self.ngram_lm, self.ngram_alpha = None, 0.0

        if fuzzy_topk_logits:
            raise NotImplementedError("fuzzy_topk_logits emulates the FPGA's packetised argmax (fuzzy_logits.py); "
                                      "it is outside this path")
        self.beam_prune_topk_thresh = INF if beam_prune_topk_thresh < 0 else beam_prune_topk_thresh
        self.beam_prune_score_thresh = INF if beam_prune_score_thresh < 0 else beam_prune_score_thresh
        assert self.beam_prune_topk_thresh > 1e-9, \
            "--beam_prune_topk_thresh=0 prunes every token but the most probable: use --decoder=greedy instead"
        assert self.beam_prune_score_thresh > 1e-9, \
            "--beam_prune_score_thresh=0 prunes every hypothesis but the most probable: use --decoder=greedy instead"
        self.serialiser = ResponseSerializer(self._sort_nbest)
        self.return_partials = return_partials
        self.expander = BeamExpander(self.model, blank_idx, beam_width, temperature, self.beam_prune_topk_thresh,
                                     logprob_correction=self._eos_prob_correction)

    # ---- scoring helpers --------------------------------------------------------------------------------------
    @staticmethod
    def normalised_score(h: Hypothesis) -> float:
        return h.score / h.y_length_tot

    def add_ys(self, hyp: Hypothesis) -> bool:
        """May this hypothesis still take non-blank tokens on the current frame?"""
        return not self.max_symbols or hyp.y_len_t < self.max_symbols

    def _sort_nbest(self, hyps) -> List[Hypothesis]:
        return sorted(hyps, key=self.normalised_score, reverse=True)

    def _best_beam_width(self, hyps: Dict[int, Hypothesis]) -> Dict[int, Hypothesis]:
        if len(hyps) <= self.beam_width:
            return hyps
        top = sorted(hyps.values(), key=lambda h: h.score, reverse=True)[: self.beam_width]
        return {h.hashval: h for h in top}

    def _prune_beam(self, hyps: Dict[int, Hypothesis]) -> Dict[int, Hypothesis]:
        floor = max(self.normalised_score(h) for h in hyps.values()) - self.beam_prune_score_thresh
        return {k: h for k, h in hyps.items() if self.normalised_score(h) >= floor}

    def _silence_terminate(self, kept: Dict[int, Hypothesis], idx: int) -> bool:
        if self.eos_vad_threshold == INF:
            return False
        last = max(h.timesteps[-1] for h in kept.values())
        if last < 0:
            return False  # nothing but SOS (emitted at frame -1) so far
        return (idx - last) * self.frame_width >= self.eos_vad_threshold

    def _update_hyps(self, logp: float, tok: int, parent: Hypothesis, closed: Dict[int, Hypothesis],
                     open_: Dict[int, Hypothesis], time_idx: int, pred_state) -> None:
        """Apply one (log-prob, token) step of `parent` to the open / closed sets (beam.py:449-516)."""
        if tok == self.blank_idx:
            twin = closed.get(parent.hashval)
            if twin is not None:
                twin.score = np.logaddexp(twin.score, parent.score + logp)
            else:
                h = parent.clone()
                h.score += logp
                closed[h.hashval] = h
            return
        h = parent.clone()
        h.score += logp
        h.p_seq.append(float(np.exp(np.float32(logp))))
        h.timesteps.append(time_idx)
        h.pred_state = pred_state
        h.y_seq.append(tok)
        h.y_len_t += 1
        if self.eos_is_terminal and isinstance(self.eos_strategy, EOSPredict) and tok == self.eos_strategy.eos_idx:
            h.is_terminal = True
        assert tok != 0, "Decoding error: '<unk>' token encountered"   # id 0 has no text to score (:621,:635)
        piece = self.detokenize(tok)
        if self.ngram_lm is not None and tok not in self.user_tokens:   # meta tokens are unknown to the LM
            lm_score, h.ngram_lm_state = self.ngram_lm.score_ngram(piece, parent.ngram_lm_state)
            h.score += self.ngram_alpha * lm_score
        delta, h.kws_state = self.keywords.steps(piece, h.kws_state)
        h.score += delta
        # a word-boundary mark directly after a word-boundary mark adds nothing to the text (:644-659)
        text = piece[1:] if ord(h.s_seq[-1][-1]) == ord(piece[0]) == SPU_UNICODE else piece
        h.s_seq.append(piece)
        if text:
            h.update_hash(text)
        twin = open_.get(h.hashval)
        if twin is None:
            open_[h.hashval] = h
        else:
            merged = np.logaddexp(twin.score, h.score)
            if h.score > twin.score:           # same text: keep the likelier tokenisation and its states
                open_[h.hashval] = h
            open_[h.hashval].score = merged

    # ---- driver -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _inner_decode(self, encs: torch.Tensor, encs_len: torch.Tensor) -> List[Dict[int, FrameResponses]]:
        was_training = self.model.training
        self.model.eval()
        lens = encs_len.tolist()
        searches = [_Search(self, int(n)) for n in lens]
        while True:
            fresh, warm = [], []          # hypotheses without / with a prediction state
            for b, s in enumerate(searches):
                req = s.request()
                if req is not None:
                    (fresh if req[0].pred_state is None else warm).append((s, b, req[0], req[1]))
            if not fresh and not warm:
                break
            for group in (fresh, warm):
                if not group:
                    continue
                rows = torch.tensor([b for _, b, _, _ in group], device=encs.device)
                cols = torch.tensor([t for _, _, _, t in group], device=encs.device)
                f = encs[rows, cols].unsqueeze(1)
                if group is fresh:
                    y, state = None, None
                else:
                    y = torch.tensor([[h.y_last] for _, _, h, _ in group], dtype=torch.long, device=encs.device)
                    state = (torch.cat([h.pred_state[0] for _, _, h, _ in group], dim=1),
                             torch.cat([h.pred_state[1] for _, _, h, _ in group], dim=1))
                for (s, _, _, _), ex in zip(group, self.expander.expand(f, y, state)):
                    s.feed(ex)
        self.model.train(was_training)
        return [s.responses for s in searches]
