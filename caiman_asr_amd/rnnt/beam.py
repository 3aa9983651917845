"""Device side of the beam-search decoder: batched prediction + joint step for a set of live
hypotheses, temperature-scaled log-softmax, top-k and adaptive pruning, one packed device->host copy.

Restates training/caiman_asr_train/rnnt/beam.py:518-612 (`_batched_decode_step`, `_batched_decode`,
`_collate`) with the reference's defaults (beam width 4, prune threshold 1.5 nats, temperature 1.4,
beam.py:121-127).  The host side of beam search (hypothesis merging by string hash, n-gram / keyword
rescoring, final / partial emission, beam.py:285-516) is a "next" row of SURVEY §8f and is not part of this
round; `BeamExpander` is what that host loop calls per expansion.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class Expansion:
    """Result for one hypothesis: surviving (log-prob, token) pairs, the blank log-prob, new pred state."""
    scores: torch.Tensor       # CPU [<= beam_width]
    tokens: torch.Tensor       # CPU [<= beam_width]
    blank_logp: float
    pred_state: Tuple[torch.Tensor, torch.Tensor]  # device (h, c) each [L, 1, H]


class BeamExpander:
    def __init__(self, model, blank_idx: int, beam_width: int = 4, temperature: float = 1.4,
                 beam_prune_topk_thresh: float = 1.5):
        self.model = getattr(model, "module", model)
        self.blank_idx = blank_idx
        self.beam_width = beam_width
        self.temperature = temperature
        self.thresh = beam_prune_topk_thresh

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
        return F.log_softmax(logits.float() / self.temperature, dim=-1), (h, c)

    @torch.no_grad()
    def expand(self, f: torch.Tensor, y_last: Optional[torch.Tensor], state) -> List[Expansion]:
        log_p, (h, c) = self.log_probs(f, y_last, state)
        top_s, top_i = log_p.topk(self.beam_width, dim=1)
        keep = top_s >= top_s.max(dim=1, keepdim=True).values - self.thresh
        counts = keep.sum(1)
        # one packed transfer: [kept scores | kept tokens | blank log-probs | counts]
        packed = torch.cat([top_s[keep], top_i[keep].to(top_s.dtype), log_p[:, self.blank_idx],
                            counts.to(top_s.dtype)]).cpu()
        n, total = f.shape[0], int(packed.numel() - 2 * f.shape[0]) // 2
        scores, tokens = packed[:total], packed[total:2 * total].long()
        blank = packed[2 * total:2 * total + n]
        cnt = packed[2 * total + n:].long().tolist()
        out, lo = [], 0
        for i in range(n):
            out.append(Expansion(scores[lo:lo + cnt[i]], tokens[lo:lo + cnt[i]], float(blank[i]),
                                 (h[:, i:i + 1], c[:, i:i + 1])))
            lo += cnt[i]
        return out
