"""Definition-level transducer loss by explicit path enumeration (tiny lattices).

TEST INFRASTRUCTURE ONLY.  This is the mathematical definition the reference's
kernel implements (Graves 2012 "Sequence Transduction with RNNs" Eq. 16-18; delay
penalty: arXiv 2211.00490 Eq. 19; the EOS and star ("uncertain label") terms as
coded in training/lib/csrc/transducer_loss.cu:120-173).  It shares no code with
oracle/rnnt_oracle.c and is used to pin it.
"""
import itertools
import math

import numpy as np


def loss_by_enumeration(lp, label, T, U, blank, delay_penalty=0.0, eos_penalty=0.0,
                        eos_idx=-1, star_lam=0.0, star_idx=-2):
    """-log sum over all monotone alignments of one utterance.

    lp : [T, U+1, V] log-softmax of the logits;  label : U ints.
    A path makes T blank ("null") moves and U label moves, ending with the blank
    move out of (T-1, U).
    """
    total = -math.inf
    moves = T - 1 + U  # the final null move is fixed
    for emit_pos in itertools.combinations(range(moves), U):
        emit_pos = set(emit_pos)
        t = u = 0
        s = 0.0
        for m in range(moves):
            if m in emit_pos:
                pen = delay_penalty * ((T - 1) / 2 - t)
                if label[u] == star_idx:
                    s += pen
                else:
                    s += lp[t, u, label[u]] + pen
                    if label[u] == eos_idx:
                        s += eos_penalty * ((T - 1) / 2 - t)
                u += 1
            else:
                s += _null(lp, label, t, u, blank, star_lam, star_idx)
                t += 1
        assert t == T - 1 and u == U
        s += _null(lp, label, t, u, blank, star_lam, star_idx)
        total = np.logaddexp(total, s)
    return -total


def _null(lp, label, t, u, blank, star_lam, star_idx):
    if u > 0 and label[u - 1] == star_idx:
        return star_lam
    return lp[t, u, blank]
