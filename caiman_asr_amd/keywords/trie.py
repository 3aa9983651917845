"""Keyword boosting automaton for beam search.

Same surface as training/caiman_asr_train/keywords/trie.py:117-203 (`Keywords(vocab)`, `.init()`, `.step`,
`.steps`): every keyword is a path in a prefix tree, walking an edge pays that edge's weight (the sum of the
per-symbol weights of all keywords sharing the edge), a thread that falls off the tree pays back whatever it
had accumulated and not yet committed, and reaching the end of a keyword commits `weight * len(keyword)`.

The tree is stored as three parallel lists (children, edge weight into the node, committed value) rather than
the reference's instruction objects.
"""
from typing import Dict, Generic, Hashable, Iterable, List, Optional, Sequence, Tuple, TypeVar

T = TypeVar("T", bound=Hashable)
State = Dict[int, float]  # node index -> score accumulated along the path but not yet committed


class Keywords(Generic[T]):
    State = State

    def __init__(self, vocab: Iterable[Tuple[Sequence[T], float]]):
        vocab = [(tuple(word), float(w)) for word, w in vocab]
        words = [w for w, _ in vocab]
        assert len(set(words)) == len(words), "Duplicate keywords"
        self.children: List[Dict[T, int]] = [{}]
        self.edge_weight: List[float] = [0.0]       # weight of the edge that enters node i
        self.committed: List[Optional[float]] = [None]  # value locked in when a keyword ends at node i
        for word, w in vocab:
            assert len(word) > 0, "Empty keyword"
            node = 0
            for sym in word:
                nxt = self.children[node].get(sym)
                if nxt is None:
                    nxt = len(self.children)
                    self.children[node][sym] = nxt
                    self.children.append({})
                    self.edge_weight.append(0.0)
                    self.committed.append(None)
                self.edge_weight[nxt] += w
                node = nxt
            assert self.committed[node] is None, "Duplicate keyword"
            self.committed[node] = w * len(word)

    @classmethod
    def init(cls) -> State:
        return {0: 0.0}

    def step(self, tok: T, state: State) -> Tuple[float, State]:
        """Advance every live thread by `tok` -> (score delta, new state)."""
        if isinstance(tok, str):
            assert len(tok) == 1, f"Did you mean to call steps() with: {tok}?"
        assert 0 in state, "All states should have the initial inst"
        nxt_state: State = {0: 0.0}  # a keyword may start at any symbol
        delta = 0.0
        for node, acc in state.items():
            if self.committed[node] is not None:
                acc -= self.committed[node]
            child = self.children[node].get(tok)
            if child is None:
                delta -= acc                       # the thread dies: uncommitted score is returned
            else:
                nxt_state[child] = acc + self.edge_weight[child]
                delta += self.edge_weight[child]
        return delta, nxt_state

    def steps(self, toks: Iterable[T], state: State) -> Tuple[float, State]:
        total = 0.0
        for tok in toks:
            d, state = self.step(tok, state)
            total += d
        return total, state

    def __bool__(self):
        return len(self.children) > 1
