"""WER, CPU.  Known answers: the reference's own tables (training/tests/evaluate/test_metrics.py:19-27,
53-96) and its pure-Python reference distance (:30-50) as a differential check over random token lists."""
import random

import pytest

from caiman_asr_amd.evaluate.metrics import ErrorRate, decide_and_split, get_error_rate, levenshtein, word_error_rate


def _dp(a, b):
    prev = list(range(len(a) + 1))
    for i, y in enumerate(b, 1):
        cur = [i]
        for j, x in enumerate(a, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[len(a)]


def test_levenshtein_known_answers_and_random():
    assert levenshtein("Mary had a little lamb".split(), "Mary had a little lamb".split()) == 0
    assert levenshtein("I have a pet dog".split(), "You have a pet cat".split()) == 2
    assert levenshtein("one two three".split(), "two three".split()) == 1
    assert levenshtein("one two three".split(), "two three one".split()) == 2
    assert levenshtein([], []) == 0 and levenshtein([], list("abc")) == 3 and levenshtein(list("abc"), []) == 3
    rng = random.Random(0)
    for _ in range(200):
        a = [rng.choice("abcd") for _ in range(rng.randrange(0, 30))]
        b = [rng.choice("abcd") for _ in range(rng.randrange(0, 30))]
        assert levenshtein(a, b) == _dp(a, b) == levenshtein(b, a)
    a = [rng.choice(["a", "b"]) for _ in range(700)]
    b = [rng.choice(["a", "b"]) for _ in range(700)]
    assert levenshtein(a, b) == _dp(a, b)


@pytest.mark.parametrize("hyps, refs, expected", [
    (["hello world"], ["hello world"], (0.0, 0, 2)),
    (["hello world"], ["hi everyone"], (1.0, 2, 2)),
    ([], [], (float("inf"), 0, 0)),
    (["hello world"], ["hello new world"], (1 / 3, 1, 3)),
    (["good morning earth"], ["good morning mars good morning"], (0.6, 3, 5)),
])
def test_word_error_rate_table(hyps, refs, expected):
    assert word_error_rate(hyps, refs, ErrorRate.WORD, standardize=True) == expected


def test_wer_modes_and_errors():
    with pytest.raises(ValueError):
        word_error_rate(["hello"], ["hello", "mars"])
    assert word_error_rate(["One two four一二四"], ["One two three一二三"], ErrorRate.MIXTURE, standardize=False) == (1 / 3, 2, 6)
    assert word_error_rate(["abd"], ["abc"], ErrorRate.CHAR, standardize=False) == (1 / 3, 1, 3)
    assert word_error_rate(["Hello, <noise> WORLD!"], ["hello world"]) == (0.0, 0, 2)
    assert decide_and_split("ab c", ErrorRate.CHAR) == ["a", "b", "c"]
    assert get_error_rate({"error_rate": "CER"}) is ErrorRate.CHAR
    with pytest.raises(ValueError):
        get_error_rate({"error_rate": "bleu"})
