"""Word error rate for reporting decode parity (reference metrics.py:110-131: corpus WER = sum of edit distances /
sum of reference lengths * 100).  The reference delegates the distance to the `editdistance` package; here it is a
plain Levenshtein on token lists (host, integer DP)."""
from typing import List, Sequence


def edit_distance(a: Sequence, b: Sequence) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def wer(hypotheses: List[str], references: List[str], tokenizer=None) -> float:
    tok = tokenizer or (lambda s: s.split())
    n_err = n_tok = 0
    for hyp, ref in zip(hypotheses, references):
        h, r = tok(hyp), tok(ref)
        n_err += edit_distance(h, r)
        n_tok += len(r)
    return (n_err / n_tok * 100) if n_tok != 0 else 0.0


def token_accuracy(hypotheses: List[List[str]], references: List[List[str]]) -> float:
    correct = total = 0
    for hyp, ref in zip(hypotheses, references):
        total += len(hyp)
        correct += sum(1 for h, r in zip(hyp, ref) if h == r)
    return (correct / total) * 100 if total > 0 else 0.0
