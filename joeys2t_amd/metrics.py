"""Evaluation tail for reporting decode parity: corpus WER, token and sequence accuracy
(reference metrics.py:64-131) and the evaluation-time tokenizer (reference tokenizers.py:511-553).

The reference delegates the edit distance to the `editdistance` package and the "13a" tokenisation to sacreBLEU
(`sacrebleu>=2.0.0`, requirements.txt); neither is present here, so both are restated from their published algorithms:
Levenshtein distance on token lists, and mteval-v13a tokenisation (the four regular-expression passes of sacreBLEU's
`Tokenizer13a` after its `&quot; &amp; &lt; &gt;` unescaping).  Pinned by the reference's own vectors
(test/unit/test_metric.py:40-64): WER 25.0 / 40.0 and token accuracy 60.0 / 75.0."""
import re
import unicodedata
from typing import Callable, List, Sequence


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


def wer(hypotheses: List[str], references: List[str], tokenizer: Callable = None) -> float:
    """Corpus-level word error rate: sum of edit distances / sum of reference lengths * 100 (metrics.py:110-131)."""
    tok = tokenizer or (lambda s: s.split())
    n_err = n_tok = 0
    for hyp, ref in zip(hypotheses, references):
        h, r = tok(hyp), tok(ref)
        n_err += edit_distance(h, r)
        n_tok += len(r)
    return (n_err / n_tok * 100) if n_tok != 0 else 0.0


def token_accuracy(hypotheses: List[str], references: List[str], tokenizer: Callable = None) -> float:
    """Correct tokens (same position) / hypothesis tokens * 100 (metrics.py:64-87)."""
    tok = tokenizer or (lambda s: s)
    assert len(hypotheses) == len(references)
    correct = total = 0
    for hyp, ref in zip(hypotheses, references):
        hyp, ref = tok(hyp), tok(ref)
        total += len(hyp)
        correct += sum(1 for h, r in zip(hyp, ref) if h == r)
    return (correct / total) * 100 if total > 0 else 0.0


def sequence_accuracy(hypotheses: List[str], references: List[str]) -> float:
    """Exactly matching hypotheses / hypotheses * 100 (metrics.py:90-107)."""
    assert len(hypotheses) == len(references)
    return (sum(1 for h, r in zip(hypotheses, references) if h == r) / len(hypotheses)) * 100 if hypotheses else 0.0


_RE_13A = [(re.compile(r"([\{-\~\[-\` -\&\(-\+\:-\@\/])"), r" \1 "), (re.compile(r"([^0-9])([\.,])"), r"\1 \2 "),
           (re.compile(r"([\.,])([^0-9])"), r" \1 \2"), (re.compile(r"([0-9])(-)"), r"\1 \2 ")]


def tokenize_13a(line: str) -> str:
    """mteval-v13a tokenisation as sacreBLEU applies it (language-independent part + the western-language passes)."""
    line = line.replace("<skipped>", "").replace("-\n", "").replace("\n", " ")
    if "&" in line:
        line = line.replace("&quot;", '"').replace("&amp;", "&").replace("&lt;", "<").replace("&gt;", ">")
    line = f" {line} "
    for pattern, repl in _RE_13A:
        line = pattern.sub(repl, line)
    return " ".join(line.split())


def remove_punctuation(s: str, space: str = " ") -> str:
    """Drop tokens that consist of Unicode punctuation only (helpers.py:445-456)."""
    return space.join(t for t in s.split(space) if not all(unicodedata.category(c)[0] == "P" for c in t))


class EvaluationTokenizer:
    """Evaluation-time tokenizer (tokenizers.py:511-553): sacreBLEU-style tokenisation ("13a" or "none"), then optional
    lower-casing and punctuation removal."""
    SPACE = " "

    def __init__(self, lowercase: bool = False, tokenize: str = "13a", **kwargs):
        if tokenize not in ("none", "13a"):
            raise NotImplementedError(f"tokenize={tokenize!r}: only 'none' and '13a' are restated here")
        self.lowercase, self.tokenize = lowercase, tokenize
        self.no_punc = kwargs.get("no_punc", False)

    def __call__(self, raw_input: str, is_train: bool = False) -> List[str]:
        tokenized = tokenize_13a(raw_input) if self.tokenize == "13a" else " ".join(raw_input.split())
        if self.lowercase:
            tokenized = tokenized.lower()
        if self.no_punc:
            tokenized = remove_punctuation(tokenized, space=self.SPACE)
        return tokenized.split()
