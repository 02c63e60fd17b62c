"""CPU: evaluation tail (WER, accuracies, 13a evaluation tokenizer) against the reference's own vectors
(reference test/unit/test_metric.py:40-64)."""
from joeys2t_amd.metrics import EvaluationTokenizer, edit_distance, sequence_accuracy, token_accuracy, tokenize_13a, wer


def test_wer_13a_reference_vectors():
    tok = EvaluationTokenizer(lowercase=True, tokenize="13a", no_punc=True)
    assert wer(["This is a test."], ["this is a Tezt!"], tokenizer=tok) == 25.0  # 1/4
    tok.no_punc = False
    assert wer(["This is a test."], ["this is a Tezt!"], tokenizer=tok) == 40.0  # 2/5


def test_token_accuracy_reference_vectors():
    assert token_accuracy(["tests"], ["tezt"], list) == 60.0
    assert token_accuracy(["test"], ["tezts"], list) == 75.0
    assert sequence_accuracy(["a b", "c"], ["a b", "d"]) == 50.0
    assert sequence_accuracy([], []) == 0.0


def test_13a_tokenizer_rules_and_edit_distance():
    assert tokenize_13a('Hello, world! 3.5 or 1,000 (x) a-b 5-6 &amp; &quot;q&quot;') == 'Hello , world ! 3.5 or 1,000 ( x ) a-b 5 - 6 & " q "'
    assert edit_distance("kitten", "sitting") == 3 and edit_distance([], [1, 2]) == 2 and edit_distance("abc", "abc") == 0
    assert wer([], []) == 0.0 and wer(["a"], [""]) == 0.0
