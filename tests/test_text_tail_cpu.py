"""CPU: evaluation tail and loader-side pieces against captures of the reference (tests/golden/text_tail.json, written by
oracle/make_golden.py:golden_text_tail): post_process of the SentencePiece / subword-nmt / word / char tokenizers
(tokenizers.py:133-165,230-260,334-366) and the batches of TokenBatchSampler / SentenceBatchSampler (datasets.py:1164-1295)."""
import json
from types import SimpleNamespace

import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def g():
    return json.loads((GOLDEN / "text_tail.json").read_text(encoding="utf-8"))


def _vocab(g, tokens):
    from joeys2t_amd.vocabulary import Vocabulary
    return Vocabulary(tokens, SimpleNamespace(**g["specials"]))


def _check(tok, cases):
    for c in cases:
        got = tok.post_process(list(c["seq"]), generate_unk=c["generate_unk"], cut_at_sep=c["cut_at_sep"])
        assert got == c["out"], (c, got)


def test_sentencepiece_post_process(g):
    from joeys2t_amd.tokenizers import SentencePieceTokenizer
    sp = g["sentencepiece"]
    tok = SentencePieceTokenizer(level="bpe", normalize=sp["normalize"])  # no model file: pieces are glued back in Python
    tok.set_vocab(_vocab(g, sorted({p for ps in sp["pieces"] for p in ps})))
    _check(tok, sp["cases"])
    # without specials the pieces decode back to the (normalised) sentence they came from
    for sent, pieces in zip(sp["sentences"], sp["pieces"]):
        assert tok.post_process(list(pieces)) == tok.post_process(sent)


def test_subword_nmt_and_basic_post_process(g):
    from joeys2t_amd.tokenizers import BasicTokenizer, SubwordNMTTokenizer
    tok = SubwordNMTTokenizer(level="bpe", normalize=g["subword_nmt"]["normalize"], separator=g["subword_nmt"]["separator"])
    tok.set_vocab(_vocab(g, ["he@@", "llo", "wor@@", "ld", "a", "te@@", "st@@"]))
    _check(tok, g["subword_nmt"]["cases"])
    for level in ("word", "char"):
        bt = BasicTokenizer(level=level, normalize=g[level]["normalize"])
        bt.set_vocab(_vocab(g, sorted({t for ts in g[level]["tokens"] for t in ts})))
        _check(bt, g[level]["cases"])
        assert [bt(" ".join(ts) if level == "word" else "".join(ts).replace(bt.SPACE_ESCAPE, " ")) for ts in g[level]["tokens"]] == g[level]["tokens"]


class ToyDataset:
    def __init__(self, src_len, trg_len, drop):
        self.src_len, self.trg_len, self.drop = src_len, trg_len, set(drop)
        self.indices = list(range(len(src_len)))
        self.random_subset, self.seed, self.split = -1, 0, "train"

    def __len__(self):
        return len(self.src_len)

    def reset_indices(self):
        self.indices = list(range(len(self.src_len)))

    def __getitem__(self, idx):
        if idx in self.drop:
            return idx, None, None
        return idx, [0] * self.src_len[idx], [0] * self.trg_len[idx]


def test_batch_samplers_match_reference(g):
    from joeys2t_amd.datasets import SentenceBatchSampler, TokenBatchSampler
    from joeys2t_amd.helpers_for_ddp import RandomSubsetSampler
    sm = g["samplers"]
    for key, case in sm["cases"].items():
        cls = TokenBatchSampler if key.startswith("token") else SentenceBatchSampler
        ds = ToyDataset(sm["src_len"], sm["trg_len"], sm["drop"])
        base = RandomSubsetSampler(ds, shuffle=True, generator=torch.Generator().manual_seed(42))
        bs = cls(base, batch_size=case["batch_size"], drop_last=case["drop_last"], seed=42)
        epochs = [[list(b) for b in bs], [list(b) for b in bs]]
        assert epochs == case["epochs"], key
        if "len" in case:
            assert len(bs) == case["len"]
        else:
            with pytest.raises(NotImplementedError):
                len(bs)
    # token batches respect the rule: closed as soon as max(len + 1) * n >= batch_size
    for batch in sm["cases"]["token_0"]["epochs"][0][:-1]:
        toks = [max(sm["src_len"][i] + 1, sm["trg_len"][i] + 1) for i in batch]
        assert max(toks) * len(batch) >= 6000 and max(toks[:-1] or [0]) * (len(batch) - 1) < 6000


def test_data_parallel_wrapper_passes_through():
    """DataParallelWrapper (model.py:323-363 of the reference) around this package's DDP stand-in: attributes of the inner
    model are reachable, state_dict carries no `module.` prefixes, load_state_dict reaches the inner model."""
    import copy

    from golden_cfg import tiny_cfg
    from joeys2t_amd.helpers_for_ddp import FlatDDP
    from joeys2t_amd.model import DataParallelWrapper, build_model
    from joeys2t_amd.vocabulary import Vocabulary
    model = build_model(copy.deepcopy(tiny_cfg("pre")), None, Vocabulary.synthetic(20))
    wrapped = DataParallelWrapper(FlatDDP(model))
    assert wrapped.pad_index == model.pad_index and wrapped.bos_index == 2 and wrapped.decoder is model.decoder
    sd = wrapped.state_dict()
    assert list(sd.keys()) == list(model.state_dict().keys()) and not any(k.startswith("module.") for k in sd)
    sd2 = {k: v + 1.0 if v.is_floating_point() else v for k, v in sd.items()}
    wrapped.load_state_dict(sd2)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd2[k]), k
