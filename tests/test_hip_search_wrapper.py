"""GPU: the options `search()` derives from the batch (reference search.py:866-873) and `predict` forwards (prediction.py:205-218):
`encoder_input = batch.src` for the source-side repetition penalty / n-gram block, `decoder_prompt` / `trg_prompt_mask` from a
prompted batch - through the repo's own `search()` / `predict()`, against the reference's `search()` on the same model and batches
(oracle/make_golden.py:golden_search_wrapper -> tests/golden/search_wrapper.npz; ids bit-exact, scores 1e-4)."""
import copy
import json
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from golden_cfg import mt_cfg

pytestmark = pytest.mark.gpu
CASES = json.loads((Path(__file__).resolve().parent / "golden" / "search_wrapper_cases.json").read_text())


def sep_vocab(size):
    from joeys2t_amd.vocabulary import Vocabulary
    cfg = SimpleNamespace(unk_token="<unk>", pad_token="<pad>", bos_token="<s>", eos_token="</s>", sep_token="<sep>", unk_id=0,
                          pad_id=1, bos_id=2, eos_id=3, sep_id=4, lang_tags=[])
    return Vocabulary([f"tok{i}" for i in range(size - 5)], cfg)


@pytest.fixture(scope="module")
def setup(device):
    from joeys2t_amd.model import build_model
    g = load_golden("search_wrapper")
    model = build_model(copy.deepcopy(mt_cfg()), sep_vocab(30), sep_vocab(30))
    assert model.sep_index == 4
    model.load_state_dict(golden_sd(g), strict=True)
    model.finalize(device, torch.float32).eval()
    return model, g


def make_batch(g, device, prompted):
    from joeys2t_amd.batch import Batch
    B = g["src"].shape[0]
    trg = trg_len = pm = None
    if prompted:
        trg, trg_len, pm = (torch.from_numpy(g[k]) for k in ("prompt", "prompt_length", "prompt_mask"))
    return Batch(src=torch.from_numpy(g["src"]), src_length=torch.from_numpy(g["src_length"]), src_prompt_mask=None, trg=trg,
                 trg_length=trg_len, trg_prompt_mask=pm, indices=torch.arange(B), device=device, pad_index=1, eos_index=3,
                 is_train=False, task="MT")


@pytest.mark.parametrize("case", sorted(CASES))
def test_search_matches_reference(setup, device, case):
    from joeys2t_amd.search import search
    model, g = setup
    kw = dict(CASES[case])
    prompted = kw.pop("prompted")
    ids, scores, _ = search(model, make_batch(g, device, prompted), max_output_length=14, return_prob="hyp", generate_unk=False, **kw)
    assert np.array_equal(ids, g[f"{case}.ids"]), (case, ids, g[f"{case}.ids"])
    np.testing.assert_allclose(scores, g[f"{case}.scores"], rtol=1e-4, atol=1e-4)


def test_options_reach_search_through_predict(setup, device):
    """predict() hands repetition_penalty / no_repeat_ngram_size on; search() adds the source tokens itself."""
    from joeys2t_amd.prediction import predict
    model, g = setup
    for case in ("rep_greedy", "ngram_beam", "both_beam"):
        kw = dict(CASES[case])
        kw.pop("prompted")
        n_best = kw.get("n_best", 1)
        ids, sentences, scores = predict(model, [make_batch(g, device, False)], max_output_length=14, return_prob="hyp",
                                         generate_unk=False, **kw)
        # the batch was sorted by source length inside predict and un-sorted again: compare in the original order
        order = np.argsort(-g["src_length"], kind="stable")
        ref = np.empty_like(g[f"{case}.ids"])
        # the reference capture ran on the UN-sorted batch; rows are independent, so original order == capture order
        ref[:] = g[f"{case}.ids"]
        width = max(len(r) for r in ids)
        got = np.full((len(ids), width), 1, dtype=np.int64)
        for r, row in enumerate(ids):
            got[r, :len(row)] = row
        assert got.shape[0] == ref.shape[0] == g["src"].shape[0] * n_best
        w = min(got.shape[1], ref.shape[1])
        assert np.array_equal(got[:, :w], ref[:, :w]) and (got[:, w:] == 1).all() and (ref[:, w:] == 1).all(), case
        assert len(sentences) == got.shape[0] and order.shape[0] == g["src"].shape[0]


def test_return_attention_through_predict(setup, device):
    from joeys2t_amd.prediction import predict
    model, g = setup
    ids, _, _, att = predict(model, [make_batch(g, device, False)], max_output_length=6, return_attention=True)
    assert att is not None and len(att) == len(ids) and att[0].shape[-1] == g["src"].shape[1]


def test_speech_batch_with_source_side_option_says_why(device):
    from joeys2t_amd.search import search
    from test_hip_model import batch_kwargs, build  # the S2T fixture helpers
    model, g = build("model_pre", device)
    b = batch_kwargs(g, device)
    with pytest.raises(ValueError, match="source TOKENS"):
        search(model, b, max_output_length=4, beam_size=1, beam_alpha=-1, repetition_penalty=1.5)


def test_long_max_output_length_is_linear_in_memory(setup, device):
    """The sync-free beam loop keeps back-pointers [L, B, k], not hypotheses [L, B*k, L]: a 3000-step budget must cost MBs."""
    from joeys2t_amd.search import search
    model, g = setup
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ids, _, _ = search(model, make_batch(g, device, False), max_output_length=3000, beam_size=5, beam_alpha=1.0, n_best=1)
    assert torch.cuda.max_memory_allocated() - base < 512 << 20
    ref, _, _ = search(model, make_batch(g, device, False), max_output_length=3000, beam_size=5, beam_alpha=1.0, n_best=1,
                       sync_free=False)
    assert np.array_equal(ids, ref)
