"""GPU: the decoding options of the reference on the HIP path - forced-decoding prompts, repetition penalty (target and
source side), n-gram blocking, generate_unk, attention export - against captures of the reference's greedy / beam_search
(tests/golden/search_options.npz, oracle/make_golden.py:golden_search_options) AND the constants its own unit tests
hard-code (test/unit/test_search.py:101-500): ids bit-exact, scores / attention within 1e-4."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden

pytestmark = pytest.mark.gpu


def build(g, bs, device):
    """the decoder-only model of TestSearchTransformer._build (test_search.py:61-99) with the captured weights"""
    from joeys2t_amd.decoders import TransformerDecoder
    from joeys2t_amd.embeddings import Embeddings
    from joeys2t_amd.model import Model
    from joeys2t_amd.vocabulary import Vocabulary
    special = SimpleNamespace(unk_token="<unk>", pad_token="<pad>", bos_token="<s>", eos_token="</s>", sep_token="<sep>", unk_id=0,
                              pad_id=1, bos_id=2, eos_id=3, sep_id=4, lang_tags=["<de>", "<en>"])
    vocab = Vocabulary(["word"], special)
    assert len(vocab) == 8
    emb = Embeddings(embedding_dim=12, vocab_size=8, padding_idx=1)
    dec = TransformerDecoder(num_layers=3, num_heads=4, hidden_size=12, ff_size=24, dropout=0.0, emb_dropout=0.0, vocab_size=8,
                             layer_norm="pre")
    model = Model(encoder=None, decoder=dec, src_embed=emb, trg_embed=emb, src_vocab=vocab, trg_vocab=vocab, task="MT")
    assert model.specials == [0, 1, 2, 3, 4] and model.lang_tags == [5, 6] and model.sep_index == 4
    missing, unexpected = model.load_state_dict(golden_sd(g, f"bs{bs}.sd."), strict=False)
    assert not unexpected and all(k.endswith("pe.pe") or k.startswith("src_embed") for k in missing), (missing, unexpected)
    model.finalize(device, torch.float32).eval()
    enc = torch.from_numpy(g[f"bs{bs}.encoder_output"]).to(device)
    return model, enc, torch.ones(bs, 1, 4, dtype=torch.bool, device=device)


def cmp(g, case, res):
    ids, scores, att = res
    assert np.array_equal(ids.numpy(), g[f"{case}.ids"]), (case, ids, g[f"{case}.ids"])
    assert np.array_equal(ids.numpy(), g[f"{case}.exp_ids"])
    if f"{case}.scores" in g and scores is not None:
        np.testing.assert_allclose(scores.numpy(), g[f"{case}.scores"], rtol=1e-4, atol=1e-4, err_msg=case)
        np.testing.assert_allclose(scores.numpy(), g[f"{case}.exp_scores"], rtol=1e-4, atol=1e-4, err_msg=case)
    if f"{case}.att" in g:
        np.testing.assert_allclose(att.numpy(), g[f"{case}.att"], rtol=1e-4, atol=1e-4, err_msg=case)
        np.testing.assert_allclose(att.numpy(), g[f"{case}.exp_att"], rtol=1e-4, atol=1e-4, err_msg=case)
    else:
        assert att is None


@pytest.mark.parametrize("incremental", [True, False])
def test_search_options_batch2(device, incremental):
    """incremental: KV-cached decoding where the option allows it (penalties / blocking); prompts and attention export run
    the full-prefix pass either way."""
    from joeys2t_amd.search import beam_search, transformer_greedy as greedy
    g = load_golden("search_options")
    model, enc, mask = build(g, 2, device)
    kw = dict(src_mask=mask, model=model, encoder_output=enc, encoder_hidden=None, incremental=incremental)
    prompt, pmask = torch.from_numpy(g["prompt"]), torch.from_numpy(g["prompt_mask"])
    cmp(g, "greedy", greedy(max_output_length=3, return_prob="hyp", **kw))
    cmp(g, "greedy_prompt", greedy(max_output_length=7, return_prob="hyp", return_attention=True, decoder_prompt=prompt,
                                   trg_prompt_mask=pmask, **kw))
    cmp(g, "beam1", beam_search(beam_size=1, max_output_length=3, alpha=0.0, n_best=1, return_prob="hyp", **kw))
    cmp(g, "beam7", beam_search(beam_size=7, max_output_length=3, alpha=1.0, n_best=5, return_prob="hyp", **kw))
    cmp(g, "beam7_prompt", beam_search(beam_size=7, max_output_length=10, alpha=1.0, n_best=5, return_prob="hyp", decoder_prompt=prompt,
                                       trg_prompt_mask=pmask, **kw))
    cmp(g, "beam7_penalty", beam_search(beam_size=7, max_output_length=3, alpha=1.0, n_best=5, return_prob="hyp",
                                        encoder_input=torch.from_numpy(g["beam7_penalty.src_tokens"]), repetition_penalty=1.5, **kw))
    cmp(g, "greedy_ngram", greedy(max_output_length=7, return_prob="hyp", encoder_input=None, no_repeat_ngram_size=3, **kw))
    cmp(g, "beam3_ngram", beam_search(beam_size=3, max_output_length=7, alpha=1.0, n_best=3, return_prob="hyp", encoder_input=None,
                                      no_repeat_ngram_size=3, **kw))


def test_search_options_batch3_penalties_and_attention(device):
    from joeys2t_amd.search import transformer_greedy as greedy
    g = load_golden("search_options")
    model, enc, mask = build(g, 3, device)
    kw = dict(model=model, encoder_output=enc, encoder_hidden=None)
    cmp(g, "greedy_nounk", greedy(src_mask=mask, max_output_length=3, generate_unk=False, **kw))
    cmp(g, "greedy_nounk_penalty", greedy(src_mask=mask, max_output_length=3, generate_unk=False, encoder_input=None,
                                          repetition_penalty=1.5, **kw))
    src = torch.from_numpy(g["greedy_src_penalty.src_tokens"])
    cmp(g, "greedy_src_penalty", greedy(src_mask=(src != 1).unsqueeze(1).to(device), max_output_length=3, generate_unk=False,
                                        encoder_input=src, repetition_penalty=1.5, return_attention=True, **kw))


def test_rep_penalty_and_logp_set_kernels(device):
    """js2t_rep_penalty against the reference's gather / where / scatter chain (search.py:987-993) incl. repeated ids and
    positive scores; js2t_logp_set; js2t_beam_step_logp == js2t_beam_step on pre-normalised rows."""
    from joeys2t_amd import ops
    gen = torch.Generator().manual_seed(3)
    rows, V, L = 37, 1000, 50
    lp = torch.randn(rows, V, generator=gen)  # both signs
    toks = torch.randint(0, V, (rows, L), generator=gen)
    toks[:, 10:20] = toks[:, :10]  # repeats: penalised once
    ref = lp.clone()
    sc = torch.gather(ref, 1, toks)
    ref.scatter_(1, toks, torch.where(sc < 0, sc * 1.7, sc / 1.7))
    got = ops.rep_penalty(lp.to(device).clone(), toks.to(device), 1.7)
    assert torch.equal(got.cpu(), ref)
    r, c = torch.randint(0, rows, (64, ), generator=gen), torch.randint(0, V, (64, ), generator=gen)
    ref[r, c] = float("-inf")
    got = ops.logp_set(got, r.tolist(), c.tolist(), float("-inf"))
    assert torch.equal(got.cpu(), ref)
    for beam in (5, 20):
        logits = torch.randn(3 * beam, V, generator=gen) * 2
        blp = torch.randn(3 * beam, generator=gen)
        a = ops.beam_step(logits.to(device), blp.to(device), 3, beam, [1, 2], 1.5)
        b = ops.beam_step(torch.log_softmax(logits, -1).to(device).contiguous(), blp.to(device), 3, beam, [1, 2], 1.5, normalized=True)
        assert torch.equal(a[1], b[1])
        torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=1e-5)
