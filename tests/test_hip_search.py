"""GPU: greedy and beam search of the HIP path against ids/scores captured from the real reference: beam indices
bit-exact, scores within 1e-4 (north_star).  Also the fused beam-step kernel against plain torch on random input."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from golden_cfg import FIXTURES
from test_hip_model import batch_kwargs, build

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(FIXTURES))
def test_greedy_and_beam_match_reference(device, name):
    from joeys2t_amd.search import search
    model, g = build(name, device)
    model.eval()
    b = batch_kwargs(g, device)
    ids, scores, _ = search(model, b, max_output_length=12, beam_size=1, beam_alpha=-1, return_prob="hyp")
    assert np.array_equal(ids, g["greedy_ids"])
    np.testing.assert_allclose(scores, g["greedy_scores"], rtol=1e-4, atol=1e-4)
    k = int(g["beam_size"])
    ids, scores, _ = search(model, b, max_output_length=12, beam_size=k, beam_alpha=float(g["beam_alpha"]), n_best=k,
                            return_prob="hyp")
    assert np.array_equal(ids, g["beam_ids"])  # bit-exact beam indices
    np.testing.assert_allclose(scores, g["beam_scores"], rtol=1e-4, atol=1e-4)
    ids, scores, _ = search(model, b, max_output_length=-1, beam_size=k, beam_alpha=0.0, n_best=1, return_prob="hyp")
    assert np.array_equal(ids, g["beam_ids_a0"])
    np.testing.assert_allclose(scores, g["beam_scores_a0"], rtol=1e-4, atol=1e-4)


def test_predict_unsorts_and_decodes(device):
    from joeys2t_amd.prediction import predict
    model, g = build("model_pre", device)
    b = batch_kwargs(g, device)
    ids, sents, scores = predict(model, [b], beam_size=int(g["beam_size"]), beam_alpha=float(g["beam_alpha"]), n_best=1,
                                 max_output_length=12, return_prob="hyp")
    # the reference capture ran search() on the unsorted batch; predict() sorts by length, decodes, and un-sorts, so its
    # rows must equal the best hypothesis (row 0 of every n_best=k group) of that capture, in the original order
    k = int(g["beam_size"])
    ref = g["beam_ids"][::k]
    assert len(ids) == ref.shape[0] == len(sents)
    for row, r in zip(ids, ref):
        n = min(len(row), len(r))
        assert np.array_equal(row[:n], r[:n]) and np.all(row[n:] == 1) and np.all(r[n:] == 1)
    np.testing.assert_allclose(np.concatenate(scores), g["beam_scores"][::k, 0], rtol=1e-4, atol=1e-4)
    assert all(isinstance(s, list) and "<pad>" not in s for s in sents)


@pytest.mark.parametrize("beam,V,alpha", [(1, 37, 0.0), (5, 501, 1.0), (20, 5000, 1.0)])
def test_beam_step_kernel(device, beam, V, alpha):
    from joeys2t_amd import ops
    g = torch.Generator().manual_seed(beam)
    nb = 3
    logits = torch.randn(nb * beam, V, generator=g) * 3
    blp = torch.randn(nb, beam, generator=g)
    blp[0, 1:] = float("-inf")
    forbid = [1, 2, 3]
    lp = torch.log_softmax(logits, -1)
    lp[:, forbid] = float("-inf")
    lp = lp + blp.view(-1, 1)
    pen = ((5.0 + 3) / 6.0)**alpha if alpha > 0 else 0.0
    cur = lp / pen if alpha > 0 else lp
    ref_s, ref_i = cur.reshape(nb, beam * V).topk(beam, dim=-1)
    s, i, lse = ops.beam_step(logits.to(device), blp.view(-1).to(device), nb, beam, forbid, pen)
    assert torch.equal(i.cpu(), ref_i)
    torch.testing.assert_close(s.cpu(), ref_s, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(logits, -1), rtol=1e-5, atol=1e-5)


def test_wer_on_synthetic_set_equals_cpu_oracle(device):
    """north_star: WER on a held-out synthetic set equal to the CPU reference.  Two synthetic batches the fixtures do not
    contain are decoded (beam 3) by the HIP path and by the CPU oracle; hypotheses must be identical, hence the WER."""
    from golden_cfg import SPECIALS, oracle_cfg
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.metrics import wer
    from joeys2t_amd.prediction import predict
    from joeys2t_amd.vocabulary import Vocabulary
    from oracle import s2t_oracle as O
    model, g = build("model_pre", device)
    model.eval()
    from conftest import golden_sd
    sd = golden_sd(g)
    cfg = oracle_cfg(FIXTURES["model_pre"]["cfg"])
    vocab = Vocabulary.synthetic(20)
    hyp_hip, hyp_cpu, refs = [], [], []
    for seed in (501, 502):
        gen = torch.Generator().manual_seed(seed)
        B, T = 4, 41
        lengths = torch.tensor([41, 33, 29, 37])
        src = torch.randn(B, T, 8, generator=gen)
        for b in range(B):
            src[b, lengths[b]:] = 1.0
        ref_ids = [torch.randint(4, 20, (int(n), ), generator=gen).tolist() for n in torch.randint(3, 7, (B, ), generator=gen)]
        refs += [" ".join(vocab.array_to_sentence(r)) for r in ref_ids]
        batch = Batch(src=src, src_length=lengths, src_prompt_mask=None, trg=None, trg_length=None, trg_prompt_mask=None,
                      indices=torch.arange(B), device=device, pad_index=1, eos_index=3, is_train=False, task="S2T", n_gpu=1)
        _, sents, _ = predict(model, [batch], beam_size=3, beam_alpha=1.0, n_best=1, max_output_length=10)
        hyp_hip += [" ".join(s) for s in sents]
        enc, mask, _ = O.encoder_forward(sd, cfg, src, lengths)
        ids, _ = O.beam_search(sd, cfg, SPECIALS, enc, mask, beam_size=3, max_output_length=10, alpha=1.0, n_best=1)
        hyp_cpu += [" ".join(vocab.array_to_sentence(row.tolist(), cut_at_eos=True)) for row in ids]
    assert hyp_hip == hyp_cpu
    assert wer(hyp_hip, refs) == wer(hyp_cpu, refs)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_incremental_decoding_equals_full_prefix_pass(device, dtype):
    """KV-cached decoding against the reference-style full-prefix decoder pass on a wider model (d = 256, head size 128,
    ragged source lengths, beam 4): identical hypotheses in fp32; in bf16 the two orders of rounding may flip a near-tie,
    so scores are compared instead."""
    import copy
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.model import build_model
    from joeys2t_amd.search import search
    from joeys2t_amd.vocabulary import Vocabulary
    cfg = copy.deepcopy(FIXTURES["model_pre"]["cfg"])
    cfg["encoder"].update(hidden_size=256, ff_size=512, num_heads=2, conv_channels=64)
    cfg["decoder"].update(hidden_size=256, ff_size=512, num_heads=2)
    cfg["decoder"]["embeddings"]["embedding_dim"] = 256
    torch.manual_seed(3)
    model = build_model(cfg, None, Vocabulary.synthetic(50))
    model.finalize(device, dtype).eval()
    gen = torch.Generator().manual_seed(9)
    B, T = 5, 64
    lengths = torch.tensor([64, 51, 40, 64, 33])
    src = torch.randn(B, T, 8, generator=gen)
    for b in range(B):
        src[b, lengths[b]:] = 1.0
    batch = Batch(src=src, src_length=lengths, src_prompt_mask=None, trg=None, trg_length=None, trg_prompt_mask=None,
                  indices=torch.arange(B), device=device, pad_index=1, eos_index=3, is_train=False, task="S2T", n_gpu=1)
    for beam in (1, 4):
        a_ids, a_sc, _ = search(model, batch, max_output_length=14, beam_size=beam, beam_alpha=1.0, n_best=1, return_prob="hyp")
        b_ids, b_sc, _ = search(model, batch, max_output_length=14, beam_size=beam, beam_alpha=1.0, n_best=1, return_prob="hyp",
                                incremental=False)
        if dtype == torch.float32:
            assert np.array_equal(a_ids, b_ids)
            np.testing.assert_allclose(a_sc, b_sc, rtol=1e-4, atol=1e-4)
        elif beam > 1:
            np.testing.assert_allclose(a_sc, b_sc, rtol=5e-2, atol=5e-2)


def test_ctc_collapse_kernel_matches_oracle(device):
    """js2t_ctc_collapse against the oracle's best-path rule on random label paths: repeats, blanks, ragged lengths, an
    empty utterance, more than 64 frames (several ballot rounds)."""
    from joeys2t_amd import ops
    from oracle import s2t_oracle as O
    rs = np.random.RandomState(0)
    B, T, V, blank, pad = 7, 150, 6, 2, 1
    best = rs.randint(0, V, size=(B, T))
    best[1, 10:40] = blank
    best[2, :] = 4
    lens = np.array([150, 97, 150, 0, 64, 65, 1])
    logits = np.full((B, T, V), -5.0, dtype=np.float32)
    np.put_along_axis(logits, best[..., None], 3.0, axis=2)
    ref_ids, ref_len = O.ctc_best_path(logits, lens, blank, pad)
    ids, n = ops.ctc_collapse(torch.from_numpy(best).to(device), torch.from_numpy(lens).to(device), blank, pad)
    assert np.array_equal(n.cpu().numpy(), ref_len)
    assert np.array_equal(ids.cpu().numpy(), ref_ids)


@pytest.mark.parametrize("name", list(FIXTURES))
def test_ctc_greedy_matches_oracle(device, name):
    """search.ctc_greedy: encoder + CTC output layer + best path on the HIP path == the oracle's encoder and
    projection followed by the oracle's collapse rule (label ids bit-exact)."""
    from golden_cfg import oracle_cfg
    from joeys2t_amd.search import ctc_greedy
    from oracle import s2t_oracle as O
    model, g = build(name, device)
    model.eval()
    b = batch_kwargs(g, device)
    ids, n = ctc_greedy(model, b)
    sd = {k: v for k, v in model.state_dict().items()}
    sd = {k: v.detach().cpu().float() for k, v in sd.items()}
    with torch.no_grad():
        x, mask, lengths = O.encoder_forward(sd, oracle_cfg(FIXTURES[name]["cfg"]), torch.from_numpy(g["src"]), torch.from_numpy(g["src_length"]))
        logits = x @ sd["decoder.ctc_output_layer.weight"].t()
    ref_ids, ref_len = O.ctc_best_path(logits.numpy(), mask.squeeze(1).sum(1).numpy(), model.bos_index, model.pad_index)
    assert np.array_equal(n, ref_len)
    assert np.array_equal(ids, ref_ids)


@pytest.mark.parametrize("name", list(FIXTURES))
def test_sync_free_beam_search_files_the_same_hypotheses(device, name):
    """The default beam search never looks at the device inside a step (every utterance stays in the batch, the steps' tensors
    are kept, finished hypotheses are filed after the loop); with sync_free=False it files them step by step and drops ended
    utterances as the reference does (search.py:683-717,757-781).  Same ids, same scores, for n_best up to the beam size."""
    from joeys2t_amd.search import search
    model, g = build(name, device)
    model.eval()
    b = batch_kwargs(g, device)
    for k, nb, alpha, L in ((3, 3, 1.0, 12), (5, 2, 0.6, 9), (2, 1, 0.0, 30)):
        a = search(model, b, max_output_length=L, beam_size=k, beam_alpha=alpha, n_best=nb, return_prob="hyp")
        c = search(model, b, max_output_length=L, beam_size=k, beam_alpha=alpha, n_best=nb, return_prob="hyp", sync_free=False)
        assert np.array_equal(a[0], c[0]), (k, nb)
        np.testing.assert_array_equal(a[1], c[1])


def test_wer_on_64_held_out_utterances_at_mustc_width_equals_cpu_oracle(device):
    """north_star: "WER on a held-out synthetic set equal to the CPU reference" - beyond the 16-wide golden model (the test above):
    configs/mustc_st.yaml WIDTH (d 512, ff 2048, 8 heads of 64, V 5000, xavier_normal with the DeepNet residual scale of the 12 + 6
    stack; depth cut to 2 + 1 so that the oracle's full-prefix beam search finishes in seconds), 64 utterances of 120 - 260 frames in
    four ragged batches, beam 5, alpha 1 (configs/mustc_st.yaml:57-58).  The HIP path (fp32, KV-cached search through `predict`)
    must return the oracle's hypotheses token for token - hence its WER against any references."""
    from golden_cfg import SPECIALS
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.metrics import wer
    from joeys2t_amd.prediction import predict
    from joeys2t_amd.vocabulary import Vocabulary
    from oracle import s2t_oracle as O
    from test_hip_config_width import MUSTC_ALPHA, make_model, width_cfg
    import copy
    V = 5000
    cfg = width_cfg(8, 2, 1, "xavier_normal")
    torch.manual_seed(23)
    base = make_model(cfg, V, None, None, None, 0.1)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    for k in sd:  # an untrained decoder all but ignores the audio (every utterance decodes to one of three strings): turn the
        if "src_trg_att.output_layer.weight" in k:  # cross-attention's contribution up until the hypotheses follow the input
            sd[k] *= 30.0
    model = make_model(cfg, V, sd, device, torch.float32, 0.1, alpha=MUSTC_ALPHA)
    model.eval()
    ocfg = copy.deepcopy(cfg)
    ocfg["encoder"]["alpha"], ocfg["decoder"]["alpha"] = MUSTC_ALPHA
    vocab = Vocabulary.synthetic(V)
    hyp_hip, hyp_cpu, refs = [], [], []
    for seed in (601, 602, 603, 604):
        gen = torch.Generator().manual_seed(seed)
        B = 16
        lengths = torch.randint(120, 261, (B, ), generator=gen)
        T = int(lengths.max())
        src = torch.randn(B, T, 80, generator=gen)
        for b in range(B):
            src[b, lengths[b]:] = 1.0
        ref_ids = [torch.randint(4, V, (int(n), ), generator=gen).tolist() for n in torch.randint(3, 9, (B, ), generator=gen)]
        refs += [" ".join(vocab.array_to_sentence(r)) for r in ref_ids]
        batch = Batch(src=src, src_length=lengths, src_prompt_mask=None, trg=None, trg_length=None, trg_prompt_mask=None,
                      indices=torch.arange(B), device=device, pad_index=1, eos_index=3, is_train=False, task="S2T", n_gpu=1)
        _, sents, _ = predict(model, [batch], beam_size=5, beam_alpha=1.0, n_best=1, max_output_length=10)
        hyp_hip += [" ".join(s) for s in sents]
        enc, mask, _ = O.encoder_forward(sd, ocfg, src, lengths)
        ids, _ = O.beam_search(sd, ocfg, SPECIALS, enc, mask, beam_size=5, max_output_length=10, alpha=1.0, n_best=1)
        hyp_cpu += [" ".join(vocab.array_to_sentence(row.tolist(), cut_at_eos=True)) for row in ids]
    assert len(hyp_hip) == 64 and len(set(hyp_hip)) > 8  # 64 utterances that do not all decode to one string (random weights: 13 distinct)
    assert hyp_hip == hyp_cpu
    assert wer(hyp_hip, refs) == wer(hyp_cpu, refs)
