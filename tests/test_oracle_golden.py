"""CPU: the oracle restatement (oracle/s2t_oracle.py) against golden vectors captured from the real reference
(oracle/make_golden.py) and against the constants the reference's own unit tests assert."""
import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from golden_cfg import FIXTURES, SPECIALS, oracle_cfg
from oracle import s2t_oracle as O

TOL = dict(rtol=1e-4, atol=1e-4)  # the reference's own tolerance (test/unit/*: rtol=atol=1e-4)


def _batch(g):
    return {"src": torch.from_numpy(g["src"]), "src_length": torch.from_numpy(g["src_length"]),
            "trg_input": torch.from_numpy(g["trg_input"]), "trg": torch.from_numpy(g["trg"]),
            "trg_length": torch.from_numpy(g["trg_length"]), "trg_mask": torch.from_numpy(g["trg_mask"])}


@pytest.mark.parametrize("name", list(FIXTURES))
def test_model_forward_loss(name):
    g = load_golden(name)
    cfg = oracle_cfg(FIXTURES[name]["cfg"])
    sd = golden_sd(g)
    b = _batch(g)
    enc, mask, lens = O.encoder_forward(sd, cfg, b["src"], b["src_length"])
    torch.testing.assert_close(enc, torch.from_numpy(g["enc_out"]), **TOL)
    assert np.array_equal(mask.numpy(), g["src_mask"])  # bit-exact length mask
    logits, hidden, att, ctc = O.decoder_forward(sd, cfg, b["trg_input"], enc, mask, b["trg_mask"], return_attention=True)
    torch.testing.assert_close(logits, torch.from_numpy(g["logits"]), **TOL)
    torch.testing.assert_close(hidden, torch.from_numpy(g["dec_hidden"]), **TOL)
    torch.testing.assert_close(att, torch.from_numpy(g["att"]), **TOL)
    torch.testing.assert_close(ctc, torch.from_numpy(g["ctc_logits"]), **TOL)
    total, xent, ctcl, ncor, _, _ = O.model_loss(sd, cfg, b, SPECIALS, 0.1, FIXTURES[name]["ctc_weight"])
    assert abs(total.item() - g["loss_total"]) <= 1e-4 * abs(g["loss_total"])
    assert abs(xent.item() - g["loss_xent"]) <= 1e-4 * abs(g["loss_xent"])
    assert abs(ctcl.item() - g["loss_ctc"]) <= 1e-4 * abs(g["loss_ctc"])
    assert int(ncor) == int(g["n_correct"])


@pytest.mark.parametrize("name", list(FIXTURES))
def test_batch_bookkeeping(name):
    g = load_golden(name)
    b = O.make_batch(torch.from_numpy(g["src"]), torch.from_numpy(g["src_length"]), torch.from_numpy(g["trg_full"]),
                     torch.from_numpy(g["trg_length_full"]), SPECIALS["pad"], SPECIALS["eos"])
    for k in ("trg_input", "trg", "trg_length", "trg_mask"):
        assert np.array_equal(b[k].numpy(), g[k]), k


@pytest.mark.parametrize("name", list(FIXTURES))
def test_search(name):
    g = load_golden(name)
    cfg = oracle_cfg(FIXTURES[name]["cfg"])
    sd = golden_sd(g)
    b = _batch(g)
    enc, mask, _ = O.encoder_forward(sd, cfg, b["src"], b["src_length"])
    ids, scores = O.greedy(sd, cfg, SPECIALS, enc, mask, 12, return_prob=True)
    assert np.array_equal(ids.numpy(), g["greedy_ids"])
    np.testing.assert_allclose(scores.numpy(), g["greedy_scores"], rtol=1e-4, atol=1e-4)
    k = int(g["beam_size"])
    ids, scores = O.beam_search(sd, cfg, SPECIALS, enc, mask, k, 12, float(g["beam_alpha"]), n_best=k)
    assert np.array_equal(ids.numpy(), g["beam_ids"])  # bit-exact beam indices
    np.testing.assert_allclose(scores.numpy(), g["beam_scores"], rtol=1e-4, atol=1e-4)
    max_len = int(max(g["src_length"]) * 1.5)  # search.py:863-864: un-subsampled frame count
    ids, scores = O.beam_search(sd, cfg, SPECIALS, enc, mask, k, max_len, 0.0, n_best=1)
    assert np.array_equal(ids.numpy(), g["beam_ids_a0"])
    np.testing.assert_allclose(scores.numpy(), g["beam_scores_a0"], rtol=1e-4, atol=1e-4)


def test_units_subsampler_and_lengths():
    g = load_golden("units")
    sd = golden_sd(g, "sub.sd.")
    y, yl = O.conv_subsample(sd, "", torch.from_numpy(g["sub_x"]), torch.tensor([9, 9]), [3, 3]) if False else (None, None)
    sd2 = {"s." + k: v for k, v in sd.items()}
    y, yl = O.conv_subsample(sd2, "s", torch.from_numpy(g["sub_x"]), torch.tensor([9, 9]), [3, 3])
    torch.testing.assert_close(y, torch.from_numpy(g["sub_y"]), **TOL)
    # constants hard-coded in the reference's test (test_transformer_encoder.py:134-136)
    np.testing.assert_allclose(y[0, 0].numpy(), g["sub_y_ref_row0"], rtol=1e-4, atol=1e-4)
    assert yl.tolist() == g["sub_len"].tolist() == [3, 3]
    lens = torch.from_numpy(g["len_in"])
    for ks in ([3, 3], [5, 5], [5], [3, 5, 3]):
        assert np.array_equal(O.subsample_lengths(lens, ks).numpy(), g["len_" + "_".join(map(str, ks))])


def test_units_losses():
    g = load_golden("units")
    predict, targets = torch.from_numpy(g["xent_predict"]), torch.from_numpy(g["xent_targets"])
    v1 = O.xent_loss(predict.log(), targets, 0, 0.4).item()
    v0 = O.xent_loss(predict.log(), targets, 0, 0.0).item()
    assert abs(v1 - g["xent_s04"]) < 1e-5 and abs(v0 - g["xent_s00"]) < 1e-5
    assert round(v1, 4) == float(g["xent_s04_ref"]) and round(v0, 4) == float(g["xent_s00_ref"])  # test_loss.py:52,95
    # smoothed target rows of test_loss.py:35-46
    st = O.smooth_targets(targets.view(-1), 5, 0, 0.4)
    np.testing.assert_allclose(st[0].numpy(), [0.0, 0.1333, 0.6, 0.1333, 0.1333], atol=1e-4)
    assert torch.all(st[3] == 0) and torch.all(st[5] == 0)
    logits = torch.from_numpy(g["xc_logits"]).requires_grad_(True)
    ctc_logits = torch.from_numpy(g["xc_ctc_logits"]).requires_grad_(True)
    trg = torch.from_numpy(g["xc_trg"])
    xe = O.xent_loss(torch.log_softmax(logits, -1), trg, 1, 0.1)
    ct = O.ctc_loss(torch.log_softmax(ctc_logits, -1), trg, torch.from_numpy(g["xc_in_len"]),
                    torch.from_numpy(g["xc_trg_len"]), 2)
    tot = 0.7 * xe + 0.3 * ct
    tot.backward()
    assert abs(tot.item() - g["xc_total"]) < 1e-3 and abs(xe.item() - g["xc_xent"]) < 1e-3
    assert abs(ct.item() - g["xc_ctc"]) < 1e-3
    np.testing.assert_allclose(logits.grad.numpy(), g["xc_dlogits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ctc_logits.grad.numpy(), g["xc_dctc"], rtol=1e-4, atol=1e-5)
    ct2 = O.ctc_loss(torch.log_softmax(torch.from_numpy(g["xc_ctc_logits"]), -1), trg,
                     torch.from_numpy(g["xc_in_len_inf"]), torch.from_numpy(g["xc_trg_len"]), 2)
    assert abs(ct2.item() - g["xc_ctc_inf"]) < 1e-3


def test_units_frontend():
    g = load_golden("units")
    np.testing.assert_allclose(O.cmvn(g["cmvn_in"].copy()), g["cmvn_out"], rtol=1e-6, atol=1e-6)
    rs = np.random.RandomState(42)
    p = O.specaugment_params(57, 80, rs, time_mask_t=100)
    np.testing.assert_array_equal(O.specaugment_apply(g["cmvn_in"], p), g["spec_out"])
    rs = np.random.RandomState(42)
    p = O.specaugment_params(7, 80, rs, time_mask_t=100)
    np.testing.assert_array_equal(O.specaugment_apply(g["cmvn_in"][:7], p), g["spec_out_short"])
    feat = g["cmvn_in"]
    padded, lengths, _ = O.pad_features([feat, feat[:20], feat[:33]])
    np.testing.assert_array_equal(padded, g["pad_out"])
    assert lengths == g["pad_len"].tolist()
    assert [O.get_n_frames(int(n), 16000) for n in g["n_frames_in"]] == g["n_frames"].tolist()
    assert O.get_n_frames(240000, 16000) == 1498


def test_fbank_known_answer():
    """Reference pin: test/unit/test_tokenizer.py:318-325 (CMVN'd frame 0, bins 0-9, atol=rtol=1e-5) and the
    n_frames column of test/data/speech/test.tsv."""
    g = load_golden("audio")
    pcm = g["pcm_260-123440-1"].astype(np.float32) / 32768.0
    feat = O.fbank(pcm)
    names = g["tsv_names"].tolist()
    assert feat.shape == (int(g["tsv_n_frames"][names.index("260-123440-1")]), 80)
    got = O.cmvn(feat)[0, :10]
    np.testing.assert_allclose(got, g["fbank_cmvn_ref_260-123440-1_frame0_bins0_9"], rtol=1e-5, atol=1e-5)
    for n, ns in zip(g["tsv_n_frames"], g["tsv_n_samples"]):
        assert 1 + (int(ns) - 400) // 160 == int(n) == O.get_n_frames(int(ns), 16000)
    for key in ("pcm_260-123440-0", "pcm_260-123440-6"):
        f = O.fbank(g[key].astype(np.float32) / 32768.0)
        assert f.shape[0] == int(g["tsv_n_frames"][names.index(key[4:])]) and np.isfinite(f).all()


@pytest.mark.parametrize("ln", ["pre", "post"])
def test_conformer_oracle_matches_reference(ln):
    """a30: the reference's ConformerEncoder (quirks included: depthwise convolution and BatchNorm over the batch axis,
    half-step residuals on top of the feed-forward modules' own) - train-mode output, every parameter gradient, the
    BatchNorm running statistics after one forward, and the eval-mode output."""
    g = load_golden("conformer")
    pre = ln + "."
    sd = {k[len(pre) + 4:]: torch.from_numpy(v).clone() for k, v in g.items() if k.startswith(pre + "sd0.")}
    sd = {"encoder." + k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    cfg = {"encoder": {"num_layers": 2, "num_heads": 2, "alpha": 1.0, "layer_norm": ln, "conv_kernel_sizes": [5, 5]}}
    src, lengths, proj = torch.from_numpy(g[pre + "src"]), torch.from_numpy(g[pre + "src_length"]), torch.from_numpy(g[pre + "proj"])
    stats = {}
    y, mask, _ = O.conformer_encoder_forward(sd, cfg, src, lengths, train=True, new_stats=stats)
    np.testing.assert_allclose(y.detach().numpy(), g[pre + "out_train"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(mask.numpy(), g[pre + "mask"])
    (y * proj).sum().backward()
    for k, v in g.items():
        if k.startswith(pre + "grad."):
            got = sd["encoder." + k[len(pre) + 5:]].grad
            np.testing.assert_allclose(got.numpy(), v, rtol=2e-4, atol=2e-4, err_msg=k)
        if k.startswith(pre + "sd1.") and "running" in k:
            np.testing.assert_allclose(stats["encoder." + k[len(pre) + 4:]].numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)
    with torch.no_grad():
        sd2 = {k: v.detach() for k, v in sd.items()}
        sd2.update(stats)
        y2, _, _ = O.conformer_encoder_forward(sd2, cfg, src, lengths, train=False)
    np.testing.assert_allclose(y2.numpy(), g[pre + "out_eval"], rtol=1e-4, atol=1e-4)


def test_ctc_best_path_known_answer():
    """The collapse rule of CTC best-path decoding on the textbook example: a a - a b b - -> a a b; frames
    beyond the input length are ignored, an all-blank utterance decodes to nothing."""
    from oracle import s2t_oracle as O
    blank, pad, V = 2, 1, 6
    paths = np.array([[4, 4, 2, 4, 5, 5, 2], [2, 2, 2, 2, 2, 2, 2], [3, 3, 3, 4, 4, 2, 5]])
    logits = np.full((3, 7, V), -1.0, dtype=np.float32)
    np.put_along_axis(logits, paths[..., None], 1.0, axis=2)
    ids, lens = O.ctc_best_path(logits, np.array([7, 7, 4]), blank, pad)
    assert lens.tolist() == [3, 0, 2]
    assert ids[0, :3].tolist() == [4, 4, 5] and ids[2, :2].tolist() == [3, 4]
    assert (ids[0, 3:] == pad).all() and (ids[1] == pad).all() and (ids[2, 2:] == pad).all()


# ---------------------------------------------------------------------------------------------- decoding options
SEARCH_CFG = {"decoder": {"num_layers": 3, "num_heads": 4, "layer_norm": "pre", "activation": "relu", "alpha": 1.0,
                          "embeddings": {"scale": False}}}
SEARCH_SPECIALS = dict(unk=0, pad=1, bos=2, eos=3, sep=4, lang_tags=[5, 6], all=[0, 1, 2, 3, 4])


def search_case(g, bs):
    sd = golden_sd(g, f"bs{bs}.sd.")
    return sd, torch.from_numpy(g[f"bs{bs}.encoder_output"]), torch.ones(bs, 1, 4, dtype=torch.bool)


def _cmp(g, case, ids, scores=None, att=None):
    assert np.array_equal(np.asarray(ids), g[f"{case}.ids"]), (case, ids, g[f"{case}.ids"])
    assert np.array_equal(g[f"{case}.ids"], g[f"{case}.exp_ids"])  # the constants of the reference's own test
    if scores is not None:
        np.testing.assert_allclose(np.asarray(scores), g[f"{case}.scores"], rtol=1e-4, atol=1e-4, err_msg=case)
        np.testing.assert_allclose(np.asarray(scores), g[f"{case}.exp_scores"], rtol=1e-4, atol=1e-4, err_msg=case)
    if att is not None:
        np.testing.assert_allclose(np.asarray(att), g[f"{case}.att"], rtol=1e-4, atol=1e-4, err_msg=case)
        np.testing.assert_allclose(np.asarray(att), g[f"{case}.exp_att"], rtol=1e-4, atol=1e-4, err_msg=case)


def test_search_options_match_reference_tests():
    """Forced-decoding prompts, repetition penalty, n-gram blocking, generate_unk, attention export: the oracle's greedy /
    beam search against the captures AND the hard-coded constants of test/unit/test_search.py:101-500."""
    g = load_golden("search_options")
    S, cfg = SEARCH_SPECIALS, SEARCH_CFG
    sd, enc, mask = search_case(g, 2)
    prompt, pmask = torch.from_numpy(g["prompt"]), torch.from_numpy(g["prompt_mask"])
    with torch.no_grad():
        _cmp(g, "greedy", *O.greedy(sd, cfg, S, enc, mask, 3, return_prob=True))
        _cmp(g, "greedy_prompt", *O.greedy(sd, cfg, S, enc, mask, 7, return_prob=True, return_attention=True, decoder_prompt=prompt,
                                           trg_prompt_mask=pmask))
        _cmp(g, "beam1", *O.beam_search(sd, cfg, S, enc, mask, 1, 3, 0.0, n_best=1))
        _cmp(g, "beam7", *O.beam_search(sd, cfg, S, enc, mask, 7, 3, 1.0, n_best=5))
        _cmp(g, "beam7_prompt", *O.beam_search(sd, cfg, S, enc, mask, 7, 10, 1.0, n_best=5, decoder_prompt=prompt, trg_prompt_mask=pmask))
        _cmp(g, "beam7_penalty", *O.beam_search(sd, cfg, S, enc, mask, 7, 3, 1.0, n_best=5, repetition_penalty=1.5,
                                                encoder_input=torch.from_numpy(g["beam7_penalty.src_tokens"])))
        _cmp(g, "greedy_ngram", *O.greedy(sd, cfg, S, enc, mask, 7, return_prob=True, no_repeat_ngram_size=3))
        _cmp(g, "beam3_ngram", *O.beam_search(sd, cfg, S, enc, mask, 3, 7, 1.0, n_best=3, no_repeat_ngram_size=3))
        sd, enc, mask = search_case(g, 3)
        _cmp(g, "greedy_nounk", O.greedy(sd, cfg, S, enc, mask, 3, generate_unk=False)[0])
        _cmp(g, "greedy_nounk_penalty", O.greedy(sd, cfg, S, enc, mask, 3, generate_unk=False, repetition_penalty=1.5)[0])
        src = torch.from_numpy(g["greedy_src_penalty.src_tokens"])
        ids, _, att = O.greedy(sd, cfg, S, enc, (src != 1).unsqueeze(1), 3, generate_unk=False, repetition_penalty=1.5, encoder_input=src,
                               return_attention=True)
        _cmp(g, "greedy_src_penalty", ids, att=att)


REF_UNIT_CFG = {"encoder": {"num_layers": 3, "num_heads": 4, "layer_norm": "pre", "activation": "relu", "alpha": 1.0, "subsample": False},
                "decoder": {"num_layers": 3, "num_heads": 4, "layer_norm": "pre", "activation": "relu", "alpha": 1.0,
                            "embeddings": {"scale": False}}}


def test_reference_unit_test_known_answers():
    """The oracle against the known-answer tests of the reference's own suite (tests/golden/ref_unit_tests.npz, written by
    oracle/make_golden.py:golden_ref_unit_tests from the reference's classes and tied there to the constants the tests
    hard-code): test/unit/test_transformer_encoder.py:31-90 and test/unit/test_transformer_decoder.py:45-172, tolerance 1e-4
    as in those tests; the DeepNet residual scales of test/unit/test_model_init.py:104-107."""
    g = load_golden("ref_unit_tests")
    sd = {"encoder." + k[len("enc.sd."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("enc.sd.")}
    x = torch.from_numpy(g["enc.x"])
    y, mask, _ = O.encoder_forward(sd, REF_UNIT_CFG, x, torch.tensor([4, 4]))
    torch.testing.assert_close(y, torch.from_numpy(g["enc.out"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y[0, 0], torch.from_numpy(g["enc.test_const_row0"]), rtol=1e-4, atol=1e-4)
    sd = {"decoder." + k[len("dec.sd."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("dec.sd.")}
    src_mask, trg_mask = torch.ones(2, 1, 4, dtype=torch.bool), torch.ones(2, 5, 1, dtype=torch.bool)
    logits, states, att, _ = O.decoder_forward_embedded(sd, REF_UNIT_CFG, torch.from_numpy(g["dec.trg_embed"]), torch.from_numpy(g["dec.memory"]),
                                                        src_mask, trg_mask, return_attention=True)
    for got, key in ((logits, "logits"), (att, "att"), (states, "states")):
        torch.testing.assert_close(got, torch.from_numpy(g["dec." + key]), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(got[0, 0], torch.from_numpy(g[f"dec.test_const_{key}_row0"]), rtol=1e-4, atol=1e-4)
    # the product's model builder computes the same residual scales (initialization.py of the reference, :176-210)
    import copy
    from golden_cfg import FIXTURES
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    cfg = copy.deepcopy(FIXTURES["model_deepnet"]["cfg"])
    cfg["encoder"]["num_layers"] = cfg["decoder"]["num_layers"] = 6
    model = build_model(cfg, None, Vocabulary.synthetic(20))
    assert {layer.alpha for layer in model.encoder.layers} == {float(g["deepnet_alpha_6_6"][0])}
    assert {layer.alpha for layer in model.decoder.layers} == {float(g["deepnet_alpha_6_6"][1])}


def _mt_batch(g):
    return {"src": torch.from_numpy(g["src"]), "src_length": torch.from_numpy(g["src_length"]),
            "trg_input": torch.from_numpy(g["trg_input"]), "trg": torch.from_numpy(g["trg"]),
            "trg_length": torch.from_numpy(g["trg_length"]), "trg_mask": torch.from_numpy(g["trg_mask"])}


def test_text_source_model_matches_reference():
    """BASELINE config 0, configs/transformer_small.yaml (task MT, reverse task): source embedding, encoder without
    sub-sampler, tied softmax, cross-entropy only - activations, loss, every gradient, greedy and beam-5 hypotheses of the
    reference (tests/golden/model_mt.npz) against the oracle's text-source path."""
    from golden_cfg import mt_cfg
    g = load_golden("model_mt")
    cfg = oracle_cfg(mt_cfg())
    sd = {k: v.clone().requires_grad_(True) for k, v in golden_sd(g).items() if not k.endswith("pe.pe")}
    sd["decoder.output_layer.weight"] = sd["trg_embed.lut.weight"]  # tied softmax: ONE parameter (model.py:470-478)
    b = _mt_batch(g)
    enc, mask = O.encoder_forward_text(sd, cfg, b["src"], SPECIALS["pad"])
    assert np.array_equal(mask.numpy(), g["src_mask"])
    torch.testing.assert_close(enc.detach(), torch.from_numpy(g["enc_out"]), **TOL)
    logits, hidden, att, ctc = O.decoder_forward(sd, cfg, b["trg_input"], enc, mask, b["trg_mask"], return_attention=True)
    assert ctc is None
    torch.testing.assert_close(logits.detach(), torch.from_numpy(g["logits"]), **TOL)
    torch.testing.assert_close(att.detach(), torch.from_numpy(g["att"]), **TOL)
    total, xent, ctcl, ncor, _, _ = O.model_loss(sd, cfg, b, SPECIALS, 0.0, None)
    assert ctcl is None and abs(total.item() - g["loss_total"]) <= 1e-4 * abs(g["loss_total"]) and int(ncor) == int(g["n_correct"])
    total.backward()
    n_checked = 0
    for k, v in g.items():
        if k.startswith("grad."):
            got = sd[k[5:]].grad
            scale = np.abs(v).max() + 1e-6
            assert np.abs(got.numpy() - v).max() <= 1e-4 * scale + 1e-5, k
            n_checked += 1
    assert n_checked > 50 and "grad.decoder.output_layer.weight" not in g  # the tied weight appears once, under trg_embed
    with torch.no_grad():
        sdd = {k: v.detach() for k, v in sd.items()}
        enc, mask = O.encoder_forward_text(sdd, cfg, b["src"], SPECIALS["pad"])
        ids, scores = O.greedy(sdd, cfg, SPECIALS, enc, mask, 31, return_prob=True)
        assert np.array_equal(ids.numpy(), g["greedy_ids"])
        np.testing.assert_allclose(scores.numpy(), g["greedy_scores"], rtol=1e-4, atol=1e-4)
        k = int(g["beam_size"])
        ids, scores = O.beam_search(sdd, cfg, SPECIALS, enc, mask, k, 31, float(g["beam_alpha"]), n_best=1)
        assert np.array_equal(ids.numpy(), g["beam_ids"])
        np.testing.assert_allclose(scores.numpy(), g["beam_scores"], rtol=1e-4, atol=1e-4)
        max_len = int(max(g["src_length"]) * 1.5)
        ids, scores = O.beam_search(sdd, cfg, SPECIALS, enc, mask, k, max_len, float(g["beam_alpha"]), n_best=k)
        assert np.array_equal(ids.numpy(), g["beam_ids_nbest"])
        np.testing.assert_allclose(scores.numpy(), g["beam_scores_nbest"], rtol=1e-4, atol=1e-4)


def test_search_wrapper_options_match_reference():
    """The options search() derives from the batch (search.py:866-873) - tests/golden/search_wrapper.npz, captured from the
    reference's own search() - against the oracle's restatement: ids bit-exact, scores 1e-4."""
    import json
    from pathlib import Path
    from golden_cfg import mt_cfg
    g = load_golden("search_wrapper")
    cases = json.loads((Path(__file__).resolve().parent / "golden" / "search_wrapper_cases.json").read_text())
    cfg = oracle_cfg(mt_cfg())
    sd = {k: v for k, v in golden_sd(g).items() if not k.endswith("pe.pe")}
    sd["decoder.output_layer.weight"] = sd["trg_embed.lut.weight"]
    specials = dict(SPECIALS, sep=4, all=[0, 1, 2, 3, 4])
    prm, pmask = torch.from_numpy(g["prompt"]), torch.from_numpy(g["prompt_mask"])
    for name, kw in sorted(cases.items()):
        kw = dict(kw)
        batch = {"src": torch.from_numpy(g["src"]), "src_length": torch.from_numpy(g["src_length"])}
        if kw.pop("prompted"):
            # Batch.__init__ (batch.py:82-96): EOS -> PAD in the teacher-forcing input, no column dropped without an EOS
            batch["trg_input"], batch["trg_prompt_mask"] = prm, pmask
        with torch.no_grad():
            ids, scores = O.search_text(sd, cfg, specials, batch, 14, kw.pop("beam_size"), kw.pop("beam_alpha"),
                                        n_best=kw.pop("n_best", 1), generate_unk=False, **kw)
        assert np.array_equal(ids.numpy(), g[f"{name}.ids"]), name
        np.testing.assert_allclose(scores.numpy(), g[f"{name}.scores"], rtol=1e-4, atol=1e-4, err_msg=name)


def _frontend_general_cases():
    import json
    from pathlib import Path
    return json.loads((Path(__file__).resolve().parent / "golden" / "frontend_general.json").read_text())


def oracle_frontend(item, cmvn, sa, max_length, rng):
    """SpeechProcessor.__call__'s cmvn / specaugment block (tokenizers.py:474-492) on the oracle's restatements."""
    if max_length is not None and item.shape[0] > max_length:
        item = item[:max_length]
    if cmvn and cmvn["before"]:
        item = O.cmvn(item, cmvn["norm_means"], cmvn["norm_vars"])
    item = O.specaugment_apply(item, O.specaugment_params(item.shape[0], item.shape[1], rng, **sa))
    if cmvn and not cmvn["before"]:
        item = O.cmvn(item, cmvn["norm_means"], cmvn["norm_vars"])
    return item


def test_frontend_orders_and_mask_counts_beside_the_configured_one():
    """CMVN after SpecAugment, three / four masks of a kind, SpecAugment without CMVN, truncation first: the oracle against the
    reference's own classes composed as SpeechProcessor.__call__ composes them (frontend_general.npz)."""
    g = load_golden("frontend_general")
    for name, case in _frontend_general_cases().items():
        rng = np.random.RandomState(77)
        for i in range(3):
            got = oracle_frontend(g[f"in{i}"].copy(), case["cmvn"], case["specaugment"], case["max_length"], rng)
            np.testing.assert_allclose(got, g[f"{name}_{i}"], rtol=1e-5, atol=1e-5, err_msg=f"{name} {i}")


def test_predict_validation_leg():
    """The validation-loss / reference-scoring leg of the reference's `predict` loop (prediction.py:165-200), captured per batch
    in predict_loss.npz: the oracle's eval-mode loss, n_correct, log-probabilities and the scores of the reference tokens."""
    g = load_golden("predict_loss")
    sd = golden_sd(load_golden("model_pre"))
    cfg = oracle_cfg(FIXTURES["model_pre"]["cfg"])
    tot = dict(loss=0.0, n_correct=0, ntokens=0, nseqs=0)
    for bi in (0, 1):
        pre = f"b{bi}."
        b = O.make_batch(torch.from_numpy(g[pre + "src"]), torch.from_numpy(g[pre + "src_length"]), torch.from_numpy(g[pre + "trg_full"]),
                         torch.from_numpy(g[pre + "trg_length_full"]), SPECIALS["pad"], SPECIALS["eos"])
        order = torch.argsort(b["src_length"], descending=True, stable=True)
        rev = g[pre + "reverse_index"]
        assert np.array_equal(np.argsort(order.numpy()), rev)  # batch.sort_by_src_length(): position of every original row
        b = {k: v[order] for k, v in b.items()}
        total, xent, ctc, ncor, out, ctc_out = O.model_loss(sd, cfg, b, SPECIALS, 0.1, 0.3)
        assert abs(total.item() - g[pre + "loss"]) <= 1e-4 * abs(g[pre + "loss"])
        # the 4-tuple of return_type="loss": slots 1-2 are the two loss components (what the reference's loop calls log_probs / attn)
        assert abs(xent.item() - g[pre + "slot1"]) <= 1e-4 * abs(g[pre + "slot1"]) and abs(ctc.item() - g[pre + "slot2"]) <= 1e-4 * abs(g[pre + "slot2"])
        assert int(ncor) == int(g[pre + "n_correct"])
        lp = torch.log_softmax(out, -1)
        torch.testing.assert_close(lp, torch.from_numpy(g[pre + "log_probs"]), **TOL)
        torch.testing.assert_close(torch.log_softmax(ctc_out, -1), torch.from_numpy(g[pre + "ctc_log_probs"]), **TOL)
        assert np.array_equal(b["trg"].numpy(), g[pre + "trg_sorted"])
        for i in range(int(g[pre + "n_rows"])):  # Batch.score (batch.py:210-223): log-probability of every non-pad reference token
            row = np.array([lp[i, j, t].item() for j, t in enumerate(b["trg"][i]) if t != SPECIALS["pad"]])
            np.testing.assert_allclose(row, g[pre + f"ref_scores.{i}"], rtol=1e-4, atol=1e-4)
        tot["loss"] += total.item()
        tot["n_correct"] += int(ncor)
        tot["ntokens"] += int((b["trg"] != SPECIALS["pad"]).sum())
        tot["nseqs"] += b["src"].shape[0]
    for k, v in tot.items():
        assert abs(v - float(g["total." + k])) <= 1e-4 * max(1.0, abs(float(g["total." + k]))), k


def test_ctc_prefix_score_against_enumeration():
    """Known-answer pin of the CTC prefix score (EXTENSION f3; Watanabe et al. 2017, Algorithm 2): on tiny random inputs the
    recursion's log psi(g + c) equals the summed probability of EVERY alignment whose labelling starts with g + c (EOS: is exactly
    g), found by brute-force enumeration - for chains of extensions, repeated labels and ragged input lengths."""
    rng = np.random.RandomState(5)
    blank, eos = 0, 3
    for trial in range(6):
        T, V = 5, 4
        in_len = T if trial % 2 == 0 else T - 1
        logp = np.log(rng.dirichlet(np.ones(V), size=T))
        lp_live = logp[:in_len]
        # hand-checkable corner: the empty prefix extended by c = probability that the first non-blank label is c
        r0 = O.ctc_prefix_init(logp, in_len, blank)
        psi, r1 = O.ctc_prefix_score(logp, in_len, [blank], [1, 2, eos], r0, blank, eos)
        for i, c in enumerate([1, 2]):
            assert abs(psi[i] - O.ctc_prefix_brute(lp_live, [c], blank)) < 1e-9
        assert abs(psi[2] - O.ctc_prefix_brute(lp_live, [], blank, whole=True)) < 1e-9  # EOS right away: the all-blank alignment
        assert abs(psi[2] - logp[:in_len, blank].sum()) < 1e-9
        # chains: g = [1], [1, 1] (a repeat: needs a blank in between), [1, 2], [1, 2, 1]
        state = {(): (r0, 0.0)}
        for g in [(1, ), (1, 1), (1, 2), (1, 2, 1), (2, ), (2, 2)]:
            parent = g[:-1]
            r_prev, _ = state[parent]
            cands = [1, 2, eos, blank]
            psi, r_new = O.ctc_prefix_score(logp, in_len, [blank] + list(parent), cands, r_prev, blank, eos)
            i = cands.index(g[-1])
            want = O.ctc_prefix_brute(lp_live, list(g), blank)
            assert abs(psi[i] - want) < 1e-9 or (np.isinf(want) and psi[i] < -1e8), (trial, g, psi[i], want)
            assert abs(psi[2] - O.ctc_prefix_brute(lp_live, list(parent), blank, whole=True)) < 1e-9  # parent + EOS
            assert np.isinf(psi[3]) and psi[3] < 0  # the blank is never a label
            state[g] = (r_new[:, :, i], psi[i])


@pytest.mark.parametrize("name", list(FIXTURES))
def test_joint_ctc_beam_search_reduces_to_beam_search(name):
    """weight 0: the candidate pre-selection (n_cand >= beam) cannot lose a winner - the joint search IS the reference's beam search"""
    g = load_golden(name)
    cfg = oracle_cfg(FIXTURES[name]["cfg"])
    sd = golden_sd(g)
    b = _batch(g)
    enc, mask, _ = O.encoder_forward(sd, cfg, b["src"], b["src_length"])
    k, alpha = int(g["beam_size"]), float(g["beam_alpha"])
    ids, scores = O.joint_ctc_beam_search(sd, cfg, SPECIALS, enc, mask, k, 12, alpha, ctc_weight=0.0, n_cand=8, n_best=k)
    assert np.array_equal(ids.numpy(), g["beam_ids"])
    np.testing.assert_allclose(scores.numpy(), g["beam_scores"], rtol=1e-4, atol=1e-4)
    ids2, scores2 = O.joint_ctc_beam_search(sd, cfg, SPECIALS, enc, mask, k, 12, alpha, ctc_weight=0.3, n_cand=8, n_best=1)
    assert ids2.shape[0] == b["src"].shape[0] and np.isfinite(scores2.numpy()).all()
