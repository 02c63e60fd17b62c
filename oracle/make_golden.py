"""Generates tests/golden/*.npz by running the REAL reference (/root/reference, read-only) on CPU.

Run in the build container only (`python oracle/make_golden.py`); the reference never travels to the GPU box,
the small fixtures it emits here do.  Third-party packages that the reference imports but that are absent from
this image and carry no arithmetic of the hot path (tensorboard, sacrebleu, subword_nmt, seaborn, editdistance,
sacremoses, torchaudio) are replaced by empty stub modules before `import joeynmt`; nothing from the reference
is copied into the repository — only inputs and outputs of its functions.
"""
import json
import sys
import types
from pathlib import Path

import copy

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return _Dummy()

        def __call__(self, *a, **k):
            return _Dummy()

    if "torch.utils.tensorboard" not in sys.modules:
        try:
            import torch.utils.tensorboard  # noqa: F401
        except Exception:
            _stub("torch.utils.tensorboard", SummaryWriter=_Dummy)
    for name in ("seaborn", "editdistance", "sacremoses"):
        try:
            __import__(name)
        except Exception:
            _stub(name)
    try:
        import sacrebleu  # noqa: F401
    except Exception:
        sb = _stub("sacrebleu")
        met = _stub("sacrebleu.metrics", BLEU=_Dummy, CHRF=_Dummy)
        bleu = _stub("sacrebleu.metrics.bleu", _get_tokenizer=lambda *a, **k: _Dummy)
        sb.metrics, met.bleu = met, bleu
    try:
        import subword_nmt  # noqa: F401
    except Exception:
        sn = _stub("subword_nmt")
        sn.apply_bpe = _stub("subword_nmt.apply_bpe", BPE=_Dummy)
    try:
        import torchaudio  # noqa: F401
    except Exception:
        ta = _stub("torchaudio")
        comp = _stub("torchaudio.compliance")
        kaldi = _stub("torchaudio.compliance.kaldi")
        sox = _stub("torchaudio.sox_effects")
        ta.compliance, comp.kaldi, ta.sox_effects = comp, kaldi, sox
    sys.path.insert(0, str(REF))
    import joeynmt  # noqa: F401


def np_sd(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}  # copy: optimizers update in place


SPECIALS = dict(unk=0, pad=1, bos=2, eos=3)


def tiny_cfg(layer_norm="pre", initializer="xavier_uniform", act="relu", heads=2):
    return {
        "initializer": initializer, "bias_initializer": "zeros", "embed_initializer": "xavier_uniform",
        "tied_embeddings": False, "tied_softmax": False,
        "encoder": {"type": "transformer", "num_layers": 2, "num_heads": heads, "embeddings": {"embedding_dim": 8},
                    "hidden_size": 16, "ff_size": 32, "dropout": 0.0, "freeze": False, "subsample": True,
                    "conv_kernel_sizes": [5, 5], "conv_channels": 24, "in_channels": 8, "layer_norm": layer_norm,
                    "activation": act},
        "decoder": {"type": "transformer", "num_layers": 2, "num_heads": heads,
                    "embeddings": {"embedding_dim": 16, "scale": True, "dropout": 0.0}, "hidden_size": 16, "ff_size": 32,
                    "dropout": 0.0, "freeze": False, "layer_norm": layer_norm, "activation": act},
    }


def make_vocab(size):
    from types import SimpleNamespace

    from joeynmt.vocabulary import Vocabulary
    cfg = SimpleNamespace(unk_token="<unk>", pad_token="<pad>", bos_token="<s>", eos_token="</s>", sep_token=None, unk_id=0,
                          pad_id=1, bos_id=2, eos_id=3, sep_id=None, lang_tags=[])
    return Vocabulary([f"tok{i}" for i in range(size - 4)], cfg)


def synth_batch(B, T, F, V, Lmin, Lmax, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    lengths = torch.randint(T // 2, T + 1, (B, ), generator=g) if ragged else torch.full((B, ), T)
    lengths[0] = T
    src = torch.randn(B, T, F, generator=g)
    for b in range(B):
        src[b, lengths[b]:] = 1.0  # pad_features pads with 1.0
    tl = torch.randint(Lmin, Lmax + 1, (B, ), generator=g)
    L = int(tl.max()) + 2
    trg = torch.full((B, L), SPECIALS["pad"], dtype=torch.long)
    for b in range(B):
        n = int(tl[b])
        trg[b, 0] = SPECIALS["bos"]
        trg[b, 1:1 + n] = torch.randint(4, V, (n, ), generator=g)
        trg[b, 1 + n] = SPECIALS["eos"]
    return src, lengths, trg, tl + 2


def golden_model(name, cfg, V=20, B=3, T=37, seed=7, smoothing=0.1, ctc_weight=0.3, beam=3, alpha=1.0):
    from joeynmt.batch import Batch
    from joeynmt.model import build_model
    from joeynmt.search import search
    torch.manual_seed(42)
    vocab = make_vocab(V)
    model = build_model(cfg, src_vocab=None, trg_vocab=vocab)
    model.loss_function = ("crossentropy-ctc", smoothing, ctc_weight)
    with torch.no_grad():  # non-trivial biases / norms so every term is exercised
        g = torch.Generator().manual_seed(123)
        for n, p in model.named_parameters():
            if "bias" in n or "layer_norm" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    model.eval()
    src, lengths, trg, trg_len = synth_batch(B, T, cfg["encoder"]["in_channels"], V, 3, 6, seed)
    batch = Batch(src=src, src_length=lengths, src_prompt_mask=None, trg=trg, trg_length=trg_len, trg_prompt_mask=None,
                  indices=torch.arange(B), device=torch.device("cpu"), pad_index=1, eos_index=3, is_train=True, task="S2T")
    out = {f"sd.{k}": v for k, v in np_sd(model.state_dict()).items()}
    out.update(src=src.numpy(), src_length=lengths.numpy(), trg_full=trg.numpy(), trg_length_full=trg_len.numpy(),
               trg_input=batch.trg_input.numpy(), trg=batch.trg.numpy(), trg_length=batch.trg_length.numpy(),
               trg_mask=batch.trg_mask.numpy(), ntokens=np.int64(batch.ntokens))
    kw = dict(vars(batch))
    kw["repad"] = False
    # encoder / decoder activations
    with torch.no_grad():
        enc, _, src_mask, _ = model(return_type="encode", **kw)
        logits, hidden, att, ctc_logits = model(return_type="decode_ctc", encoder_output=enc, encoder_hidden=None,
                                                src_mask=src_mask, trg_input=batch.trg_input, unroll_steps=None,
                                                trg_mask=batch.trg_mask, return_attention=True)
    out.update(enc_out=enc.numpy(), src_mask=src_mask.numpy(), logits=logits.numpy(), dec_hidden=hidden.numpy(),
               att=att.numpy(), ctc_logits=ctc_logits.numpy())
    # loss + gradients
    model.zero_grad()
    total, xent, ctc, n_correct = model(return_type="loss", **kw)
    total.backward()
    out.update(loss_total=total.item(), loss_xent=xent.item(), loss_ctc=ctc.item(), n_correct=n_correct.item())
    for n, p in model.named_parameters():
        out[f"grad.{n}"] = p.grad.numpy().copy()
    # decoding
    with torch.no_grad():
        gids, gscores, _ = search(model, batch, max_output_length=12, beam_size=1, beam_alpha=-1, return_prob="hyp")
        bids, bscores, _ = search(model, batch, max_output_length=12, beam_size=beam, beam_alpha=alpha, n_best=beam,
                                  return_prob="hyp")
        bids1, bscores1, _ = search(model, batch, max_output_length=-1, beam_size=beam, beam_alpha=0.0, n_best=1,
                                    return_prob="hyp")
    out.update(greedy_ids=gids, greedy_scores=gscores, beam_ids=bids, beam_scores=bscores, beam_ids_a0=bids1,
               beam_scores_a0=bscores1, beam_size=np.int64(beam), beam_alpha=np.float64(alpha))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(name, "loss", total.item(), xent.item(), ctc.item(), n_correct.item(), "beam", bids.shape)


def small_mt_cfg():
    """The `model:` section of the reference's configs/transformer_small.yaml (BASELINE.json configs[0]), read from the file."""
    import yaml
    cfg = yaml.safe_load((REF / "configs/transformer_small.yaml").read_text())["model"]
    assert cfg["tied_softmax"] and not cfg["tied_embeddings"] and cfg["encoder"]["hidden_size"] == 64
    return cfg


def golden_model_mt(name="model_mt", V=30, B=5, seed=11, beam=5, alpha=1.0):
    """configs/transformer_small.yaml as a text-to-text model (task "MT": source Embeddings, no sub-sampler, no CTC layer,
    tied softmax) on the reverse task its tutorial trains (docs/source/tutorial.rst: the target is the source reversed),
    with the config's own loss (crossentropy, label_smoothing 0.0) and decoding settings (beam 5, alpha 1.0, max length 31)."""
    from joeynmt.batch import Batch
    from joeynmt.model import build_model
    from joeynmt.search import search
    torch.manual_seed(42)
    cfg = small_mt_cfg()
    vocab = make_vocab(V)
    model = build_model(copy.deepcopy(cfg), src_vocab=vocab, trg_vocab=make_vocab(V))
    model.loss_function = ("crossentropy", 0.0, 0.0)  # (type, label_smoothing, ctc_weight): training.loss / label_smoothing of the config
    assert model.decoder.ctc_output_layer is None and model.decoder.output_layer.weight is model.trg_embed.lut.weight
    with torch.no_grad():
        g = torch.Generator().manual_seed(123)
        for n, p in model.named_parameters():
            if "bias" in n or "layer_norm" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    model.eval()
    g = torch.Generator().manual_seed(seed)
    n_tok = torch.randint(3, 13, (B, ), generator=g)
    n_tok[0] = 12
    S = int(n_tok.max()) + 1
    src = torch.full((B, S), SPECIALS["pad"], dtype=torch.long)
    trg = torch.full((B, S + 1), SPECIALS["pad"], dtype=torch.long)
    for b in range(B):
        n = int(n_tok[b])
        words = torch.randint(4, V, (n, ), generator=g)
        src[b, :n], src[b, n] = words, SPECIALS["eos"]
        trg[b, 0], trg[b, 1:1 + n], trg[b, 1 + n] = SPECIALS["bos"], words.flip(0), SPECIALS["eos"]
    batch = Batch(src=src, src_length=n_tok + 1, src_prompt_mask=None, trg=trg, trg_length=n_tok + 2, trg_prompt_mask=None,
                  indices=torch.arange(B), device=torch.device("cpu"), pad_index=1, eos_index=3, is_train=True, task="MT")
    out = {f"sd.{k}": v for k, v in np_sd(model.state_dict()).items()}
    out.update(src=src.numpy(), src_length=(n_tok + 1).numpy(), trg_full=trg.numpy(), trg_length_full=(n_tok + 2).numpy(),
               trg_input=batch.trg_input.numpy(), trg=batch.trg.numpy(), trg_length=batch.trg_length.numpy(),
               trg_mask=batch.trg_mask.numpy(), src_mask=batch.src_mask.numpy(), ntokens=np.int64(batch.ntokens))
    kw = dict(vars(batch))
    with torch.no_grad():
        enc, _, src_mask, _ = model(return_type="encode", **kw)
        logits, hidden, att, _ = model(return_type="decode", encoder_output=enc, encoder_hidden=None, src_mask=batch.src_mask,
                                       trg_input=batch.trg_input, unroll_steps=None, trg_mask=batch.trg_mask, return_attention=True)
    out.update(enc_out=enc.numpy(), logits=logits.numpy(), dec_hidden=hidden.numpy(), att=att.numpy())
    model.zero_grad()
    total, nll, ctc, n_correct = model(return_type="loss", **kw)
    assert nll is None and ctc is None
    total.backward()
    out.update(loss_total=total.item(), n_correct=n_correct.item())
    for n, p in model.named_parameters():
        out[f"grad.{n}"] = p.grad.numpy().copy()
    with torch.no_grad():
        gids, gscores, _ = search(model, batch, max_output_length=31, beam_size=1, beam_alpha=-1, return_prob="hyp")
        bids, bscores, _ = search(model, batch, max_output_length=31, beam_size=beam, beam_alpha=alpha, n_best=1, return_prob="hyp")
        bidsn, bscoresn, _ = search(model, batch, max_output_length=-1, beam_size=beam, beam_alpha=alpha, n_best=beam,
                                    return_prob="hyp")
    out.update(greedy_ids=gids, greedy_scores=gscores, beam_ids=bids, beam_scores=bscores, beam_ids_nbest=bidsn,
               beam_scores_nbest=bscoresn, beam_size=np.int64(beam), beam_alpha=np.float64(alpha))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    import json
    (OUT / "model_mt_cfg.json").write_text(json.dumps(cfg, indent=1, sort_keys=True) + "\n")
    print(name, "loss", total.item(), n_correct.item(), "greedy", gids.shape, "beam", bids.shape, bidsn.shape)


def make_vocab_sep(size):
    """Vocabulary with a prompt marker (sep) - the models that accept forced prefixes need one (model.py:271-282)."""
    from types import SimpleNamespace

    from joeynmt.vocabulary import Vocabulary
    cfg = SimpleNamespace(unk_token="<unk>", pad_token="<pad>", bos_token="<s>", eos_token="</s>", sep_token="<sep>", unk_id=0,
                          pad_id=1, bos_id=2, eos_id=3, sep_id=4, lang_tags=[])
    return Vocabulary([f"tok{i}" for i in range(size - 5)], cfg)


def golden_search_wrapper(name="search_wrapper", V=30, B=5, seed=13, beam=5, alpha=1.0):
    """The options `search()` DERIVES FROM THE BATCH (search.py:866-873): `encoder_input = batch.src` when a repetition
    penalty / n-gram block is on, `decoder_prompt` / `trg_prompt_mask` from a prompted batch - through the reference's own
    `search()` on the transformer_small.yaml model (text source, so that batch.src holds token ids) with a vocabulary that has
    a prompt marker."""
    from joeynmt.batch import Batch
    from joeynmt.model import build_model
    from joeynmt.search import search
    torch.manual_seed(43)
    cfg = small_mt_cfg()
    model = build_model(copy.deepcopy(cfg), src_vocab=make_vocab_sep(V), trg_vocab=make_vocab_sep(V))
    assert model.sep_index == 4
    with torch.no_grad():
        g = torch.Generator().manual_seed(321)
        for n, p in model.named_parameters():
            if "bias" in n or "layer_norm" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif "lut" in n:  # a flatter output distribution: the penalty has to be able to change the arg-max
                p.mul_(0.35)
    model.eval()
    g = torch.Generator().manual_seed(seed)
    n_tok = torch.randint(4, 11, (B, ), generator=g)
    n_tok[0] = 10
    S = int(n_tok.max()) + 1
    src = torch.full((B, S), SPECIALS["pad"], dtype=torch.long)
    for b in range(B):
        n = int(n_tok[b])
        src[b, :n], src[b, n] = torch.randint(5, 12, (n, ), generator=g), SPECIALS["eos"]  # few distinct words: repeats
    out = {f"sd.{k}": v for k, v in np_sd(model.state_dict()).items()}
    out.update(src=src.numpy(), src_length=(n_tok + 1).numpy())

    def plain_batch():
        return Batch(src=src.clone(), src_length=n_tok + 1, src_prompt_mask=None, trg=None, trg_length=None, trg_prompt_mask=None,
                     indices=torch.arange(B), device=torch.device("cpu"), pad_index=1, eos_index=3, is_train=False, task="MT")

    # forced prefixes: <s> w w <sep> for some rows, shorter for others (mask 0 behind the prefix), as datasets.py builds them
    P = 4
    prm = torch.full((B, P), SPECIALS["pad"], dtype=torch.long)
    prm_mask = torch.zeros((B, P), dtype=torch.long)
    prm_len = torch.tensor([4, 3, 4, 2, 3])[:B]
    for b in range(B):
        n = int(prm_len[b])
        prm[b, 0] = SPECIALS["bos"]
        prm[b, 1:n - 1] = torch.randint(5, V, (n - 2, ), generator=g)
        prm[b, n - 1] = 4  # <sep>
        prm_mask[b, :n] = 1
    out.update(prompt=prm.numpy(), prompt_mask=prm_mask.numpy(), prompt_length=prm_len.numpy())

    def prompted_batch():
        return Batch(src=src.clone(), src_length=n_tok + 1, src_prompt_mask=None, trg=prm.clone(), trg_length=prm_len.clone(),
                     trg_prompt_mask=prm_mask.clone(), indices=torch.arange(B), device=torch.device("cpu"), pad_index=1,
                     eos_index=3, is_train=False, task="MT")

    cases = {
        "plain_greedy": (plain_batch, dict(beam_size=1, beam_alpha=-1)),
        "plain_beam": (plain_batch, dict(beam_size=beam, beam_alpha=alpha, n_best=1)),
        "rep_greedy": (plain_batch, dict(beam_size=1, beam_alpha=-1, repetition_penalty=1.5)),
        "rep_beam": (plain_batch, dict(beam_size=beam, beam_alpha=alpha, n_best=2, repetition_penalty=1.5)),
        "ngram_greedy": (plain_batch, dict(beam_size=1, beam_alpha=-1, no_repeat_ngram_size=2)),
        "ngram_beam": (plain_batch, dict(beam_size=beam, beam_alpha=alpha, n_best=1, no_repeat_ngram_size=2)),
        "both_beam": (plain_batch, dict(beam_size=3, beam_alpha=alpha, n_best=1, no_repeat_ngram_size=3, repetition_penalty=1.2)),
        "prompt_greedy": (prompted_batch, dict(beam_size=1, beam_alpha=-1)),
        "prompt_beam": (prompted_batch, dict(beam_size=beam, beam_alpha=alpha, n_best=1)),
        "prompt_rep_beam": (prompted_batch, dict(beam_size=beam, beam_alpha=alpha, n_best=1, repetition_penalty=1.5)),
    }
    meta = {}
    with torch.no_grad():
        for cname, (mk, kw) in cases.items():
            ids, scores, _ = search(model, mk(), max_output_length=14, return_prob="hyp", generate_unk=False, **kw)
            out[f"{cname}.ids"], out[f"{cname}.scores"] = ids, scores
            meta[cname] = {"prompted": mk is prompted_batch, **kw}
    # the options must have changed something, otherwise the fixture pins nothing
    assert not np.array_equal(out["plain_greedy.ids"], out["rep_greedy.ids"])
    assert not np.array_equal(out["plain_greedy.ids"], out["ngram_greedy.ids"])
    assert not np.array_equal(out["plain_beam.ids"], out["ngram_beam.ids"])
    assert not np.array_equal(out["plain_beam.ids"], out["prompt_beam.ids"])
    np.savez_compressed(OUT / f"{name}.npz", **out)
    (OUT / "search_wrapper_cases.json").write_text(json.dumps(meta, indent=1, sort_keys=True) + "\n")
    print(name, {k: out[f"{k}.ids"].shape for k in cases})


def golden_units():
    """Operator-level captures + constants that the reference's own unit tests assert."""
    from joeynmt.data_augmentation import CMVN, SpecAugment
    from joeynmt.encoders import Conv1dSubsampler
    from joeynmt.helpers_for_audio import get_n_frames, pad_features
    from joeynmt.loss import XentCTCLoss, XentLoss
    out = {}
    # test_transformer_encoder.py:118-154 procedure (seed 42, uniform(-0.5,0.5)) -> subsampler known answer
    torch.manual_seed(42)
    sub = Conv1dSubsampler(in_channels=10, mid_channels=24, out_channels=12, kernel_sizes=[3, 3])
    for p in sub.parameters():
        torch.nn.init.uniform_(p, -0.5, 0.5)
    x = torch.rand(size=(2, 9, 10))
    xl = torch.Tensor([9, 9]).int()
    y, yl = sub(x, xl)
    out.update({f"sub.sd.{k}": v for k, v in np_sd(sub.state_dict()).items()})
    out.update(sub_x=x.numpy(), sub_y=y.detach().numpy(), sub_len=yl.numpy())
    # the first row the reference test hard-codes (test_transformer_encoder.py:134-136), as data
    out["sub_y_ref_row0"] = np.array([-0.4831, -0.0188, -0.0643, 0.2323, 0.1843, -0.0599, 0.0333, -0.0295, 0.0926, 0.0629,
                                      0.4416, -0.3737], dtype=np.float32)
    # subsampled lengths for a sweep of input lengths, k in {3,5}
    lens = torch.arange(1, 400)
    for ks in ([3, 3], [5, 5], [5], [3, 5, 3]):
        s = Conv1dSubsampler(4, 8, 4, ks)
        out["len_" + "_".join(map(str, ks))] = s.get_out_seq_lens_tensor(lens).numpy()
    out["len_in"] = lens.numpy()
    # loss constants of test_loss.py:14-95
    predict = torch.FloatTensor([[[0.1, 0.1, 0.6, 0.1, 0.1]] * 2] * 3)
    targets = torch.LongTensor([[2, 1], [2, 0], [1, 0]])
    v1, = XentLoss(pad_index=0, smoothing=0.4)(predict.log(), trg=targets)
    v0, = XentLoss(pad_index=0, smoothing=0.0)(predict.log(), trg=targets)
    out.update(xent_predict=predict.numpy(), xent_targets=targets.numpy(), xent_s04=v1.item(), xent_s00=v0.item(),
               xent_s04_ref=2.1326, xent_s00_ref=5.6268)
    # XentCTCLoss on random logits (no reference unit test exists; pinned by this capture)
    g = torch.Generator().manual_seed(5)
    B, L, V, T = 4, 6, 11, 15
    logits = torch.randn(B, L, V, generator=g, requires_grad=True)
    ctc_logits = torch.randn(B, T, V, generator=g, requires_grad=True)
    trg = torch.randint(4, V, (B, L), generator=g)
    tl = torch.tensor([6, 4, 5, 2])
    for b in range(B):
        trg[b, tl[b] - 1] = 3
        trg[b, tl[b]:] = 1
    trg[3, 0] = trg[3, 0]  # keep
    in_len = torch.tensor([15, 12, 9, 3])
    src_mask = (torch.arange(T)[None, :] < in_len[:, None]).unsqueeze(1)
    crit = XentCTCLoss(pad_index=1, bos_index=2, smoothing=0.1, ctc_weight=0.3)
    tot, xe, ct = crit(torch.log_softmax(logits, -1), trg=trg, trg_length=tl, src_mask=src_mask,
                       ctc_log_probs=torch.log_softmax(ctc_logits, -1))
    tot.backward()
    out.update(xc_logits=logits.detach().numpy(), xc_ctc_logits=ctc_logits.detach().numpy(), xc_trg=trg.numpy(),
               xc_trg_len=tl.numpy(), xc_in_len=in_len.numpy(), xc_total=tot.item(), xc_xent=xe.item(), xc_ctc=ct.item(),
               xc_dlogits=logits.grad.numpy(), xc_dctc=ctc_logits.grad.numpy())
    # infeasible CTC (target longer than input) -> zero_infinity
    in_len2 = torch.tensor([15, 12, 9, 1])
    mask2 = (torch.arange(T)[None, :] < in_len2[:, None]).unsqueeze(1)
    l2 = ctc_logits.detach().clone().requires_grad_(True)
    _, _, ct2 = crit(torch.log_softmax(logits.detach(), -1), trg=trg, trg_length=tl, src_mask=mask2,
                     ctc_log_probs=torch.log_softmax(l2, -1))
    ct2.backward()
    out.update(xc_in_len_inf=in_len2.numpy(), xc_ctc_inf=ct2.item(), xc_dctc_inf=l2.grad.numpy())
    # CMVN / SpecAugment (no reference unit test: test_tokenizer.py:335 TODO)
    rs = np.random.RandomState(3)
    feat = (rs.randn(57, 80) * 3 + 1).astype(np.float32)
    out.update(cmvn_in=feat, cmvn_out=CMVN()(feat.copy()))
    np.random.seed(42)
    sa = SpecAugment(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=100, time_mask_p=1.0)
    out.update(spec_out=sa(feat.copy()))
    np.random.seed(42)
    feat_short = feat[:7]
    out.update(spec_out_short=sa(feat_short.copy()))
    padded, lengths, _ = pad_features([feat, feat[:20], feat[:33]], embed_size=80, pad_index=1)
    out.update(pad_out=padded, pad_len=np.array(lengths))
    out["n_frames"] = np.array([get_n_frames(n, 16000) for n in (34640, 240000, 16000, 400, 160000, 272000)])
    out["n_frames_in"] = np.array([34640, 240000, 16000, 400, 160000, 272000])
    np.savez_compressed(OUT / "units.npz", **out)
    print("units: xent", v1.item(), v0.item(), "xentctc", tot.item(), xe.item(), ct.item(), "inf ctc", ct2.item())


def golden_audio():
    """fbank pins: the reference's known-answer (test/unit/test_tokenizer.py:318-325: first 10 bins of frame 0 of
    utterance 260-123440-1 after CMVN) and its frame counts (test/data/speech/test.tsv); the wav files themselves
    are test DATA of the reference and are stored as int16 sample arrays."""
    import wave
    out = {}
    names, frames = [], []
    for line in (REF / "test/data/speech/test.tsv").read_text().splitlines()[1:]:
        cols = line.split("\t")
        names.append(cols[0])
        frames.append(int(cols[2]))
    for i in (1, 0, 6):  # keep the fixture small: three utterances
        with wave.open(str(REF / "test/data/speech/wav" / f"{names[i]}.wav"), "rb") as w:
            assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        out[f"pcm_{names[i]}"] = pcm
    out["tsv_n_frames"] = np.array(frames)
    out["tsv_names"] = np.array(names)
    out["tsv_n_samples"] = np.array([0] * len(names))
    for i, n in enumerate(names):
        with wave.open(str(REF / "test/data/speech/wav" / f"{n}.wav"), "rb") as w:
            out["tsv_n_samples"][i] = w.getnframes()
    out["fbank_cmvn_ref_260-123440-1_frame0_bins0_9"] = np.array(
        [-1.0788909, -1.0076448, -1.0421542, -1.0393586, -1.0239305, -0.9921213, -0.95107234, -0.9340749, -0.9119267,
         -0.8962079], dtype=np.float32)
    np.savez_compressed(OUT / "audio.npz", **out)
    print("audio fixture:", {k: getattr(v, 'shape', None) for k, v in out.items()})


def golden_train_steps(name="train_steps", n_updates=3, batch_multiplier=2):
    """The reference's own _train_step + update tail (training.py:541-596,436-456) replayed with its builders: tiny
    pre-LN model, 6 micro-batches = 3 updates (batch_multiplier 2, normalization 'batch'), clip 1.0, AdamW,
    warmupinversesquareroot (warmup 2).  Captures per-micro-batch losses, per-update grad norms and learning rates, and
    the final parameters."""
    from joeynmt.batch import Batch
    from joeynmt.builders import build_gradient_clipper, build_optimizer, build_scheduler
    from joeynmt.model import build_model
    torch.manual_seed(42)
    cfg = tiny_cfg("pre")
    model = build_model(cfg, src_vocab=None, trg_vocab=make_vocab(20))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    tcfg = {"optimizer": "adamw", "learning_rate": 2.0e-3, "weight_decay": 0.0, "adam_betas": [0.9, 0.98],
            "scheduling": "warmupinversesquareroot", "learning_rate_warmup": 2, "learning_rate_min": 1.0e-6,
            "clip_grad_norm": 1.0, "clip_grad_val": None}
    clipper = build_gradient_clipper(tcfg)
    opt = build_optimizer(tcfg, model.parameters())
    sched, step_at = build_scheduler(tcfg, optimizer=opt, scheduler_mode="min", hidden_size=16)
    assert step_at == "step"
    out = {f"sd0.{k}": v for k, v in np_sd(model.state_dict()).items()}
    steps, losses, norms, lrs = 0, [], [], []
    model.train()  # dropout is 0 in tiny_cfg, so train mode is deterministic
    for i in range(n_updates * batch_multiplier):
        src, lengths, trg, trg_len = synth_batch(3, 37, 8, 20, 3, 6, seed=100 + i)
        out.update({f"mb{i}.src": src.numpy(), f"mb{i}.src_length": lengths.numpy(), f"mb{i}.trg": trg.numpy(),
                    f"mb{i}.trg_length": trg_len.numpy()})
        batch = Batch(src=src, src_length=lengths, src_prompt_mask=None, trg=trg, trg_length=trg_len, trg_prompt_mask=None,
                      indices=torch.arange(3), device=torch.device("cpu"), pad_index=1, eos_index=3, is_train=True, task="S2T")
        batch.sort_by_src_length()
        kw = dict(vars(batch))
        kw["repad"] = False
        total, nll, ctc, ncor = model(return_type="loss", **kw)
        norm = batch.normalize(total, "batch", 1, batch_multiplier)
        norm.backward()
        losses.append([norm.item(), batch.normalize(nll, "batch", 1, batch_multiplier).item(),
                       batch.normalize(ctc, "batch", 1, batch_multiplier).item(), ncor.item()])
        if (i + 1) % batch_multiplier == 0:
            norms.append(float(clipper(parameters=model.parameters())))
            lrs.append(opt.param_groups[0]["lr"])  # the rate this update is taken with
            opt.step()
            sched.step(steps)
            model.zero_grad(set_to_none=True)
            steps += 1
            if steps == n_updates - 1:
                # a checkpoint in the layout TrainManager._save_checkpoint writes (training.py:166-177), taken after the
                # second update: resuming from it and running the third must land on sd1 (tests/test_hip_train_step.py)
                state = {"model_state": {k: v.clone() for k, v in model.state_dict().items()},
                         "optimizer_state": copy.deepcopy(opt.state_dict()), "scaler_state": None,
                         "scheduler_state": dict(sched.state_dict()), "train_iter_state": None,
                         "stats_state": {"steps": steps, "is_min_lr": False, "is_max_update": False, "total_tokens": 0,
                                         "best_ckpt_iter": 0, "minimize_metric": True, "total_correct": 0}}
                torch.save(state, OUT / "ref_checkpoint_after2.ckpt")
    out["param_order"] = np.array([n for n, _ in model.named_parameters()])
    out.update({f"sd1.{k}": v for k, v in np_sd(model.state_dict()).items()})
    out.update(losses=np.array(losses), grad_norms=np.array(norms), lrs=np.array(lrs), lr_next=np.float64(opt.param_groups[0]["lr"]))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(name, "losses", np.array(losses)[:, 0], "norms", norms, "lrs", lrs)


def golden_conformer(name="conformer"):
    """The reference's ConformerEncoder (encoders.py:376-445; not reachable from build_model, constructed directly): tiny
    2-layer encoder, train-mode forward (BatchNorm batch statistics) with all parameter gradients of a random linear
    functional of the output, the running statistics after that forward, and the eval-mode forward."""
    from joeynmt.encoders import ConformerEncoder
    torch.manual_seed(42)
    out = {}
    for ln in ("pre", "post"):
        enc = ConformerEncoder(hidden_size=16, ff_size=32, num_layers=2, num_heads=2, dropout=0.0, emb_dropout=0.0, in_channels=8,
                               conv_channels=24, conv_kernel_sizes=[5, 5], depthwise_conv_kernel_size=5, alpha=1.0, layer_norm=ln)
        with torch.no_grad():
            g = torch.Generator().manual_seed(123)
            for n, p in enc.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.3 * torch.randn(p.shape, generator=g))
        src, lengths, _, _ = synth_batch(3, 37, 8, 20, 3, 6, seed=11)
        proj = torch.randn(16, generator=torch.Generator().manual_seed(5))
        pre = f"{ln}."
        out.update({f"{pre}sd0.{k}": v for k, v in np_sd(enc.state_dict()).items() if not k.endswith("pe.pe")})  # pe: analytic
        out.update({f"{pre}src": src.numpy(), f"{pre}src_length": lengths.numpy(), f"{pre}proj": proj.numpy()})
        enc.train()
        y, _, mask = enc(src, lengths, None)
        (y * proj).sum().backward()
        out.update({f"{pre}out_train": y.detach().numpy(), f"{pre}mask": mask.numpy()})
        for n, p in enc.named_parameters():
            out[f"{pre}grad.{n}"] = p.grad.numpy().copy()
        out.update({f"{pre}sd1.{k}": v for k, v in np_sd(enc.state_dict()).items() if "running" in k or "num_batches" in k})
        enc.eval()
        with torch.no_grad():
            y2, _, _ = enc(src, lengths, None)
        out[f"{pre}out_eval"] = y2.numpy()
        print(name, ln, "train out", float(y.abs().mean()), "eval out", float(y2.abs().mean()))
    np.savez_compressed(OUT / f"{name}.npz", **out)


def golden_search_options(name="search_options"):
    """The decoding options the reference's own unit tests pin (test/unit/test_search.py:101-500): forced-decoding prompts,
    repetition penalty (target and source side), n-gram blocking, generate_unk, attention export - on the decoder-only
    model those tests build (TestSearchTransformer._build after set_seed(42); vocabulary of 8 with <sep> = 4 and two
    language tags).  Stored per case: the arguments, the outputs of the reference's greedy / beam_search (full precision)
    and the constants the reference test hard-codes (transcribed as data, `exp_*`); the generator asserts that the two agree
    the way the reference test does (ids exact, scores / attention 1e-4)."""
    from types import SimpleNamespace

    from joeynmt.decoders import TransformerDecoder
    from joeynmt.embeddings import Embeddings
    from joeynmt.helpers import set_seed
    from joeynmt.model import Model
    from joeynmt.search import beam_search, greedy
    from joeynmt.vocabulary import Vocabulary
    special = SimpleNamespace(unk_token="<unk>", pad_token="<pad>", bos_token="<s>", eos_token="</s>", sep_token="<sep>", unk_id=0,
                              pad_id=1, bos_id=2, eos_id=3, sep_id=4, lang_tags=["<de>", "<en>"])
    vocab = Vocabulary(tokens=["word"], cfg=special)
    assert len(vocab) == 8
    autocast = {"device_type": "cpu", "enabled": False}

    def build(batch_size):
        set_seed(42)
        emb = Embeddings(embedding_dim=12, vocab_size=8, padding_idx=1)
        dec = TransformerDecoder(num_layers=3, num_heads=4, hidden_size=12, ff_size=24, dropout=0.0, emb_dropout=0.0, vocab_size=8,
                                 layer_norm="pre")
        enc_out = torch.rand(size=(batch_size, 4, 12))
        for p in dec.parameters():
            torch.nn.init.trunc_normal_(p, mean=0.0, std=1.0, a=-2.0, b=2.0)
        src_mask = torch.ones(size=(batch_size, 1, 4)) == 1
        model = Model(encoder=None, decoder=dec, src_embed=emb, trg_embed=emb, src_vocab=vocab, trg_vocab=vocab)
        model.eval()
        return src_mask, model, enc_out

    out = {"vocab_size": np.int64(8), "specials": np.array(model_specials := [0, 1, 2, 3, 4]), "lang_tags": np.array([5, 6])}
    T = torch.tensor

    def t2n(x):
        return None if x is None else (x.numpy() if torch.is_tensor(x) else np.asarray(x))

    def record(case, res, exp):
        names = ("ids", "scores", "att")
        for nm, r in zip(names, res):
            if r is not None:
                out[f"{case}.{nm}"] = t2n(r)
        for nm, e in exp.items():
            out[f"{case}.exp_{nm}"] = t2n(e)
            got = torch.as_tensor(out[f"{case}.{nm}"])
            if nm == "ids":
                assert torch.equal(got.long(), e.long()), (case, got, e)
            else:
                torch.testing.assert_close(got.float(), e.float(), rtol=1e-4, atol=1e-4)

    for bs in (2, 3):
        src_mask, model, enc_out = build(bs)
        assert model.specials == model_specials and model.lang_tags == [5, 6] and model.sep_index == 4
        out.update({f"bs{bs}.sd.{k}": v for k, v in np_sd(model.state_dict()).items() if not k.endswith("pe.pe") and not k.startswith("src_embed")})
        out[f"bs{bs}.encoder_output"] = enc_out.numpy()
    exp_ids = T([[0, 0, 0], [0, 0, 7]])
    exp_scores = T([[-0.5425, -0.4908, -0.5439], [-0.8726, -0.9898, -0.9668]])
    src_mask, model, enc_out = build(2)
    common = dict(src_mask=src_mask, model=model, encoder_output=enc_out, encoder_hidden=None, autocast=autocast)
    with torch.no_grad():
        record("greedy", greedy(max_output_length=3, return_prob="hyp", **common), dict(ids=exp_ids, scores=exp_scores))
        prompt, pmask = T([[2, 7, 7, 4], [0, 7, 4, 1]]), T([[1, 1, 1, 1], [1, 1, 1, 0]])
        out["prompt"], out["prompt_mask"] = prompt.numpy(), pmask.numpy()
        record("greedy_prompt", greedy(max_output_length=7, return_prob="hyp", return_attention=True, decoder_prompt=prompt,
                                       trg_prompt_mask=pmask, **common),
               dict(ids=T([[7, 7, 4, 0, 0, 0, 0], [7, 4, 0, 0, 0, 0, 0]]),
                    scores=T([[0.0000, 0.0000, 0.0000, -0.4631, -0.3289, -0.3042, -0.3404],
                              [0.0000, 0.0000, -0.7532, -0.6751, -0.5730, -0.4957, -0.5718]]),
                    att=T([[[0.0000, 0.0000, 0.0000, 0.0000], [0.0000, 0.0000, 0.0000, 0.0000], [0.0000, 0.0000, 0.0000, 0.0000],
                            [0.3926, 0.2844, 0.3191, 0.0039], [0.4019, 0.2798, 0.3156, 0.0027], [0.4072, 0.2809, 0.3093, 0.0026],
                            [0.4004, 0.2799, 0.3169, 0.0027]],
                           [[0.0000, 0.0000, 0.0000, 0.0000], [0.0000, 0.0000, 0.0000, 0.0000], [0.3194, 0.0042, 0.4271, 0.2492],
                            [0.3523, 0.0036, 0.3957, 0.2484], [0.3335, 0.0034, 0.4143, 0.2488], [0.3135, 0.0031, 0.4346, 0.2488],
                            [0.3322, 0.0034, 0.4158, 0.2486]]])))
        record("beam1", beam_search(beam_size=1, max_output_length=3, alpha=0.0, n_best=1, return_prob="hyp", **common),
               dict(ids=exp_ids, scores=T([[-1.5772], [-2.8292]])))
        record("beam7", beam_search(beam_size=7, max_output_length=3, alpha=1.0, n_best=5, return_prob="hyp", **common),
               dict(ids=T([[0, 0, 0], [0, 0, 7], [0, 7, 0], [7, 0, 0], [7, 0, 7], [0, 0, 7], [7, 0, 0], [0, 0, 0], [0, 7, 0], [7, 0, 7]]),
                    scores=T([[-1.1829], [-1.6948], [-1.7128], [-2.0805], [-2.9899], [-2.1219], [-2.1881], [-2.1931], [-2.3707],
                              [-2.4195]])))
        record("beam7_prompt", beam_search(beam_size=7, max_output_length=10, alpha=1.0, n_best=5, return_prob="hyp",
                                           decoder_prompt=prompt, trg_prompt_mask=pmask, **common),
               dict(ids=T([[7, 7, 4, 0, 0, 0, 0, 0, 0, 0], [7, 7, 4, 0, 0, 0, 0, 0, 7, 0], [7, 7, 4, 0, 0, 0, 0, 0, 0, 7],
                           [7, 7, 4, 0, 0, 0, 0, 0, 0, 0], [7, 7, 4, 0, 0, 0, 0, 7, 0, 0], [7, 4, 0, 0, 0, 0, 0, 0, 0, 0],
                           [7, 4, 0, 0, 0, 0, 0, 0, 0, 7], [7, 4, 0, 0, 0, 0, 0, 0, 7, 7], [7, 4, 0, 0, 0, 0, 0, 0, 7, 0],
                           [7, 4, 0, 0, 0, 0, 0, 7, 7, 0]]),
                    scores=T([[-1.2273], [-1.3972], [-1.3999], [-1.4480], [-1.6088], [-2.2729], [-2.2850], [-2.3435], [-2.4353],
                              [-2.4680]])))
        src_tokens_b = T([[5, 5, 4], [5, 6, 6]]).long()
        out["beam7_penalty.src_tokens"] = src_tokens_b.numpy()
        record("beam7_penalty", beam_search(beam_size=7, max_output_length=3, alpha=1.0, n_best=5, return_prob="hyp",
                                            encoder_input=src_tokens_b, repetition_penalty=1.5, **common),
               dict(ids=T([[0, 0, 0], [0, 7, 0], [0, 0, 7], [7, 0, 0], [7, 0, 7], [7, 0, 0], [0, 0, 7], [7, 0, 7], [0, 7, 0], [0, 0, 0]]),
                    scores=T([[-1.5709], [-1.8617], [-1.8788], [-2.2284], [-3.5925], [-2.4791], [-2.4931], [-2.8261], [-2.8357],
                              [-2.9624]])))
        record("greedy_ngram", greedy(max_output_length=7, return_prob="hyp", encoder_input=None, no_repeat_ngram_size=3, **common),
               dict(ids=T([[0, 0, 0, 0, 0, 0, 0], [0, 0, 7, 0, 1, 0, 0]]),
                    scores=T([[-0.5425, -0.4908, -0.5439, -0.7328, -0.6922, -0.6422, -0.6464],
                              [-0.8726, -0.9898, -0.9668, -1.3988, -0.6783, -1.0269, -0.6804]])))
        record("beam3_ngram", beam_search(beam_size=3, max_output_length=7, alpha=1.0, n_best=3, return_prob="hyp", encoder_input=None,
                                          no_repeat_ngram_size=3, **common),
               dict(ids=T([[0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 7], [0, 0, 0, 0, 0, 7, 7], [7, 0, 0, 0, 0, 0, 7],
                           [0, 0, 0, 7, 3, 1, 1], [7, 0, 0, 0, 0, 0, 0]]),
                    scores=T([[-2.1454], [-2.4287], [-2.4680], [-3.2931], [-3.3489], [-3.4080]])))
        # batch of 3 (a different random decoder: the encoder-output draw precedes the parameter draw)
        src_mask, model, enc_out = build(3)
        common = dict(src_mask=src_mask, model=model, encoder_output=enc_out, encoder_hidden=None, autocast=autocast)
        record("greedy_nounk", greedy(max_output_length=3, generate_unk=False, **common), dict(ids=T([[1, 1, 1], [1, 1, 1], [1, 1, 1]])))
        record("greedy_nounk_penalty", greedy(max_output_length=3, generate_unk=False, encoder_input=None, repetition_penalty=1.5, **common),
               dict(ids=T([[1, 1, 1], [1, 1, 1], [1, 1, 3]])))
        src_tokens = T([[4, 3, 1, 1], [5, 4, 3, 1], [5, 5, 6, 3]]).long()
        out["greedy_src_penalty.src_tokens"] = src_tokens.numpy()
        common["src_mask"] = (src_tokens != 1).unsqueeze(1)
        record("greedy_src_penalty", greedy(max_output_length=3, generate_unk=False, encoder_input=src_tokens, repetition_penalty=1.5,
                                            return_attention=True, **common),
               dict(ids=T([[1, 7, 3], [1, 7, 1], [1, 1, 1]]),
                    att=T([[[0.5292, 0.4708, 0.0000, 0.0000], [0.5269, 0.4731, 0.0000, 0.0000], [0.5264, 0.4736, 0.0000, 0.0000]],
                           [[0.3075, 0.6322, 0.0602, 0.0000], [0.2890, 0.6350, 0.0760, 0.0000], [0.3343, 0.6314, 0.0343, 0.0000]],
                           [[0.2648, 0.1326, 0.5174, 0.0852], [0.2642, 0.1167, 0.5365, 0.0825], [0.2646, 0.1125, 0.5421, 0.0809]]])))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(name, sorted(k for k in out if k.endswith(".ids")))


def _ddp_worker(rank, world, port, q):
    import os

    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import_reference()
    from joeynmt.helpers_for_ddp import DistributedSubsetSampler, ddp_merge, ddp_reduce
    dist.init_process_group("gloo", rank=rank, world_size=world)  # ddp_setup hard-codes nccl (helpers_for_ddp.py:17-38)
    res = {}
    # ddp_merge: ragged 2-D ids and 3-D attention-like tensors, pad_index -1 / 1 / 0.0
    g = torch.Generator().manual_seed(100 + rank)
    a2 = torch.randint(4, 20, (2 + rank, 5 - 2 * rank), generator=g)
    a3 = torch.rand((3 - rank, 2 + rank, 4), generator=g)
    res["merge2_in"], res["merge3_in"] = a2.numpy(), a3.numpy()
    res["merge2"] = ddp_merge(a2, -1).numpy()
    res["merge2_pad1"] = ddp_merge(a2, 1).numpy()
    res["merge3"] = ddp_merge(a3, 0.0).numpy()
    # ddp_reduce: 0-d tensor, 1-d tensor, python int
    res["reduce0"] = ddp_reduce(torch.tensor(1.5 + rank)).numpy()
    res["reduce1"] = ddp_reduce(torch.tensor([1.0 + rank, 2.0, -3.0 * rank])).numpy()
    res["reduce_int"] = ddp_reduce(7 + rank, torch.device("cpu"), torch.long).numpy()

    class DS:
        def __init__(self, n):
            self.indices = list(range(n))
            self.random_subset = -1

        def __len__(self):
            return len(self.indices)

        def reset_indices(self):
            self.indices = list(range(len(self.indices)))

    for n in (11, 16):
        ds = DS(n)
        sampler = DistributedSubsetSampler(ds, shuffle=True, drop_last=True, generator=torch.Generator().manual_seed(42))
        res[f"sampler{n}_epoch0"] = np.array(list(iter(sampler)))
        res[f"sampler{n}_epoch1"] = np.array(list(iter(sampler)))
        res[f"sampler{n}_len"] = np.int64(len(sampler))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def golden_ddp(name="ddp", world=2):
    """The reference's ddp_merge / ddp_reduce / DistributedSubsetSampler (helpers_for_ddp.py:58-174,244-342) run under a
    2-rank gloo group on the CPU: per-rank inputs and outputs."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    out = {"world": np.int64(world)}
    for r in range(world):
        out.update({f"rank{r}.{k}": v for k, v in got[r].items()})
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items() if k.startswith("rank0.")})


class _ToyDataset:
    """what the batch samplers need of a dataset: `indices`, `__len__`, `__getitem__` -> (idx, src, trg), src None = filtered"""

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


def golden_text_tail(name="text_tail"):
    """f1 / f4 captures (JSON, they are strings and index lists):
    * `post_process` of the reference's SentencePieceTokenizer (its own toy model test/data/toy/sp200.model), SubwordNMTTokenizer
      and BasicTokenizer (word / char) on hypothesis token lists with special tokens, a prompt marker and unknowns;
    * the batches its TokenBatchSampler / SentenceBatchSampler cut from a shuffled toy dataset over two epochs."""
    import json

    from types import SimpleNamespace

    from joeynmt.datasets import SentenceBatchSampler, TokenBatchSampler
    from joeynmt.helpers_for_ddp import RandomSubsetSampler
    from joeynmt.tokenizers import BasicTokenizer, SentencePieceTokenizer, SubwordNMTTokenizer
    from joeynmt.vocabulary import Vocabulary
    toy = REF / "test" / "data" / "toy"
    special = SimpleNamespace(unk_token="<unk>", pad_token="<pad>", bos_token="<s>", eos_token="</s>", sep_token="<sep>", unk_id=0,
                              pad_id=1, bos_id=2, eos_id=3, sep_id=4, lang_tags=["<de>", "<en>"])
    out = {"specials": special.__dict__}
    sents = [l.strip() for l in (toy / "dev.en").read_text(encoding="utf-8").splitlines()[:12] if l.strip()]
    # ---- SentencePiece
    spt = SentencePieceTokenizer(level="bpe", normalize=True, model_file=str(toy / "sp200.model"))
    pieces = [spt(x) for x in sents]
    vocab = Vocabulary(tokens=sorted({p for ps in pieces for p in ps}), cfg=special)
    BasicTokenizer.set_vocab(spt, vocab)  # SentencePieceTokenizer.set_vocab also calls spm.SetVocabulary (gone in sentencepiece 0.2)
    cases = []
    for i, ps in enumerate(pieces):
        seq = list(ps)
        if i % 3 == 0:
            seq = ["<en>"] + seq + ["</s>", "<pad>", "<pad>"]
        if i % 3 == 1:
            seq = ["<de>", seq[0], "<sep>"] + seq[1:] + ["<unk>", "</s>"]
        if i % 4 == 2:
            seq.insert(len(seq) // 2, "<unk>")
        for gu in (True, False):
            for cut in (True, False):
                cases.append({"seq": seq, "generate_unk": gu, "cut_at_sep": cut,
                              "out": spt.post_process(list(seq), generate_unk=gu, cut_at_sep=cut)})
    cases.append({"seq": ["<pad>", "</s>"], "generate_unk": True, "cut_at_sep": True, "out": spt.post_process(["<pad>", "</s>"])})
    out["sentencepiece"] = {"normalize": True, "cases": cases, "sentences": sents, "pieces": pieces}
    # ---- subword-nmt (post_process does not touch the BPE object: bypass the constructor, which needs the absent package)
    bpt = SubwordNMTTokenizer.__new__(SubwordNMTTokenizer)
    bpt.level, bpt.lowercase, bpt.normalize, bpt.max_length, bpt.min_length = "bpe", False, False, -1, -1
    bpt.pretokenizer, bpt.separator = "none", "@@"
    bv = Vocabulary(tokens=["he@@", "llo", "wor@@", "ld", "a", "te@@", "st@@"], cfg=special)
    BasicTokenizer.set_vocab(bpt, bv)
    cases = []
    for seq in (["he@@", "llo", "wor@@", "ld", "</s>"], ["<en>", "a", "<sep>", "te@@", "st@@", "</s>", "<pad>"],
                ["he@@", "<unk>", "llo", "wor@@"], ["<s>", "</s>"]):
        for gu in (True, False):
            for cut in (True, False):
                cases.append({"seq": seq, "generate_unk": gu, "cut_at_sep": cut,
                              "out": bpt.post_process(list(seq), generate_unk=gu, cut_at_sep=cut)})
    out["subword_nmt"] = {"normalize": False, "separator": "@@", "cases": cases}
    # ---- word / char
    for level in ("word", "char"):
        bt = BasicTokenizer(level=level, normalize=True)
        toks = [bt(x) for x in sents[:4]]
        wv = Vocabulary(tokens=sorted({t for ts in toks for t in ts}), cfg=special)
        bt.set_vocab(wv)
        cases = []
        for i, ts in enumerate(toks):
            seq = (["<de>", ts[0], "<sep>"] if i % 2 else []) + list(ts) + ["</s>", "<pad>"]
            for gu in (True, False):
                for cut in (True, False):
                    cases.append({"seq": seq, "generate_unk": gu, "cut_at_sep": cut,
                                  "out": bt.post_process(list(seq), generate_unk=gu, cut_at_sep=cut)})
        out[level] = {"normalize": True, "cases": cases, "tokens": toks}
    # ---- batch samplers
    g = torch.Generator().manual_seed(5)
    src_len = torch.randint(50, 1500, (57, ), generator=g).tolist()
    trg_len = torch.randint(3, 80, (57, ), generator=g).tolist()
    drop = [3, 17, 40]
    samplers = {}
    for kind, cls, bs in (("token", TokenBatchSampler, 6000), ("sentence", SentenceBatchSampler, 8)):
        for drop_last in (False, True):
            ds = _ToyDataset(src_len, trg_len, drop)
            base = RandomSubsetSampler(ds, shuffle=True, generator=torch.Generator().manual_seed(42))
            bsamp = cls(base, batch_size=bs, drop_last=drop_last, seed=42)
            epochs = [[list(b) for b in bsamp], [list(b) for b in bsamp]]
            entry = {"batch_size": bs, "drop_last": drop_last, "epochs": epochs}
            if kind == "sentence":
                entry["len"] = len(bsamp)
            samplers[f"{kind}_{int(drop_last)}"] = entry
    out["samplers"] = {"src_len": src_len, "trg_len": trg_len, "drop": drop, "cases": samplers}
    (OUT / f"{name}.json").write_text(json.dumps(out, ensure_ascii=False, indent=0), encoding="utf-8")
    print(name, "sp cases", len(out["sentencepiece"]["cases"]), "token batches", len(samplers["token_0"]["epochs"][0]))


def golden_ref_unit_tests(name="ref_unit_tests"):
    """The known-answer tests of the reference's own suite for the Transformer stacks, replayed with the reference's classes:
    test/unit/test_transformer_encoder.py:31-90 (3 layers, 4 heads of 3, pre-LN, parameters ~ U(-0.5, 0.5) under seed 42) and
    test/unit/test_transformer_decoder.py:45-172 (logits, last-layer cross-attention, hidden states).  Stored: parameters, inputs,
    the reference's outputs - and the first rows of the constants the tests hard-code, which the outputs are asserted to match
    here (the tests' tolerance 1e-4), so the fixture is tied to those constants and not only to this replay."""
    from joeynmt.decoders import TransformerDecoder
    from joeynmt.encoders import TransformerEncoder
    out = {}
    torch.manual_seed(42)  # setUp
    torch.manual_seed(42)  # test_transformer_encoder_forward
    enc = TransformerEncoder(hidden_size=12, ff_size=24, num_layers=3, num_heads=4, dropout=0.0, emb_dropout=0.0, alpha=1.0, layer_norm="pre")
    for p in enc.parameters():
        torch.nn.init.uniform_(p, -0.5, 0.5)
    x = torch.rand(size=(2, 4, 12))
    x_length = torch.Tensor([4, 4]).int()
    mask = torch.ones([2, 1, 4]) == 1
    y, hidden, _ = enc(x, x_length, mask)
    assert hidden is None
    enc_row0 = torch.tensor([1.9728e-01, -1.2042e-01, 8.0998e-02, 1.3411e-03, -3.5960e-01, -5.2988e-01, -5.6056e-01, -3.5297e-01,
                             2.6680e-01, 2.8343e-01, -3.7342e-01, -5.9112e-03])  # output_target[0, 0] of the reference test
    torch.testing.assert_close(y[0, 0].detach(), enc_row0, rtol=1e-4, atol=1e-4)
    out.update({"enc.sd." + k: v.detach().numpy() for k, v in enc.state_dict().items()})
    out.update({"enc.x": x.numpy(), "enc.out": y.detach().numpy(), "enc.test_const_row0": enc_row0.numpy()})

    torch.manual_seed(42)  # setUp of test_transformer_decoder.py
    trg_embed = torch.rand(size=(2, 5, 12))
    dec = TransformerDecoder(num_layers=3, num_heads=4, hidden_size=12, ff_size=24, dropout=0.0, emb_dropout=0.0, vocab_size=7, alpha=1.0,
                             layer_norm="pre")
    encoder_output = torch.rand(size=(2, 4, 12))
    for p in dec.parameters():
        torch.nn.init.uniform_(p, -0.5, 0.5)
    src_mask = torch.ones(size=(2, 1, 4)) == 1
    trg_mask = torch.ones(size=(2, 5, 1)) == 1
    logits, states, att, _, _ = dec(trg_embed, encoder_output, None, src_mask, None, None, trg_mask, return_attention=True)
    dec_row0 = torch.tensor([0.1718, 0.5595, -0.1996, -0.6924, 0.4351, -0.0850, 0.2805])  # output_target[0, 0]
    att_row0 = torch.tensor([0.2494, 0.2482, 0.2419, 0.2605])  # att_target[0, 0]
    st_row0 = torch.tensor([3.7535e-02, 5.3508e-01, 4.9478e-02, -9.1961e-01, -5.3966e-01, -1.0065e-01, 4.3053e-01, -3.0671e-01,
                            -1.2724e-02, -4.1879e-01, 5.9625e-01, 1.1887e-01])  # states_target[0, 0]
    torch.testing.assert_close(logits[0, 0].detach(), dec_row0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(att[0, 0].detach(), att_row0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(states[0, 0].detach(), st_row0, rtol=1e-4, atol=1e-4)
    out.update({"dec.sd." + k: v.detach().numpy() for k, v in dec.state_dict().items()})
    out.update({"dec.trg_embed": trg_embed.numpy(), "dec.memory": encoder_output.numpy(), "dec.logits": logits.detach().numpy(),
                "dec.att": att.detach().numpy(), "dec.states": states.detach().numpy(), "dec.test_const_logits_row0": dec_row0.numpy(),
                "dec.test_const_att_row0": att_row0.numpy(), "dec.test_const_states_row0": st_row0.numpy()})
    # test_model_init.py:81-117: DeepNet residual scale of a 6 + 6 stack under `xavier_normal`
    out["deepnet_alpha_6_6"] = np.array([1.417938140685523, 2.0597671439071177])
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, len(out), "arrays")


def golden_frontend_general():
    """SpeechProcessor.__call__'s cmvn / specaugment block (tokenizers.py:480-492) on the reference's own classes, for the
    settings beside the configured one: CMVN after SpecAugment, more than two masks of a kind, SpecAugment without CMVN
    (local-mean fill), evaluation-time truncation in front of it all.  Three utterances, np.random seeded once per case."""
    from joeynmt.data_augmentation import CMVN, SpecAugment
    rs = np.random.RandomState(8)
    # (long enough that two 30-frame time masks leave most frames alone: with nearly every frame at the fill value the reference's
    # float32 `square_sums / n - mean**2` is cancellation noise, and no implementation that sums differently can reproduce it)
    utts = [(rs.randn(n, 80) * 2.5 + 0.7).astype(np.float32) for n in (161, 123, 140)]
    cases = {
        "after": (dict(norm_means=True, norm_vars=True, before=False), dict(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=30, time_mask_p=1.0), None),
        "many": (dict(norm_means=True, norm_vars=True, before=True), dict(freq_mask_n=3, freq_mask_f=15, time_mask_n=4, time_mask_t=12, time_mask_p=1.0), None),
        "many_after": (dict(norm_means=True, norm_vars=False, before=False), dict(freq_mask_n=4, freq_mask_f=10, time_mask_n=3, time_mask_t=9, time_mask_p=0.5), None),
        "nocmvn": (None, dict(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=30, time_mask_p=1.0), None),
        "after_trunc": (dict(norm_means=True, norm_vars=True, before=False), dict(freq_mask_n=3, freq_mask_f=20, time_mask_n=1, time_mask_t=10, time_mask_p=1.0), 135),
    }
    out = {f"in{i}": u for i, u in enumerate(utts)}
    for name, (ck, sk, max_len) in cases.items():
        cmvn = CMVN(**ck) if ck is not None else None
        sa = SpecAugment(**sk)
        np.random.seed(77)
        for i, item in enumerate(utts):
            item = item.copy()
            if max_len is not None and item.shape[0] > max_len:
                item = item[:max_len, :]
            if cmvn and cmvn.before:
                item = cmvn(item)
            item = sa(item)
            if cmvn and not cmvn.before:
                item = cmvn(item)
            out[f"{name}_{i}"] = item.astype(np.float32)
    np.savez_compressed(OUT / "frontend_general.npz", **out)
    (OUT / "frontend_general.json").write_text(json.dumps({k: dict(cmvn=v[0], specaugment=v[1], max_length=v[2]) for k, v in cases.items()}, indent=1))
    print("frontend_general:", sorted(cases))


def golden_predict_loss(name="predict_loss", V=20):
    """The validation-loss / reference-scoring leg of the reference's `predict` batch loop (prediction.py:165-200) on the tiny
    golden model (weights of model_pre.npz), two batches, eval mode, no process group (ddp_reduce / ddp_merge are identities then,
    helpers_for_ddp.py:88-174; their 2-rank behaviour is pinned by ddp.npz).  What the loop computes per batch:
      batch.sort_by_src_length(); model(return_type="loss", return_prob=..., return_attention=..., **vars(batch)) under no_grad;
      ddp_reduce of the loss, n_correct and batch.ntokens; batch.normalize(x, "sum", n_gpu); running totals;
      return_prob == "ref": ddp_merge(log_probs), ddp_merge(batch.trg) -> Batch.score(log_probs, trg, pad_index).
    NOTE (reference quirk, kept out of the fixture's expectations): the loop asks for return_type="loss", whose 4-tuple carries the
    loss components in slots 1-2, not log-probabilities - `Batch.score` on them cannot work; log-probabilities come with
    return_type="loss_probs" (model.py:148-150).  The fixture stores both calls' outputs; the scoring uses "loss_probs"."""
    from joeynmt.batch import Batch
    from joeynmt.helpers_for_ddp import ddp_merge, ddp_reduce
    from joeynmt.model import build_model
    g0 = dict(np.load(OUT / "model_pre.npz"))
    cfg = tiny_cfg("pre")
    model = build_model(cfg, src_vocab=None, trg_vocab=make_vocab(V))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g0.items() if k.startswith("sd.")})
    model.eval()
    dev = torch.device("cpu")
    out = {}
    totals = dict(loss=0.0, n_correct=0, ntokens=0, nseqs=0)
    for bi, (B, T, seed) in enumerate([(3, 37, 7), (4, 29, 21)]):
        src, lengths, trg, trg_len = synth_batch(B, T, cfg["encoder"]["in_channels"], V, 3, 6, seed)
        batch = Batch(src=src, src_length=lengths, src_prompt_mask=None, trg=trg, trg_length=trg_len, trg_prompt_mask=None,
                      indices=torch.arange(B), device=dev, pad_index=1, eos_index=3, is_train=False, task="S2T")
        out.update({f"b{bi}.src": src.numpy(), f"b{bi}.src_length": lengths.numpy(), f"b{bi}.trg_full": trg.numpy(),
                    f"b{bi}.trg_length_full": trg_len.numpy()})
        batch_nseqs = ddp_reduce(batch.nseqs, dev, torch.long).item()
        reverse_index = batch.sort_by_src_length()
        with torch.no_grad():
            batch_loss, s1, s2, n_correct = model(return_type="loss", return_prob="ref", return_attention=False, **vars(batch))
            _, log_probs, ctc_log_probs, n_correct2 = model(return_type="loss_probs", return_prob="ref", return_attention=False, **vars(batch))
        assert int(n_correct) == int(n_correct2)
        batch_loss, n_correct = ddp_reduce(batch_loss), ddp_reduce(n_correct)
        batch_ntokens = ddp_reduce(batch.ntokens, dev, torch.long)
        batch_loss = batch.normalize(batch_loss, "sum", n_gpu=1)
        n_correct = batch.normalize(n_correct, "sum", n_gpu=1)
        log_probs_m, trg_m = ddp_merge(log_probs, 0.0), ddp_merge(batch.trg, model.pad_index)
        ref_scores = batch.score(log_probs_m, trg_m, model.pad_index)
        totals["loss"] += batch_loss.item()
        totals["n_correct"] += n_correct.item()
        totals["ntokens"] += batch_ntokens.item()
        totals["nseqs"] += batch_nseqs
        out.update({f"b{bi}.reverse_index": np.asarray(reverse_index), f"b{bi}.loss": np.float64(batch_loss.item()),
                    f"b{bi}.slot1": np.float64(s1.item()), f"b{bi}.slot2": np.float64(s2.item()),
                    f"b{bi}.n_correct": np.int64(n_correct.item()), f"b{bi}.ntokens": np.int64(batch_ntokens.item()),
                    f"b{bi}.log_probs": log_probs.numpy(), f"b{bi}.ctc_log_probs": ctc_log_probs.numpy(),
                    f"b{bi}.trg_sorted": batch.trg.numpy(), f"b{bi}.n_rows": np.int64(len(ref_scores))})
        for i, row in enumerate(ref_scores):  # ragged: one array per sentence, in the SORTED order (un-sorted by the caller)
            out[f"b{bi}.ref_scores.{i}"] = np.asarray(row, dtype=np.float64)
    out.update({f"total.{k}": np.float64(v) for k, v in totals.items()})
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(name, totals)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    import_reference()
    torch.set_num_threads(4)
    jobs = {
        "units": golden_units, "audio": golden_audio,
        "model_pre": lambda: golden_model("model_pre", tiny_cfg("pre")),
        "model_post": lambda: golden_model("model_post", tiny_cfg("post", act="gelu")),
        "model_deepnet": lambda: golden_model("model_deepnet", tiny_cfg("pre", initializer="xavier_normal", heads=4), ctc_weight=0.1),
        "train_steps": golden_train_steps, "conformer": golden_conformer, "search_options": golden_search_options, "ddp": golden_ddp, "text_tail": golden_text_tail,
        "ref_unit_tests": golden_ref_unit_tests, "model_mt": golden_model_mt, "frontend_general": golden_frontend_general, "search_wrapper": golden_search_wrapper,
        "predict_loss": golden_predict_loss,
    }
    for name in (sys.argv[1:] or list(jobs)):  # `python oracle/make_golden.py search_options ddp` regenerates only those
        jobs[name]()


if __name__ == "__main__":
    main()
