"""GPU: BASELINE config 0 - configs/transformer_small.yaml as the reference builds it (task MT: source embedding table, encoder
without sub-sampler, tied softmax, cross-entropy only) on the reverse task, through the HIP path in fp32, against a capture of
the reference (oracle/make_golden.py:golden_model_mt -> tests/golden/model_mt.npz).  There is no CPU product path: the "CPU"
half of that config is the oracle (tests/test_oracle_golden.py::test_text_source_model_matches_reference)."""
import copy

import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from golden_cfg import mt_cfg

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


def build_mt(device, dtype=torch.float32, dropout=None):
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    g = load_golden("model_mt")
    cfg = copy.deepcopy(mt_cfg())
    if dropout is not None:  # the config trains with dropout 0.1; the capture is an eval-mode pass
        cfg["encoder"]["dropout"] = cfg["decoder"]["dropout"] = dropout
    model = build_model(cfg, Vocabulary.synthetic(30), Vocabulary.synthetic(30))
    model.loss_function = ("crossentropy", 0.0, 0.0)
    assert model.task == "MT" and model.decoder.ctc_output_layer is None
    assert model.decoder.output_layer.weight is model.trg_embed.lut.weight  # tied_softmax: True
    model.load_state_dict(golden_sd(g), strict=True)
    model.finalize(device, dtype).eval()
    return model, g


def mt_batch(g, device):
    from joeys2t_amd.batch import Batch
    b = Batch(src=torch.from_numpy(g["src"]), src_length=torch.from_numpy(g["src_length"]), src_prompt_mask=None,
              trg=torch.from_numpy(g["trg_full"]), trg_length=torch.from_numpy(g["trg_length_full"]), trg_prompt_mask=None,
              indices=torch.arange(g["src"].shape[0]), device=device, pad_index=1, eos_index=3, is_train=True, task="MT")
    for k in ("trg_input", "trg", "trg_length", "trg_mask", "src_mask"):
        assert np.array_equal(getattr(b, k).cpu().numpy(), g[k]), k
    return b


def test_forward_loss_and_gradients(device):
    model, g = build_mt(device)
    b = mt_batch(g, device)
    with torch.no_grad():
        enc, _, src_mask, _ = model(return_type="encode", **vars(b))
        logits, hidden, att, _ = model(return_type="decode", encoder_output=enc, encoder_hidden=None, src_mask=b.src_mask,
                                       trg_input=b.trg_input, unroll_steps=None, trg_mask=b.trg_mask, return_attention=True)
    torch.testing.assert_close(enc.cpu(), torch.from_numpy(g["enc_out"]), **TOL)
    torch.testing.assert_close(logits.cpu(), torch.from_numpy(g["logits"]), **TOL)
    torch.testing.assert_close(hidden.cpu(), torch.from_numpy(g["dec_hidden"]), **TOL)
    torch.testing.assert_close(att.cpu(), torch.from_numpy(g["att"]), **TOL)
    total, nll, ctc, ncor = model(return_type="loss", **vars(b))
    assert nll is None and ctc is None  # plain cross-entropy returns (loss, None, None, n_correct), model.py:137-148
    total.backward()
    assert abs(total.item() - g["loss_total"]) <= 1e-4 * abs(g["loss_total"])
    assert int(ncor.item()) == int(g["n_correct"])
    names = [n for n, _ in model.named_parameters()]
    assert "decoder.output_layer.weight" not in names and "src_embed.lut.weight" in names
    for n, p in model.named_parameters():
        ref = torch.from_numpy(g[f"grad.{n}"])
        scale = ref.abs().max().item() + 1e-6
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 1e-4 * scale + 1e-5, (n, err, scale)


def test_greedy_and_beam5(device):
    from joeys2t_amd.search import search
    model, g = build_mt(device)
    b = mt_batch(g, device)
    ids, scores, _ = search(model, b, max_output_length=31, beam_size=1, beam_alpha=-1, return_prob="hyp")
    assert np.array_equal(ids, g["greedy_ids"])
    np.testing.assert_allclose(scores, g["greedy_scores"], rtol=1e-4, atol=1e-4)
    k = int(g["beam_size"])
    ids, scores, _ = search(model, b, max_output_length=31, beam_size=k, beam_alpha=float(g["beam_alpha"]), n_best=1, return_prob="hyp")
    assert np.array_equal(ids, g["beam_ids"])  # bit-exact beam indices
    np.testing.assert_allclose(scores, g["beam_scores"], rtol=1e-4, atol=1e-4)
    ids, scores, _ = search(model, b, max_output_length=-1, beam_size=k, beam_alpha=float(g["beam_alpha"]), n_best=k, return_prob="hyp")
    assert np.array_equal(ids, g["beam_ids_nbest"])  # max length from the source length x 1.5 (search.py:863-864)
    np.testing.assert_allclose(scores, g["beam_scores_nbest"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_train_step_accumulates_tied_weight_gradient(device, dtype):
    """Through TrainStep (flat gradient store, deferred grouped weight-gradient products): the tied embedding / softmax matrix
    collects the embedding gradient AND the projection's weight gradient in one slice of the flat gradient."""
    from joeys2t_amd.training import TrainStep
    model, g = build_mt(device, dtype, dropout=0.0)  # TrainStep puts the model in train mode
    ts = TrainStep(model, learning_rate=5e-3, adam_betas=(0.9, 0.999), clip_grad_norm=None, scheduling=None, normalization="sum")
    b = mt_batch(g, device)
    ts.micro_step(b, update=False)
    torch.cuda.synchronize()
    gmax = max(float(np.abs(g[f"grad.{n}"]).max()) for n, _ in model.named_parameters())
    for n, p in model.named_parameters():
        ref = torch.from_numpy(g[f"grad.{n}"])
        scale = ref.abs().max().item() + 1e-6
        if dtype == torch.float32:
            err = (p.grad.cpu() - ref).abs().max().item()
            assert err <= 1e-4 * scale + 1e-5, (n, err, scale)
        elif scale >= 1e-2 * gmax:  # bf16 products: direction of every non-negligible gradient (as tests/test_hip_model.py)
            cos = torch.nn.functional.cosine_similarity(p.grad.cpu().flatten(), ref.flatten(), dim=0).item()
            assert cos > 0.97, (n, cos)
    before = model.trg_embed.lut.weight.detach().clone()
    ts.update()
    assert not torch.equal(before, model.trg_embed.lut.weight.detach())
    assert model.decoder.output_layer.weight.data_ptr() == model.trg_embed.lut.weight.data_ptr()
