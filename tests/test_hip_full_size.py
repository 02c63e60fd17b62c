"""GPU: the train step at BASELINE.json's FULL size (configs/librispeech_100h.yaml: 16 + 8 layers, d 512, 4 heads, ff 2048,
V 5000, 32 utterances of up to 15 s = 1498 frames, bf16) - too large for the CPU oracle, so checked through properties the
computation has at any size (reference training.py:541-596, model.py:95-168):

  * the losses and every parameter gradient do not depend on the ORDER of the utterances in the batch;
  * the gradient of a batch is the sum of the gradients of its two halves (normalization "sum": nothing in the path couples
    utterances - no batch statistics in the Transformer stack);
  * normalization "batch" is normalization "sum" divided by nseqs = 32, a power of two: identical bits up to that scale;
  * two runs from the same state agree (no run-to-run drift beyond the atomics' reduction order).

Dropout is off (the masks are a function of row indices, i.e. of the batch order)."""

import pytest
import torch

from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg

pytestmark = pytest.mark.gpu

V, B = 5000, 32


@pytest.fixture(scope="module")
def full_case(device):
    cfg = width_cfg(4, 16, 8)  # librispeech_100h.yaml:110-137: 16 encoder + 8 decoder layers
    torch.manual_seed(21)
    base = make_model(cfg, V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    lengths = [1498] + torch.randint(900, 1499, (B - 1, ), generator=g).tolist()  # ragged, the longest first
    tl = torch.randint(20, 80, (B, ), generator=g).tolist()
    return cfg, sd, synth_batch(V, lengths, tl, seed=13)


def run(cfg, sd, batch, device, normalization="sum", overlap_ctc=True):
    """one micro-step of a fresh model: (loss, nll, ctc, n_correct) sums and the flat gradient"""
    from joeys2t_amd.training import TrainStep
    model = make_model(cfg, V, sd, device, torch.bfloat16, 0.3, train=True)
    step = TrainStep(model, learning_rate=2e-3, adam_betas=(0.9, 0.98), clip_grad_norm=10.0, normalization=normalization,
                     batch_multiplier=1, n_gpu=1, overlap_ctc=overlap_ctc)
    step.micro_step(hip_batch(*batch, device), update=False, sort=False)
    torch.cuda.synchronize()
    stats = step.read_stats()
    return stats, step.store.flat_grad.clone(), step


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def subset(batch, idx):
    src, lengths, trg, tlen = batch
    idx = torch.as_tensor(idx)
    T, L = int(lengths[idx].max()), int(tlen[idx].max())
    return src[idx][:, :T].contiguous(), lengths[idx], trg[idx][:, :L].contiguous(), tlen[idx]


def test_full_size_step_is_invariant_to_batch_order_and_repeatable(device, full_case):
    cfg, sd, batch = full_case
    s0, g0, _ = run(cfg, sd, batch, device)
    s1, g1, _ = run(cfg, sd, batch, device)
    assert rel(g1, g0) <= 1e-5, rel(g1, g0)  # same state, same batch: only the atomics' order may differ
    assert s1["loss"] == pytest.approx(s0["loss"], rel=1e-6)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    sp, gp, _ = run(cfg, sd, subset(batch, perm), device)
    for k in ("loss", "nll", "ctc"):
        assert sp[k] == pytest.approx(s0[k], rel=2e-4), (k, sp[k], s0[k])
    assert sp["n_correct"] == s0["n_correct"] and sp["ntokens"] == s0["ntokens"] and sp["nseqs"] == s0["nseqs"] == B
    # every row goes through the same arithmetic wherever it sits; the sums over rows (weight gradients, losses) change order
    assert rel(gp, g0) <= 2e-3, rel(gp, g0)


def test_full_size_gradient_is_the_sum_over_half_batches(device, full_case):
    # (full batch and halves take the same kernels: the LayerNorm fold applies at every row count)
    cfg, sd, batch = full_case
    s0, g0, _ = run(cfg, sd, batch, device)
    halves = [run(cfg, sd, subset(batch, list(range(h, B, 2))), device) for h in (0, 1)]  # odd / even utterances: both ragged
    gsum = halves[0][1] + halves[1][1]
    for k in ("loss", "nll", "ctc", "n_correct", "ntokens", "nseqs"):
        assert halves[0][0][k] + halves[1][0][k] == pytest.approx(s0[k], rel=2e-4), k
    assert rel(gsum, g0) <= 3e-3, rel(gsum, g0)


@pytest.mark.parametrize("overlap_ctc", [True, False])
def test_full_size_batch_normalisation_is_an_exact_scale(device, full_case, overlap_ctc):
    cfg, sd, batch = full_case
    ss, gs, _ = run(cfg, sd, batch, device, "sum", overlap_ctc)
    sb, gb, step = run(cfg, sd, batch, device, "batch", overlap_ctc)
    assert sb["loss"] * B == pytest.approx(ss["loss"], rel=1e-6)
    # 1/32 commutes with every rounding in the backward pass (bf16 and f32): the same bits, scaled - up to the order in which
    # the atomically accumulated pieces (split-K weight gradients of the sub-sampler, bias row sums) arrive
    assert rel(gb * B, gs) <= 1e-5, rel(gb * B, gs)
    # and the whole update runs at this size: finite parameters, a gradient norm the clip sees
    step.update()
    torch.cuda.synchronize()
    assert torch.isfinite(step.store.flat).all()
    assert float(step.optimizer.norm_clip[0]) == pytest.approx(float(gb.double().norm()), rel=1e-3)


# ------------------------------------------------------------------------------------------------ decoding at full size
def _mustc_decode_case(device, dtype):
    """configs/mustc_st.yaml shapes (12 + 6 layers, d 512, 8 heads of 64), 32 ragged utterances of up to 1498 frames"""
    from test_hip_config_width import MUSTC_ALPHA, set_alpha
    cfg = width_cfg(8, 12, 6, "xavier_normal")
    torch.manual_seed(17)
    model = make_model(cfg, V, None, None, None, 0.1)
    set_alpha(model, *MUSTC_ALPHA)
    model.finalize(device, dtype).eval()
    g = torch.Generator().manual_seed(4)
    lengths = torch.tensor([1498] + torch.randint(700, 1499, (B - 1, ), generator=g).tolist())
    src = torch.randn(B, 1498, 80, generator=g)
    for b in range(B):
        src[b, lengths[b]:] = 1.0
    return model, src, lengths


def _decode_batch(src, lengths, idx, device):
    from joeys2t_amd.batch import Batch
    idx = torch.as_tensor(idx)
    T = int(lengths[idx].max())
    return Batch(src=src[idx][:, :T].contiguous(), src_length=lengths[idx], src_prompt_mask=None, trg=None, trg_length=None,
                 trg_prompt_mask=None, indices=torch.arange(len(idx)), device=device, pad_index=1, eos_index=3, is_train=False, task="S2T",
                 n_gpu=1)


def test_full_size_decoding_properties(device):
    """Beam search / greedy / CTC best path at full size in fp32 (reference search.py:345-912): hypotheses do not depend on
    the order of the utterances in the batch, a beam of one is greedy search, and the key/value-cached step produces the
    hypotheses of the reference's full-prefix decoder pass."""
    import numpy as np
    from joeys2t_amd.search import ctc_greedy, search
    model, src, lengths = _mustc_decode_case(device, torch.float32)
    order = list(range(B))
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(8)).tolist()
    kw = dict(max_output_length=12, beam_alpha=1.0, n_best=1, return_prob="hyp")
    ids0, sc0, _ = search(model, _decode_batch(src, lengths, order, device), beam_size=5, **kw)
    idsp, scp, _ = search(model, _decode_batch(src, lengths, perm, device), beam_size=5, **kw)
    assert np.array_equal(ids0[perm], idsp)  # row i of the permuted batch is utterance perm[i]
    np.testing.assert_allclose(np.asarray(sc0)[perm], np.asarray(scp), rtol=1e-4, atol=1e-4)
    # beam size 1 takes the greedy path in search() (search.py:891-900); the beam machinery with a single hypothesis agrees
    from joeys2t_amd.search import beam_search
    b = _decode_batch(src, lengths, order[:8], device)
    with torch.no_grad():
        enc, hid, mask, _ = model(return_type="encode", **vars(b))
    gids, _, _ = search(model, b, beam_size=1, **kw)
    bids, _, _ = beam_search(model, 1, enc, hid, mask, max_output_length=12, alpha=1.0, n_best=1)
    L = min(gids.shape[1], bids.shape[1])
    assert np.array_equal(np.asarray(gids)[:, :L], np.asarray(bids)[:, :L])
    # KV-cached step against the full-prefix pass
    fids, fsc, _ = search(model, b, beam_size=5, incremental=False, **kw)
    assert np.array_equal(ids0[:8], fids)
    np.testing.assert_allclose(np.asarray(sc0)[:8], np.asarray(fsc), rtol=1e-4, atol=1e-4)
    # CTC best path
    c0, n0 = ctc_greedy(model, _decode_batch(src, lengths, order, device))
    cp, n1 = ctc_greedy(model, _decode_batch(src, lengths, perm, device))
    for i, u in enumerate(perm):
        assert list(cp[i][:n1[i]]) == list(c0[u][:n0[u]]), (i, u)
