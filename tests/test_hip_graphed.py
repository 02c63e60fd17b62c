"""GPU: graphed.GraphedTrainStep - train steps over VARYING batches replayed from one hipGraph per shape bucket - against
TrainStep.micro_step on the same batches un-padded (the reference's per-step procedure, training.py:541-596 over
datasets.py:1249-1295 batches).  Dropout off: the padded, replayed step must give the un-padded step's numbers."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
V = 300


def _cfg():
    from test_hip_config_width import width_cfg
    return width_cfg(4, 2, 1)


def _batches(seed, n, B=4, lo=30000, hi=41000):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        ns = torch.randint(lo, hi, (B, ), generator=g).tolist()
        wave = 0.1 * torch.randn(B, max(ns), generator=g)
        for i, k in enumerate(ns):
            wave[i, k:] = 0.0
        tl = torch.randint(5, 14, (B, ), generator=g).tolist()
        L = max(tl) + 2
        trg = torch.full((B, L), 1, dtype=torch.long)
        for i, k in enumerate(tl):
            trg[i, 0], trg[i, 1 + k] = 2, 3
            trg[i, 1:1 + k] = torch.randint(4, V, (k, ), generator=g)
        out.append((wave, ns, trg, [k + 2 for k in tl]))
    return out


def _proc():
    from joeys2t_amd.tokenizers import SpeechProcessor
    return SpeechProcessor(num_freq=80, min_length=10, max_length=6000,
                           specaugment=dict(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=40, time_mask_p=1.0),
                           cmvn=dict(norm_means=True, norm_vars=True, before=True))


def _make(sd, device, dtype):
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import make_model
    model = make_model(_cfg(), V, sd, device, dtype, 0.3, train=True)
    return TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3, normalization="batch",
                     overlap_ctc=True)


def _plain_run(step, proc, batches, device, dtype):
    """the un-bucketed procedure: features of the batch as it is, Batch(), sort, micro_step with update"""
    from joeys2t_amd.batch import Batch
    np.random.seed(11)
    losses = []
    for wave, ns, trg, tl in batches:
        order = sorted(range(len(ns)), key=lambda i: -ns[i])  # the graphed step sorts on the host; draw the masks in that order too
        wave, ns, trg, tl = wave[order], [ns[i] for i in order], trg[order], [tl[i] for i in order]
        feats, lengths = proc.batch_from_waveforms(wave.to(device), ns, is_train=True, out_dtype=dtype)
        b = Batch(src=feats, src_length=torch.tensor(lengths), src_prompt_mask=None, trg=trg, trg_length=torch.tensor(tl), trg_prompt_mask=None,
                  indices=torch.arange(len(ns)), device=device, pad_index=1, eos_index=3, is_train=True, task="S2T", n_gpu=1)
        step.micro_step(b, sort=False)
        s = step.read_stats()
        losses.append((s["loss"], s["nll"], s["ctc"], s["n_correct"], s["ntokens"], s["lr"]))
    torch.cuda.synchronize()
    return losses, step.store.flat.detach().clone()


def _graphed_run(step, proc, batches, device, dtype, **kw):
    from joeys2t_amd.graphed import GraphedTrainStep
    np.random.seed(11)
    gs = GraphedTrainStep(step, proc, compute_dtype=dtype, **kw)
    losses, how = [], []
    for wave, ns, trg, tl in batches:
        how.append(gs.run(wave.to(device), ns, trg, tl))
        s = gs.read_stats()
        losses.append((s["loss"], s["nll"], s["ctc"], s["n_correct"], s["ntokens"], s["lr"]))
    torch.cuda.synchronize()
    return losses, step.store.flat.detach().clone(), how, gs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_steps_equal_plain_steps(device, dtype):
    from test_hip_config_width import make_model
    torch.manual_seed(3)
    base = make_model(_cfg(), V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    batches = _batches(5, 6)
    proc = _proc()
    ref, flat_ref = _plain_run(_make(sd, device, dtype), proc, batches, device, dtype)
    # one wide bucket: every batch after the first is a replay of the first one's graph
    got, flat, how, gs = _graphed_run(_make(sd, device, dtype), proc, batches, device, dtype, frame_bucket=128, target_bucket=16)
    assert how == ["eager"] + ["replay"] * 5, how
    assert len(gs.buckets) == 1 and gs.counts["captured"] == 1
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    for i, (r, g) in enumerate(zip(ref, got)):
        for k in range(3):
            assert abs(r[k] - g[k]) <= tol * abs(r[k]), (i, k, r, g)
        assert r[4] == g[4] and r[5] == pytest.approx(g[5], rel=1e-6), (i, r, g)  # tokens counted, learning-rate schedule
        if dtype == torch.float32:
            assert r[3] == g[3], (i, r, g)
    # six updates later: the same parameters.  Padded rows change the order of the sums over rows, and the first Adam steps move
    # every coordinate by ~lr whatever its gradient's size: where a gradient is rounding noise around zero (a ReLU unit of the
    # decoder's feed-forward layer that 52 target rows barely switch on) the two runs step apart by up to lr per update.  fp32: such
    # coordinates (measured: 1.4 k of 12.5 M, nearly all in decoder.layers.0 pwff_layer.0.weight, 3.4e-5 of the norm) are set
    # aside and counted, everything else agrees to 1e-5; bf16: all of it within 8e-3
    diff = (flat - flat_ref).abs()
    if dtype == torch.float32:
        apart = diff > 1e-5
        assert float(apart.float().mean()) < 5e-4, int(apart.sum())
        rel = (((flat - flat_ref) * ~apart).norm() / flat_ref.norm()).item()
        assert rel < 1e-5, rel
    else:
        rel = ((flat - flat_ref).norm() / flat_ref.norm()).item()
        assert rel < 8e-3, rel


def test_buckets_are_cut_by_shape_and_evicted(device):
    from test_hip_config_width import make_model
    torch.manual_seed(3)
    base = make_model(_cfg(), V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    proc = _proc()
    batches = _batches(9, 3, lo=30000, hi=33000) + _batches(10, 3, lo=52000, hi=56000) + _batches(11, 2, B=3) + _batches(9, 1, lo=30000, hi=33000)
    got, _, how, gs = _graphed_run(_make(sd, device, torch.bfloat16), proc, batches, device, torch.bfloat16, frame_bucket=64, target_bucket=16,
                                   max_graphs=2)
    assert all(np.isfinite(v[0]) for v in got)
    assert how[0] == "eager" and how[3] == "eager" and how[6] == "eager"  # a new frame bucket, a new utterance count
    assert gs.counts["evicted"] >= 1 and len(gs.buckets) <= 2
    assert how[-1] == "eager"  # its bucket had been evicted by then (max_graphs 2): captured again


def test_bucket_padding_matches_unpadded_features(device):
    """front-end + sub-sampler on a batch padded to a bucket (zeros from the longest utterance on, crop in the first GLU) give
    the un-padded batch's encoder input and mask, position by position"""
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.graphed import GraphedTrainStep
    from test_hip_config_width import make_model
    torch.manual_seed(3)
    base = make_model(_cfg(), V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    step = _make(sd, device, torch.float32)
    proc = _proc()
    proc.specaugment = None
    (wave, ns, trg, tl), = _batches(21, 1)
    order = sorted(range(len(ns)), key=lambda i: -ns[i])
    wave, ns = wave[order], [ns[i] for i in order]
    model = step.model.eval()
    feats, lengths = proc.batch_from_waveforms(wave.to(device), ns, is_train=False)
    with torch.no_grad():
        x_ref, len_ref, mask_ref = model.encoder.subsampler(feats, torch.tensor(lengths, device=device))
    gs = GraphedTrainStep(step, proc, compute_dtype=torch.float32, frame_bucket=128)
    bk = gs._bucket((len(ns), -(-max(lengths) // 128) * 128, 16, 0))
    fr = torch.tensor(lengths)
    bk.host_view("soff").copy_(torch.arange(len(ns)) * bk.wave.shape[1])
    bk.host_view("foff").copy_(torch.cat([torch.zeros(1, dtype=torch.long), fr.cumsum(0)]))
    crop = [int(fr[0])]
    for k in gs.kernel_sizes:
        crop.append((crop[-1] + 2 * (k // 2) - (k - 1) - 1) // 2 + 1)
    bk.host_view("crop").copy_(torch.tensor(crop))
    bk.upload()
    bk.wave[:, :wave.shape[1]].copy_(wave.to(device))
    padded = proc.batch_from_tables(bk.wave, bk.soff, bk.foff, len(ns), bk.key[1], bk.crop[0:1], is_train=False)
    assert padded.shape[1] == bk.key[1] > feats.shape[1]
    assert torch.equal(padded[:, :feats.shape[1]], feats) and float(padded[:, feats.shape[1]:].abs().max()) == 0.0
    with torch.no_grad():
        x, lens, mask = model.encoder.subsampler(padded, fr.to(device), bk.crop[1:])
    T = x_ref.shape[1]
    assert torch.equal(lens, len_ref) and torch.equal(mask[:, :, :T], mask_ref) and not bool(mask[:, :, T:].any())
    torch.testing.assert_close(x[:, :T], x_ref, rtol=1e-6, atol=1e-6)


def test_update_count_and_checkpoint_after_graphed_steps(device, tmp_path):
    """The host's update count follows the REAL updates (first sights run eagerly + a capture pass, replays never reach
    FlatAdamW.clip_and_step): optimizer.t == step.steps == step_dev; a checkpoint written after graphed steps resumes onto the
    parameters of the run that went on (Adam's bias correction reads the count)."""
    from joeys2t_amd.graphed import GraphedTrainStep
    from test_hip_config_width import make_model
    torch.manual_seed(3)
    base = make_model(_cfg(), V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    proc = _proc()
    proc.specaugment = None
    batches = _batches(5, 3) + _batches(10, 2, lo=52000, hi=56000) + _batches(6, 2)  # two buckets: 2 first sights, 5 replays
    step = _make(sd, device, torch.float32)
    # two plain updates first: the graphed driver must start counting from them
    np.random.seed(11)
    _plain_run(step, proc, _batches(4, 2), device, torch.float32)
    assert step.optimizer.t == 2
    gs = GraphedTrainStep(step, proc, compute_dtype=torch.float32, frame_bucket=128, target_bucket=16)
    assert int(step.optimizer.step_dev) == 2
    how = [gs.run(w.to(device), ns, trg, tl) for w, ns, trg, tl in batches[:5]]
    assert how.count("eager") == 2 and how.count("replay") == 3, how
    assert step.optimizer.t == step.steps == int(step.optimizer.step_dev) == 7
    path = tmp_path / "after_graphed.ckpt"
    step.save_checkpoint(path)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert {float(s["step"]) for s in ckpt["optimizer_state"]["state"].values()} == {7.0}
    for w, ns, trg, tl in batches[5:]:
        gs.run(w.to(device), ns, trg, tl)
    torch.cuda.synchronize()
    flat_on = step.store.flat.detach().clone()
    # resume a fresh driver from the checkpoint and make the same two updates
    step2 = _make(sd, device, torch.float32)
    step2.init_from_checkpoint(path)
    assert step2.optimizer.t == 7 and step2.steps == 7
    gs2 = GraphedTrainStep(step2, proc, compute_dtype=torch.float32, frame_bucket=128, target_bucket=16)
    for w, ns, trg, tl in batches[5:]:
        gs2.run(w.to(device), ns, trg, tl)
    torch.cuda.synchronize()
    rel = ((step2.store.flat - flat_on).norm() / flat_on.norm()).item()
    assert rel < 1e-6, rel
    assert step2.optimizer.t == step2.steps == int(step2.optimizer.step_dev) == 9


def test_graphed_steps_on_packed_encoder_rows(device):
    """Ragged batches: the bucket key carries the packed row count (sum of the sub-sampled lengths, rounded up), the row offsets
    arrive with the batch, and a replayed step over OTHER lengths of the same bucket gives the un-bucketed step's numbers."""
    from test_hip_config_width import make_model
    torch.manual_seed(3)
    base = make_model(_cfg(), V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    batches = []
    for w, ns, trg, tl in _batches(7, 5, lo=20000, hi=24000):  # one long utterance, three short ones: ~35 % dead positions
        ns = [40000] + ns[1:]
        wave = 0.1 * torch.randn(len(ns), 40000, generator=g)
        for i, k in enumerate(ns):
            wave[i, k:] = 0.0
        batches.append((wave, ns, trg, tl))
    proc = _proc()
    dtype = torch.bfloat16
    ref, flat_ref = _plain_run(_make(sd, device, dtype), proc, batches, device, dtype)
    got, flat, how, gs = _graphed_run(_make(sd, device, dtype), proc, batches, device, dtype, frame_bucket=128, target_bucket=16, row_bucket=64)
    assert how == ["eager"] + ["replay"] * 4, how
    (key, bk), = gs.buckets.items()
    assert key[3] == 192 and bk.batch.src_pack.rows == 192 and bk.batch.src_pack.T == 64
    segs = {tuple(int(v) for v in np.cumsum([gs._sub_len(1 + (n - 400) // 160) for n in sorted(ns, reverse=True)])) for _, ns, _, _ in batches}
    assert len(segs) > 1  # the replays really saw other row offsets
    for i, (r, gt) in enumerate(zip(ref, got)):
        for k in range(3):
            assert abs(r[k] - gt[k]) <= 3e-2 * abs(r[k]), (i, k, r, gt)
        assert r[4] == gt[4]
    rel = ((flat - flat_ref).norm() / flat_ref.norm()).item()
    assert rel < 8e-3, rel


def test_precaptured_bucket_replays_on_first_sight(device):
    """GraphedTrainStep.precapture: a bucket captured before any batch of it has arrived - capturing runs nothing (parameters, step
    count, RNG untouched) - and the bucket's first batch is a replay with the plain step's numbers."""
    from test_hip_config_width import make_model
    torch.manual_seed(3)
    base = make_model(_cfg(), V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    dtype = torch.float32
    first = _batches(5, 1)                          # ~ 30000-41000 samples: frame bucket 256
    # two further buckets (longer utterances) that share their target-length bucket and their utterance count: what a capture builds
    # lazily per target length / per normaliser (causal mask, backward seed) must not be handed from one captured graph to the next -
    # the batches below arrive in the REVERSE of the capture order, so a constant filled only by the first graph's replays would be
    # read before it was ever written (bench.py --varying showed exactly that as a NaN loss)
    later = _batches(7, 1, lo=72000, hi=80000) + _batches(6, 2, lo=52000, hi=60000)
    proc = _proc()
    ref, flat_ref = _plain_run(_make(sd, device, dtype), proc, first + later, device, dtype)
    from joeys2t_amd.graphed import GraphedTrainStep
    np.random.seed(11)
    step = _make(sd, device, dtype)
    gs = GraphedTrainStep(step, proc, compute_dtype=dtype, frame_bucket=128, target_bucket=16)
    with pytest.raises(Exception):
        gs.precapture(gs.bucket_key(later[0][1], later[0][3]))  # nothing has run yet
    assert gs.run(first[0][0].to(device), first[0][1], first[0][2], first[0][3]) == "eager"
    s0 = gs.read_stats()
    keys = {gs.bucket_key(ns, tl) for _, ns, _, tl in later}
    assert gs.bucket_key(first[0][1], first[0][3]) not in keys and len(keys) == 2 and len({k[2] for k in keys}) == 1
    assert gs.bucket_key(later[0][1], later[0][3]) == max(keys)  # captured last, replayed first
    before = step.store.flat.detach().clone()
    t_before = (step.optimizer.t, step.steps, step.micro, int(step.optimizer.step_dev.item()))
    for key in sorted(keys):
        assert gs.precapture(key) and not gs.precapture(key)  # the second call finds it captured
    torch.cuda.synchronize()
    assert torch.equal(before, step.store.flat) and t_before == (step.optimizer.t, step.steps, step.micro, int(step.optimizer.step_dev.item()))
    got = [(s0["loss"], s0["nll"], s0["ctc"])]
    for wave, ns, trg, tl in later:
        assert gs.run(wave.to(device), ns, trg, tl) == "replay"  # first sight of the bucket, and no eager step
        s = gs.read_stats()
        got.append((s["loss"], s["nll"], s["ctc"]))
    assert gs.counts["eager"] == 1 and gs.counts["precaptured"] == len(keys)
    for r, g in zip(ref, got):
        for k in range(3):
            assert abs(r[k] - g[k]) <= 2e-5 * abs(r[k]), (r, g)
    torch.cuda.synchronize()
    diff = (step.store.flat - flat_ref).abs()
    apart = diff > 1e-5
    assert float(apart.float().mean()) < 5e-4 and ((((step.store.flat - flat_ref) * ~apart).norm() / flat_ref.norm()).item() < 1e-5)
