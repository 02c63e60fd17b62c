"""GPU: the encoder stack of a ragged batch on its live rows only (js2t_pack_rows, js2t_attn_desc.seg, encoders.TransformerEncoder.
_packing).  Positions behind an utterance's sub-sampled length are dead in the reference (joeynmt/encoders.py:348-373 builds the
padding mask, transformer_layers.py:86-105 masks them as keys, loss.py:156-161 cuts the CTC input at the lengths), so dropping them
must change nothing: losses and gradients equal the padded path's."""
import pytest
import torch

from joeys2t_amd import ops

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("lens,T,C,round_to", [([5, 3, 7, 1], 7, 64, 1), ([300, 425, 77, 250, 1], 425, 512, 64), ([9], 16, 24, 8), ([3, 2], 4, 32, 192)])
def test_pack_unpack_rows(device, lens, T, C, round_to):
    B = len(lens)
    pk = ops.PackedRows.from_lengths(lens, T, device, round_to=round_to)
    assert pk.rows % round_to == 0 and pk.rows >= sum(lens) and pk.seg.tolist()[-1] == sum(lens)
    x = rnd(B, T, C, seed=1).bfloat16()
    xp = ops.pack_rows(x.view(B * T, C).to(device), pk).cpu()
    ref = torch.cat([x[b, :n] for b, n in enumerate(lens)] + [torch.zeros(pk.rows - sum(lens), C, dtype=x.dtype)])
    assert torch.equal(xp, ref)
    back = ops.unpack_rows(xp.to(device), pk).cpu().view(B, T, C)
    live = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).unsqueeze(-1)
    assert torch.equal(back, torch.where(live, x, torch.zeros_like(x)))
    buf = torch.full((pk.rows, C), float("nan"), dtype=torch.bfloat16, device=device)
    ops.zero_tail_rows(buf, pk)
    assert torch.isnan(buf[:sum(lens)].float()).all() and float(buf[sum(lens):].float().abs().sum()) == 0.0
    f32 = rnd(B * T, C // 4 * 4, seed=2).to(device)  # any row of a multiple of 16 bytes
    assert torch.equal(ops.unpack_rows(ops.pack_rows(f32, pk), pk).view(B, T, -1)[0, :lens[0]], f32.view(B, T, -1)[0, :lens[0]])


@pytest.mark.parametrize("dh,H", [(128, 4), (64, 4)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_flash_attention_over_packed_rows_equals_padded(device, dh, H, p):
    """the same keys, queries, tiles and dropout counters: bit for bit on every live row, forward and backward (also through the
    one-grid backward with delta from partial sums)"""
    lens, T = [375, 301, 64, 129, 1, 200], 375
    B, d = len(lens), H * dh
    pk = ops.PackedRows.from_lengths(lens, T, device, round_to=64)
    live = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None])
    qkv = rnd(B * T, 3 * d, seed=1).bfloat16().to(device)
    go = (rnd(B * T, d, seed=2) * live.reshape(-1, 1)).bfloat16().to(device)  # dead rows carry no gradient (they are masked downstream)
    mask = live.unsqueeze(1).to(device)
    rng = ops.DropoutRng(device, seed=3) if p else None
    out, lse = ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, dh, mask, p, rng, 5)
    qkv_p, go_p = ops.pack_rows(qkv, pk), ops.pack_rows(go, pk)
    out_p, lse_p = ops.flash_attn_fwd(qkv_p, 2 * d, qkv_p, 0, qkv_p, d, B, H, T, T, dh, mask, p, rng, 5, seg=pk)
    assert out_p.shape == (pk.rows, d) and pk.rows > sum(lens)
    assert torch.equal(out_p, ops.pack_rows(out, pk))  # tail rows zero on both sides
    lv = live.to(device)[:, None, :].expand(B, H, T).reshape(B * H, T)
    assert torch.equal(lse_p[lv], lse[lv])
    for with_partial in (False, True):
        kw, kw_p = {}, {}
        if with_partial:
            kw["delta_partial"] = (go.float() * out.float()).view(B * T, d // 64, 64).sum(-1).contiguous()
            kw_p["delta_partial"] = (go_p.float() * out_p.float()).view(pk.rows, d // 64, 64).sum(-1).contiguous()
        dqkv = torch.zeros_like(qkv)
        ops.flash_attn_bwd(go, out, lse, qkv, 2 * d, qkv, 0, qkv, d, dqkv, 2 * d, dqkv, 0, dqkv, d, B, H, T, T, dh, mask, p, rng, 5, **kw)
        dqkv_p = torch.full_like(qkv_p, float("nan"))  # the rows behind the last entry are zeroed by the kernels (seg_rows)
        ops.flash_attn_bwd(go_p, out_p, lse_p, qkv_p, 2 * d, qkv_p, 0, qkv_p, d, dqkv_p, 2 * d, dqkv_p, 0, dqkv_p, d, B, H, T, T, dh, mask,
                           p, rng, 5, seg=pk, **kw_p)
        assert torch.isfinite(dqkv_p.float()).all()
        assert torch.equal(dqkv_p, ops.pack_rows(dqkv, pk)), with_partial


@pytest.mark.parametrize("dh,H", [(128, 4), (64, 8)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_cross_attention_over_packed_keys_equals_padded(device, dh, H, p):
    """js2t_attn_desc.seg_keys: the decoder's cross-attention reading the encoder states of a ragged batch as packed rows - the same
    keys, tiles, masks and dropout counters as over the padded memory: output, log-sum-exp, dQ bit for bit, dK / dV bit for bit on
    every live key row, the rows behind the last entry zeroed by the kernels (NaN-prefilled buffers); one grid and two launches."""
    lens, Tk, Tq = [375, 301, 64, 129, 1, 200], 375, 27
    B, d = len(lens), H * dh
    pk = ops.PackedRows.from_lengths(lens, Tk, device, round_to=64)
    live = (torch.arange(Tk)[None, :] < torch.tensor(lens)[:, None])
    q = rnd(B * Tq, d, seed=1).bfloat16().to(device)
    kv = (rnd(B * Tk, 2 * d, seed=2) * live.reshape(-1, 1)).bfloat16().to(device)
    go = rnd(B * Tq, d, seed=3).bfloat16().to(device)
    mask = live.unsqueeze(1).to(device)
    rng = ops.DropoutRng(device, seed=3) if p else None
    out, lse = ops.flash_attn_fwd(q, 0, kv, 0, kv, d, B, H, Tq, Tk, dh, mask, p, rng, 5)
    kv_p = ops.pack_rows(kv, pk)
    out_p, lse_p = ops.flash_attn_fwd(q, 0, kv_p, 0, kv_p, d, B, H, Tq, Tk, dh, mask, p, rng, 5, seg=pk, seg_keys=True)
    assert out_p.shape == out.shape and torch.equal(out_p, out) and torch.equal(lse_p, lse)
    part = (go.float() * out.float()).view(B * Tq, d // 64, 64).sum(-1).contiguous()
    for kw in ({}, {"delta_partial": part}):
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        ops.flash_attn_bwd(go, out, lse, q, 0, kv, 0, kv, d, dq, 0, dkv, 0, dkv, d, B, H, Tq, Tk, dh, mask, p, rng, 5, **kw)
        dq_p, dkv_p = torch.full_like(q, float("nan")), torch.full_like(kv_p, float("nan"))
        ops.flash_attn_bwd(go, out_p, lse_p, q, 0, kv_p, 0, kv_p, d, dq_p, 0, dkv_p, 0, dkv_p, d, B, H, Tq, Tk, dh, mask, p, rng, 5,
                           seg=pk, seg_keys=True, **kw)
        assert torch.isfinite(dq_p.float()).all() and torch.isfinite(dkv_p.float()).all()
        assert torch.equal(dq_p, dq), bool(kw)
        assert torch.equal(dkv_p, ops.pack_rows(dkv, pk)), bool(kw)  # (pack_rows zeroes the tail: so did the kernels)
    with pytest.raises(ops.Js2tError):  # the key buffer must be the packed one
        ops.flash_attn_fwd(q, 0, kv, 0, kv, d, B, H, Tq, Tk, dh, mask, p, rng, 5, seg=pk, seg_keys=True)


def test_packed_rows_are_refused_where_they_cannot_work(device):
    pk = ops.PackedRows.from_lengths([5, 3], 8, device)
    q = torch.zeros(8, 384, dtype=torch.bfloat16, device=device)
    mem = torch.zeros(2 * 20, 384, dtype=torch.bfloat16, device=device)
    with pytest.raises(ops.Js2tError):  # cross-attention shapes
        ops.flash_attn_fwd(q, 256, mem, 0, mem, 128, 2, 1, 8, 20, 128, None, 0.0, None, 0, seg=pk)
    with pytest.raises(ops.Js2tError):
        ops.PackedRows.from_lengths([9, 3], 8, device)


def _grads(device, packed, data, dropout=0.0, seed=5, dtype=torch.bfloat16):
    from joeys2t_amd import encoders
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import hip_batch, make_model, width_cfg
    torch.manual_seed(seed)
    cfg = width_cfg(4, 3, 2)
    cfg["encoder"]["dropout"] = cfg["decoder"]["dropout"] = dropout
    model = make_model(cfg, 300, None, device, dtype, 0.3, train=True)
    step = TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3)
    taken = []
    real = ops.pack_rows

    def spy(x, pk):
        taken.append(pk.rows)
        return real(x, pk)

    ops.pack_rows, encoders.PACK_RAGGED = spy, packed
    try:
        out = step.micro_step(hip_batch(*data, device), update=False)
        torch.cuda.synchronize()
    finally:
        ops.pack_rows, encoders.PACK_RAGGED = real, True
    stats = step.read_stats()
    return step.store.flat_grad.detach().clone(), stats, taken, out


def test_train_step_on_packed_rows_equals_padded(device):
    from test_hip_config_width import synth_batch
    data = synth_batch(300, [400, 330, 170, 150, 90], [9, 7, 5, 6, 3], 1)  # frames: T' = 100, 83, 43, 38, 23
    g_pad, s_pad, took_pad, _ = _grads(device, False, data)
    g_pad2, _, _, _ = _grads(device, False, data)
    g_pk, s_pk, took, _ = _grads(device, True, data)
    assert took_pad == [] and len(took) >= 1 and took[0] < 5 * 100  # forward packs once (backward of the un-pack packs again)
    assert s_pk["loss"] == pytest.approx(s_pad["loss"], rel=1e-5)
    noise = (g_pad - g_pad2).norm().item()  # the padded path twice: split-K atomics
    err = (g_pk - g_pad).norm().item()
    assert torch.isfinite(g_pk).all() and err <= max(4.0 * noise, 2e-3 * g_pad.norm().item()), (err, noise, g_pad.norm().item())


def test_decoder_reads_packed_encoder_states(device):
    """The decoder side of a ragged batch: K | V projections of all layers, their gradients and the cross-attention keys on the live
    positions (decoders.PACK_MEMORY, js2t_attn_desc.seg_keys) against the same step reading the padded [B, T', d] states: the cross-
    attention calls really take packed keys, the loss agrees to 1e-5, the flat gradient within the padded path's run-to-run noise."""
    from joeys2t_amd import decoders
    from test_hip_config_width import synth_batch
    data = synth_batch(300, [400, 330, 170, 150, 90], [9, 7, 5, 6, 3], 1)
    seen = []
    real = ops.flash_attn_fwd

    def spy(*a, **kw):
        seen.append((bool(kw.get("seg_keys", False)), None if kw.get("seg") is None else kw["seg"].rows, a[2].shape[0]))
        return real(*a, **kw)

    ops.flash_attn_fwd = spy
    try:
        g_pk, s_pk, _, _ = _grads(device, True, data)
        n_keys_packed = sum(1 for k, rows, krows in seen if k and rows == krows)
        del seen[:]
        decoders.PACK_MEMORY = False
        g_pad, s_pad, _, _ = _grads(device, True, data)
        g_pad2, _, _, _ = _grads(device, True, data)
        assert not any(k for k, _, _ in seen)
    finally:
        ops.flash_attn_fwd, decoders.PACK_MEMORY = real, True
    assert n_keys_packed == 2  # width_cfg(4, 3, 2): two decoder layers, each cross-attention once
    assert s_pk["loss"] == pytest.approx(s_pad["loss"], rel=1e-5)
    noise = (g_pad - g_pad2).norm().item()
    err = (g_pk - g_pad).norm().item()
    assert torch.isfinite(g_pk).all() and err <= max(4.0 * noise, 2e-3 * g_pad.norm().item()), (err, noise, g_pad.norm().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_ctc_over_packed_rows_equals_padded(device, dtype):
    """js2t_ctc_alpha / js2t_ctc_bwd with row_offsets: the recursions and the gradient read the packed logits rows of every live
    frame - alpha, beta, the negative log-likelihoods BIT for bit those of the padded [B, T, V] logits, the gradient bit for bit on
    every live row (an infeasible utterance and one of a single frame included); through loss._CtcFn: loss and gradient."""
    from joeys2t_amd.loss import _CtcFn
    lens, T, V = [40, 33, 7, 12, 1, 25], 40, 24
    B = len(lens)
    pk = ops.PackedRows.from_lengths(lens, T, device, round_to=16)
    live = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None])
    logits = (rnd(B, T, V, seed=1) * 2.0).to(dtype)
    targets = torch.tensor([[5, 6, 6, 7, 3, 1, 1, 1], [4, 4, 9, 3, 1, 1, 1, 1], [5, 6, 7, 8, 9, 10, 11, 3], [8, 3, 1, 1, 1, 1, 1, 1],
                            [9, 3, 1, 1, 1, 1, 1, 1], [10, 11, 12, 13, 3, 1, 1, 1]])
    tgt_len = torch.tensor([5, 4, 8, 2, 2, 5])  # entry 2: 8 labels in 7 frames (infeasible), entry 4: 2 labels in 1 frame (infeasible)
    in_len = torch.tensor(lens)
    lg, tg, il, tl = logits.to(device), targets.to(device), in_len.to(device), tgt_len.to(device)
    lg_p = ops.pack_rows(lg.view(B * T, V), pk) if dtype == torch.bfloat16 else ops.pack_rows(lg.view(B * T, V).contiguous(), pk)
    lse, _ = ops.row_lse(lg.view(B * T, V))
    lse_p, _ = ops.row_lse(lg_p)
    a, nll, rows, b = ops.ctc_alpha(lg, lse, tg, il, tl, 2, True, with_beta=True)
    a_p, nll_p, rows_p, b_p = ops.ctc_alpha(lg_p, lse_p, tg, il, tl, 2, True, with_beta=True, pack=pk)
    assert torch.equal(nll, nll_p) and torch.equal(rows, rows_p) and torch.isinf(nll[2]) and torch.isinf(nll[4])
    for i, (n, L) in enumerate(zip(lens, tgt_len.tolist())):  # (states behind 2 L + 1 and frames behind the length are not defined)
        assert torch.equal(a[i, :n, :2 * L + 1], a_p[i, :n, :2 * L + 1]) and torch.equal(b[i, :n, :2 * L + 1], b_p[i, :n, :2 * L + 1])
    g = torch.ones((), device=device)
    d = ops.ctc_bwd(lg, lse, tg, il, tl, a, nll, g, 1.0, 2, True, beta=b)
    d_p = ops.ctc_bwd(lg_p, lse_p, tg, il, tl, a_p, nll_p, g, 1.0, 2, True, beta=b_p, pack=pk)
    assert d_p.shape == lg_p.shape and torch.isfinite(d_p.float()).all()
    assert torch.equal(d_p, ops.pack_rows(d.view(B * T, V), pk))  # dead frames carry zeros in the padded form; the packed tail is zeroed
    # through the autograd function
    x = lg.clone().requires_grad_(True)
    xp = lg_p.clone().unsqueeze(0).requires_grad_(True)
    l0 = _CtcFn.apply(x, tg, il, tl, 2, True)
    l1 = _CtcFn.apply(xp, tg, il, tl, 2, True, pk)
    assert torch.equal(l0, l1)
    l0.backward(), l1.backward()
    assert torch.equal(xp.grad[0], ops.pack_rows(x.grad.view(B * T, V), pk))


def test_ctc_branch_on_packed_rows_in_the_train_step(device):
    """model.PACK_CTC: projection, row log-sum-exp, recursions and gradient of the CTC branch on the packed encoder rows against the
    padded branch - loss to 1e-5, flat gradient within the padded path's run-to-run noise."""
    from joeys2t_amd import model as model_mod
    from test_hip_config_width import synth_batch
    data = synth_batch(300, [400, 330, 170, 150, 90], [9, 7, 5, 6, 3], 1)
    seen = []
    real = ops.ctc_alpha

    def spy(*a, **kw):
        seen.append(kw.get("pack") is not None)
        return real(*a, **kw)

    ops.ctc_alpha = spy
    try:
        g_pk, s_pk, _, _ = _grads(device, True, data)
        took_packed = list(seen)
        del seen[:]
        model_mod.PACK_CTC = False
        g_pad, s_pad, _, _ = _grads(device, True, data)
        g_pad2, _, _, _ = _grads(device, True, data)
        assert seen == [False, False]
    finally:
        ops.ctc_alpha, model_mod.PACK_CTC = real, True
    assert took_packed == [True]
    assert s_pk["loss"] == pytest.approx(s_pad["loss"], rel=1e-5)
    noise = (g_pad - g_pad2).norm().item()
    err = (g_pk - g_pad).norm().item()
    assert torch.isfinite(g_pk).all() and err <= max(4.0 * noise, 2e-3 * g_pad.norm().item()), (err, noise, g_pad.norm().item())


def test_packed_rows_with_dropout_stay_finite_and_close(device):
    """dropout on: the row-wise masks differ between the layouts (the counter is the row index), the attention masks do not;
    the loss of one step stays within the spread dropout gives it anyway"""
    from test_hip_config_width import synth_batch
    data = synth_batch(300, [400, 330, 170, 150, 90], [9, 7, 5, 6, 3], 1)
    g_pk, s_pk, took, _ = _grads(device, True, data, dropout=0.1)
    _, s_pad, _, _ = _grads(device, False, data, dropout=0.1)
    assert took and torch.isfinite(g_pk).all()
    assert s_pk["loss"] == pytest.approx(s_pad["loss"], rel=0.1)


def test_inference_encode_on_packed_rows_equals_padded(device):
    """no_grad forward (what search() runs): the encoder states of every live position are those of the padded layout, the
    positions behind a length are zero"""
    from joeys2t_amd import encoders
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    torch.manual_seed(7)
    model = make_model(width_cfg(4, 3, 1), 300, None, device, torch.bfloat16, 0.3, train=False)
    data = synth_batch(300, [400, 330, 170, 150, 90], [9, 7, 5, 6, 3], 1)
    outs = {}
    for packed in (True, False):
        encoders.PACK_RAGGED = packed
        try:
            b = hip_batch(*data, device)
            with torch.no_grad():
                enc, _, mask, _ = model(return_type="encode", **vars(b))
            outs[packed] = (enc.float().cpu(), mask.cpu())
        finally:
            encoders.PACK_RAGGED = True
    (e_pk, m_pk), (e_pad, m_pad) = outs[True], outs[False]
    assert torch.equal(m_pk, m_pad) and e_pk.shape == e_pad.shape
    live = m_pad.squeeze(1).unsqueeze(-1)
    assert float(e_pk.masked_select(~live.expand_as(e_pk)).abs().sum()) == 0.0  # zeros behind every length
    a, r = e_pk * live, e_pad * live
    assert (a - r).norm() <= 2e-2 * r.norm(), ((a - r).norm().item(), r.norm().item())  # bf16: the first block's LayerNorm runs stand-alone on packed rows
