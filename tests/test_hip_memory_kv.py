"""GPU: the cross-attention K | V projections of the encoder states for ALL decoder layers as one product
(functional.MemoryKVFn, TransformerDecoder.memory_kv) against the per-layer projections the reference runs
(transformer_layers.py:66-68 under :383): same forward numbers, same gradients, and the fall-backs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
V = 300


def _setup(device, dtype, dec_layers=3, train=True):
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    cfg = width_cfg(4, 2, dec_layers)
    torch.manual_seed(5)
    base = make_model(cfg, V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    model = make_model(cfg, V, sd, device, dtype, 0.3, train=train)
    batch = hip_batch(*synth_batch(V, [400, 333, 290], [12, 9, 15], seed=4), device)
    return model, batch


def test_store_keeps_the_cross_kv_weights_of_all_layers_adjacent(device):
    model, _ = _setup(device, torch.bfloat16)
    st, dec = model.runtime.store, model.decoder
    ws, bs = dec._cross_kv_params()
    d, L = 512, len(dec.layers)
    w, b = st.view(ws, torch.bfloat16), st.view(bs, torch.float32)
    assert w is not None and tuple(w.shape) == (L * 2 * d, d) and b is not None and tuple(b.shape) == (L * 2 * d, )
    for i, layer in enumerate(dec.layers):
        a = layer.src_trg_att
        assert torch.equal(w[i * 2 * d:i * 2 * d + d].float(), a.k_layer.weight.data.bfloat16().float())
        assert torch.equal(w[i * 2 * d + d:(i + 1) * 2 * d].float(), a.v_layer.weight.data.bfloat16().float())
        assert st.view([a.k_layer.weight, a.v_layer.weight], torch.bfloat16) is not None  # the per-layer fall-back's view
        s = layer.trg_trg_att  # the self-attention [k; v; q] fusion is untouched
        assert st.view([s.k_layer.weight, s.v_layer.weight, s.q_layer.weight], torch.bfloat16) is not None
    wt = st.view_t(ws)  # ... and so is the transposed shadow the one gradient product multiplies by
    assert wt is not None and tuple(wt.shape) == (d, L * 2 * d) and torch.equal(wt, w.t())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_projections_give_the_per_layer_step(device, dtype):
    """TrainStep.micro_step with and without the grouping: loss, every gradient, and the launch count of the projections."""
    from joeys2t_amd import functional as Fn
    from joeys2t_amd.training import TrainStep
    out = {}
    for grouped in (False, True):
        Fn.GROUP_MEMORY_KV = grouped
        try:
            model, batch = _setup(device, dtype)
            step = TrainStep(model, learning_rate=1e-3, clip_grad_norm=1.0, normalization="batch", overlap_ctc=False)
            calls = []
            real = Fn.MemoryKVFn.forward

            def counted(ctx, *a, _real=real, _calls=calls):
                _calls.append(1)
                return _real(ctx, *a)

            Fn.MemoryKVFn.forward = staticmethod(counted)
            try:
                loss = step.micro_step(batch, sort=True, update=False)
            finally:
                Fn.MemoryKVFn.forward = staticmethod(real)
            torch.cuda.synchronize()
            out[grouped] = (loss.item(), step.store.flat_grad.clone(), len(calls), dict(step.store.offsets), model)
        finally:
            Fn.GROUP_MEMORY_KV = True
    (la, ga, na, _, ma), (lb, gb, nb, _, mb) = out[False], out[True]
    assert na == 0 and nb == 1
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert abs(la - lb) <= tol * abs(la)
    # parameter by parameter (both stores have the same layout: it does not depend on the switch)
    sta, stb = ma.runtime.store, mb.runtime.store
    floor = 1e-5 * ga.norm().item()
    for (name, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        a = ga[sta.offsets[id(pa)]:sta.offsets[id(pa)] + pa.numel()]
        b = gb[stb.offsets[id(pb)]:stb.offsets[id(pb)] + pb.numel()]
        den = a.norm().item()
        if den < floor:  # e.g. the key biases: zero in exact arithmetic (softmax does not see a shift of all scores), noise here
            assert (a - b).norm().item() < floor, name
            continue
        rel = ((a - b).norm() / den).item()
        # bf16: the per-layer chain rounds the running encoder-state gradient to bf16 after every layer, the one product does not
        assert rel < (2e-5 if dtype == torch.float32 else 3e-2), (name, rel)


def test_forward_columns_are_the_per_layer_projections(device):
    """Inference (no gradients wanted): the grouped path is taken, and every layer reads what its own k_layer / v_layer give."""
    from joeys2t_amd import functional as Fn
    model, batch = _setup(device, torch.bfloat16, train=False)
    dec, rt = model.decoder, model.runtime
    with torch.no_grad():
        enc, _, src_mask, _ = model(return_type="encode", **vars(batch))
        kv = dec.memory_kv(enc)
        assert kv is not None and tuple(kv.shape) == (enc.shape[0] * enc.shape[1], len(dec.layers) * 1024)
        m2 = enc.reshape(-1, 512)
        for i, layer in enumerate(dec.layers):
            wc = layer.src_trg_att._weights(rt, "cross")
            own = Fn.linear_fwd(m2, wc["w_kv"], wc["b_kv"])
            assert torch.equal(kv[:, i * 1024:(i + 1) * 1024], own), i
        outs = {}
        for grouped in (False, True):
            Fn.GROUP_MEMORY_KV = grouped
            try:
                outs[grouped] = model(return_type="decode", encoder_output=enc, encoder_hidden=None, src_mask=src_mask,
                                      trg_input=batch.trg_input, unroll_steps=batch.trg_input.size(1), trg_mask=batch.trg_mask)[0]
            finally:
                Fn.GROUP_MEMORY_KV = True
        assert torch.equal(outs[False], outs[True])


def test_gradients_outside_a_train_step_take_the_per_layer_path(device):
    """loss.backward() without TrainStep around it: nobody would check that every layer's backward ran, so the layers project
    for themselves - and the gradients are there."""
    from joeys2t_amd import functional as Fn
    model, batch = _setup(device, torch.float32)
    calls = []
    real = Fn.MemoryKVFn.forward
    Fn.MemoryKVFn.forward = staticmethod(lambda ctx, *a: (calls.append(1), real(ctx, *a))[1])
    try:
        total = model(return_type="loss", **vars(batch))[0]
        total.backward()
    finally:
        Fn.MemoryKVFn.forward = staticmethod(real)
    assert not calls
    for layer in model.decoder.layers:
        g = layer.src_trg_att.k_layer.weight.grad
        assert g is not None and torch.isfinite(g).all() and g.abs().sum().item() > 0


def test_a_cross_block_that_never_runs_backward_is_reported(device):
    """The shared gradient buffer goes to autograd when the LAST layer has written its columns; a layer whose backward never
    ran would lose it silently - end_memory_chain() says so."""
    from joeys2t_amd import functional as Fn
    model, batch = _setup(device, torch.float32)
    model.train()
    dec = model.decoder
    Fn.begin_memory_chain()
    try:
        enc, _, src_mask, _ = model(return_type="encode", **vars(batch))
        kv = dec.memory_kv(enc)
        assert kv is not None and kv.requires_grad
        x = torch.randn(enc.shape[0], 7, 512, device=device, requires_grad=True)
        trg_mask = torch.ones(enc.shape[0], 7, 7, dtype=torch.bool, device=device)
        ys = [layer(x=x, memory=enc, src_mask=src_mask, trg_mask=trg_mask, memory_kv=(kv, i * 1024))[0] for i, layer in enumerate(dec.layers)]
        ys[0].sum().backward()  # only the first layer's branch
        with pytest.raises(RuntimeError, match="incomplete"):
            Fn.end_memory_chain()
    finally:
        Fn.end_memory_chain(check=False)


def test_one_utterance_and_two_layers(device):
    """Edge sizes: a single utterance, the smallest decoder the grouping applies to (two layers), ragged target lengths."""
    from joeys2t_amd import functional as Fn
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    cfg = width_cfg(4, 1, 2)
    torch.manual_seed(9)
    base = make_model(cfg, V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    out = {}
    for grouped in (False, True):
        Fn.GROUP_MEMORY_KV = grouped
        try:
            model = make_model(cfg, V, sd, device, torch.float32, 0.3, train=True)
            step = TrainStep(model, learning_rate=1e-3, clip_grad_norm=1.0, normalization="batch")
            loss = step.micro_step(hip_batch(*synth_batch(V, [211], [6], seed=1), device), update=False)
            torch.cuda.synchronize()
            out[grouped] = (loss.item(), step.store.flat_grad.clone())
        finally:
            Fn.GROUP_MEMORY_KV = True
    assert abs(out[False][0] - out[True][0]) <= 1e-5 * abs(out[False][0])
    rel = ((out[False][1] - out[True][1]).norm() / out[False][1].norm()).item()
    assert rel < 2e-5, rel


def test_single_layer_decoder_projects_for_itself(device):
    from test_hip_config_width import make_model, width_cfg
    model = make_model(width_cfg(4, 1, 1), V, None, device, torch.float32, 0.3, train=False)
    enc = torch.randn(2, 50, 512, device=device)
    with torch.no_grad():
        assert model.decoder.memory_kv(enc) is None
