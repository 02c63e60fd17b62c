"""GPU: BASELINE.json configs[4] as a COMPOSED model - ConformerEncoder with relative-position attention, in fp32 against the
oracle, the fused bf16 kernels against the fp32 path, and the e4m3 forward mode against bf16 on the same model.

Both parts of config 5 are extensions (the reference has neither a relative-position term nor fp8: transformer_layers.py:
478-565, config.py:223-225), so the checker for the bias is the oracle's own restatement (oracle.rel_pos_scores inside
oracle.mha) on top of the reference-captured Conformer weights; everything else in the layer is pinned by conformer.npz."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import s2t_oracle as O

pytestmark = pytest.mark.gpu


def _bind(enc, device, dtype, store=False):
    from joeys2t_amd.runtime import ParamStore, Runtime, install_runtime
    enc.to(device)
    rt = Runtime(device, dtype)
    if store:
        rt.store = ParamStore(enc, device)
        rt.rng.seed(42)
    install_runtime(enc, rt)
    if store and dtype == torch.bfloat16:
        rt.store.refresh(force=True)
    return enc


def test_conformer_relpos_fp32_matches_oracle(device):
    """The golden Conformer (weights captured from the reference) + a random relative-position table per layer, train mode,
    fp32 compute: output within 1e-4 of the oracle, every gradient - the tables' included - within 2e-4."""
    from joeys2t_amd.encoders import ConformerEncoder
    g = load_golden("conformer")
    pre, R = "pre.", 3
    enc = ConformerEncoder(hidden_size=16, ff_size=32, num_layers=2, num_heads=2, dropout=0.0, emb_dropout=0.0, in_channels=8,
                           conv_channels=24, conv_kernel_sizes=[5, 5], depthwise_conv_kernel_size=5, alpha=1.0, layer_norm="pre",
                           rel_pos_clip=R)
    sd = {k[len(pre) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre + "sd0.")}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("pe.pe") or k.endswith("rel_pos_bias") for k in missing)
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for layer in enc.layers:
            layer.src_src_att.rel_pos_bias.copy_(0.5 * torch.randn(2, 2 * R + 1, generator=gen))
    osd = {"encoder." + k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and "pe.pe" not in k)
           for k, v in enc.state_dict().items()}
    _bind(enc, device, torch.float32).train()
    src, lengths = torch.from_numpy(g[pre + "src"]), torch.from_numpy(g[pre + "src_length"])
    proj = torch.from_numpy(g[pre + "proj"])
    cfg = {"encoder": {"num_layers": 2, "num_heads": 2, "alpha": 1.0, "layer_norm": "pre", "conv_kernel_sizes": [5, 5]}}
    ref, rmask, _ = O.conformer_encoder_forward(osd, cfg, src, lengths, train=True, new_stats={})
    y, _, mask = enc(src.to(device), lengths.to(device), None)
    assert np.array_equal(mask.cpu().numpy(), rmask.numpy())
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    # the table matters: without it the output is the plain capture, which differs
    assert np.abs(ref.detach().numpy() - g[pre + "out_train"]).max() > 1e-2
    (ref * proj).sum().backward()
    (y * proj.to(device)).sum().backward()
    n_bias = 0
    for n, p in enc.named_parameters():
        want = osd["encoder." + n].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(want).max())), err_msg=n)
        n_bias += n.endswith("rel_pos_bias")
        if n.endswith("rel_pos_bias"):
            assert np.abs(want).max() > 1e-3
    assert n_bias == 2


def _wide(device, dtype, R=8, seed=5, dropout=0.0, layers=3):
    """A Conformer wide enough for the fused attention kernels (2 heads of 64) and the e4m3 products (K, N >= 128)."""
    from joeys2t_amd.encoders import ConformerEncoder
    torch.manual_seed(seed)
    enc = ConformerEncoder(hidden_size=128, ff_size=256, num_layers=layers, num_heads=2, dropout=dropout, emb_dropout=dropout,
                           in_channels=16, conv_channels=128, conv_kernel_sizes=[5, 5], depthwise_conv_kernel_size=7, alpha=1.0,
                           layer_norm="pre", rel_pos_clip=R)
    with torch.no_grad():
        for layer in enc.layers:
            layer.src_src_att.rel_pos_bias.normal_(0.0, 0.3)
        for n, p in enc.named_parameters():
            if n.endswith("bias") and "rel_pos" not in n:
                p.normal_(0.0, 0.05)
    return _bind(enc, device, dtype, store=True)


def _inputs(device, B=3, T=203, F=16, seed=9):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, T, F, generator=g)
    lengths = torch.tensor([T, T - 40, T - 77])
    for b in range(B):
        src[b, lengths[b]:] = 1.0
    return src.to(device), lengths.to(device)


def test_conformer_relpos_bf16_fused_close_to_fp32(device):
    """The same composed model in bf16 takes the fused attention kernels (bias added on chip, gradient by LDS histogram);
    output and the tables' gradients stay close to the fp32 (materialised) path."""
    from joeys2t_amd import functional as Fn
    src, lengths = _inputs(device)
    outs, grads = {}, {}
    calls = {"n": 0}
    real = Fn.ops.flash_attn_fwd

    def spy(*a, **k):
        calls["n"] += 1
        assert k.get("rel_bias", a[15] if len(a) > 15 else None) is not None
        return real(*a, **k)

    for dt in (torch.float32, torch.bfloat16):
        enc = _wide(device, dt).train()
        if dt == torch.bfloat16:
            Fn.ops.flash_attn_fwd = spy
        try:
            y, _, _ = enc(src, lengths, None)
        finally:
            Fn.ops.flash_attn_fwd = real
        w = torch.linspace(-1, 1, y.shape[-1], device=device)
        (y.float() * w).sum().backward()
        outs[dt] = y.detach().float().cpu()
        grads[dt] = {n: p.grad.detach().float().cpu() for n, p in enc.named_parameters() if n.endswith("rel_pos_bias")}
    assert calls["n"] == 3  # one fused launch per layer
    a, b = outs[torch.float32], outs[torch.bfloat16]
    assert ((a - b).norm() / a.norm()).item() < 3e-2
    for n, ga in grads[torch.float32].items():
        gb = grads[torch.bfloat16][n]
        cos = torch.nn.functional.cosine_similarity(ga.flatten(), gb.flatten(), dim=0).item()
        assert cos > 0.98 and abs(gb.norm().item() / ga.norm().item() - 1) < 0.1, (n, cos)


@pytest.mark.parametrize("separate_pass", [False, True])
def test_fp8_forward_close_to_bf16(device, separate_pass):
    """functional.FP8_FORWARD on the composed model: nn.Linear forwards on e4m3 operands (per-tensor scales) - by default those
    whose input the LayerNorm kernel quantises as it writes it (q/k/v projection and the first layer of both feed-forward
    modules) or the preceding e4m3 product writes as e4m3 next to bf16 (the second feed-forward layer): 5 per Conformer layer, with FP8_SEPARATE_PASS every eligible one (8 per layer + the input Linear).
    Bound: e4m3 keeps 3 mantissa bits, so an element is off by at most 2^-4 relative (uniform: 2^-4 / sqrt(3) = 3.6 % rms); a
    product of two rounded operands by 5.1 % rms, and a dot product of terms with independent errors keeps that RELATIVE rms
    only if all terms had one sign - with mixed signs the sum's error relative to the sum is larger by |terms|_2 sqrt(K) /
    |sum|, a factor of 1 - 1.5 for these activations.  So each block's branch output carries <= ~8 % noise; the residual stream
    adds the (independent) branch errors in quadrature over 3 layers x 4 branches while the signal adds coherently, which
    keeps the encoder output's relative L2 error below the per-branch figure: 8 % asserted, ~3 - 5 % measured."""
    from joeys2t_amd import functional as Fn
    src, lengths = _inputs(device)
    enc = _wide(device, torch.bfloat16).eval()
    seen, quant = [], []
    real, real_q = Fn.ops.gemm, Fn.ops.quantize_fp8_delayed

    def spy(A, Bm, Cc, **k):
        seen.append(A.dtype)
        return real(A, Bm, Cc, **k)

    def spy_q(*a, **k):
        quant.append(1)
        return real_q(*a, **k)

    with torch.no_grad():
        ref, _, _ = enc(src, lengths, None)
        old = (Fn.FP8_FORWARD, Fn.FP8_SEPARATE_PASS)
        Fn.FP8_FORWARD, Fn.FP8_SEPARATE_PASS = True, separate_pass
        Fn.ops.gemm, Fn.ops.quantize_fp8_delayed = spy, spy_q
        try:
            out, _, _ = enc(src, lengths, None)
            out2, _, _ = enc(src, lengths, None)  # second call: delayed scaling uses the first call's maxima
        finally:
            Fn.FP8_FORWARD, Fn.FP8_SEPARATE_PASS = old
            Fn.ops.gemm, Fn.ops.quantize_fp8_delayed = real, real_q
    n8 = sum(d == torch.float8_e4m3fn for d in seen)
    if separate_pass:
        assert n8 >= 2 * 3 * 7 and len(quant) >= 2 * 3 * 3, (n8, len(quant))  # own passes: output projection, two pointwise convolutions
    else:
        # LayerNorm-fed products + the second feed-forward product (fed by the first one's e4m3 output): 5 per layer and forward,
        # plus the two bf16-output calibration passes of each feed-forward module's first use; no quantisation pass at all
        assert n8 == 2 * 3 * 5 + 3 * 2 and not quant, (n8, len(quant))
    a = ref.float()
    for o in (out, out2):
        rel = ((o.float() - a).norm() / a.norm()).item()
        cos = torch.nn.functional.cosine_similarity(o.float().flatten(), a.flatten(), dim=0).item()
        assert rel < 0.08 and cos > 0.996, (rel, cos)
    assert not torch.equal(out, ref)


def test_layernorm_fp8_output_matches_quantised_layernorm(device):
    """js2t_layernorm_fwd_fp8: the e4m3 bytes equal js2t_quantize_fp8_delayed of the bf16 LayerNorm output up to the bf16
    rounding the separate path puts in between; mean / rstd / bf16 output identical to the plain kernel; the state adapts."""
    from joeys2t_amd import ops
    g = torch.Generator().manual_seed(8)
    rows, D = 1000, 512
    x = (2 * torch.randn(rows, D, generator=g)).bfloat16().to(device)
    gamma, beta = (1 + 0.1 * torch.randn(D, generator=g)).to(device), (0.1 * torch.randn(D, generator=g)).to(device)
    y_ref, mean_ref, rstd_ref = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    state = ops.new_fp8_state(y_ref)
    s0 = float(state[0])
    mul = torch.tensor([0.5], device=device)
    y, mean, rstd, y8, sc = ops.layernorm_fwd_fp8(x, gamma, beta, 1e-6, state, mul=mul)
    assert torch.equal(y, y_ref) and torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)
    assert float(sc) == pytest.approx(s0 * 0.5, rel=1e-6)
    deq = y8.float() * s0
    assert ((deq - y_ref.float()).abs() <= 0.0626 * y_ref.float().abs() + 2 * s0 * 2**-9 + 1e-3).all()  # 2^-4 relative + subnormal step
    # the call's maximum waits in state[1]; the consuming e4m3 product hands it over (js2t_gemm fp8_state)
    assert float(state[0]) == s0 and float(state[1]) == pytest.approx(float(y_ref.float().abs().max()), rel=1e-2)
    w8 = torch.randn(256, D, generator=g).to(device).to(torch.float8_e4m3fn)
    out = torch.empty((rows, 256), dtype=torch.bfloat16, device=device)
    ops.gemm(y8, w8, out, M=rows, N=256, K=D, lda=D, ldb=D, ldc=256, alpha_dev=sc, fp8_state=state)
    amax = float(y_ref.float().abs().max())
    assert float(state[0]) == pytest.approx(amax / 448.0, rel=1e-2) and float(state[1]) == pytest.approx(amax * 0.9375, rel=1e-2)
    n, _, _, y8b, _ = ops.layernorm_fwd_fp8(x, gamma, beta, 1e-6, state, want_y=False)
    assert n is None and y8b.shape == y8.shape


@pytest.mark.parametrize("M,N,K,with_c", [(12000, 2048, 512, True), (700, 256, 256, False)])
def test_e4m3_product_writes_the_e4m3_operand_of_the_next_one(device, M, N, K, with_c):
    """js2t_gemm c8: an e4m3 product (bias + ReLU + dropout epilogue) also writes its result as e4m3 with the delayed scale of
    c8_state - what js2t_quantize_fp8_delayed would make of the bf16 result, minus the bf16 rounding in between; the launch's
    max |v| is collected for the next scale; C may be left out."""
    from joeys2t_amd import ops
    g = torch.Generator().manual_seed(12)
    a8, sa = ops.quantize_fp8(torch.randn(M, K, generator=g).bfloat16().to(device))
    w8, sw = ops.quantize_fp8((torch.randn(N, K, generator=g) * 0.05).bfloat16().to(device), mul=sa)
    bias = torch.randn(N, generator=g).to(device)
    ref = ((a8.float() @ w8.float().t()) * float(sw) + bias).relu()
    state = torch.zeros(4, device=device)
    state[0] = float(ref.max()) / 448.0
    mul, sc = torch.tensor([0.25], device=device), torch.zeros(1, device=device)
    y = torch.empty((M, N), dtype=torch.bfloat16, device=device) if with_c else None
    y8 = torch.empty((M, N), dtype=torch.float8_e4m3fn, device=device)
    ops.gemm(a8, w8, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, alpha_dev=sw, bias=bias, act="relu", c8=(y8, state, mul, sc))
    s0 = float(ref.max()) / 448.0
    assert float(sc) == pytest.approx(s0 * 0.25, rel=1e-6)
    deq = y8.float() * s0
    assert ((deq - ref).abs() <= 0.0626 * ref.abs() + 2 * s0 * 2**-9 + 2e-2).all()
    assert float(state[1]) == pytest.approx(float(ref.max()), rel=2e-2) and float(state[0]) == pytest.approx(s0)
    if with_c:
        torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=1e-2)
