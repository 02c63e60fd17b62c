"""GPU: Conformer encoder of the HIP path against a capture of the reference's ConformerEncoder
(tests/golden/conformer.npz): train-mode output, every parameter gradient, BatchNorm running statistics, eval-mode output;
and the convolution-module kernels against plain torch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _build(ln, g, device, dtype=torch.float32):
    from joeys2t_amd.encoders import ConformerEncoder
    from joeys2t_amd.runtime import Runtime, install_runtime
    enc = ConformerEncoder(hidden_size=16, ff_size=32, num_layers=2, num_heads=2, dropout=0.0, emb_dropout=0.0, in_channels=8,
                           conv_channels=24, conv_kernel_sizes=[5, 5], depthwise_conv_kernel_size=5, alpha=1.0, layer_norm=ln)
    pre = ln + "."
    sd = {k[len(pre) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre + "sd0.")}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("pe.pe") for k in missing), (missing, unexpected)  # state_dict keys identical
    enc.to(device)
    install_runtime(enc, Runtime(device, dtype))
    return enc


@pytest.mark.parametrize("ln", ["pre", "post"])
def test_conformer_encoder_matches_reference(device, ln):
    g = load_golden("conformer")
    pre = ln + "."
    enc = _build(ln, g, device)
    src, lengths = torch.from_numpy(g[pre + "src"]).to(device), torch.from_numpy(g[pre + "src_length"]).to(device)
    proj = torch.from_numpy(g[pre + "proj"]).to(device)
    enc.train()
    y, _, mask = enc(src, lengths, None)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[pre + "out_train"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(mask.cpu().numpy(), g[pre + "mask"])  # bit-exact length mask
    (y * proj).sum().backward()
    for n, p in enc.named_parameters():
        ref = g[pre + "grad." + n]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(ref).max())), err_msg=n)
    for k, v in g.items():
        if k.startswith(pre + "sd1."):
            got = dict(enc.named_buffers())[k[len(pre) + 4:]].cpu().numpy()
            np.testing.assert_allclose(got, v, rtol=1e-5, atol=1e-6, err_msg=k)
    enc.eval()
    with torch.no_grad():
        y2, _, _ = enc(src, lengths, None)
    np.testing.assert_allclose(y2.cpu().numpy(), g[pre + "out_eval"], rtol=1e-4, atol=1e-4)


def test_conformer_bf16_runs_close_to_fp32(device):
    g = load_golden("conformer")
    enc32, enc16 = _build("pre", g, device), _build("pre", g, device, torch.bfloat16)
    src, lengths = torch.from_numpy(g["pre.src"]).to(device), torch.from_numpy(g["pre.src_length"]).to(device)
    enc32.eval(), enc16.eval()
    with torch.no_grad():
        a, _, _ = enc32(src, lengths, None)
        b, _, _ = enc16(src, lengths, None)
    assert torch.nn.functional.cosine_similarity(a.flatten(), b.float().flatten(), dim=0) > 0.995


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_outer_and_bn_kernels(device, dtype):
    from joeys2t_amd import ops
    gen = torch.Generator().manual_seed(3)
    L, N, C, K = 7, 13, 40, 5
    x = torch.randn(L, N, C, generator=gen).to(dtype)
    w, b = torch.randn(C, K, generator=gen), torch.randn(C, generator=gen)
    xr = x.float().clone().detach().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv1d(xr.permute(1, 2, 0), wr.unsqueeze(1), br, padding=(K - 1) // 2, groups=C).permute(2, 0, 1)  # conv along L
    y = ops.dwconv_outer_fwd(x.to(device), w.to(device), b.to(device))
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(y.float().cpu(), ref.detach(), **tol)
    dy = torch.randn(L, N, C, generator=gen).to(dtype)
    ref.backward(dy.float())
    dx, dw = ops.dwconv_outer_bwd(dy.to(device), x.to(device), w.to(device))
    torch.testing.assert_close(dx.float().cpu(), xr.grad, **tol)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=2e-3, atol=2e-3)
    # batch norm + hardswish, train and eval
    rows = L * N
    x2 = x.view(rows, C)
    gam, bet = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    for train in (True, False):
        bn = torch.nn.BatchNorm1d(C)
        with torch.no_grad():
            bn.weight.copy_(gam), bn.bias.copy_(bet)
            bn.running_mean.copy_(torch.randn(C, generator=gen) * 0.1), bn.running_var.copy_(torch.rand(C, generator=gen) + 0.5)
        rm, rv = bn.running_mean.clone().to(device), bn.running_var.clone().to(device)
        bn.train(train)
        xr2 = x2.float().clone().detach().requires_grad_(True)
        yr = F.hardswish(bn(xr2))
        yk, mean, invstd = ops.bn_act_fwd(x2.to(device), gam.to(device), bet.to(device), rm, rv, 1e-5, 0.1, train, "hardswish")
        torch.testing.assert_close(yk.float().cpu(), yr.detach(), **tol)
        torch.testing.assert_close(rm.cpu(), bn.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(rv.cpu(), bn.running_var, rtol=1e-4, atol=1e-5)
        dy2 = torch.randn(rows, C, generator=gen).to(dtype)
        yr.backward(dy2.float())
        dxk, dg, db = ops.bn_act_bwd(dy2.to(device), x2.to(device), gam.to(device), bet.to(device), mean, invstd, train, "hardswish")
        torch.testing.assert_close(dxk.float().cpu(), xr2.grad, **tol)
        torch.testing.assert_close(dg.cpu(), bn.weight.grad, rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(db.cpu(), bn.bias.grad, rtol=2e-3, atol=2e-3)
