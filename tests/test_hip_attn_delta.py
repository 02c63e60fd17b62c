"""delta = rowsum(dO * O) of the attention backward handed over as partial sums from the epilogue of the product that made dO
(js2t_gemm_desc.dot_src / dot_partial), and the two backward passes as one grid (js2t_attn_desc.delta_partial).
Reference behaviour is unchanged: the gradients must equal the two-launch path's (joeynmt/transformer_layers.py:24-131)."""
import math

import pytest
import torch

from joeys2t_amd import _lib, ops

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("M,N,K", [(2592, 512, 512), (200, 256, 384), (77, 128, 64), (3000, 1024, 256)])
def test_gemm_dot_partial(device, M, N, K):
    a = rnd(M, K, seed=1).bfloat16().to(device)
    w_t = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().to(device)  # C = A W_t^T
    src = rnd(M, N + 64, seed=3).bfloat16().to(device)[:, :N]  # a row pitch of its own
    c = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    part = torch.full((M, N // 64), float("nan"), device=device)
    ops.gemm(a, w_t, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, dot=(src, part))
    plain = torch.empty_like(c)
    ops.gemm(a, w_t, plain, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    assert torch.equal(c, plain)  # the product itself is untouched
    ref = (c.float() * src.float()).view(M, N // 64, 64).sum(-1)
    assert torch.allclose(part, ref, rtol=1e-4, atol=1e-3 * ref.abs().max().item())


def test_gemm_dot_partial_rejects_what_it_cannot_do(device):
    M, N, K = 256, 192, 128  # N % 128 != 0
    a, w = torch.zeros(M, K, dtype=torch.bfloat16, device=device), torch.zeros(N, K, dtype=torch.bfloat16, device=device)
    c, src = torch.empty(M, N, dtype=torch.bfloat16, device=device), torch.zeros(M, N, dtype=torch.bfloat16, device=device)
    with pytest.raises(_lib.Js2tError):
        ops.gemm(a, w, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, dot=(src, torch.empty(M, N // 64, device=device)))


@pytest.mark.parametrize("dh,H,Tq,Tk,cross", [(64, 4, 150, 150, False), (128, 2, 333, 333, False), (64, 8, 27, 150, True), (64, 4, 700, 700, False)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_flash_bwd_with_delta_partial_equals_two_launches(device, dh, H, Tq, Tk, cross, p):
    B, d = 3, H * dh
    q = rnd(B * Tq, d, seed=1).bfloat16().to(device)
    kv = rnd(B * Tk, 2 * d, seed=2).bfloat16().to(device)
    lens = torch.tensor([Tk, Tk - 49 if Tk > 60 else Tk - 5, max(1, Tk // 2)])
    mask = (torch.arange(Tk)[None, :] < lens[:, None]).unsqueeze(1).to(device)
    rng = ops.DropoutRng(device, seed=3) if p else None
    out, lse = ops.flash_attn_fwd(q, 0, kv, 0, kv, d, B, H, Tq, Tk, dh, mask, p, rng, 5)
    go = rnd(B * Tq, d, seed=4).bfloat16().to(device)
    part = (go.float() * out.float()).view(B * Tq, d // 64, 64).sum(-1).contiguous()

    def run(**kw):
        dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
        ops.flash_attn_bwd(go, out, lse, q, 0, kv, 0, kv, d, dq, 0, dkv, 0, dkv, d, B, H, Tq, Tk, dh, mask, p, rng, 5, **kw)
        return dq, dkv

    dq0, dkv0 = run()
    dq1, dkv1 = run(delta_partial=part)
    _lib.lib().js2t_debug_attn_bwd_merge(0)
    try:
        dq2, dkv2 = run(delta_partial=part)
    finally:
        _lib.lib().js2t_debug_attn_bwd_merge(1)
    # one grid or two: the same bodies on the same delta -> the same bits
    assert torch.equal(dq1, dq2) and torch.equal(dkv1, dkv2)
    # delta from partial sums vs in-kernel: fp32 summation order only
    for x, y in ((dq0, dq1), (dkv0, dkv1)):
        assert torch.isfinite(y.float()).all()
        assert (x.float() - y.float()).norm() <= 2e-3 * x.float().norm()


def test_train_step_gradients_with_and_without_the_hand_over(device):
    """A model at LS100 width (transposed shadows exist, fused attention kernels): the flat gradient with the hand-over equals
    the one without within run-to-run noise, and the hand-over IS taken in every attention block of both stacks."""
    from joeys2t_amd import functional as F
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    V = 300
    torch.manual_seed(5)
    model = make_model(width_cfg(4, 2, 2), V, None, device, torch.bfloat16, 0.3, train=True)
    data = synth_batch(V, [200, 170, 150], [9, 7, 5], 1)
    step = TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3)
    taken = []
    real = ops.flash_attn_bwd

    def spy(*a, **kw):
        taken.append(kw.get("delta_partial") is not None)
        return real(*a, **kw)

    grads = {}
    ops.flash_attn_bwd = spy
    try:
        for on in (True, False, True):
            F.FUSE_ATTN_DELTA = on
            del taken[:]
            step.store.flat_grad.zero_()
            step.micro_step(hip_batch(*data, device), update=False)
            torch.cuda.synchronize()
            assert len(taken) == 6 and all(t == on for t in taken), taken  # 2 encoder self + 2 decoder self + 2 cross
            grads.setdefault(on, []).append(step.store.flat_grad.detach().clone())
    finally:
        ops.flash_attn_bwd = real
        F.FUSE_ATTN_DELTA = True
    ref = grads[False][0]
    noise = (grads[True][0] - grads[True][1]).norm()  # the same path twice: split-K atomics
    err = (grads[True][0] - ref).norm()
    assert torch.isfinite(ref).all() and err <= max(3.0 * noise.item(), 2e-3 * ref.norm().item()), (err, noise, ref.norm())
