"""GPU: BASELINE.json configs[4] as a TRAINABLE model - `encoder.type: conformer` (+ `rel_pos_clip`, `depthwise_conv_kernel_size`)
through build_model, Model.forward("loss"), backward and TrainStep.

EXTENSION WITHOUT A REFERENCE TARGET FOR THE COMPOSITION: the reference ships ConformerEncoder (encoders.py:376-445,
transformer_layers.py:410-565) but its build_model refuses it (model.py:417-421), it has no relative-position term and no fp8
(config.py:223-225).  What is pinned: the encoder's weights and behaviour by the capture of the reference class
(tests/golden/conformer.npz, tests/test_hip_conformer.py), decoder / losses by the S2T captures; the composition is checked
against the oracle's restatement of Model._encode_decode (model.py:170-239) over that encoder - fp32 at 1e-4, every gradient."""
import copy

import numpy as np
import pytest
import torch

from conftest import load_golden
from golden_cfg import SPECIALS
from oracle import s2t_oracle as O

pytestmark = pytest.mark.gpu
R = 3


def conformer_cfg(d=16, ff=32, heads=2, layers=2, in_ch=8, conv_ch=24, dwk=5, rel=R, dropout=0.0, dec_layers=2):
    return {
        "initializer": "xavier_uniform", "bias_initializer": "zeros", "embed_initializer": "xavier_uniform",
        "tied_embeddings": False, "tied_softmax": False,
        "encoder": {"type": "conformer", "num_layers": layers, "num_heads": heads, "embeddings": {"embedding_dim": in_ch},
                    "hidden_size": d, "ff_size": ff, "dropout": dropout, "freeze": False, "subsample": True,
                    "conv_kernel_sizes": [5, 5], "conv_channels": conv_ch, "in_channels": in_ch, "layer_norm": "pre",
                    "depthwise_conv_kernel_size": dwk, "rel_pos_clip": rel},
        "decoder": {"type": "transformer", "num_layers": dec_layers, "num_heads": heads,
                    "embeddings": {"embedding_dim": d, "scale": True, "dropout": dropout}, "hidden_size": d, "ff_size": ff,
                    "dropout": dropout, "freeze": False, "layer_norm": "pre", "activation": "relu"},
    }


def _batch_from(src, lengths, V, seed, device):
    from joeys2t_amd.batch import Batch
    g = torch.Generator().manual_seed(seed)
    B = src.shape[0]
    tl = torch.randint(2, 5, (B, ), generator=g)
    L = int(tl.max()) + 2
    trg = torch.full((B, L), 1, dtype=torch.long)
    for b in range(B):
        n = int(tl[b])
        trg[b, 0], trg[b, 1 + n] = 2, 3
        trg[b, 1:1 + n] = torch.randint(4, V, (n, ), generator=g)
    hb = Batch(src=src.clone(), src_length=lengths.clone(), src_prompt_mask=None, trg=trg, trg_length=tl + 2, trg_prompt_mask=None,
               indices=torch.arange(B), device=device, pad_index=1, eos_index=3, is_train=True, task="S2T", n_gpu=1)
    ob = O.make_batch(src, lengths, trg, tl + 2, SPECIALS["pad"], SPECIALS["eos"])
    return hb, ob


def _golden_model(device, dtype=torch.float32):
    """build_model over the config above; encoder weights = the capture of the reference's ConformerEncoder, relative-position
    tables and decoder random (seeded)."""
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    g = load_golden("conformer")
    V = 20
    torch.manual_seed(7)
    cfg = conformer_cfg()
    model = build_model(copy.deepcopy(cfg), None, Vocabulary.synthetic(V))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    pre = "pre."
    sd = {"encoder." + k[len(pre) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre + "sd0.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(not k.startswith("encoder.") or k.endswith("pe.pe") or k.endswith("rel_pos_bias") for k in missing)
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for layer in model.encoder.layers:
            layer.src_src_att.rel_pos_bias.copy_(0.5 * torch.randn(2, 2 * R + 1, generator=gen))
        for n, p in model.named_parameters():
            if n.startswith("decoder") and ("bias" in n or "layer_norm" in n):
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    osd = {k: v.clone() for k, v in model.state_dict().items()}
    model.finalize(device, dtype)
    ocfg = copy.deepcopy(cfg)
    ocfg["encoder"]["alpha"] = ocfg["decoder"]["alpha"] = 1.0
    return model, osd, ocfg, g, V


def test_build_model_conformer_is_an_s2t_extension():
    from joeys2t_amd.builders import ConfigurationError
    from joeys2t_amd.encoders import ConformerEncoder
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    model = build_model(conformer_cfg(), None, Vocabulary.synthetic(20))
    assert isinstance(model.encoder, ConformerEncoder) and model.encoder.layers[0].src_src_att.rel_pos_bias.shape == (2, 2 * R + 1)
    assert model.encoder.layers[0].conv_module.depthwise_conv.weight.shape[-1] == 5
    assert model.decoder.ctc_output_layer is not None
    with pytest.raises(ConfigurationError):
        build_model(conformer_cfg(), Vocabulary.synthetic(20), Vocabulary.synthetic(20))  # text source: it always sub-samples


@pytest.mark.parametrize("train", [True, False])
def test_conformer_s2t_loss_and_gradients_match_oracle(device, train):
    model, osd, ocfg, g, V = _golden_model(device)
    model.train(train)
    src, lengths = torch.from_numpy(g["pre.src"]), torch.from_numpy(g["pre.src_length"])
    hb, ob = _batch_from(src, lengths, V, 3, device)
    for k, v in osd.items():
        if v.is_floating_point() and "running" not in k and "pe.pe" not in k:
            v.requires_grad_(True)
    stats = {}
    rt, rx, rc, rn, _, _ = O.model_loss(osd, ocfg, ob, SPECIALS, 0.1, 0.3, train=train, new_stats=stats)
    rt.backward()
    total, xent, ctc, ncor = model(return_type="loss", **vars(hb))
    total.backward()
    for got, want in ((total, rt), (xent, rx), (ctc, rc)):
        assert abs(got.item() - want.item()) <= 1e-4 * abs(want.item()), (got.item(), want.item())
    assert int(ncor) == int(rn)
    n_checked = 0
    for n, p in model.named_parameters():
        want = osd[n].grad
        assert want is not None and p.grad is not None, n
        scale = float(want.abs().max()) + 1e-6
        err = float((p.grad.cpu() - want).abs().max())
        assert err <= 1e-4 * scale + 2e-5, (n, err, scale)
        n_checked += 1
    assert n_checked > 80
    assert float(osd["encoder.layers.0.src_src_att.rel_pos_bias"].grad.abs().max()) > 1e-4
    if train:  # BatchNorm running statistics moved as the reference's module moves them
        for k, v in stats.items():
            torch.testing.assert_close(model.state_dict()[k].cpu(), v, rtol=1e-4, atol=1e-5)


def _wide_model(device, dtype, seed=5, dropout=0.0):
    """wide enough for the fused attention kernels (2 heads of 64), the persistent products and the e4m3 products"""
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    torch.manual_seed(seed)
    cfg = conformer_cfg(d=128, ff=256, heads=2, layers=3, in_ch=16, conv_ch=128, dwk=7, rel=8, dropout=dropout, dec_layers=2)
    model = build_model(copy.deepcopy(cfg), None, Vocabulary.synthetic(40))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    with torch.no_grad():
        for layer in model.encoder.layers:
            layer.src_src_att.rel_pos_bias.normal_(0.0, 0.3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.finalize(device, dtype)
    return model, sd, cfg


def _wide_batches(device, n, V=40):
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(100 + i)
        B, T = 4, 203
        lengths = torch.tensor([T, T - 30, T - 61, T - 90])
        src = torch.randn(B, T, 16, generator=g)
        for b in range(B):
            src[b, lengths[b]:] = 1.0
        out.append(_batch_from(src, lengths, V, 200 + i, device))
    return out


def _train(model, batches, n_updates, **kw):
    from joeys2t_amd.training import TrainStep
    step = TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=2,
                     normalization="batch", **kw)
    losses = []
    for i in range(n_updates):
        step.micro_step(batches[i % len(batches)][0])
        losses.append(step.read_stats()["loss"])
    torch.cuda.synchronize()
    return losses, step


def test_conformer_train_step_fp32_follows_oracle_autograd(device):
    """Three updates through TrainStep (deferred grouped weight gradients, flat AdamW) against torch's clip + AdamW on the
    oracle's gradients of the same Conformer model."""
    model, sd, cfg = _wide_model(device, torch.float32)
    batches = _wide_batches(device, 2)
    losses, step = _train(model, batches, 3)
    ocfg = copy.deepcopy(cfg)
    ocfg["encoder"]["alpha"] = ocfg["decoder"]["alpha"] = 1.0
    osd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and "pe.pe" not in k) for k, v in sd.items()}
    params = [v for v in osd.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.0)
    from joeys2t_amd.builders import WarmupInverseSquareRootScheduler
    ref_losses = []

    class _G:  # the scheduler only touches param_groups
        param_groups = opt.param_groups

    sched = WarmupInverseSquareRootScheduler(_G, peak_rate=1e-3, warmup=2, min_rate=1e-6)
    for i in range(3):
        ob = batches[i % 2][1]
        stats = {}
        opt.zero_grad()
        total, _, _, _, _, _ = O.model_loss(osd, ocfg, ob, SPECIALS, 0.1, 0.3, train=True, new_stats=stats)
        (total / ob["src"].shape[0]).backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        sched.step(i)
        with torch.no_grad():
            for k, v in stats.items():
                osd[k].copy_(v)
        ref_losses.append(total.item() / ob["src"].shape[0])
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-4 * abs(b), (losses, ref_losses)
    # the three UPDATES agree: Adam moves a coordinate by ~lr whatever its gradient's size, so single coordinates whose tiny gradient
    # differs in sign end 2 lr apart per step; over all 0.6 M parameters the movement must be the oracle's
    num = den = 0.0
    for n, p in model.named_parameters():
        moved_ref = osd[n].detach() - sd[n]
        num += float((p.detach().cpu() - sd[n] - moved_ref).pow(2).sum())
        den += float(moved_ref.pow(2).sum())
    assert (num / den)**0.5 < 0.05, (num / den)**0.5


def test_conformer_train_step_bf16_and_fp8_forward(device):
    """bf16 TrainStep on the Conformer model stays near fp32; FP8_FORWARD on against off: the e4m3 products are taken, the loss
    stays within the bound test_hip_config5 derives for the encoder output (8 % relative on the states -> a few % on the loss),
    training goes down in both modes."""
    from joeys2t_amd import functional as Fn
    batches = _wide_batches(device, 2)
    m32, _, _ = _wide_model(device, torch.float32)
    l32, _ = _train(m32, batches, 6)
    m16, _, _ = _wide_model(device, torch.bfloat16)
    l16, _ = _train(m16, batches, 6, overlap_ctc=True)
    seen = []
    real = Fn.ops.gemm

    def spy(A, Bm, Cc, **k):
        seen.append(A.dtype)
        return real(A, Bm, Cc, **k)

    m8, _, _ = _wide_model(device, torch.bfloat16)
    old = Fn.FP8_FORWARD
    Fn.FP8_FORWARD = True
    Fn.ops.gemm = spy
    try:
        l8, _ = _train(m8, batches, 6, overlap_ctc=True)
    finally:
        Fn.FP8_FORWARD = old
        Fn.ops.gemm = real
    assert sum(d == torch.float8_e4m3fn for d in seen) >= 6 * 3 * 5 - 12, len(seen)
    for a, b in zip(l16, l32):
        assert abs(a - b) <= 3e-2 * abs(b), (l16, l32)
    for a, b in zip(l8, l16):
        assert abs(a - b) <= 6e-2 * abs(b), (l8, l16)
    assert l16[-1] < l16[0] and l8[-1] < l8[0] and all(np.isfinite(l8))


def _soak(device, dtype, steps=10, deterministic=True):
    """a small config-5 model (head size 64: the fused attention kernels with the relative-position bias; depthwise convolution,
    BatchNorm) trained for a few updates on two fixed batches -> (losses per update, parameters, Adam moments)"""
    from joeys2t_amd._lib import lib
    from joeys2t_amd.model import build_model
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    V = 40
    torch.manual_seed(11)
    cfg = conformer_cfg(d=128, ff=256, heads=2, layers=2, in_ch=16, conv_ch=48, dwk=7, rel=6, dropout=0.1, dec_layers=1)
    model = build_model(copy.deepcopy(cfg), None, Vocabulary.synthetic(V))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    with torch.no_grad():
        for layer in model.encoder.layers:
            layer.src_src_att.rel_pos_bias.normal_(0.0, 0.1, generator=torch.Generator().manual_seed(3))
    model.finalize(device, dtype, seed=11)
    g = torch.Generator().manual_seed(23)
    data = []
    for s in range(2):
        lengths = torch.tensor([150, 141, 120, 97, 96])
        src = torch.randn(5, 150, 16, generator=g)
        for b in range(5):
            src[b, lengths[b]:] = 1.0
        data.append(_batch_from(src if dtype == torch.float32 else src.to(dtype).float(), lengths, V, 40 + s, device)[0])
    try:
        step = TrainStep(model, learning_rate=2e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=4, overlap_ctc=True,
                         deterministic=deterministic)
        losses = []
        for i in range(steps):
            b = copy.copy(data[i % 2])
            if dtype != torch.float32:
                b.src = data[i % 2].src.to(dtype)
            step.micro_step(b)
            losses.append(step.read_stats()["loss"])
        torch.cuda.synchronize()
        return losses, step.store.flat.clone(), step.optimizer.exp_avg.clone(), step.optimizer.exp_avg_sq.clone()
    finally:
        lib().js2t_set_deterministic(0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_config5_deterministic_runs_agree_and_train(device, dtype):
    """js2t_set_deterministic now also orders the extension kernels' sums - the relative-position bias gradient (2^-32 fixed-point
    histogram: integer atomics commute), the depthwise-convolution weight gradient and the BatchNorm statistics / parameter
    gradients (one block per column group): two runs of ten updates agree BIT FOR BIT (bf16: the fused attention kernels; fp32: the
    materialised path), and the loss goes down."""
    a = _soak(device, dtype)
    b = _soak(device, dtype)
    assert a[0] == b[0], (a[0], b[0])
    for x, y, name in zip(a[1:], b[1:], ("parameters", "exp_avg", "exp_avg_sq")):
        assert torch.equal(x, y), (name, (x - y).abs().max().item(), int((x != y).sum()))
    losses = a[0]
    assert all(np.isfinite(losses)) and min(losses[-2:]) < 0.8 * max(losses[:2]), losses
    # the ordered forms compute the default step's numbers (other orders of the same sums)
    c = _soak(device, dtype, deterministic=False)
    for u, v in zip(a[0][:3], c[0][:3]):
        assert abs(u - v) <= (2e-2 if dtype == torch.bfloat16 else 1e-4) * abs(v), (a[0], c[0])
