"""GPU: the fused update (js2t_adamw_items: AdamW + bf16 shadow + TRANSPOSED bf16 shadow + LayerNorm-fold weights in one pass
per matrix) against the separate passes it replaces (js2t_adamw, js2t_transpose_groups, js2t_fold_ln_weights) - bit for bit on
every buffer, over several updates of a model at LS100 width (folds and transposed shadows exist) and of the tiny golden model
(row counts that are no multiple of the kernel's units / of 8).  The golden three-update comparison with the reference
(tests/test_hip_train_step.py) runs through the fused kernel as well: it is the default (reference builders.py:112-114)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(fused, device, dtype, n_updates, make, batches):
    from joeys2t_amd import builders
    from joeys2t_amd.training import TrainStep
    builders.FUSED_UPDATE = fused
    try:
        model = make()
        step = TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3,
                         weight_decay=0.01, normalization="batch")
        for i in range(n_updates):
            step.micro_step(batches[i % len(batches)]())
        torch.cuda.synchronize()
        st, opt = step.store, step.optimizer
        out = {"flat": st.flat, "m": opt.exp_avg, "v": opt.exp_avg_sq, "grad": st.flat_grad}
        if st.flat_lp is not None:
            out["lp"] = st.flat_lp
        if st.flat_lp_t is not None:
            out["lp_t"] = st.flat_lp_t
        for i, f in enumerate(v for v in st._folds.values() if v is not None):
            out[f"fold{i}.w"], out[f"fold{i}.bias"] = f.w, f.bias
        took_fused = opt._plan is not None
        return {k: v.detach().clone() for k, v in out.items()}, took_fused, len(st._fold_rows)
    finally:
        builders.FUSED_UPDATE = True


def _compare(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), (k, (a[k].float() - b[k].float()).abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fused_update_equals_separate_passes_ls100_width(device, dtype):
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    V = 300
    torch.manual_seed(5)
    base = make_model(width_cfg(4, 2, 2), V, None, None, None, 0.3)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    data = [synth_batch(V, [200, 170, 150], [9, 7, 5], 1), synth_batch(V, [180, 180, 120, 90], [6, 8, 5, 4], 2)]
    batches = [lambda d=d: hip_batch(*d, device) for d in data]
    make = lambda: make_model(width_cfg(4, 2, 2), V, sd, device, dtype, 0.3, train=True)  # noqa: E731
    ref, took, _ = _run(False, device, dtype, 3, make, batches)
    got, took_fused, n_folds = _run(True, device, dtype, 3, make, batches)
    assert not took and took_fused
    if dtype == torch.bfloat16:
        assert n_folds >= 6 and "lp_t" in got  # QKV / FFN1 of two encoder layers, self-QKV / cross-Q / FFN1 of two decoder layers
    _compare(ref, got)


def test_fused_update_on_odd_shapes(device):
    """the golden tiny model: 16-wide layers, a 20-row vocabulary (rows % 8 != 0: scalar stores of the transposed image)"""
    from test_hip_model import batch_kwargs, build
    for dtype in (torch.bfloat16, torch.float32):
        def make():
            model, _ = build("model_pre", device, dtype, train=True)
            return model
        from conftest import load_golden
        g = load_golden("model_pre")
        batches = [lambda: batch_kwargs(g, device)]
        ref, took, _ = _run(False, device, dtype, 2, make, batches)
        got, took_fused, _ = _run(True, device, dtype, 2, make, batches)
        assert not took and took_fused
        _compare(ref, got)


def test_frozen_parameters_keep_the_plain_kernel(device):
    """`freeze: True` sub-networks: per-range launches of js2t_adamw (torch's AdamW skips parameters without a gradient)"""
    from joeys2t_amd.training import TrainStep
    from test_hip_model import build
    model, _ = build("model_pre", device, torch.bfloat16, train=True)
    for p in model.encoder.parameters():
        p.requires_grad_(False)
    step = TrainStep(model, learning_rate=1e-3)
    assert step.optimizer._fused_plan() is None
