"""GPU: the fused update (js2t_adamw_items: AdamW + bf16 shadow + TRANSPOSED bf16 shadow + LayerNorm-fold weights in one pass
per matrix) against the separate passes it replaces (js2t_adamw, js2t_transpose_groups, js2t_fold_ln_weights) - bit for bit on
every buffer, over several updates of a model at LS100 width (folds and transposed shadows exist) and of the tiny golden model
(row counts that are no multiple of the kernel's units / of 8).  The golden three-update comparison with the reference
(tests/test_hip_train_step.py) runs through the fused kernel as well: it is the default (reference builders.py:112-114)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both_ways(step, batch_fns, n_updates):
    """One TrainStep; per update: forward + backward once (two runs of the backward pass differ in the last place - split-K
    atomics), then the SAME gradient through the separate passes and through the fused kernel from the SAME state."""
    from joeys2t_amd import builders
    st, opt = step.store, step.optimizer
    # this comparison is about the update's arithmetic: every gradient is cleared by both forms (the un-cleared, overwritten weight
    # gradients have their own test below)
    opt.keep = step.rt.wgrad_queue.cand = None

    def state():
        out = {"flat": st.flat, "m": opt.exp_avg, "v": opt.exp_avg_sq, "grad": st.flat_grad, "norm": opt.norm_clip}
        if st.flat_lp is not None:
            out["lp"] = st.flat_lp
        if st.flat_lp_t is not None:
            out["lp_t"] = st.flat_lp_t
        for i, f in enumerate(v for v in st._folds.values() if v is not None):
            out[f"fold{i}.w"], out[f"fold{i}.bias"] = f.w, f.bias
        return out

    took = []
    for u in range(n_updates):
        step.micro_step(batch_fns[u % len(batch_fns)](), update=False)
        torch.cuda.synchronize()
        before = {k: v.detach().clone() for k, v in state().items()}
        t0 = opt.t
        results = []
        for fused in (False, True):
            for k, v in state().items():
                v.copy_(before[k])
            opt.t = t0
            builders.FUSED_UPDATE = fused
            try:
                opt.clip_and_step(step.clip_grad_norm, zero_grad=True)
            finally:
                builders.FUSED_UPDATE = True
            torch.cuda.synchronize()
            took.append(opt._fused_plan() is not None if fused else False)
            results.append({k: v.detach().clone() for k, v in state().items()})
        ref, got = results
        assert ref.keys() == got.keys()
        for k in ref:
            assert torch.equal(ref[k], got[k]), (u, k, (ref[k].float() - got[k].float()).abs().max().item())
        assert float(got["grad"].abs().max()) == 0.0 and not torch.equal(got["flat"], before["flat"])
        step.after_update()
    return took, state()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fused_update_equals_separate_passes_ls100_width(device, dtype):
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    V = 300
    torch.manual_seed(5)
    model = make_model(width_cfg(4, 2, 2), V, None, device, dtype, 0.3, train=True)
    data = [synth_batch(V, [200, 170, 150], [9, 7, 5], 1), synth_batch(V, [180, 180, 120, 90], [6, 8, 5, 4], 2)]
    batches = [lambda d=d: hip_batch(*d, device) for d in data]
    step = TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3, weight_decay=0.01)
    took, final = _both_ways(step, batches, 3)
    assert took == [False, True] * 3
    if dtype == torch.bfloat16:
        # QKV / FFN1 of two encoder layers, self-QKV / cross-Q / FFN1 of two decoder layers; all of them inside the kernel
        assert len(step.store._fold_rows) >= 6 and "lp_t" in final and step.optimizer._plan["left_folds"] is None


def test_fused_update_on_odd_shapes(device):
    """the golden tiny model: 16-wide layers, a 20-row vocabulary (rows % 8 != 0: scalar stores of the transposed image)"""
    from joeys2t_amd.training import TrainStep
    from conftest import load_golden
    from test_hip_model import batch_kwargs, build
    g = load_golden("model_pre")
    for dtype in (torch.bfloat16, torch.float32):
        model, _ = build("model_pre", device, dtype, train=True)
        step = TrainStep(model, learning_rate=1e-3, clip_grad_norm=1.0, weight_decay=0.01)
        took, _ = _both_ways(step, [lambda: batch_kwargs(g, device)], 2)
        assert took == [False, True] * 2


def _three_updates(device, dtype, overwrite, monkeypatch, batch_multiplier=2):
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    monkeypatch.setenv("JS2T_WGRAD_OVERWRITE", "1" if overwrite else "0")
    V = 300
    torch.manual_seed(5)
    # six encoder layers: their QKV / feed-forward weight gradients have > 256 output tiles per group - un-split whatever the batch
    model = make_model(width_cfg(4, 6, 1), V, None, device, dtype, 0.3, train=True)
    data = [synth_batch(V, [200, 170, 150], [9, 7, 5], 1), synth_batch(V, [180, 180, 120, 90], [6, 8, 5, 4], 2)]
    step = TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=0.5, learning_rate_warmup=3,
                     batch_multiplier=batch_multiplier)
    norms, modes = [], []
    real_take = step.rt.wgrad_queue.take

    def take():
        plan = real_take()
        modes.append([k[6] for k, _ in plan])
        return plan

    step.rt.wgrad_queue.take = take
    for i in range(3 * batch_multiplier):
        step.micro_step(hip_batch(*data[i % 2], device))
        if (i + 1) % batch_multiplier == 0:
            norms.append(step.read_stats()["grad_norm"])
    torch.cuda.synchronize()
    return step, norms, modes


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_overwritten_weight_gradients_and_epilogue_norm(device, dtype, monkeypatch):
    """Un-split weight-gradient products overwrite their dW in the first micro-batch of an update (the update does not clear
    those pieces) and leave the sums of squares for clip_grad_norm_ behind in the last one: same gradient norms and parameters as
    clearing + accumulating + a pass over the whole gradient (two runs of one procedure differ by the arrival order of the
    split-K atomics; the bound is that of tests/test_hip_full_size.py)."""
    ref, norms_ref, modes_ref = _three_updates(device, dtype, False, monkeypatch)
    got, norms, modes = _three_updates(device, dtype, True, monkeypatch)
    assert all(m == 0 for ms in modes_ref for m in ms)
    kept = got.optimizer.keep
    if dtype == torch.bfloat16:  # (fp32 compute multiplies at once, product by product: nothing is queued)
        # first micro-batch of an update: overwrite (1); last: sums of squares (2); six flushes, alternating
        assert any(m & 1 for m in modes[0]) and not any(m & 2 for m in modes[0])
        assert any(m & 2 for m in modes[1]) and not any(m & 1 for m in modes[1])
        assert got.optimizer.collector is not None and ref.optimizer.collector is None
        assert kept and any(it[6] for launch in got.optimizer._plan["launches"] for it in launch[0].tolist())
    assert abs(norms[0] - norms_ref[0]) <= 2e-6 * abs(norms_ref[0]), (norms, norms_ref)  # same gradient, two ways to its norm
    for a, b in zip(norms, norms_ref):  # later updates: two runs drift apart by their atomics' arrival order (also without the switch)
        assert abs(a - b) <= (2e-4 if dtype == torch.float32 else 2e-2) * abs(b), (norms, norms_ref)
    rel = ((got.store.flat - ref.store.flat).norm() / ref.store.flat.norm()).item()
    assert rel < (1e-4 if dtype == torch.float32 else 2e-3), rel
    # the pieces the update keeps hold the last gradient, everything else is cleared
    g = got.store.flat_grad
    mask = torch.zeros_like(g, dtype=torch.bool)
    for lo, hi in kept.r:
        mask[lo:hi] = True
    assert float(g[~mask].abs().max()) == 0.0
    if dtype == torch.bfloat16:
        assert float(g[mask].abs().max()) > 0.0
    else:  # nothing was queued, so nothing was overwritten: the update clears everything as before
        assert not kept and float(g.abs().max()) == 0.0


def test_overwrite_single_micro_batch_modes(device, monkeypatch):
    """batch_multiplier 1: every flush both overwrites and collects (mode 3) for the groups that qualify"""
    step, norms, modes = _three_updates(device, torch.bfloat16, True, monkeypatch, batch_multiplier=1)
    assert all(any(m == 3 for m in ms) for ms in modes) and all(np.isfinite(norms))


def test_frozen_parameters_keep_the_plain_kernel(device):
    """`freeze: True` sub-networks: per-range launches of js2t_adamw (torch's AdamW skips parameters without a gradient)"""
    from joeys2t_amd.training import TrainStep
    from test_hip_model import build
    model, _ = build("model_pre", device, torch.bfloat16, train=True)
    for p in model.encoder.parameters():
        p.requires_grad_(False)
    step = TrainStep(model, learning_rate=1e-3)
    assert step.optimizer._fused_plan() is None
