"""GPU: TrainStep(deterministic=True) - the reference's set_seed asks cuDNN for deterministic kernels (helpers.py:93-104); here
js2t_set_deterministic puts every floating-point-atomic sum of the Transformer S2T train step on an ordered form (un-split weight
gradients, LayerNorm parameter gradients through the partial slab, ordered embedding / CTC gradients).  Two runs of three
updates from one state must agree BIT FOR BIT - parameters, both Adam moments, losses; the default mode only to the last place."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(device, dtype, deterministic, dropout):
    from joeys2t_amd._lib import lib
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    V = 300
    torch.manual_seed(5)
    cfg = width_cfg(4, 3, 2)
    cfg["encoder"]["dropout"] = cfg["decoder"]["dropout"] = dropout
    cfg["decoder"]["embeddings"]["dropout"] = dropout
    model = make_model(cfg, V, None, device, dtype, 0.3, train=True)
    # repeated target tokens and several utterances: the embedding and CTC gradients have rows with more than one contribution
    data = [synth_batch(V, [400, 370, 350, 300], [19, 17, 15, 12], 1), synth_batch(V, [380, 380, 320, 290], [16, 18, 15, 14], 2)]
    try:
        step = TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3,
                         batch_multiplier=2, overlap_ctc=True, deterministic=deterministic)
        assert bool(step.ctx.get("deterministic")) == deterministic and not lib().js2t_get_deterministic()  # the step's own setting, no process switch
        losses = []
        for i in range(6):
            step.micro_step(hip_batch(*data[i % 2], device))
            if i % 2:
                losses.append(step.read_stats()["loss"])
        torch.cuda.synchronize()
        return losses, step.store.flat.clone(), step.optimizer.exp_avg.clone(), step.optimizer.exp_avg_sq.clone()
    finally:
        lib().js2t_set_deterministic(0)


@pytest.mark.parametrize("dtype,dropout", [(torch.bfloat16, 0.1), (torch.bfloat16, 0.0), (torch.float32, 0.1)])
def test_two_deterministic_runs_agree_bit_for_bit(device, dtype, dropout):
    a = _run(device, dtype, True, dropout)
    b = _run(device, dtype, True, dropout)
    assert a[0] == b[0], (a[0], b[0])
    for x, y, name in zip(a[1:], b[1:], ("parameters", "exp_avg", "exp_avg_sq")):
        assert torch.equal(x, y), (name, (x - y).abs().max().item(), int((x != y).sum()))


def test_deterministic_mode_computes_the_same_step(device):
    """the ordered kernels are other summation orders of the same gradients: losses and parameters of the default mode to rounding"""
    det = _run(device, torch.bfloat16, True, 0.0)
    dflt = _run(device, torch.bfloat16, False, 0.0)
    for a, b in zip(det[0], dflt[0]):
        assert abs(a - b) <= 2e-3 * abs(b), (det[0], dflt[0])
    rel = ((det[1] - dflt[1]).norm() / dflt[1].norm()).item()
    assert rel < 2e-3, rel


def test_two_steps_in_one_process_one_deterministic(device):
    """the mode belongs to the step (js2t_ctx bound around its launches), not to the process - a deterministic
    TrainStep and a default one INTERLEAVED micro-batch by micro-batch in one process: the deterministic one must end on the bits of
    a deterministic run on its own (round 5's process-wide switch was whatever the last constructor wrote: both ran in one mode)."""
    from joeys2t_amd import _lib
    from joeys2t_amd._lib import lib
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import hip_batch, make_model, synth_batch, width_cfg
    alone = _run(device, torch.bfloat16, True, 0.1)
    V = 300
    steps, data = [], None
    for det in (True, False):
        torch.manual_seed(5)
        cfg = width_cfg(4, 3, 2)
        cfg["encoder"]["dropout"] = cfg["decoder"]["dropout"] = 0.1
        cfg["decoder"]["embeddings"]["dropout"] = 0.1
        model = make_model(cfg, V, None, device, torch.bfloat16, 0.3, train=True)
        steps.append(TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3,
                               batch_multiplier=2, overlap_ctc=True, deterministic=det))
    data = [synth_batch(V, [400, 370, 350, 300], [19, 17, 15, 12], 1), synth_batch(V, [380, 380, 320, 290], [16, 18, 15, 14], 2)]
    assert _lib.effective("deterministic") == 0  # nothing bound, no override
    seen = []
    for i in range(6):
        for st in steps:  # the default step runs between the deterministic one's micro-batches
            st.micro_step(hip_batch(*data[i % 2], device))
        with steps[0].ctx:
            seen.append(_lib.effective("deterministic"))
        with steps[1].ctx:
            seen.append(_lib.effective("deterministic"))
    torch.cuda.synchronize()
    assert seen == [1, 0] * 6 and _lib.effective("deterministic") == 0
    assert torch.equal(steps[0].store.flat, alone[1]) and torch.equal(steps[0].optimizer.exp_avg, alone[2])
    # and the process-wide setter is still a test override that wins over a context
    lib().js2t_set_deterministic(1)
    try:
        with steps[1].ctx:
            assert _lib.effective("deterministic") == 1
    finally:
        lib().js2t_set_deterministic(0)
