"""GPU: the train-step driver (joeys2t_amd.training.TrainStep) against a replay of the reference's own
`_train_step` + update tail (training.py:541-596,436-456) captured with the reference's builders
(tests/golden/train_steps.npz): per-micro-batch normalised losses, global gradient norms, the learning-rate sequence
(first update at the un-warmed configured rate) and the parameters after 3 updates with batch_multiplier = 2."""
import copy

import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from golden_cfg import tiny_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap_ctc", [False, True])
def test_three_updates_match_reference(device, overlap_ctc):
    """overlap_ctc: the CTC branch (projection, loss, their backward) on the runtime's second stream."""
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.model import build_model
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    g = load_golden("train_steps")
    model = build_model(copy.deepcopy(tiny_cfg("pre")), None, Vocabulary.synthetic(20))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.load_state_dict(golden_sd(g, "sd0."))
    model.finalize(device, torch.float32)
    step = TrainStep(model, learning_rate=2.0e-3, adam_betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=1.0,
                     learning_rate_warmup=2, learning_rate_min=1.0e-6, normalization="batch", batch_multiplier=2, n_gpu=1,
                     overlap_ctc=overlap_ctc)
    lrs, norms = [], []
    for i in range(6):
        b = Batch(src=torch.from_numpy(g[f"mb{i}.src"]), src_length=torch.from_numpy(g[f"mb{i}.src_length"]),
                  src_prompt_mask=None, trg=torch.from_numpy(g[f"mb{i}.trg"]), trg_length=torch.from_numpy(g[f"mb{i}.trg_length"]),
                  trg_prompt_mask=None, indices=torch.arange(3), device=device, pad_index=1, eos_index=3, is_train=True,
                  task="S2T", n_gpu=1)
        lr_before = step.optimizer.param_groups[0]["lr"]
        loss = step.micro_step(b)
        assert abs(loss.item() - g["losses"][i, 0]) <= 1e-4 * abs(g["losses"][i, 0]), (i, loss.item(), g["losses"][i, 0])
        if (i + 1) % 2 == 0:
            lrs.append(lr_before)
            norms.append(float(step.optimizer.norm_clip[0]))
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    assert step.optimizer.param_groups[0]["lr"] == pytest.approx(float(g["lr_next"]), rel=1e-12)
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4)
    stats = step.read_stats()
    assert stats["loss"] == pytest.approx(g["losses"][:, 0].sum(), rel=1e-4)
    assert stats["nll"] == pytest.approx(g["losses"][:, 1].sum(), rel=1e-4)
    assert stats["ctc"] == pytest.approx(g["losses"][:, 2].sum(), rel=1e-4)
    assert int(stats["n_correct"]) == int(g["losses"][:, 3].sum())
    ref = golden_sd(g, "sd1.")
    worst = 0.0
    for n, p in model.named_parameters():
        if "k_layer.bias" in n:
            continue  # true gradient is exactly zero: Adam turns rounding noise into +-lr steps in any implementation
        err = (p.detach().cpu() - ref[n]).abs().max().item()
        worst = max(worst, err)
        assert err <= 3e-4, (n, err)
    print("worst parameter deviation after 3 updates:", worst)


def test_resume_from_reference_checkpoint(device, tmp_path):
    """A checkpoint written by the reference's code after its second update (model, torch AdamW state, scheduler, counters)
    is loaded into the HIP train step; the third update then lands on the reference's parameters.  Saving and re-loading
    through this package's own writer (same layout) gives the same state."""
    from conftest import GOLDEN
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.model import build_model
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    g = load_golden("train_steps")

    def fresh():
        torch.manual_seed(1)
        model = build_model(copy.deepcopy(tiny_cfg("pre")), None, Vocabulary.synthetic(20))
        model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
        model.finalize(device, torch.float32)
        return model, TrainStep(model, learning_rate=2.0e-3, adam_betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=1.0,
                                learning_rate_warmup=2, learning_rate_min=1.0e-6, normalization="batch", batch_multiplier=2, n_gpu=1)

    model, step = fresh()
    step.init_from_checkpoint(GOLDEN / "ref_checkpoint_after2.ckpt")
    assert step.steps == 2 and step.optimizer.t == 2
    assert step.optimizer.param_groups[0]["lr"] == pytest.approx(float(g["lrs"][2]), rel=1e-12)
    path = tmp_path / "resaved.ckpt"
    step.save_checkpoint(path)
    model2, step2 = fresh()
    step2.init_from_checkpoint(path)
    torch.testing.assert_close(step2.optimizer.exp_avg_sq, step.optimizer.exp_avg_sq)
    for (n, a), (_, b) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.equal(a, b), n
    for st, mdl in ((step, model), (step2, model2)):
        for i in (4, 5):
            b = Batch(src=torch.from_numpy(g[f"mb{i}.src"]), src_length=torch.from_numpy(g[f"mb{i}.src_length"]), src_prompt_mask=None,
                      trg=torch.from_numpy(g[f"mb{i}.trg"]), trg_length=torch.from_numpy(g[f"mb{i}.trg_length"]), trg_prompt_mask=None,
                      indices=torch.arange(3), device=device, pad_index=1, eos_index=3, is_train=True, task="S2T", n_gpu=1)
            loss = st.micro_step(b)
            assert abs(loss.item() - g["losses"][i, 0]) <= 1e-4 * abs(g["losses"][i, 0])
        assert float(st.optimizer.norm_clip[0]) == pytest.approx(float(g["grad_norms"][2]), rel=1e-4)
        ref = golden_sd(g, "sd1.")
        for n, p in mdl.named_parameters():
            if "k_layer.bias" in n:
                continue
            assert (p.detach().cpu() - ref[n]).abs().max().item() <= 3e-4, n


def test_checkpoint_written_here_has_the_reference_loaders_keys(device, tmp_path):
    """Two-way compatibility: the reference's loader indexes six stats keys (training.py:819-826) and calls .cpu() on
    train_iter_state (:287); a checkpoint written by TrainStep must carry them in the reference's types."""
    from joeys2t_amd.model import build_model
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    model = build_model(copy.deepcopy(tiny_cfg("pre")), None, Vocabulary.synthetic(20))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.finalize(device, torch.float32)
    step = TrainStep(model, batch_multiplier=1)
    path = tmp_path / "ours.ckpt"
    step.save_checkpoint(path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    st = ck["stats_state"]
    for key in ("epochs", "steps", "total_tokens", "total_correct", "best_ckpt_score", "best_ckpt_iter"):
        assert key in st, key  # TrainStatistics.load_state_dict reads exactly these
    assert isinstance(st["steps"], int) and isinstance(st["total_tokens"], int) and isinstance(st["best_ckpt_score"], float)
    assert torch.is_tensor(ck["train_iter_state"]) and ck["train_iter_state"].cpu().dtype == torch.uint8  # a generator state
    torch.Generator().set_state(ck["train_iter_state"])  # what batch_sampler.set_state does with it (datasets.py:1243-1246)
    assert set(ck) == {"model_state", "optimizer_state", "scaler_state", "scheduler_state", "train_iter_state", "stats_state"}
