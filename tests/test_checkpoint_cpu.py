"""CPU: checkpoint compatibility with the reference: parameter names AND order of this package's model equal the
reference's (so torch-optimizer state indices line up), a checkpoint written by the reference's code loads, checkpoint
averaging and partial (encoder / decoder) initialisation follow the reference scripts."""
import copy

import numpy as np
import torch

from conftest import GOLDEN, load_golden
from golden_cfg import tiny_cfg

CKPT = GOLDEN / "ref_checkpoint_after2.ckpt"


def _model():
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    return build_model(copy.deepcopy(tiny_cfg("pre")), None, Vocabulary.synthetic(20))


def test_parameter_names_and_order_equal_the_reference():
    g = load_golden("train_steps")
    assert [n for n, _ in _model().named_parameters()] == list(g["param_order"])


def test_reference_checkpoint_loads_strictly():
    from joeys2t_amd.helpers import load_checkpoint
    ck = load_checkpoint(CKPT)
    assert set(ck) >= {"model_state", "optimizer_state", "scheduler_state", "stats_state"}
    m = _model()
    missing, unexpected = m.load_state_dict(ck["model_state"], strict=True)
    assert not missing and not unexpected
    assert len(ck["optimizer_state"]["state"]) == len(list(m.parameters()))


def test_average_checkpoints_and_init_layers(tmp_path):
    from joeys2t_amd.helpers import average_checkpoints, init_layers, load_checkpoint
    a = load_checkpoint(CKPT)
    b = copy.deepcopy(a)
    for k, v in b["model_state"].items():
        if v.is_floating_point():
            v.mul_(3.0)
    b["model_state"]["steps_like_int"] = torch.tensor(7)
    a["model_state"]["steps_like_int"] = torch.tensor(2)
    pa, pb = tmp_path / "a.ckpt", tmp_path / "b.ckpt"
    torch.save(a, pa), torch.save(b, pb)
    avg = average_checkpoints([str(pa), str(pb)])
    k = "decoder.output_layer.weight"
    torch.testing.assert_close(avg["model_state"][k], a["model_state"][k] * 2.0)
    assert int(avg["model_state"]["steps_like_int"]) == 4  # integers: floor division
    del a["model_state"]["steps_like_int"]
    torch.save(a, pa)
    m = _model()
    before = copy.deepcopy(m.state_dict())
    init_layers(m, pa, "encoder")
    after = m.state_dict()
    assert torch.equal(after["encoder.layers.0.src_src_att.k_layer.weight"], a["model_state"]["encoder.layers.0.src_src_att.k_layer.weight"])
    assert torch.equal(after["decoder.output_layer.weight"], before["decoder.output_layer.weight"])  # decoder untouched
