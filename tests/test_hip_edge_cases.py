"""GPU: edge cases around the hot path - a single utterance, a one-step decode, zero-sized launches, an utterance that
sub-samples to one encoder position."""
import copy

import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from golden_cfg import FIXTURES, SPECIALS, oracle_cfg

pytestmark = pytest.mark.gpu


def _model(device):
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    g = load_golden("model_pre")
    model = build_model(copy.deepcopy(FIXTURES["model_pre"]["cfg"]), None, Vocabulary.synthetic(20))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.load_state_dict(golden_sd(g))
    model.finalize(device, torch.float32).eval()
    return model, golden_sd(g), oracle_cfg(FIXTURES["model_pre"]["cfg"])


@pytest.mark.parametrize("T", [37, 5, 3])
def test_single_utterance_loss_and_search_match_oracle(device, T):
    """B = 1; T = 5 / 3 sub-sample (k = 5, stride 2, twice) to 2 / 1 encoder positions."""
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.search import search
    from oracle import s2t_oracle as O
    model, sd, cfg = _model(device)
    gen = torch.Generator().manual_seed(T)
    src = torch.randn(1, T, 8, generator=gen)
    lengths = torch.tensor([T])
    trg = torch.tensor([[2, 7, 3]]) if T < 10 else torch.tensor([[2, 7, 9, 11, 3]])
    tl = torch.tensor([trg.shape[1]])
    b = Batch(src=src, src_length=lengths, src_prompt_mask=None, trg=trg, trg_length=tl, trg_prompt_mask=None, indices=torch.arange(1),
              device=device, pad_index=1, eos_index=3, is_train=True, task="S2T", n_gpu=1)
    with torch.no_grad():
        total, xent, ctc, ncor = model(return_type="loss", **vars(b))
    ob = O.make_batch(src, lengths, trg, tl, 1, 3)
    o_total, o_xent, o_ctc, o_ncor, _, _ = O.model_loss(sd, cfg, ob, SPECIALS, 0.1, 0.3)
    if torch.isfinite(o_total):
        assert total.item() == pytest.approx(o_total.item(), rel=1e-4)
    assert xent.item() == pytest.approx(o_xent.item(), rel=1e-4) and int(ncor) == int(o_ncor)
    assert (ctc.item() == pytest.approx(o_ctc.item(), rel=1e-4)) or (not np.isfinite(o_ctc.item()) and ctc.item() == 0.0) or o_ctc.item() == 0.0
    enc, mask, _ = O.encoder_forward(sd, cfg, src, lengths)
    for beam, L in ((1, 1), (3, 6)):
        ids, _, _ = search(model, b, max_output_length=L, beam_size=beam, beam_alpha=1.0 if beam > 1 else -1, n_best=1)
        if beam == 1:
            ref, _ = O.greedy(sd, cfg, SPECIALS, enc, mask, L)
        else:
            ref, _ = O.beam_search(sd, cfg, SPECIALS, enc, mask, beam, L, 1.0, 1)
        assert np.array_equal(ids, ref.numpy())


def test_zero_sized_launches_are_no_ops(device):
    from joeys2t_amd import ops
    z = torch.empty(0, 16, device=device)
    w = torch.randn(8, 16, device=device)
    out = torch.empty(0, 8, device=device)
    ops.gemm(z, w, out, M=0, N=8, K=16, lda=16, ldb=16, ldc=8)
    y, mean, rstd = ops.layernorm_fwd(z, torch.ones(16, device=device), torch.zeros(16, device=device), 1e-6)
    assert y.shape == (0, 16) and mean.shape == (0, )
    assert ops.glu_fwd(torch.empty(0, 32, device=device)).shape == (0, 16)
    assert ops.colsum(z).tolist() == [0.0] * 16
    ops.gemm_grouped([], [], [], M=4, N=4, K=8, lda=8, ldb=8, ldc=4)
