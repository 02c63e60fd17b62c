"""GPU: joint CTC / attention decoding (EXTENSION, SURVEY 8 f3 - the reference returns ctc_out for return_type="decode_ctc",
joeynmt/model.py:162-166, and has no consumer).  The checker is the oracle's restatement of Watanabe et al. 2017, Algorithm 2
(oracle/s2t_oracle.py: ctc_prefix_score / joint_ctc_beam_search), itself pinned by brute-force enumeration of alignments
(tests/test_oracle_golden.py::test_ctc_prefix_score_against_enumeration)."""
import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from golden_cfg import FIXTURES, SPECIALS, oracle_cfg
from oracle import s2t_oracle as O
from test_hip_model import batch_kwargs, build

pytestmark = pytest.mark.gpu


def test_beam_pick_kernel(device):
    from joeys2t_amd import ops
    g = torch.Generator().manual_seed(3)
    rows, V, C = 11, 5000, 8
    logits = torch.randn(rows, V, generator=g) * 3
    forbid = [1, 2, 7]
    lp = torch.log_softmax(logits, -1)
    lp[:, forbid] = float("-inf")
    want_s, want_i = lp.topk(C, dim=-1)
    s, i, lse = ops.beam_pick(logits.to(device), C, forbid)
    assert np.array_equal(i.cpu().numpy(), want_i.numpy())
    torch.testing.assert_close(s.cpu(), want_s, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(logits, -1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n_out", [0, 1, 3, 9])
def test_ctc_prefix_step_kernel(device, n_out):
    """js2t_ctc_prefix_step against the oracle's recursion: ragged input lengths, a candidate that repeats the hypothesis' last label,
    EOS and the blank among the candidates, a hypothesis longer than one utterance has frames"""
    from joeys2t_amd import ops
    rs = np.random.RandomState(n_out)
    B, k, T, V, C = 3, 2, 8, 12, 5
    blank, eos = 2, 3
    logp = np.log(rs.dirichlet(np.ones(V), size=(B, T)))
    in_len = np.array([8, 5, 3])
    rows = B * k
    # hypotheses: built by extending the empty prefix n_out times through the oracle (so that r_prev / psi_prev are real states)
    ys = [[blank] for _ in range(rows)]
    r_prev = [O.ctc_prefix_init(logp[r // k], int(in_len[r // k]), blank) for r in range(rows)]
    psi_prev = [0.0] * rows
    for _ in range(n_out):
        for r in range(rows):
            c = int(rs.choice([4, 5, 6]))
            psi, rn = O.ctc_prefix_score(logp[r // k], int(in_len[r // k]), ys[r], [c], r_prev[r], blank, eos)
            ys[r], r_prev[r], psi_prev[r] = ys[r] + [c], rn[:, :, 0], float(psi[0])
    cand = np.stack([np.array([ys[r][-1] if n_out else 4, eos, blank, 5 + (r % 3), 9]) for r in range(rows)])
    cand_lp = np.log(rs.dirichlet(np.ones(C), size=rows))
    cand_lp[0, 4] = -np.inf
    w = 0.3
    to = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(device)
    rp = np.stack(r_prev)
    rp[~np.isfinite(rp)] = ops.CTC_LOG0
    pp = np.array([p if np.isfinite(p) else ops.CTC_LOG0 for p in psi_prev])
    local, psi, r_new = ops.ctc_prefix_step(to(logp, torch.float32), to(in_len, torch.int64), to(rp, torch.float32),
                                            to([y[-1] for y in ys], torch.int64), to(cand, torch.int64), to(cand_lp, torch.float32),
                                            to(pp, torch.float32), n_out, k, blank, eos, w)
    local, psi, r_new = local.cpu().numpy(), psi.cpu().numpy(), r_new.cpu().numpy()
    for r in range(rows):
        want_psi, want_r = O.ctc_prefix_score(logp[r // k], int(in_len[r // k]), ys[r], cand[r].tolist(), r_prev[r], blank, eos)
        for c in range(C):
            if np.isfinite(want_psi[c]):
                assert abs(psi[r, c] - want_psi[c]) <= 1e-4 * max(1.0, abs(want_psi[c])), (r, c, psi[r, c], want_psi[c])
            else:
                assert psi[r, c] < -1e29
            wr = want_r[:, :, c]
            fin = np.isfinite(wr)
            np.testing.assert_allclose(r_new[r, c][fin], wr[fin], rtol=1e-4, atol=1e-4)
            assert (r_new[r, c][~fin] < -1e29).all()
            if np.isfinite(want_psi[c]) and np.isfinite(psi_prev[r]) and np.isfinite(cand_lp[r, c]):
                want = (1 - w) * cand_lp[r, c] + w * (want_psi[c] - psi_prev[r])
                assert abs(local[r, c] - want) <= 1e-4 * max(1.0, abs(want))
            else:
                assert local[r, c] == -np.inf


@pytest.mark.parametrize("name", list(FIXTURES))
@pytest.mark.parametrize("weight", [0.3, 0.7])
def test_joint_ctc_beam_search_matches_oracle(device, name, weight):
    from joeys2t_amd.search import search
    model, g = build(name, device)
    model.eval()
    b = batch_kwargs(g, device)
    k, alpha = int(g["beam_size"]), float(g["beam_alpha"])
    ids, scores, _ = search(model, b, max_output_length=12, beam_size=k, beam_alpha=alpha, n_best=k, return_prob="hyp", ctc_weight=weight,
                            ctc_candidates=8)
    sd = golden_sd(g)
    cfg = oracle_cfg(FIXTURES[name]["cfg"])
    enc, mask, _ = O.encoder_forward(sd, cfg, torch.from_numpy(g["src"]), torch.from_numpy(g["src_length"]))
    want_ids, want_scores = O.joint_ctc_beam_search(sd, cfg, SPECIALS, enc, mask, k, 12, alpha, ctc_weight=weight, n_cand=8, n_best=k)
    assert np.array_equal(ids, want_ids.numpy())  # bit-exact ids
    np.testing.assert_allclose(scores, want_scores.numpy(), rtol=1e-4, atol=1e-4)
    # the CTC term changes the outcome on these models (otherwise the test would not see it) ...
    assert not np.array_equal(ids, g["beam_ids"]) or weight < 0.5
    # ... and weight 0 is the reference's beam search, untouched
    ids0, scores0, _ = search(model, b, max_output_length=12, beam_size=k, beam_alpha=alpha, n_best=k, return_prob="hyp", ctc_weight=0.0)
    assert np.array_equal(ids0, g["beam_ids"])
    np.testing.assert_allclose(scores0, g["beam_scores"], rtol=1e-4, atol=1e-4)
