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


def test_ctc_prefix_step_lds_kernel_against_thread_per_pair_at_full_size(device):
    """Round 6: a block per hypothesis with the operands staged through LDS and logaddexp on the hardware's exp / log against the
    thread-per-pair kernel of round 5 (library log1pf / expf) - T' = 375 frames, V = 5000, 32 utterances x beam 5, 8 candidates (the
    decode the bench times), ragged input lengths, three extension steps chained through the winners' variables; a repeated label,
    EOS and the blank among the candidates: the same numbers within f32 rounding of sums of magnitude 10^3 (spacing 1e-4)."""
    from joeys2t_amd import ops
    from joeys2t_amd._lib import lib
    g = torch.Generator().manual_seed(12)
    B, k, T, V, C = 32, 5, 375, 5000, 8
    blank, eos = 2, 3
    logp = torch.log_softmax(torch.randn(B, T, V, generator=g) * 2.0, -1).to(device)
    in_len = torch.cat([torch.tensor([T, 1, 2]), torch.randint(150, T + 1, (B - 3, ), generator=g)]).to(device)
    rows = B * k
    r_prev = ops.ctc_prefix_init(logp, in_len, k, blank)
    psi_prev = torch.zeros(rows, device=device)
    last = torch.full((rows, ), blank, dtype=torch.int64, device=device)
    for n_out in range(3):
        cand = torch.randint(4, V, (rows, C), generator=g)
        cand[:, 0] = last.cpu() if n_out else cand[:, 0]  # a repeat of the hypothesis' last label
        cand[:, 1], cand[:, 2] = eos, blank
        cand = cand.to(device)
        cand_lp = torch.log_softmax(torch.randn(rows, C, generator=g), -1).to(device)
        outs = []
        for mode in (1, 0):
            lib().js2t_debug_ctc_prefix_thread_per_pair(mode)
            try:
                outs.append(ops.ctc_prefix_step(logp, in_len, r_prev, last, cand, cand_lp, psi_prev, n_out, k, blank, eos, 0.3))
            finally:
                lib().js2t_debug_ctc_prefix_thread_per_pair(0)
        torch.cuda.synchronize()
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(torch.isinf(a), torch.isinf(b)) and torch.equal(a < -1e29, b < -1e29)  # log 0 / -inf in the same places
            live = ~(torch.isinf(a) | (a < -1e29))
            torch.testing.assert_close(a[live], b[live], rtol=2e-6, atol=2e-3)
        assert torch.isfinite(outs[1][1][:, 3:]).all() and not torch.isnan(outs[1][0]).any()
        pick = 3 + n_out  # the next step extends every hypothesis by one of its ordinary candidates
        r_prev = outs[1][2][:, pick].contiguous()
        psi_prev = outs[1][1][:, pick].contiguous()
        last = cand[:, pick].contiguous()


def test_joint_ctc_decoding_properties_at_full_size(device):
    """search(ctc_weight=...) at configs/mustc_st.yaml size (12 + 6 layers, 8 heads of 64, 32 ragged utterances of up to 1498 frames,
    fp32) - beyond what the CPU oracle finishes in seconds, so through properties: hypotheses and scores do not depend on the order of
    the utterances in the batch; the thread-per-pair kernel gives the same search; weight 0 IS the reference's beam search; and the
    CTC term does change the outcome (the property test sees it)."""
    from joeys2t_amd._lib import lib
    from joeys2t_amd.search import search
    from test_hip_full_size import B, _decode_batch, _mustc_decode_case
    model, src, lengths = _mustc_decode_case(device, torch.float32)
    order = list(range(B))
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(8)).tolist()
    kw = dict(max_output_length=12, beam_size=5, beam_alpha=1.0, n_best=1, return_prob="hyp")
    ids0, sc0, _ = search(model, _decode_batch(src, lengths, order, device), ctc_weight=0.3, ctc_candidates=8, **kw)
    idsp, scp, _ = search(model, _decode_batch(src, lengths, perm, device), ctc_weight=0.3, ctc_candidates=8, **kw)
    assert np.array_equal(ids0[perm], idsp)
    np.testing.assert_allclose(np.asarray(sc0)[perm], np.asarray(scp), rtol=1e-4, atol=1e-4)
    lib().js2t_debug_ctc_prefix_thread_per_pair(1)
    try:
        ids1, sc1, _ = search(model, _decode_batch(src, lengths, order, device), ctc_weight=0.3, ctc_candidates=8, **kw)
    finally:
        lib().js2t_debug_ctc_prefix_thread_per_pair(0)
    assert np.array_equal(ids0, ids1) and np.array_equal(np.asarray(sc0), np.asarray(sc1))
    ida, sca, _ = search(model, _decode_batch(src, lengths, order, device), **kw)
    idw, scw, _ = search(model, _decode_batch(src, lengths, order, device), ctc_weight=0.0, **kw)
    assert np.array_equal(ida, idw) and np.array_equal(np.asarray(sca), np.asarray(scw))
    assert not np.array_equal(ids0, ida)
