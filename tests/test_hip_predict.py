"""GPU: `predict`'s validation-loss / reference-scoring leg (joeynmt/prediction.py:165-200) on the HIP model against the capture of
the reference's statements (tests/golden/predict_loss.npz, oracle/make_golden.py:golden_predict_loss), single process and two
ranks on one card over gloo (ddp_reduce of loss / n_correct / ntokens, ddp_merge of log-probabilities, targets and indices)."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import golden_sd, load_golden
from golden_cfg import FIXTURES

pytestmark = pytest.mark.gpu


def _model(dev):
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    model = build_model(copy.deepcopy(FIXTURES["model_pre"]["cfg"]), None, Vocabulary.synthetic(20))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.load_state_dict(golden_sd(load_golden("model_pre")))
    model.finalize(dev, torch.float32).eval()
    return model


def _batch(g, bi, dev, first_index=0):
    from joeys2t_amd.batch import Batch
    pre = f"b{bi}."
    B = g[pre + "src"].shape[0]
    return Batch(src=torch.from_numpy(g[pre + "src"]), src_length=torch.from_numpy(g[pre + "src_length"]), src_prompt_mask=None,
                 trg=torch.from_numpy(g[pre + "trg_full"]), trg_length=torch.from_numpy(g[pre + "trg_length_full"]), trg_prompt_mask=None,
                 indices=torch.arange(first_index, first_index + B), device=dev, pad_index=1, eos_index=3, is_train=False, task="S2T", n_gpu=1)


def _expected_rows(g, bi):
    """reference scores and target rows of batch bi in the ORIGINAL order (the capture holds them in the sorted order)"""
    pre = f"b{bi}."
    rev = g[pre + "reverse_index"]
    return [g[pre + f"ref_scores.{int(p)}"] for p in rev], [g[pre + "trg_sorted"][int(p)] for p in rev]


def test_predict_validation_loss_and_reference_scores(device):
    from joeys2t_amd.prediction import predict
    g = load_golden("predict_loss")
    model = _model(device)
    ids, sentences, scores, rec = predict(model, [_batch(g, 0, device), _batch(g, 1, device, 3)], return_prob="ref", compute_loss=True)
    tot = rec["totals"]
    assert abs(tot["loss"] - float(g["total.loss"])) <= 1e-4 * float(g["total.loss"])
    assert tot["n_correct"] == int(g["total.n_correct"]) and tot["ntokens"] == int(g["total.ntokens"]) and tot["nseqs"] == int(g["total.nseqs"])
    assert rec["normalizer"] == tot["nseqs"]
    assert abs(rec["valid_scores"]["loss"] - float(g["total.loss"]) / 7) <= 1e-4 * float(g["total.loss"]) / 7
    assert abs(rec["valid_scores"]["ppl"] - np.exp(float(g["total.loss"]) / tot["ntokens"])) <= 1e-3 * rec["valid_scores"]["ppl"]
    exp_scores, exp_trg = _expected_rows(g, 0)
    e1, t1 = _expected_rows(g, 1)
    exp_scores, exp_trg = exp_scores + e1, exp_trg + t1
    assert len(scores) == len(exp_scores) == 7 and len(sentences) == 7
    for got, want, gid, wid in zip(scores, exp_scores, ids, exp_trg):
        np.testing.assert_allclose(np.asarray(got, dtype=np.float64), want, rtol=1e-4, atol=1e-4)
        assert np.array_equal(np.asarray(gid)[:len(wid)], wid)  # no search: the outputs are the reference tokens
    # the loss leg in front of a search: same totals, hypotheses instead of references
    ids2, _, _, rec2 = predict(model, [_batch(g, 0, device), _batch(g, 1, device, 3)], beam_size=1, max_output_length=12, compute_loss=True)
    assert abs(rec2["totals"]["loss"] - tot["loss"]) <= 1e-5 * tot["loss"] and rec2["totals"]["n_correct"] == tot["n_correct"]
    assert np.array_equal(np.asarray(ids2[:3]), load_golden("model_pre")["greedy_ids"])  # batch 0 is the golden model's batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    from joeys2t_amd.prediction import predict
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("predict_loss")
        model = _model(dev)
        # rank 0 holds batch 0 (dataset rows 0-2), rank 1 batch 1 (rows 3-6): one step of a sharded validation set
        ids, sentences, scores, rec = predict(model, [_batch(g, rank, dev, 0 if rank == 0 else 3)], return_prob="ref", compute_loss=True)
        ret[rank] = dict(ids=[np.asarray(i) for i in ids], scores=[np.asarray(s, dtype=np.float64) for s in scores], rec=rec)
    finally:
        dist.destroy_process_group()


def test_predict_validation_leg_two_ranks(device):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    g = load_golden("predict_loss")
    e0, t0 = _expected_rows(g, 0)
    e1, t1 = _expected_rows(g, 1)
    for r in (0, 1):
        tot = ret[r]["rec"]["totals"]
        assert abs(tot["loss"] - float(g["total.loss"])) <= 1e-4 * float(g["total.loss"])  # summed over ranks on every rank
        assert tot["n_correct"] == int(g["total.n_correct"]) and tot["ntokens"] == int(g["total.ntokens"]) and tot["nseqs"] == 7
        scores, ids = ret[r]["scores"], ret[r]["ids"]
        assert len(scores) == 7  # all ranks' sentences, in dataset order (batch.indices)
        for got, want, gid, wid in zip(scores, e0 + e1, ids, t0 + t1):
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
            n = min(len(gid), len(wid))
            assert np.array_equal(gid[:n], wid[:n]) and (np.asarray(wid[n:]) == 1).all() and (gid[n:] == 1).all()


SEARCH_MODES = [dict(beam_size=1, return_prob="none"), dict(beam_size=1, return_prob="hyp"), dict(beam_size=1, return_prob="none", return_attention=True),
                dict(beam_size=3, n_best=2, return_prob="hyp", beam_alpha=1.0), dict(beam_size=3, return_prob="none")]


def _search_worker(rank, world, port, ret):
    import torch.distributed as dist
    from joeys2t_amd.prediction import predict
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("predict_loss")
        model = _model(dev)
        out = []
        for mode in SEARCH_MODES:
            # step 1: rank 0 holds rows 0-2, rank 1 rows 3-6; step 2 hands every row out a second time, to the other rank - what a
            # sampler's padding to a multiple of the world size does to a few rows: each sentence must come back once
            mine = [_batch(g, rank, dev, 0 if rank == 0 else 3), _batch(g, 1 - rank, dev, 3 if rank == 0 else 0)]
            res = predict(model, mine, max_output_length=12, **mode)
            out.append([[np.asarray(i) for i in res[0]], res[1], None if res[2] is None else [np.asarray(s) for s in res[2]],
                        None if len(res) < 4 or res[3] is None else [np.asarray(a) for a in res[3]]])
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_predict_search_two_ranks_greedy_and_beam(device):
    """greedy search merges its rows itself (search.py:333-335), so `predict` must not merge them again; hypothesis
    scores and attention travel with the ids; a sentence handed out twice is kept once.  Every mode against the single-process
    `predict` over the same sentences."""
    from joeys2t_amd.prediction import predict
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_search_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    g = load_golden("predict_loss")
    model = _model(device)
    for mi, mode in enumerate(SEARCH_MODES):
        want = predict(model, [_batch(g, 0, device), _batch(g, 1, device, 3)], max_output_length=12, **mode)
        n_rows = 7 * mode.get("n_best", 1)
        assert len(want[0]) == n_rows
        for r in (0, 1):
            ids, sentences, scores, att = ret[r][mi]
            assert len(ids) == n_rows and sentences == want[1], (mode, r)
            for a, b in zip(ids, want[0]):  # rows are padded to their batch's (here: the merged batch's) longest hypothesis
                a, b = np.asarray(a), np.asarray(b)
                n = min(len(a), len(b))
                assert np.array_equal(a[:n], b[:n]) and (a[n:] == 1).all() and (b[n:] == 1).all(), (mode, r)
            if mode["return_prob"] == "hyp":
                assert scores is not None and len(scores) == n_rows
                for a, b in zip(scores, want[2]):
                    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
                    n = min(len(a), len(b))
                    np.testing.assert_allclose(a[:n], b[:n], rtol=1e-4, atol=1e-4)
                    assert (a[n:] == 0).all() and (b[n:] == 0).all()
            else:
                assert scores is None
            if mode.get("return_attention"):
                assert att is not None and len(att) == 7
                for a, b in zip(att, want[3]):
                    t, s = min(a.shape[0], b.shape[0]), min(a.shape[1], b.shape[1])
                    np.testing.assert_allclose(a[:t, :s], b[:t, :s], rtol=1e-4, atol=1e-5)
