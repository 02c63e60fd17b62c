"""GPU: the graph drivers COMPOSED - accumulation x varying shapes x data parallelism (the reference's loop:
joeynmt/training.py:416-456 accumulates `batch_multiplier` micro-batches per update, configs/librispeech_100h.yaml:85 = 4,
librispeech_960h.yaml:85 = 8, and its DistributedDataParallel exchanges gradients of per-rank batches of different shapes,
training.py:584-588).  graphed.GraphedTrainStep keeps one capture per (bucket, phase of the accumulation) and, under a process
group, cuts the last micro-batch's capture where the collectives go.
 (i)   batch_multiplier 4, micro-batches of several shapes: replayed == launched eagerly BIT FOR BIT (deterministic mode), and the
       un-padded plain TrainStep's numbers in fp32;
 (ii)  two ranks (gloo, one card) with DIFFERENT B / T / L per rank and per step (ADVICE r4's open ask): both ranks end on identical
       parameters, those of the eager data-parallel run - while the ranks meet new buckets (eager) at different steps;
 (iii) a bucket whose capture fails on ONE rank: that rank goes on eagerly, the other replays, same parameters."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from test_hip_graphed import V, _batches, _cfg, _plain_run, _proc

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _step(sd, device, dtype, k, deterministic=True):
    from joeys2t_amd.training import TrainStep
    from test_hip_config_width import make_model
    model = make_model(_cfg(), V, sd, device, dtype, 0.3, train=True)
    return TrainStep(model, learning_rate=1e-3, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=3, normalization="batch",
                     overlap_ctc=True, batch_multiplier=k, deterministic=deterministic)


def _sd():
    from test_hip_config_width import make_model
    torch.manual_seed(3)
    base = make_model(_cfg(), V, None, None, None, 0.3)
    return {k: v.clone() for k, v in base.state_dict().items()}


def _run(step, batches, device, dtype, use_graphs, inject=None):
    from joeys2t_amd.graphed import GraphedTrainStep
    np.random.seed(11)
    gs = GraphedTrainStep(step, _proc(), compute_dtype=dtype, frame_bucket=128, target_bucket=16, use_graphs=use_graphs)
    gs.inject_failure = inject
    how = [gs.run(w.to(device), ns, trg, tl) for w, ns, trg, tl in batches]
    torch.cuda.synchronize()
    return how, step.store.flat.detach().clone().cpu(), gs


def _mixed_batches():
    """micro-batches of three shape families in an order that meets every (bucket, phase) pair late and early"""
    a, b, c = _batches(5, 8), _batches(6, 6, lo=52000, hi=56000), _batches(7, 6, B=3)
    order = [a[0], b[0], a[1], c[0], a[2], a[3], b[1], b[2], c[1], a[4], b[3], c[2], a[5], a[6], b[4], c[3], c[4], a[7], b[5], c[5]]
    return order


def test_accumulated_updates_replayed_equal_eager_bit_for_bit(device):
    sd = _sd()
    batches = _mixed_batches()  # 20 micro-batches = 5 updates of 4
    how_g, flat_g, gs = _run(_step(sd, device, torch.bfloat16, 4), batches, device, torch.bfloat16, True)
    how_e, flat_e, ge = _run(_step(sd, device, torch.bfloat16, 4), batches, device, torch.bfloat16, False)
    assert how_e == ["eager"] * 20 and how_g.count("replay") >= 6, how_g
    assert gs.step.steps == ge.step.steps == 5 and gs.step.optimizer.t == ge.step.optimizer.t == 5 and gs.step.micro == 20
    assert torch.equal(flat_g, flat_e)  # same kernels on the same data: replayed or launched, bit for bit
    assert not torch.equal(flat_g, gs.step.store.flat.new_tensor(0).cpu()) and torch.isfinite(flat_g).all()
    # captures exist per phase: first / middle / last micro-batch of an update
    phases = {ph for bk in gs.buckets.values() for ph in bk.graphs}
    assert phases == {(True, False), (False, False), (False, True)}, phases


def test_accumulated_updates_equal_plain_unpadded_steps_fp32(device):
    """the padded, bucketed, replayed accumulation against TrainStep.micro_step on the un-padded batches (fp32, the mode of the 1e-4
    parity tests): losses of every micro-batch and the parameters after three updates of two"""
    sd = _sd()
    batches = _batches(5, 6)
    ref, flat_ref = _plain_run(_step(sd, device, torch.float32, 2, deterministic=False), _proc(), batches, device, torch.float32)
    from joeys2t_amd.graphed import GraphedTrainStep
    np.random.seed(11)
    step = _step(sd, device, torch.float32, 2, deterministic=False)
    gs = GraphedTrainStep(step, _proc(), compute_dtype=torch.float32, frame_bucket=128, target_bucket=16)
    got, how = [], []
    for w, ns, trg, tl in batches:
        how.append(gs.run(w.to(device), ns, trg, tl))
        s = gs.read_stats()
        got.append((s["loss"], s["nll"], s["ctc"]))
    torch.cuda.synchronize()
    assert how == ["eager", "eager", "replay", "replay", "replay", "replay"], how  # one bucket, two phases
    for i, (r, g) in enumerate(zip(ref, got)):
        for k in range(3):
            assert abs(r[k] - g[k]) <= 2e-5 * abs(r[k]), (i, k, r, g)
    flat = step.store.flat.detach()
    apart = (flat - flat_ref).abs() > 1e-5  # (coordinates Adam's first steps amplify: see test_hip_graphed.py)
    assert float(apart.float().mean()) < 5e-4
    assert (((flat - flat_ref) * ~apart).norm() / flat_ref.norm()).item() < 1e-5
    assert step.steps == 3


def _rank_batches(rank):
    """per rank AND per step different utterance counts, lengths and target lengths (8 micro-batches = 4 updates of 2)"""
    if rank == 0:
        a, b = _batches(21, 5), _batches(22, 3, lo=52000, hi=56000)
        return [a[0], a[1], b[0], a[2], b[1], a[3], b[2], a[4]]
    a, b = _batches(31, 4, B=3, lo=44000, hi=47000), _batches(32, 4, B=5, lo=30000, hi=33000)
    return [a[0], b[0], b[1], a[1], a[2], b[2], a[3], b[3]]


def _ddp_worker(rank, world, port, inject, ret):
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = _sd()
        batches = _rank_batches(rank)
        how_e, flat_e, _ = _run(_step(sd, dev, torch.bfloat16, 2), batches, dev, torch.bfloat16, False)
        how_g, flat_g, gs = _run(_step(sd, dev, torch.bfloat16, 2), batches, dev, torch.bfloat16, True, inject=inject if rank == 0 else None)
        cut = sum(1 for bk in gs.buckets.values() for c in bk.graphs.values() if isinstance(c, dict))
        ret[rank] = dict(how_e=how_e, how_g=how_g, eager=flat_e, graph=flat_g, steps=gs.step.steps, cut=cut, errors=list(gs.capture_errors),
                         keys=sorted(gs.buckets), n_launched=sum(1 for l in gs.step.reducer.launched if l), n_ranges=len(gs.step.reducer.ranges))
    finally:
        dist.destroy_process_group()


def _run2(inject=None):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, _free_port(), inject, ret), nprocs=2, join=True)
    return ret[0], ret[1]


def test_two_ranks_different_shapes_per_rank_and_step(device):
    r0, r1 = _run2()
    assert set(r0["keys"]).isdisjoint(r1["keys"])  # no shape in common: B, frame bucket or packed rows differ between the ranks
    for r in (r0, r1):
        assert r["how_e"] == ["eager"] * 8 and r["steps"] == 4 and not r["errors"], r["errors"]
        assert r["how_g"].count("replay") >= 2 and r["cut"] >= 1, r["how_g"]  # replays happened, through captures cut at the collectives
        assert r["n_launched"] == r["n_ranges"]  # every range of the flat gradient went out in the last update
        assert torch.equal(r["graph"], r["eager"])  # replayed == launched eagerly, bit for bit (deterministic mode)
    # the ranks met their new buckets at different steps (one replaying while the other ran eagerly) and still hold the same model
    assert r0["how_g"] != r1["how_g"], (r0["how_g"], r1["how_g"])
    assert torch.equal(r0["graph"], r1["graph"])


def test_capture_failure_on_one_rank_only(device):
    r0, r1 = _run2(inject="pieces")
    assert r0["errors"] and r0["cut"] == 0 and "injected" in r0["errors"][0]  # rank 0 never got a cut capture ...
    assert not r1["errors"] and r1["cut"] >= 1                                  # ... rank 1 did and replayed it
    for r in (r0, r1):
        assert r["steps"] == 4 and torch.equal(r["graph"], r["eager"])
    assert torch.equal(r0["graph"], r1["graph"])


def test_precaptured_bucket_first_replay_equals_eager_bit_for_bit(device):
    """a bucket captured AHEAD of its first batch never ran eagerly - any per-shape device constant built lazily inside
    that capture is filled only by that graph's replays.  One eager step at shape A, precapture of shape B (other utterance count,
    frame bucket and packed row count), then B's batches arrive as replays from their first sight: parameters after five updates
    bit for bit those of the eager-only run (deterministic mode)."""
    from joeys2t_amd.graphed import GraphedTrainStep
    sd = _sd()
    a, b = _batches(5, 2), _batches(7, 3, B=3, lo=52000, hi=56000)
    order = [a[0], b[0], b[1], a[1], b[2]]
    how_e, flat_e, _ = _run(_step(sd, device, torch.bfloat16, 1), order, device, torch.bfloat16, False)
    np.random.seed(11)
    step = _step(sd, device, torch.bfloat16, 1)
    gs = GraphedTrainStep(step, _proc(), compute_dtype=torch.bfloat16, frame_bucket=128, target_bucket=16)
    how = [gs.run(order[0][0].to(device), *order[0][1:])]
    key_b = gs.bucket_key(b[0][1], b[0][3])
    assert key_b not in gs.buckets and gs.precapture(key_b) and not gs.precapture(key_b)
    flat_before = step.store.flat.detach().clone()
    how += [gs.run(w.to(device), ns, trg, tl) for w, ns, trg, tl in order[1:]]
    torch.cuda.synchronize()
    assert how == ["eager", "replay", "replay", "replay", "replay"], how  # B replays from its FIRST sight; A's second batch replays A's capture
    assert not torch.equal(flat_before.cpu(), step.store.flat.cpu())
    assert torch.equal(step.store.flat.detach().cpu(), flat_e)
    assert step.steps == 5 and step.optimizer.t == 5


def _rccl_single_worker(rank, port, ret):
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["JS2T_DDP_SINGLE"] = "1"  # a one-rank communicator exchanges nothing unless asked to
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sd = _sd()
        batches = _rank_batches(0)
        how_e, flat_e, _ = _run(_step(sd, dev, torch.bfloat16, 2), batches, dev, torch.bfloat16, False)
        how_g, flat_g, gs = _run(_step(sd, dev, torch.bfloat16, 2), batches, dev, torch.bfloat16, True)
        cut = sum(1 for bk in gs.buckets.values() for c in bk.graphs.values() if isinstance(c, dict))
        ret["res"] = dict(how_g=how_g, eager=flat_e, graph=flat_g, cut=cut, errors=list(gs.capture_errors), steps=gs.step.steps,
                          n_launched=sum(1 for l in gs.step.reducer.launched if l), n_ranges=len(gs.step.reducer.ranges))
    finally:
        dist.destroy_process_group()


def test_composed_driver_on_a_real_rccl_communicator(device):
    """The same driver over RCCL itself (one rank: the 1-GPU box cannot host two; JS2T_DDP_SINGLE makes the communicator issue its
    all-reduces anyway): captures under RCCL's watchdog thread, the side-stream collectives between replays of the cut graphs."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_single_worker, args=(_free_port(), ret), nprocs=1, join=True)
    r = ret["res"]
    assert not r["errors"] and r["cut"] >= 1 and r["steps"] == 4 and r["how_g"].count("replay") >= 2, (r["errors"], r["how_g"])
    assert r["n_launched"] == r["n_ranges"]
    assert torch.equal(r["graph"], r["eager"])
