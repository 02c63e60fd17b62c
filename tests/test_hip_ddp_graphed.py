"""GPU: graphed.GraphedDDPStep - the data-parallel train step replayed from hipGraphs cut where the collectives go (what
`bench.py --gpus N` times; the reference's counterpart is DistributedDataParallel's bucketed reducer, joeynmt/prediction.py:508-515,
training.py:584-588).  (i) one rank on a real RCCL communicator: replayed steps against the same class launched eagerly (bit for
bit in deterministic mode) and against the plain single-GPU step; (ii) two ranks on one card over gloo: every rank ends on the
same parameters, those of the eager data-parallel step; (iii) a capture that fails on ONE rank: every rank falls back to the
eager step, says so, and still trains."""
import copy
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(dev, dtype, rank, comm_dtype=None):
    """model (same on every rank), deterministic TrainStep, a static batch (different per rank), and the step body"""
    from golden_cfg import FIXTURES
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.model import build_model
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    torch.manual_seed(0)
    model = build_model(copy.deepcopy(FIXTURES["model_pre"]["cfg"]), None, Vocabulary.synthetic(20))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.finalize(dev, dtype)
    step = TrainStep(model, n_gpu=1, learning_rate=1e-2, learning_rate_warmup=2, deterministic=True, comm_dtype=comm_dtype)
    g = torch.Generator().manual_seed(100 + rank)
    B, T = 3, 37
    src = torch.randn(B, T, 8, generator=g)
    trg = torch.tensor([[2, 5, 6, 7, 3, 1], [2, 8, 9, 3, 1, 1], [2, 10 + rank, 11, 12, 13, 3]])
    batch = Batch(src=src, src_length=torch.tensor([37, 30, 25]), src_prompt_mask=None, trg=trg, trg_length=torch.tensor([5, 4, 6]),
                  trg_prompt_mask=None, indices=torch.arange(B), device=dev, pad_index=1, eos_index=3, is_train=True, task="S2T", n_gpu=1)
    batch.sort_by_src_length()
    if dtype != torch.float32:
        batch.src = batch.src.to(dtype)

    def body(cut_hook):
        return step.micro_step(batch, sort=False, update=False, overlap=False, flush=False, cut_hook=cut_hook)

    return model, step, batch, body


def _params(step):
    torch.cuda.synchronize()
    return step.store.flat.detach().clone().cpu()


def _single_rank_worker(rank, port, dtype_name, ret):
    import torch.distributed as dist
    from joeys2t_amd.graphed import GraphedDDPStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dtype = getattr(torch, dtype_name)
    # the plain single-GPU step first (no process group): five updates
    _, plain, batch, _ = _make(dev, dtype, 0)
    for _ in range(5):
        plain.micro_step(batch, sort=False)
    p_plain = _params(plain)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["JS2T_DDP_SINGLE"] = "1"  # a one-rank communicator exchanges nothing unless asked to
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        _, s_e, _, body_e = _make(dev, dtype, 0)
        eager = GraphedDDPStep(s_e, body_e)
        for _ in range(5):
            eager.run()
        p_eager = _params(s_e)
        _, s_g, _, body_g = _make(dev, dtype, 0)
        graphed = GraphedDDPStep(s_g, body_g)
        err = graphed.try_capture(warm=2)  # two eager steps (real updates), then the captures (nothing runs)
        for _ in range(3):
            graphed.run()
        p_graph = _params(s_g)
        n_launched = sum(1 for l in s_g.reducer.launched if l)
        ret["res"] = dict(err=err, counts=dict(graphed.counts), captured=graphed.captured, plain=p_plain, eager=p_eager, graph=p_graph,
                          n_pieces=(len(graphed.pieces_dec), len(graphed.pieces)), two_halves=graphed.graphs.get("step2") is not None,
                          n_ranges=len(s_g.reducer.ranges), n_launched=n_launched, t=(s_g.optimizer.t, s_g.steps, s_e.optimizer.t, s_e.steps))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_graphed_ddp_step_single_rank_rccl(device, dtype_name):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_single_rank_worker, args=(_free_port(), dtype_name, ret), nprocs=1, join=True)
    r = ret["res"]
    assert r["err"] is None and r["captured"], r["err"]
    assert r["counts"] == {"eager": 2, "replay": 3}
    assert r["two_halves"]  # cut at the encoder's output
    if dtype_name == "bfloat16":  # (the fp32 parity mode forms its weight gradients inside backward: nothing is queued)
        assert r["n_pieces"][0] >= 1 and r["n_pieces"][1] >= 1, r["n_pieces"]  # ranges complete on both sides of the cut
    assert r["n_launched"] == r["n_ranges"]  # every range of the flat gradient went to RCCL in the last step
    assert r["t"][0] == r["t"][2] == 5 and r["t"][1] == r["t"][3] == 5  # host-side update count and schedule position follow the replays
    # replayed against eagerly launched: the same kernels on the same data in deterministic mode - bit for bit
    assert torch.equal(r["graph"], r["eager"])
    # against the single-GPU step: the gradient norm is summed in another order there (product epilogues), nothing else differs
    diff = (r["graph"] - r["plain"]).abs()
    tol = 1e-5 if dtype_name == "float32" else 2e-3
    assert (diff.norm() / r["plain"].norm()).item() < tol, (diff.norm() / r["plain"].norm()).item()


def _two_rank_worker(rank, world, port, inject, ret):
    import torch.distributed as dist
    from joeys2t_amd.graphed import GraphedDDPStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # reference: the data-parallel step launched eagerly, five updates (both ranks in lock step)
        _, s_e, _, body_e = _make(dev, torch.bfloat16, rank)  # bf16: the deferred weight-gradient products (the pieces) are in play
        eager = GraphedDDPStep(s_e, body_e)
        for _ in range(5):
            eager.run()
        p_eager = _params(s_e)
        _, s_g, _, body_g = _make(dev, torch.bfloat16, rank)
        graphed = GraphedDDPStep(s_g, body_g, inject_failure=inject if rank == 0 else None)  # the failure strikes ONE rank
        err = graphed.try_capture(warm=2)
        for _ in range(3):  # replayed where the capture stands on every rank, eager otherwise
            graphed.run()
        ret[rank] = dict(err=err, captured=graphed.captured, counts=dict(graphed.counts), eager=p_eager, graph=_params(s_g))
    finally:
        dist.destroy_process_group()


def _run2(inject):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_two_rank_worker, args=(2, _free_port(), inject, ret), nprocs=2, join=True)
    return ret[0], ret[1]


def test_graphed_ddp_step_two_ranks_end_on_the_mean(device):
    r0, r1 = _run2(None)
    for r in (r0, r1):
        assert r["err"] is None and r["captured"] and r["counts"] == {"eager": 2, "replay": 3}, (r["err"], r["counts"])
        assert torch.equal(r["graph"], r["eager"])  # replayed == eagerly launched (deterministic mode)
    # every rank applied the same averaged gradient to the same parameters: identical replicas after five updates ...
    assert torch.equal(r0["graph"], r1["graph"])
    # ... and the data really differed: a replica trained on its own batch alone ends somewhere else
    # (checked through the eager reference of tests/test_hip_ddp.py::test_overlapped_exchange_averages_gradients)


@pytest.mark.parametrize("stage", ["forward", "pieces", "update"])
def test_graphed_ddp_step_capture_failure_falls_back_everywhere(device, stage):
    r0, r1 = _run2(stage)
    assert r0["err"] is not None and "injected" in r0["err"]
    assert r1["err"] == "capture failed on another rank"  # rank 1's own capture went through: it drops its graphs all the same
    for r in (r0, r1):
        assert not r["captured"] and r["counts"] == {"eager": 5, "replay": 0}, r["counts"]
        assert torch.equal(r["graph"], r["eager"])  # five eager data-parallel updates: the reference run's parameters
    assert torch.equal(r0["graph"], r1["graph"])


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_two_ranks_end_to_end(device, ranks):
    """`python bench.py --gpus N` as the driver runs it (bench.py starts its ranks itself), N ranks on one card over gloo: every
    rank must make the same collective calls from start to finish - a step run by rank 0 alone (a side measurement, say) leaves
    the other rank's collectives unmatched and the line is lost.  One JSON line, N ranks seen, graphs replayed, a finite loss - the
    headline (one update per batch) and the config's own procedure (`config_faithful`: batch_multiplier 4 through the composed graph
    driver over the same process group).  Four ranks: what a 1-GPU box admits on its card next to the test process itself (six GPU
    processes at most; the driver's node runs eight ranks) - ports,
    rendezvous, time-outs and the N-way capture / fall-back on more than a pair."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, JS2T_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(ranks), "--steps", "3" if ranks == 2 else "2", "--warmup", "2" if ranks == 2 else "1",
                          "--no-cpu-baseline", "--no-decode", "--no-roofline"], cwd=str(root), env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["n_ranks_seen"] == ranks and out["config"]["backend"] == "gloo"
    assert out["config"]["capture_error"] is None and "hipGraph replay in pieces" in out["config"]["launch"]
    assert out["value"] > 0 and out["config"]["loss"] == out["config"]["loss"]  # finite, not NaN
    assert out["config"]["grad_exchange"]["headline"].startswith("fp32")  # the scored line exchanges as the reference does
    cf = out["config_faithful"]
    assert cf and "error" not in cf, cf
    assert cf["batch_multiplier"] == 4 and cf["micro_batches"]["replay"] == 12 and cf["micro_batches"]["eager"] == 0 and not cf["capture_errors"], cf
    assert cf["frames_per_s"] > 0
