"""Distributed helpers with the reference's names (joeynmt/helpers_for_ddp.py): ddp_setup (:17-38), use_ddp (:41),
ddp_cleanup (:46), ddp_synchronize (:52), ddp_merge (:58-154), ddp_reduce (:157-174), DistributedSubsetSampler
(:244-342), RandomSubsetSampler (:345-391) — plus what replaces torch's DistributedDataParallel on MI355X:

`FlatGradReducer`: gradients already live in one flat fp32 buffer (runtime.ParamStore), cut into a few large
contiguous buckets.  Each bucket is all-reduced (average) with RCCL on a side HIP stream as soon as backward has
produced all of its gradients, overlapping the exchange with the rest of backward.  xGMI is point-to-point
(7 links/GPU): few, large messages keep every link busy; there is no per-parameter traffic and no bucket copy.
One process per GPU; backend "nccl" is RCCL on ROCm, "gloo" serves the CPU tests."""
import math
import os
from typing import List, Optional, Union

import torch
import torch.distributed as dist
from torch import Tensor
from torch.utils.data import Dataset, Sampler


def ddp_setup(rank: int, world_size: int, master_addr: str = "127.0.0.1", master_port: int = 12355,
              backend: str = "nccl") -> None:
    """init_process_group + set_device (reference :17-38; the reference hard-codes "nccl" and "localhost")."""
    if dist.is_available():
        os.environ.setdefault("MASTER_ADDR", master_addr)
        os.environ.setdefault("MASTER_PORT", str(master_port))
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
        if backend == "nccl":
            torch.cuda.set_device(rank % max(1, torch.cuda.device_count()))


def use_ddp() -> bool:
    return dist.is_available() and dist.is_initialized()


def ddp_cleanup() -> None:
    if use_ddp():
        dist.destroy_process_group()


def ddp_synchronize() -> None:
    if use_ddp():
        dist.barrier()


def ddp_merge(data: Tensor, pad_index: int = 1) -> Tensor:
    """Gather 2-D/3-D tensors of differing sizes from all ranks: pad to the per-dim maxima with pad_index,
    all_gather, concatenate rank-major and drop each rank's padding-only tail rows (reference :58-154)."""
    if data is None:
        return None
    assert torch.is_tensor(data), data
    if not use_ddp():
        return data
    if data.dim() not in (2, 3):
        raise ValueError
    world = dist.get_world_size()
    local_size = torch.tensor(data.size(), device=data.device)
    all_sizes = [torch.zeros_like(local_size) for _ in range(world)]
    dist.all_gather(all_sizes, local_size)
    sizes = torch.stack(all_sizes).cpu()
    max_dims = sizes.max(dim=0).values.tolist()
    padded = torch.full(max_dims, pad_index, device=data.device, dtype=data.dtype)
    padded[tuple(slice(0, s) for s in data.size())] = data
    gathered = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    return torch.cat([t[: int(sizes[r, 0])] for r, t in enumerate(gathered)], dim=0)


def ddp_reduce(data: Union[Tensor, int], device=None, dtype=None) -> Tensor:
    """SUM all-reduce; Python numbers are lifted to tensors; a 0-d input comes back with shape [1] under DDP
    (reference :157-174)."""
    if data is None:
        return None
    if not torch.is_tensor(data):
        assert device is not None and dtype is not None
        data = torch.tensor(data, device=device, dtype=dtype)
    if use_ddp():
        if data.dim() < 1:
            data = data.unsqueeze(0)
        dist.all_reduce(data, op=dist.ReduceOp.SUM)
    return data


class FlatDDP(torch.nn.Module):
    """What stands where torch's DistributedDataParallel stands in the reference (training.py:131-139): `.module` is the
    Model, forward passes through.  The gradient exchange itself is not hooked into autograd here - TrainStep drives a
    FlatGradReducer over the model's flat gradient buffer - so this class only provides the `.module` indirection that
    model.DataParallelWrapper (and code written against `model.module`) expects."""

    def __init__(self, module: torch.nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class FlatGradReducer:
    """Bucketed, overlapped all-reduce(average) of ParamStore.flat_grad.

    Usage per update:  reducer.begin()  ->  backward passes (hooks fire)  ->  reducer.finish().
    With `sync_every_backward=False` (default) only the LAST micro-batch of an update arms the hooks, i.e. one
    exchange per optimizer step; the reference's DDP exchanges on every micro-batch (training.py:584-588 enters
    no_sync() after the forward, which does not disarm the reducer) — the summed gradient is identical."""

    def __init__(self, store, n_buckets: int = 4, average: bool = True, comm_stream: Optional["torch.cuda.Stream"] = None,
                 ranges=None, comm_dtype: Optional[torch.dtype] = None, comm=None):
        self.store = store
        self.average = average
        # comm (joeys2t_amd.comm.Communicator): the collectives go through the C boundary (js2t_comm_allreduce_async on the
        # communicator's own stream) instead of torch.distributed.all_reduce; same buckets, same order, same staging
        self.comm = comm
        if comm is not None:
            comm_stream = comm.stream
        # comm_dtype=torch.bfloat16 (GPU + RCCL only): a range goes over the links as bf16 - cast into a staging buffer, averaged
        # there, cast back into the fp32 flat gradient, all on the communication stream.  Half the bytes on xGMI (point-to-point
        # links: a ring all-reduce is bound by one of them) for a rounding of 2^-9 relative on gradients that come out of bf16
        # products anyway.  None (default): the fp32 exchange of torch's DistributedDataParallel (prediction.py:508-515).
        self.comm_dtype = comm_dtype
        self._stage = None
        self.world = dist.get_world_size() if use_ddp() else 1
        total = store.total
        # buckets are contiguous flat ranges cut at parameter boundaries, in REVERSE flat order (backward order)
        bounds = sorted({store.offsets[id(p)] for p in store.params} | {total})
        target = math.ceil(total / max(1, n_buckets))
        cuts, last = [total], total
        for b in reversed(bounds[:-1]):
            if last - b >= target:
                cuts.append(b)
                last = b
        if cuts[-1] != 0:
            cuts.append(0)
        self.ranges = [(cuts[i + 1], cuts[i]) for i in range(len(cuts) - 1)]  # (lo, hi), first = tail of flat
        if ranges is not None:
            # explicit buckets: the store's type ranges (prefix, then one per Linear shape) - see exchange_begin()
            self.ranges = [(int(lo), int(hi)) for lo, hi in ranges]
            assert self.ranges[0][0] == 0 and self.ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(self.ranges, self.ranges[1:]))
        self.exchange_mode = False  # True between exchange_begin() and finish(): buckets go out by entries_done(), not by hooks
        self.bucket_of = {}
        self.need = [0] * len(self.ranges)
        for p in store.params:
            if not p.requires_grad:
                continue
            off = store.offsets[id(p)]
            for bi, (lo, hi) in enumerate(self.ranges):
                if lo <= off < hi:
                    self.bucket_of[id(p)] = bi
                    self.need[bi] += 1
                    break
        self.on_gpu = store.device.type == "cuda"
        self.comm_stream = comm_stream or (torch.cuda.Stream(device=store.device) if self.on_gpu else None)
        self.armed = False
        self._partial_open = False
        self.n_early = 0
        self.count = [0] * len(self.ranges)
        self.launched = [False] * len(self.ranges)
        self.works = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._hook) for p in store.params if p.requires_grad]

    def begin(self, armed: bool = True):
        # a one-rank communicator exchanges nothing - unless JS2T_DDP_SINGLE=1 asks for the calls anyway (rehearsal of the
        # RCCL path, its side stream and its event ordering on a 1-GPU box: bench.py JS2T_BENCH_FORCE_DDP)
        self.armed = armed and (self.world > 1 or (use_ddp() and os.environ.get("JS2T_DDP_SINGLE", "0") == "1"))
        self.n_early = 0
        self.count = [0] * len(self.ranges)
        self.launched = [False] * len(self.ranges)
        self.works = []

    def armed_for_exchange(self) -> bool:
        """would begin(armed=True) arm?  (a one-rank group exchanges nothing unless JS2T_DDP_SINGLE=1 asks for the calls)"""
        return self.world > 1 or (use_ddp() and os.environ.get("JS2T_DDP_SINGLE", "0") == "1")

    def params_ready(self, params):
        """Gradients of `params` were accumulated in place by a kernel (no autograd hook fires for them)."""
        for p in params:
            if p is not None and id(p) in self.bucket_of:
                self._hook(p)

    def _hook(self, p):
        if not self.armed or self.exchange_mode:
            return
        bi = self.bucket_of[id(p)]
        self.count[bi] += 1
        if self.count[bi] == self.need[bi]:
            self._launch(bi)

    def _launch(self, bi: int):
        if self.launched[bi]:
            return
        self.launched[bi] = True
        self.n_early += int(self._partial_open)  # left while the encoder's backward had not started yet
        lo, hi = self.ranges[bi]
        buf = self.store.flat_grad[lo:hi]
        if self.comm is not None:
            if self.comm_dtype is not None and self.comm_dtype != torch.float32:
                from joeys2t_amd import ops
                if self._stage is None:
                    self._stage = torch.empty(self.store.total, dtype=self.comm_dtype, device=self.store.device)
                st = self._stage[lo:hi]
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):  # cast, collective, cast back: one after the other on the communicator's stream
                    self.comm_stream.wait_event(ev)
                    ops.cast(buf, self.comm_dtype, out=st)
                    self.comm.all_reduce_async(st, average=self.average, producer=self.comm_stream)
                    ops.cast(st, torch.float32, out=buf)
            else:
                self.comm.all_reduce_async(buf, average=self.average)
            self.works.append((None, buf, None))
            return
        op = dist.ReduceOp.AVG if (self.average and dist.get_backend() == "nccl") else dist.ReduceOp.SUM
        if self.on_gpu and self.comm_dtype is not None and self.comm_dtype != torch.float32 and dist.get_backend() == "nccl":
            from joeys2t_amd import ops
            if self._stage is None:
                self._stage = torch.empty(self.store.total, dtype=self.comm_dtype, device=self.store.device)
            st = self._stage[lo:hi]
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                ops.cast(buf, self.comm_dtype, out=st)
                w = dist.all_reduce(st, op=op, async_op=True)
                w.wait()  # a stream-side wait: the cast back is ordered behind the collective, the host goes on
                ops.cast(st, torch.float32, out=buf)
        elif self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                w = dist.all_reduce(buf, op=op, async_op=True)
        else:
            w = dist.all_reduce(buf, op=op, async_op=True)
        self.works.append((w, buf, op))

    # ---- exchange driven by the deferred weight-gradient products (runtime.WgradQueue) ------------------------------
    def bucket_of_tensor(self, t) -> int:
        off = (t.data_ptr() - self.store.flat_grad.data_ptr()) // 4
        for bi, (lo, hi) in enumerate(self.ranges):
            if lo <= off < hi:
                return bi
        raise ValueError("tensor is not a view of the flat gradient")

    def exchange_begin(self, plan, partial: bool = False):
        """Start the exchange of one optimizer step: `plan` = WgradQueue plan about to run.  Buckets no queued product writes
        into (the non-Linear prefix: LayerNorm, convolution, embedding gradients, complete once backward is over) go out
        at once; the others as soon as the last product writing into them has been launched (entries_done).
        partial=True: the backward pass is only done behind the encoder's output - `plan` holds the decoder side's products, the
        buckets they complete (ParamStore.late_ranges) go out as those products are launched, everything else waits for the
        second call (partial=False) at the end of backward."""
        if not self._partial_open:
            self.begin(armed=True)
            if not self.armed:
                return
            self.exchange_mode = True
            self.pending = [0] * len(self.ranges)
        elif not self.armed:
            return
        for _, items in plan:
            for it in items:
                self.pending[self.bucket_of_tensor(it[2])] += 1
        self._partial_open = partial
        if partial:
            return
        for bi, n in enumerate(self.pending):
            if n == 0:
                self._launch(bi)

    def entries_done(self, items):
        if not self.armed:
            return
        for it in items:
            bi = self.bucket_of_tensor(it[2])
            self.pending[bi] -= 1
            if self.pending[bi] == 0:
                self._launch(bi)

    def reduce_all(self):
        """Exchange the whole flat gradient now (all buckets back to back on the side stream), e.g. after a
        hipGraph replay of forward+backward where no per-parameter hooks fire; blocks the compute stream until done."""
        if self.world <= 1:
            return
        self.begin(armed=True)
        self.finish()

    def finish(self):
        """Launch whatever has not been launched (frozen / unused parameters) and make the compute stream wait."""
        if not self.armed:
            return
        for bi in range(len(self.ranges)):
            self._launch(bi)
        for w, buf, op in self.works:
            if w is None:  # through the C boundary: ordered by the stream join below
                continue
            w.wait()
            if self.average and op == dist.ReduceOp.SUM:
                buf.div_(self.world)
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.armed = False
        self.exchange_mode = False
        self._partial_open = False


def _epoch_order(pool: List[int], shuffle: bool, generator: Optional[torch.Generator]) -> List[int]:
    """The dataset's index list in the order one epoch visits it (one randperm draw from `generator` when shuffling)."""
    if not shuffle:
        return list(pool)
    return [pool[j] for j in torch.randperm(len(pool), generator=generator).tolist()]


def _rank_share(order: List[int], rank: int, world: int, drop_last: bool):
    """-> (kept, mine): `order` cut to a multiple of `world` (an uneven list is an error unless drop_last), and every
    world-th entry of it starting at `rank`."""
    extra = len(order) % world
    if extra and not drop_last:
        raise RuntimeError("`len(dataset)` must be divisible by `world_size`.")
    kept = order[:len(order) - extra]
    return kept, kept[rank::world]


class _SubsetSampler(Sampler):
    """Index sampler over `data_source.indices` (the reference's BaseDataset keeps the active subset there) with one shared
    torch.Generator: `set_seed` re-seeds it and re-draws the dataset's `random_subset`, `reset` restores the full index list."""

    def __init__(self, data_source: Dataset, shuffle: bool, generator: Optional[torch.Generator]):
        self.data_source, self.shuffle, self.generator = data_source, shuffle, generator

    @property
    def num_samples(self) -> int:
        return len(self.data_source.indices)

    def __len__(self) -> int:
        return self.num_samples

    def reset(self) -> None:
        self.data_source.reset_indices()

    def set_seed(self, seed: int) -> None:
        self.generator.manual_seed(seed)
        self._subsample()

    def _subsample(self) -> None:
        """Honour `dataset.random_subset` = k: keep k indices drawn from the generator, in ascending order."""
        size = len(self.data_source)
        k = getattr(self.data_source, "random_subset", -1)
        if k <= 0 or k >= size:
            return
        drawn = torch.randperm(n=size, generator=self.generator)[:k]
        self.data_source.indices = sorted(drawn.tolist())


class DistributedSubsetSampler(_SubsetSampler):
    """One rank's share of an epoch (contract of the reference's helpers_for_ddp.py:244-342, pinned by a 2-rank capture in
    tests/golden/ddp.npz): every rank shuffles with the same seed, the order is cut to a multiple of the world size and
    dealt out round-robin.  Two things the capture shows and this class keeps: the dataset's index list is REPLACED by the
    cut order (so the next epoch permutes that list), and `len()` is the total number of kept indices, not the share."""

    def __init__(self, dataset: Dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True,
                 drop_last: bool = True, generator: torch.Generator = None):
        if (num_replicas is None or rank is None) and not use_ddp():
            raise RuntimeError("Requires distributed package to be available")
        world = dist.get_world_size() if num_replicas is None else num_replicas
        me = dist.get_rank() if rank is None else rank
        if not 0 <= me < world:
            raise ValueError(f"Invalid rank {me}, rank should be in the interval [0, {world - 1}]")
        super().__init__(dataset, shuffle, generator)
        self.num_replicas, self.rank, self.drop_last = world, me, drop_last

    def __iter__(self):
        order = _epoch_order(self.data_source.indices, self.shuffle, self.generator)
        kept, mine = _rank_share(order, self.rank, self.num_replicas, self.drop_last)
        self.data_source.indices = kept
        return iter(mine)


class RandomSubsetSampler(_SubsetSampler):
    """Single-process counterpart (contract of the reference's helpers_for_ddp.py:345-391): the whole index list, shuffled
    per epoch when asked to."""

    def __iter__(self):
        return iter(_epoch_order(self.data_source.indices, self.shuffle, self.generator))
