"""Batch samplers and the host->HBM prefetcher around the hot path.

`SentenceBatchSampler` / `TokenBatchSampler` have the reference's names, arguments and batching rule
(joeynmt/datasets.py:1164-1295): they wrap an index sampler (helpers_for_ddp.RandomSubsetSampler /
DistributedSubsetSampler, or any iterable with `.data_source`) and yield lists of dataset indices; an instance whose
`dataset[idx]` comes back with `src is None` (length filter, tokenizers.py:461-478) is dropped.  Every S2T config trains
with `batch_type: token` (librispeech_100h.yaml:55-57): a batch closes as soon as max(len) * n_instances reaches
batch_size, where len = max(src_len + 1, trg_len + 1) - no bucketing, exactly as in the reference.

`PrefetchLoader` is what replaces the DataLoader worker hand-over for the GPU front-end: batches of raw waveforms are
staged in pinned host memory and copied to HBM on a side HIP stream ONE step ahead, so the PCIe transfer of step i + 1
(~0.8 ms for 32 x 15 s of float32 samples) hides under the compute of step i; the consumer only waits on an event."""
import logging
from typing import Callable, Iterable, Iterator, List, Optional

import torch
from torch.utils.data import BatchSampler, Sampler

logger = logging.getLogger(__name__)


class _InstanceBatcher(BatchSampler):
    """What the two batch samplers share: walk the index sampler, look every instance up in its dataset, skip the ones the
    length filter emptied, and cut a batch whenever `_closed(n_instances, widest)` says so.  `widest` is the largest
    `_weigh(src, trg)` seen in the open batch; what is left over at the end of an epoch is handed out unless `drop_last`."""

    def __init__(self, sampler: Sampler, batch_size: int, drop_last: bool, seed: int):
        super().__init__(sampler, batch_size, drop_last)
        self.seed = seed

    # -- the batching rule (overridden per sampler) ------------------------------------------------------------------------
    def _weigh(self, src, trg) -> int:
        return 1

    def _closed(self, n_instances: int, widest: int) -> bool:
        raise NotImplementedError

    def _usable_instances(self):
        dataset = self.sampler.data_source
        for index in self.sampler:
            _, src, trg = dataset[index]
            if src is not None:
                yield index, self._weigh(src, trg)

    def __iter__(self) -> Iterator[List[int]]:
        open_batch: List[int] = []
        widest = 0
        for index, weight in self._usable_instances():
            open_batch.append(index)
            widest = max(widest, weight)
            if self._closed(len(open_batch), widest):
                yield open_batch
                open_batch, widest = [], 0
        if open_batch and self.drop_last:
            logger.warning("Drop indices %s.", open_batch)
        elif open_batch:
            yield open_batch

    # -- sampler plumbing ----------------------------------------------------------------------------------------------------
    @property
    def num_samples(self) -> int:
        """Instances one epoch draws: the index sampler's own length where it defines one, else the dataset's index list."""
        pool = self.sampler.data_source.indices
        if pool is None:
            raise AssertionError("the dataset has no active index list")
        try:
            n = len(self.sampler)
        except NotImplementedError:
            n = len(pool)
        return n

    def set_seed(self, seed: int) -> None:
        """Re-seed dataset and index sampler (a sampler with its own `set_seed` also re-draws its random subset)."""
        assert seed is not None, seed
        inner = self.sampler
        inner.data_source.seed = seed
        reseed = getattr(inner, "set_seed", None)
        if reseed is not None:
            reseed(seed)
        elif getattr(inner, "generator", None) is not None:
            inner.generator.manual_seed(seed)

    def reset(self) -> None:
        undo = getattr(self.sampler, "reset", None)
        if undo is not None:
            undo()

    def _generator(self):
        return getattr(self.sampler, "generator", None)

    def get_state(self):
        gen = self._generator()
        return None if gen is None else gen.get_state()

    def set_state(self, state) -> None:
        gen = self._generator()
        if gen is not None:
            gen.set_state(state)


class SentenceBatchSampler(_InstanceBatcher):
    """Mini-batches of `batch_size` instances (contract of the reference's datasets.py:1164-1246)."""

    def _closed(self, n_instances: int, widest: int) -> bool:
        return n_instances >= self.batch_size

    def __len__(self) -> int:
        whole, rest = divmod(self.num_samples, self.batch_size)
        return whole if (self.drop_last or rest == 0) else whole + 1


class TokenBatchSampler(_InstanceBatcher):
    """Mini-batches by padded token count (contract of the reference's datasets.py:1249-1295): an instance weighs
    max(len(src), len(trg)) + 1 (0 for an empty source), and the batch closes once widest * n_instances reaches batch_size -
    no bucketing.  The number of batches is not known ahead of an epoch: `len()` raises, as in the reference."""

    def _weigh(self, src, trg) -> int:
        if len(src) == 0:
            return 0
        return 1 + max(len(src), 0 if trg is None else len(trg))

    def _closed(self, n_instances: int, widest: int) -> bool:
        return widest * n_instances >= self.batch_size

    def __len__(self):
        raise NotImplementedError


class PrefetchLoader:
    """Iterate over `batches` (an iterable of index lists, e.g. a TokenBatchSampler) and hand out device-resident inputs,
    copying one batch ahead on a side stream.

    `load(indices) -> dict of CPU tensors / arbitrary values`: every CPU tensor is staged in a pinned buffer and copied
    with non_blocking=True on the copy stream; other values pass through.  `__next__` makes the CURRENT stream wait for
    the copy's event (no host sync) and immediately enqueues the copy of the following batch."""

    def __init__(self, batches: Iterable[List[int]], load: Callable[[List[int]], dict], device, depth: int = 1):
        self.batches, self.load, self.device = batches, load, torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("PrefetchLoader copies into HBM: it needs a cuda (ROCm) device")
        self.stream = torch.cuda.Stream(device=self.device)
        self.depth = max(1, int(depth))

    def _stage(self, item: dict):
        out = {}
        with torch.cuda.stream(self.stream):
            for k, v in item.items():
                if torch.is_tensor(v) and v.device.type == "cpu":
                    pinned = v if v.is_pinned() else v.pin_memory()
                    out[k] = pinned.to(self.device, non_blocking=True)
                    out.setdefault("_pinned", []).append(pinned)  # keep the staging buffer alive until the copy ran
                elif isinstance(v, (list, tuple)) and len(v) > 0 and all(torch.is_tensor(t) and t.device.type == "cpu" and t.dim() == 1 for t in v):
                    # rows of differing length (e.g. utterances that already sit in pinned memory): one device tensor
                    # [len(v), longest], zero behind each row, filled by one asynchronous copy per row - no host-side gather
                    width = max(int(t.numel()) for t in v)
                    dst = torch.zeros((len(v), width), dtype=v[0].dtype, device=self.device)
                    for r, t in enumerate(v):
                        src = t if t.is_pinned() else t.pin_memory()
                        dst[r, :t.numel()].copy_(src, non_blocking=True)
                        out.setdefault("_pinned", []).append(src)
                    out[k] = dst
                else:
                    out[k] = v
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.batches)
        queue = []

        def fill():
            while len(queue) < self.depth:
                try:
                    idx = next(it)
                except StopIteration:
                    return
                queue.append(self._stage(self.load(idx)))

        fill()
        while queue:
            item, ev = queue.pop(0)
            fill()  # the next batch's copy is in flight while the consumer works on this one
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in item.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(torch.cuda.current_stream(self.device))
            item.pop("_pinned", None)
            yield item
