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


class SentenceBatchSampler(BatchSampler):
    """Mini-batches of `batch_size` instances (reference datasets.py:1164-1246)."""

    def __init__(self, sampler: Sampler, batch_size: int, drop_last: bool, seed: int):
        super().__init__(sampler, batch_size, drop_last)
        self.seed = seed

    @property
    def num_samples(self) -> int:
        assert self.sampler.data_source.indices is not None
        try:
            return len(self.sampler)
        except NotImplementedError:
            return len(self.sampler.data_source.indices)

    def __iter__(self) -> Iterator[List[int]]:
        batch = []
        d = self.sampler.data_source
        for idx in self.sampler:
            _, src, _ = d[idx]
            if src is not None:  # otherwise drop the instance
                batch.append(idx)
                if len(batch) >= self.batch_size:
                    yield batch
                    batch = []
        if len(batch) > 0:
            if not self.drop_last:
                yield batch
            else:
                logger.warning("Drop indices %s.", batch)

    def __len__(self) -> int:
        if self.drop_last:
            return self.num_samples // self.batch_size
        return (self.num_samples + self.batch_size - 1) // self.batch_size

    def set_seed(self, seed: int) -> None:
        assert seed is not None, seed
        self.sampler.data_source.seed = seed
        if hasattr(self.sampler, "set_seed"):
            self.sampler.set_seed(seed)  # set seed and resample
        elif hasattr(self.sampler, "generator"):
            self.sampler.generator.manual_seed(seed)

    def reset(self) -> None:
        if hasattr(self.sampler, "reset"):
            self.sampler.reset()

    def get_state(self):
        return self.sampler.generator.get_state() if hasattr(self.sampler, "generator") else None

    def set_state(self, state) -> None:
        if hasattr(self.sampler, "generator"):
            self.sampler.generator.set_state(state)


class TokenBatchSampler(SentenceBatchSampler):
    """Mini-batches by token count incl. padding (reference datasets.py:1249-1295): the batch closes when
    max_tokens_so_far * len(batch) >= batch_size; `len()` is undefined, as in the reference."""

    def __iter__(self) -> Iterator[List[int]]:
        batch, max_tokens = [], 0
        d = self.sampler.data_source
        for idx in self.sampler:
            _, src, trg = d[idx]
            if src is not None:
                src_len = len(src)
                trg_len = 0 if trg is None else len(trg)
                n_tokens = 0 if src_len == 0 else max(src_len + 1, trg_len + 1)
                batch.append(idx)
                if n_tokens > max_tokens:
                    max_tokens = n_tokens
                if max_tokens * len(batch) >= self.batch_size:
                    yield batch
                    batch, max_tokens = [], 0
        if len(batch) > 0:
            if not self.drop_last:
                yield batch
            else:
                logger.warning("Drop indices %s.", batch)

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
