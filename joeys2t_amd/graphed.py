"""Train steps over VARYING batches replayed from hipGraphs: one graph per shape bucket.

The reference's loader cuts a new (B, T, L) every step (TokenBatchSampler, joeynmt/datasets.py:1249-1295; the step itself is
training.py:541-596).  Launching the ~530 kernels of a step from Python costs ~19 ms against ~12 ms of GPU work, and a
captured hipGraph only replays ONE shape.  So shapes are bucketed - frames up to the next multiple of `frame_bucket`, target
length up to the next multiple of `target_bucket`, the packed encoder rows (sum of the sub-sampled lengths) up to the next multiple
of `row_bucket`, the utterance count as it comes - and everything that differs between two
batches of a bucket lives in device memory the graph reads:

  * utterance offsets / frame counts of the fbank front-end (the kernels take them from device tables already),
  * sub-sampled lengths and masks (computed on the device from the length vector), the row offsets of the packed encoder,
  * targets, target lengths, SpecAugment parameters, the learning rate of the step,
  * the crop lengths that make a batch padded to the bucket look to the sub-sampler's convolutions like the reference's
    batch, which ends at its longest utterance (js2t_feature_finalize_crop, js2t_glu_*_crop).

All of it travels in ONE pinned buffer -> ONE host-to-device copy per step.  A bucket seen for the first time runs its batch
eagerly (that is the training step) and is captured right after; graphs share one memory pool and are kept in an LRU list.
Results are those of TrainStep.micro_step on the un-padded batch (tests/test_hip_graphed.py; dropout masks differ because the
row index of a position depends on the padded length - as they differ from the reference's torch RNG anyway).

Round 6 - the drivers compose: accumulation x varying shapes x data parallelism (the reference's configs accumulate,
configs/librispeech_100h.yaml:85 `batch_multiplier: 4`, librispeech_960h.yaml:85 = 8; loop at joeynmt/training.py:416-456, exchange
at :584-588).  A bucket holds one capture per PHASE of the accumulation - (first, last) of the micro-batch inside its update: the
first one overwrites weight-gradient slices instead of adding into them, the last one is followed by the update - so an update of k
micro-batches of k different shapes is k replays.  Under a process group the last micro-batch's capture is CUT where the collectives
go (`capture_cut_step`, the scheme of GraphedDDPStep): forward + decoder-side backward, the decoder side's weight-gradient pieces,
the encoder's backward, its pieces; the update graph is shape-independent and shared.  Collectives are never captured, and the eager
step of a bucket's first sight issues EXACTLY the collective calls of a replay (same ranges, same order: the plan's order is
rank- and shape-independent, runtime.WgradQueue.take) - so ranks may replay different buckets, or one rank may run eagerly while the
others replay, without any agreement between them."""
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from joeys2t_amd import ops
from joeys2t_amd.batch import Batch
from joeys2t_amd.helpers_for_audio import get_extractor
from joeys2t_amd.tokenizers import SpeechProcessor
from joeys2t_amd.training import TrainStep


def _round_up(x: int, m: int) -> int:
    return -(-x // m) * m


class _Bucket:
    """Static device inputs of one (B, T bucket, L bucket) and the graph captured over them."""

    def __init__(self, key, n_conv: int, win: int, shift: int, device, pad_index: int, t_sub: int = 0):
        B, Tb, Lb, rows = key
        self.key = key
        self.n_samples_cap = win + (Tb - 1) * shift
        Lc = Lb - 1  # columns of trg_input / trg (batch.py:82-86: BOS dropped / last column dropped)
        # one packed buffer of int64 words; every static tensor is a view of its device copy
        fields = [("soff", B), ("foff", B + 1), ("src_length", B), ("crop", 1 + n_conv), ("trg_length", B), ("trg_input", B * Lc),
                  ("trg", B * Lc), ("masks", B * 4), ("lr", 1), ("order", B), ("seg", (B + 2) // 2)]
        self.slices: Dict[str, slice] = {}
        off = 0
        for name, n in fields:
            self.slices[name] = slice(off, off + n)
            off += n
        # the host side is a ring: the copy of step i may still be queued when the host fills in step i + 1
        self.hosts = [torch.zeros((off, ), dtype=torch.int64).pin_memory() for _ in range(3)]
        self.copied = [None, None, None]
        self.turn = 0
        self.host = self.hosts[0]
        self.dev = torch.zeros((off, ), dtype=torch.int64, device=device)
        self.wave = torch.zeros((B, self.n_samples_cap), dtype=torch.float32, device=device)
        d = self.dev
        self.soff, self.foff, self.crop = d[self.slices["soff"]], d[self.slices["foff"]], d[self.slices["crop"]]
        self.masks = d[self.slices["masks"]].view(torch.int32).view(B, 8)
        self.lr = d[self.slices["lr"]].view(torch.float32)[:1]
        self.order = d[self.slices["order"]]
        b = Batch.__new__(Batch)  # filled in by hand: Batch.__init__ syncs with the device (EOS search, token count)
        b.src, b.src_length, b.src_mask, b.src_prompt_mask = None, d[self.slices["src_length"]], None, None
        b.trg_input, b.trg = d[self.slices["trg_input"]].view(B, Lc), d[self.slices["trg"]].view(B, Lc)
        b.trg_length, b.trg_mask, b.trg_prompt_mask = d[self.slices["trg_length"]], None, None
        b.indices = torch.arange(B)
        b.nseqs, b.ntokens, b.has_trg, b.is_train, b.task = B, 0, True, True, "S2T"
        b.src_max_len, b.repad = Tb, False
        b.src_crop = self.crop[1:]  # per sub-sampler layer: output positions of the longest real utterance
        # rows > 0: the encoder stack runs on the live sub-sampled positions, packed into `rows` rows (encoders.TransformerEncoder._packing);
        # the row offsets of the utterances arrive with the batch like everything else
        b.src_pack = ops.PackedRows(d[self.slices["seg"]].view(torch.int32)[:B + 1], B, t_sub, rows) if rows > 0 else None
        self.batch = b
        self.pad_index = pad_index
        # captures by phase (first, last) of the micro-batch inside its update: a CUDAGraph, or - the last micro-batch under a
        # process group - the dict of capture_cut_step
        self.graphs: Dict[tuple, object] = {}
        self.replays = 0

    @property
    def graph(self):
        """any capture of this bucket (None: nothing captured yet)"""
        return next(iter(self.graphs.values()), None)

    @graph.setter
    def graph(self, value):
        if value is not None:
            raise ValueError("captures are stored per phase (_Bucket.graphs); only None (drop them all) can be assigned")
        self.graphs = {}

    def next_host(self):
        """the staging buffer for this step (waits, on the host, only if its copy of three steps ago has not run yet)"""
        self.turn = (self.turn + 1) % len(self.hosts)
        if self.copied[self.turn] is not None:
            self.copied[self.turn].synchronize()
        self.host = self.hosts[self.turn]

    def upload(self):
        self.dev.copy_(self.host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.copied[self.turn] = ev

    def host_view(self, name: str) -> torch.Tensor:
        return self.host[self.slices[name]]


class GraphedTrainStep:
    """step = GraphedTrainStep(TrainStep(model, ...), SpeechProcessor(...)); step.run(wave, n_samples, trg, trg_len) per batch.

    wave f32 [B, N] on the device (raw 16 kHz samples, e.g. from datasets.PrefetchLoader), n_samples: host list, trg int64
    [B, L] on the host (BOS ... EOS, padded with pad_index), trg_len: host list / tensor (including BOS and EOS).

    `run` takes one MICRO-batch: with TrainStep(batch_multiplier=k) every k-th call ends in the update (training.py:436-456); with a
    process group up (TrainStep.reducer) the k-th call also carries the gradient exchange - one exchange per update, on the summed
    gradient (the reference's DistributedDataParallel exchanges in every backward pass, training.py:584-588: the same sum)."""

    def __init__(self, step: TrainStep, proc: SpeechProcessor, compute_dtype=torch.bfloat16, frame_bucket: int = 64,
                 target_bucket: int = 8, max_graphs: int = 128, pad_index: int = 1, eos_index: int = 3, use_graphs: bool = True,
                 row_bucket: int = 384, pack_min_saving: float = 0.04):
        if step.reducer is not None and step.rt.wgrad_queue is None:
            raise ops.Js2tError("GraphedTrainStep under a process group needs the deferred weight-gradient products (TrainStep(defer_wgrads=True))")
        if step.normalization == "tokens":
            raise NotImplementedError("GraphedTrainStep: 'tokens' normalisation changes a captured constant per batch; use 'batch'")
        self.step, self.proc, self.dtype = step, proc, compute_dtype
        self.device = step.store.device
        self.frame_bucket, self.target_bucket, self.max_graphs = int(frame_bucket), int(target_bucket), int(max_graphs)
        self.pad_index, self.eos_index = pad_index, eos_index
        self.use_graphs = use_graphs
        self.ex = get_extractor(self.device, proc.sample_rate, proc.num_freq)
        self.kernel_sizes = list(step.model.encoder.subsampler.kernel_sizes)
        # packed encoder rows (ragged batches): sum of the sub-sampled lengths up to the next multiple of row_bucket is part of the
        # bucket key; 0 = padded layout (the encoder cannot pack, or the batch has fewer than pack_min_saving dead positions)
        from joeys2t_amd import encoders as _enc
        enc = step.model.encoder
        self.row_bucket = int(row_bucket) if (_enc.PACK_RAGGED and type(enc) is _enc.TransformerEncoder and compute_dtype == torch.bfloat16 and
                                              (enc.layers[0].size // enc.layers[0].src_src_att.num_heads) in (64, 128)) else 0
        self.pack_min_saving = float(pack_min_saving)
        self.buckets: "OrderedDict[tuple, _Bucket]" = OrderedDict()
        self.pool = torch.cuda.graph_pool_handle()
        step.optimizer.device_schedule = True  # update count and learning rate are read from device memory
        step.optimizer.step_dev.fill_(step.optimizer.t)  # ... starting from the updates this optimizer has already made
        step.external_lr = True                # ... and the rate arrives with the batch (one copy), not by a fill per step
        self.ntokens = 0
        self.counts = {"eager": 0, "replay": 0, "captured": 0, "evicted": 0}
        self.update_graph: Optional[torch.cuda.CUDAGraph] = None  # clip + AdamW behind the last collective (process group only; any shape)
        self.capture_errors: List[str] = []  # a bucket whose capture failed keeps running eagerly (same collectives: no agreement needed)
        self.inject_failure: Optional[str] = None  # tests: "forward" / "pieces" make the cut capture raise at that stage
        self._plan_generation = step.optimizer.plan_generation  # update plan the captured graphs were made with (builders._fused_plan)

    # ------------------------------------------------------------------------------------------------------------------
    def _bucket(self, key) -> _Bucket:
        bk = self.buckets.get(key)
        if bk is None:
            while len(self.buckets) >= self.max_graphs:
                self.buckets.popitem(last=False)
                self.counts["evicted"] += 1
            bk = _Bucket(key, len(self.kernel_sizes), self.ex.win_len, self.ex.shift, self.device, self.pad_index, t_sub=self._sub_len(key[1]))
            self.buckets[key] = bk
        else:
            self.buckets.move_to_end(key)
        return bk

    def _sub_len(self, frames: int) -> int:
        """Conv1dSubsampler.get_out_seq_lens_tensor (encoders.py:348-352) of one length, on the host"""
        for k in self.kernel_sizes:
            frames = (frames + 2 * (k // 2) - (k - 1) - 1) // 2 + 1
        return frames

    def _phase(self) -> tuple:
        """(first, last) of the coming micro-batch inside its update"""
        k = self.step.batch_multiplier
        j = self.step.micro % k
        return (j == 0, j == k - 1)

    def _phases(self) -> List[tuple]:
        """the distinct phases of an update with a micro counter that has each: [((first, last), micro index)]"""
        k = self.step.batch_multiplier
        seen, out = set(), []
        for j in range(k):
            ph = (j == 0, j == k - 1)
            if ph not in seen:
                seen.add(ph)
                out.append((ph, j))
        return out

    def _body(self, bk: _Bucket, how: str = "plain", cut_hook=None):
        """One micro-batch over the bucket's static inputs.  how: "plain" - micro_step with its flush (and, behind the last
        micro-batch of an update on a single GPU, the update); "exchange" - the last micro-batch under a process group, launched
        eagerly: the cut backward pass, the overlapped exchange, the update (what a replay of the cut capture does, call for call);
        "cut" - the same micro-batch for capture_cut_step: products stay queued, `cut_hook` between the halves, no update."""
        B, Tb = bk.key[0], bk.key[1]
        step = self.step
        step.optimizer.lr_dev.copy_(bk.lr)  # the step's learning rate, as it arrived in the packed buffer
        feats = self.proc.batch_from_tables(bk.wave, bk.soff, bk.foff, B, Tb, bk.crop[0:1], is_train=True, out_dtype=self.dtype,
                                            masks_dev=bk.masks)
        b = bk.batch
        b.src = feats
        b.trg_mask = (b.trg != self.pad_index).unsqueeze(1)
        if how == "cut":
            return step.micro_step(b, sort=False, update=False, overlap=False, flush=False, cut_hook=cut_hook)
        return step.micro_step(b, sort=False, update=True, overlap=(how == "exchange"))

    def _capture(self, bk: _Bucket, phase: tuple, micro_at: int) -> bool:
        """Capture `phase` of the bucket (nothing runs; host-side counters are put back).  A failure leaves the bucket without that
        capture - it keeps running eagerly - and is recorded in `capture_errors`."""
        step = self.step
        micro, t, steps = step.micro, step.optimizer.t, step.steps
        step.micro = micro_at
        try:
            if step.reducer is not None and phase[1]:
                cap = capture_cut_step(step, lambda hook: self._body(bk, "cut", hook), pool=self.pool, fail=self.inject_failure)
                if self.update_graph is None:
                    gu = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gu, pool=self.pool, capture_error_mode="thread_local"), step.ctx:
                        step.optimizer.clip_and_step(step.clip_grad_norm, zero_grad=True)
                    self.update_graph = gu
            else:
                cap = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(cap, pool=self.pool, capture_error_mode="thread_local" if step.reducer is not None else "global"):
                        self._body(bk)
                except BaseException:
                    if step.rt.wgrad_queue is not None:
                        step.rt.wgrad_queue.take()  # nothing of a half-made capture may stay queued
                    raise
        except Exception as exc:  # noqa: BLE001
            torch.cuda.synchronize()
            if step.reducer is None:
                raise  # a single process fails loudly (counters and queue are back where they were: `finally` below, cleanup above)
            # under a process group the eager step is still there and issues the collectives a replay would: this rank goes on
            # eagerly for this (bucket, phase), the others need not know
            self.capture_errors.append(f"{bk.key} {phase}: {repr(exc)[:200]}")
            self.counts["capture_failed"] = self.counts.get("capture_failed", 0) + 1
            bk.failed = getattr(bk, "failed", set()) | {phase}
            return False
        finally:
            step.micro, step.optimizer.t, step.steps = micro, t, steps
        bk.graphs[phase] = cap
        self.counts["captured"] += 1
        return True

    def _replay(self, bk: _Bucket, phase: tuple):
        step, cap = self.step, bk.graphs[phase]
        if isinstance(cap, dict):  # the last micro-batch under a process group: graphs cut where the collectives go
            replay_cut_step(step, cap)
            self.update_graph.replay()
        else:
            cap.replay()
        bk.replays += 1
        step.micro += 1
        if phase[1]:
            # the host's count of updates (checkpoints write it, builders.py torch_state_dict) moves once per REAL update:
            # the replayed kernel counts on the device (step_dev) and never passes FlatAdamW.clip_and_step's `t += 1`
            step.optimizer.t += 1
            step.after_update()
        self.counts["replay"] += 1

    def bucket_key(self, n_samples: Sequence[int], trg_len) -> tuple:
        """The bucket a batch falls into - host arithmetic on lengths only, so a loader that knows its epoch (samplers are
        deterministic given their seed) can ask for every bucket ahead of time and have them captured before they are needed."""
        frames = [self.ex.n_frames(int(n)) for n in n_samples]
        trg_len = [int(v) for v in (trg_len.tolist() if torch.is_tensor(trg_len) else trg_len)]
        B = len(frames)
        Tb = _round_up(max(frames), self.frame_bucket)
        rows = 0
        if self.row_bucket:
            rows = _round_up(sum(self._sub_len(f) for f in frames), self.row_bucket)
            if rows > (1.0 - self.pack_min_saving) * B * self._sub_len(Tb):
                rows = 0
        return (B, Tb, _round_up(max(trg_len), self.target_bucket), rows)

    def precapture(self, key: tuple) -> bool:
        """Capture the graphs of a bucket (every phase of the accumulation) BEFORE its first batch arrives.  Capturing executes
        nothing (no kernel runs, no collective is issued, the model, the optimizer and the RNG stay where they are - the host-side
        counters the capture pass touches are put back), so it can be done at any point between two micro-batches; it needs one
        earlier eager update of any shape (lazily built tables, the allocator's pools, the weight-gradient ranges the first update
        learns to overwrite).  Returns False when the bucket was captured already (or cannot be)."""
        if not self.use_graphs:
            return False
        if not self.counts["eager"] or self.step.micro < self.step.batch_multiplier:
            raise ops.Js2tError("GraphedTrainStep.precapture: run one update first (the first steps of a run build what captures reuse)")
        key = tuple(key)
        if key not in self.buckets and len(self.buckets) >= self.max_graphs:
            # a loader that lists more buckets than the LRU holds would silently push out captures it made a moment ago
            self.counts["precapture_refused"] = self.counts.get("precapture_refused", 0) + 1
            return False
        bk = self._bucket(key)
        made = False
        for phase, j in self._phases():
            if phase not in bk.graphs and phase not in getattr(bk, "failed", ()):
                made = self._capture(bk, phase, j) or made
        if made:
            self.counts["precaptured"] = self.counts.get("precaptured", 0) + 1
        return made

    def run(self, wave: torch.Tensor, n_samples: Sequence[int], trg: torch.Tensor, trg_len) -> str:
        """One training step on this batch; returns how it ran ("eager" on a bucket's first sight, else "replay")."""
        ex, step = self.ex, self.step
        B = len(n_samples)
        if not wave.is_cuda or wave.dtype != torch.float32 or wave.dim() != 2 or wave.shape[0] != B:
            raise ops.Js2tError("GraphedTrainStep.run: wave must be a float32 [B, N] tensor on the GPU")
        frames = [ex.n_frames(int(n)) for n in n_samples]
        if min(frames) < 1:
            raise ops.Js2tError("GraphedTrainStep.run: an utterance shorter than one frame (filter with SpeechProcessor.keep_mask)")
        order = sorted(range(B), key=lambda i: -frames[i])  # batch.sort_by_src_length() of training.py:555, on the host
        trg_len = [int(v) for v in (trg_len.tolist() if torch.is_tensor(trg_len) else trg_len)]
        L = max(trg_len)
        sub = [self._sub_len(frames[i]) for i in order]  # live encoder positions per utterance
        key = self.bucket_key(n_samples, trg_len)
        Tb = key[1]
        bk = self._bucket(key)
        Lb = key[2]
        # ---- everything that varies inside the bucket, into the pinned buffer
        bk.next_host()
        fr = np.asarray([frames[i] for i in order], dtype=np.int64)
        bk.host_view("soff").copy_(torch.arange(B, dtype=torch.int64) * bk.wave.shape[1])
        bk.host_view("foff").copy_(torch.from_numpy(np.concatenate([[0], np.cumsum(fr)])))
        bk.host_view("src_length").copy_(torch.from_numpy(fr))
        crop = [int(fr[0])]
        for k in self.kernel_sizes:  # Conv1dSubsampler.get_out_seq_lens_tensor (encoders.py:348-352) of the longest utterance
            crop.append((crop[-1] + 2 * (k // 2) - (k - 1) - 1) // 2 + 1)
        bk.host_view("crop").copy_(torch.tensor(crop, dtype=torch.int64))
        t = torch.full((B, Lb), self.pad_index, dtype=torch.int64)
        t[:, :min(L, trg.shape[1])] = trg[order][:, :L]
        # Batch.__init__ (batch.py:79-96): input = EOS -> pad, last column dropped; target = BOS dropped; length - 1
        bk.host_view("trg_input").copy_(torch.where(t == self.eos_index, torch.full_like(t, self.pad_index), t)[:, :-1].reshape(-1))
        bk.host_view("trg").copy_(t[:, 1:].reshape(-1))
        bk.host_view("trg_length").copy_(torch.tensor([trg_len[i] - 1 for i in order], dtype=torch.int64))
        masks = self.proc.draw_masks(fr.tolist()) if self.proc.specaugment is not None else np.zeros((B, 8), dtype=np.int32)
        bk.host_view("masks").view(torch.int32).copy_(torch.from_numpy(np.ascontiguousarray(masks)).reshape(-1))
        bk.host_view("lr").view(torch.float32)[0] = float(step.optimizer.param_groups[0]["lr"])
        bk.host_view("order").copy_(torch.tensor(order, dtype=torch.int64))
        seg = np.zeros((2 * ((B + 2) // 2), ), dtype=np.int32)
        seg[1:B + 1] = np.cumsum(sub)
        bk.host_view("seg").view(torch.int32).copy_(torch.from_numpy(seg))
        bk.upload()
        ncol = min(wave.shape[1], bk.wave.shape[1])
        if ncol < ex.win_len + (int(fr[0]) - 1) * ex.shift:  # (samples behind an utterance's last whole frame are never read)
            raise ops.Js2tError("GraphedTrainStep.run: waveform buffer shorter than its longest utterance")
        bk.wave[:, :ncol].copy_(wave.index_select(0, bk.order)[:, :ncol])  # rows in sorted order (samples past an utterance are never read)
        self.ntokens += int((t[:, 1:] != self.pad_index).sum())
        # ---- run
        if step.optimizer.plan_generation != self._plan_generation:
            # the optimizer rebuilt its update plan (a new LayerNorm fold, another set of un-cleared gradient ranges) after graphs were
            # captured: those graphs bake in the old tables and keep flags - a stale-plus-new gradient waiting to happen.  Drop them;
            # every bucket is captured again at its next batch.
            for other in self.buckets.values():
                other.graphs = {}
            self.update_graph = None
            self.counts["recaptured_after_plan_change"] = self.counts.get("recaptured_after_plan_change", 0) + 1
            self._plan_generation = step.optimizer.plan_generation
        phase = self._phase()
        if phase in bk.graphs:
            self._replay(bk, phase)
            return "replay"
        # first sight of this (bucket, phase): the micro-batch runs eagerly (micro_step -> [exchange ->] update -> after_update) ...
        self._body(bk, "exchange" if (step.reducer is not None and phase[1]) else "plain")
        self.counts["eager"] += 1
        if self.use_graphs and phase not in getattr(bk, "failed", ()):
            # ... and is captured over the same static inputs for every later batch of the bucket (the capture pass is not a step)
            self._capture(bk, phase, (step.micro - 1) % step.batch_multiplier)
        return "eager"

    def read_stats(self, reset: bool = True):
        out = self.step.read_stats(reset=reset)
        out["ntokens"] = float(self.ntokens)  # counted on the host: the captured statistics kernel carries a per-bucket constant
        if reset:
            self.ntokens = 0
        return out


def _cut_pieces(red, plan_part):
    """one piece per point at which a range of the flat gradient becomes complete (and its all-reduce can start): groups that
    finish no range ride with the next one that does"""
    pend = [0] * len(red.ranges)
    for _, items in plan_part:
        for it in items:
            pend[red.bucket_of_tensor(it[2])] += 1
    pieces, cur = [], []
    for entry in plan_part:
        cur.append(entry)
        done = False
        for it in entry[1]:
            bi = red.bucket_of_tensor(it[2])
            pend[bi] -= 1
            done = done or pend[bi] == 0
        if done:
            pieces.append(cur)
            cur = []
    if cur:
        pieces.append(cur)
    return pieces


def capture_cut_step(step: TrainStep, body, pool=None, fail: Optional[str] = None, mode: str = "thread_local") -> Dict[str, object]:
    """One micro-batch under a process group as graphs CUT where the collectives go: `body(cut_hook)` must end in
    `step.micro_step(batch, sort=False, update=False, overlap=False, flush=False, cut_hook=cut_hook)`.  Returns
    {"step": graph 1 (front end, forward, backward down to the encoder's output), "step2": the encoder's backward (None when the
    pass could not be cut), "plan_dec" / "pieces_dec" / "wgrad_dec": the decoder side's deferred weight-gradient products, cut into
    pieces at the points where a range of the flat gradient completes, and a graph per piece, "plan" / "pieces" / "wgrad": the same
    for what is queued at the end of backward, "pool"}.  Capturing executes nothing and issues no collective; on failure the queue
    is emptied and the error propagates (host-side counters are the caller's to put back).  thread_local: the RCCL watchdog thread
    may query its events while this thread captures.  `fail` (tests): "forward" / "pieces" raise at that stage."""
    from joeys2t_amd.runtime import WgradQueue
    red = step.reducer
    device = step.store.device
    g, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    st = {"plan_dec": None, "open": None}

    def inject(stage):
        if fail == stage:
            raise RuntimeError(f"injected capture failure at '{stage}'")

    def at_cut():
        g.capture_end()
        st["open"] = None
        st["plan_dec"] = step.rt.wgrad_queue.take(final=False)
        g2.capture_begin(pool=g.pool(), capture_error_mode=mode)
        st["open"] = g2

    cap = torch.cuda.Stream(device=device)
    cap.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(cap):
            if pool is None:
                g.capture_begin(capture_error_mode=mode)
            else:
                g.capture_begin(pool=pool, capture_error_mode=mode)
            st["open"] = g
            try:
                inject("forward")
                body(at_cut)
            finally:
                if st["open"] is not None:  # only the graph that is open is ended (a failure inside at_cut() between
                    st["open"].capture_end()  # the two captures used to end the first one a second time and mask the real error)
                    st["open"] = None
        torch.cuda.current_stream().wait_stream(cap)
        # the products queued during capture reference the graphs' static buffers: they are the per-step plan
        plan = step.rt.wgrad_queue.take()

        def capture_pieces(pieces):
            out = []
            for piece in pieces:
                gw = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gw, pool=g.pool(), capture_error_mode=mode):
                    inject("pieces")
                    WgradQueue.run(piece)
                out.append(gw)
            return out

        pieces = _cut_pieces(red, plan)
        pieces_dec = _cut_pieces(red, st["plan_dec"]) if st["plan_dec"] is not None else []
        with step.ctx:  # the deferred products are launched outside micro_step: this step's settings (deterministic: un-split) apply to them too
            gw, gw_dec = capture_pieces(pieces), capture_pieces(pieces_dec)
    except BaseException:
        step.rt.wgrad_queue.take()  # nothing of a half-made capture may stay queued
        torch.cuda.synchronize()
        raise
    return {"step": g, "step2": g2 if st["plan_dec"] is not None else None, "plan": plan, "plan_dec": st["plan_dec"], "pieces": pieces,
            "pieces_dec": pieces_dec, "wgrad": gw, "wgrad_dec": gw_dec, "pool": g.pool()}


def replay_cut_step(step: TrainStep, cap: Dict[str, object], exchange: bool = True):
    """Replay what capture_cut_step captured, with the collectives between the graphs: each range of the flat gradient goes to
    RCCL (the reducer's side stream) as soon as the piece that completes it has been launched; returns behind the last collective
    (the caller replays its update graph)."""
    red = step.reducer
    cap["step"].replay()
    if cap.get("step2") is not None:  # the decoder side's products and ranges, then the encoder's backward
        if exchange:
            red.exchange_begin(cap["plan_dec"], partial=True)
        for piece, gw in zip(cap["pieces_dec"], cap["wgrad_dec"]):
            gw.replay()
            if exchange:
                for _, items in piece:
                    red.entries_done(items)
        cap["step2"].replay()
    if exchange:
        red.exchange_begin(cap["plan"])
    for piece, gw in zip(cap["pieces"], cap["wgrad"]):
        gw.replay()
        if exchange:
            for _, items in piece:
                red.entries_done(items)
    if exchange:
        red.finish()


class GraphedDDPStep:
    """The DATA-PARALLEL train step over one fixed batch shape, replayed from hipGraphs that are cut where the collectives go.

    The reference wraps the model in DistributedDataParallel, whose reducer sends gradient buckets while backward still runs
    (joeynmt/prediction.py:508-515, training.py:584-588).  Here the single-GPU step (one captured graph) becomes, per update:

        graph 1   front end, forward, backward down to the encoder's output        (TrainStep.micro_step, cut_hook)
        pieces    the decoder side's deferred weight-gradient products, one small graph per point at which a range of the
                  flat gradient completes; between two replays that range's all-reduce is handed to RCCL (side stream)
        graph 2   the encoder's backward - while the decoder-side ranges travel
        pieces    the encoder side's weight-gradient products, range by range as above
        graph u   clip + AdamW behind the last collective

    Collectives are never captured: RCCL calls sit between replays, so a capture can only fail on its own (memory, an
    unsupported call) and `try_capture()` then lets EVERY rank fall back to the eager step (same kernels, same order, launched
    from Python) - agreed over the process group, because a rank replaying graphs and a rank launching eagerly would still issue
    the same collectives in the same order, but the caller wants to know what it timed.

    `body(cut_hook)` runs ONE micro-batch over static device inputs and must end in
    `step.micro_step(batch, sort=False, update=False, overlap=False, flush=False, cut_hook=cut_hook)`; `pre_step()` (optional)
    refreshes those inputs before every step, outside the graphs (host-to-device copies, SpecAugment draws)."""

    def __init__(self, step: TrainStep, body, pre_step=None, exchange: bool = True, inject_failure: Optional[str] = None):
        if step.reducer is None:
            raise ops.Js2tError("GraphedDDPStep: the TrainStep has no gradient reducer (no process group is up)")
        if step.rt.wgrad_queue is None:
            raise ops.Js2tError("GraphedDDPStep: needs the deferred weight-gradient products (TrainStep(defer_wgrads=True))")
        if step.batch_multiplier != 1:
            raise NotImplementedError("GraphedDDPStep: one optimizer update per batch (batch_multiplier 1)")
        self.step, self.body, self.pre_step = step, body, (pre_step or (lambda: None))
        self.exchange = bool(exchange)         # False: measurement only - the price of the cuts without the collectives
        self.inject_failure = inject_failure   # tests: "forward" / "pieces" / "update" make capture() raise at that stage
        self.device = step.store.device
        self.graphs: Dict[str, object] = {}
        self.plan = self.plan_dec = None
        self.pieces: List[list] = []
        self.pieces_dec: List[list] = []
        self.capture_error: Optional[str] = None
        self.counts = {"eager": 0, "replay": 0}
        step.optimizer.device_schedule = True  # update count and learning rate are read from device memory (replayable)
        step.optimizer.step_dev.fill_(step.optimizer.t)

    # ------------------------------------------------------------------------------------------------------------------
    def eager_step(self):
        """The step launched from Python: micro-batch, overlapped exchange + deferred products, update."""
        step = self.step
        self.pre_step()
        out = self.body(None)
        step.exchange_and_flush()
        with step.ctx:
            step.optimizer.clip_and_step(step.clip_grad_norm, zero_grad=True)
        step.after_update()
        self.counts["eager"] += 1
        return out

    def capture(self, warm: int = 2):
        """`warm` eager steps (real training steps: allocator, autotuned choices, RCCL's first calls), then the captures.  Capturing
        executes nothing and issues no collective - a rank whose capture fails has made exactly the collective calls of the ranks whose
        capture went through (try_capture relies on that)."""
        import gc
        step = self.step
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.pre_step()
        gc.collect()
        torch.cuda.empty_cache()
        micro, t_opt = step.micro, step.optimizer.t  # the capture pass is not a step: host-side counters stay
        try:
            cap = capture_cut_step(step, self.body, fail=self.inject_failure if self.inject_failure in ("forward", "pieces") else None)
            gu = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gu, pool=cap["pool"], capture_error_mode="thread_local"), step.ctx:
                if self.inject_failure == "update":
                    raise RuntimeError("GraphedDDPStep: injected capture failure at 'update'")
                step.optimizer.clip_and_step(step.clip_grad_norm, zero_grad=True)
        except BaseException:
            # leave the step usable for eager_step(): nothing of a half-made capture may stay queued or counted
            step.rt.wgrad_queue.take()
            torch.cuda.synchronize()
            raise
        finally:
            step.micro, step.optimizer.t = micro, t_opt
        self.cut = cap
        self.plan, self.plan_dec, self.pieces, self.pieces_dec = cap["plan"], cap["plan_dec"], cap["pieces"], cap["pieces_dec"]
        self.graphs = {"step": cap["step"], "step2": cap["step2"], "wgrad": cap["wgrad"], "wgrad_dec": cap["wgrad_dec"], "update": gu}

    def try_capture(self, warm: int = 2) -> Optional[str]:
        """capture(); on failure - here or on ANY rank - drop the graphs everywhere and return what went wrong (None: captured).
        Every rank must call it (one small all-reduce settles the outcome)."""
        err = None
        try:
            self.capture(warm)
        except Exception as exc:  # noqa: BLE001 - whatever the capture trips over, the eager step is still there
            err = repr(exc)[:300]
            torch.cuda.synchronize()
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            flag = torch.tensor([1.0 if err else 0.0], device=self.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if flag.item() > 0 and err is None:
                err = "capture failed on another rank"
        if err is not None:
            self.graphs = {}
        self.capture_error = err
        return err

    @property
    def captured(self) -> bool:
        return bool(self.graphs)

    def replay_step(self):
        step = self.step
        self.pre_step()
        replay_cut_step(step, self.cut, exchange=self.exchange)
        self.graphs["update"].replay()
        step.optimizer.t += 1  # the replayed kernel counts on the device; checkpoints write the host's count
        step.after_update()
        self.counts["replay"] += 1

    def run(self):
        """One training step: replayed if the capture stands, else eager."""
        if self.graphs:
            return self.replay_step()
        return self.eager_step()
