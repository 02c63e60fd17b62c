"""Minimal train-step driver for the HIP path.

The reference's TrainManager (joeynmt/training.py:47-826) is the *caller* of the hot path and is out of scope; its
Python never travels to the GPU box, so this module reproduces exactly the numerically relevant part:
`_train_step` (:541-596) and the update tail (:436-456):

  1. batch.sort_by_src_length(); model(return_type="loss", **vars(batch))
  2. batch.normalize(loss, normalization, n_gpu, batch_multiplier)          (batch.py:135-175)
  3. backward (gradients accumulate in the flat store)
  4. every batch_multiplier-th micro-batch: clip_grad_norm_ -> AdamW -> scheduler.step(steps) -> zero grads -> steps += 1
     (scheduler stepped AFTER the optimizer with the pre-increment counter: the first update runs at the configured LR)

MI355X-first differences (results unchanged): statistics stay on the device (no .item() per micro-batch; the reference
syncs 6 times, :422-423,591-594); clip coefficient, AdamW, bf16 re-cast and gradient clearing are one fused pass over
the flat store; under DDP the gradient exchange is bucketed RCCL on a side stream (helpers_for_ddp.FlatGradReducer)."""
import os
from typing import Dict, Optional

import torch

from joeys2t_amd import functional, ops
from joeys2t_amd.batch import Batch
from joeys2t_amd.builders import FlatAdamW, WarmupInverseSquareRootScheduler
from joeys2t_amd.helpers_for_ddp import FlatGradReducer, use_ddp
from joeys2t_amd.model import Model
from joeys2t_amd.runtime import WgradQueue


# the backward pass cut at the encoder's output under data parallelism (JS2T_EARLY_EXCHANGE=0: one piece, exchange after it)
EARLY_EXCHANGE = os.environ.get("JS2T_EARLY_EXCHANGE", "1") != "0"


class TrainStep:
    def __init__(self, model: Model, *, learning_rate: float = 2.0e-3, adam_betas=(0.9, 0.98), weight_decay: float = 0.0,
                 clip_grad_norm: Optional[float] = 10.0, scheduling: Optional[str] = "warmupinversesquareroot",
                 learning_rate_warmup: int = 10000, learning_rate_min: float = 1.0e-6, normalization: str = "batch",
                 batch_multiplier: int = 1, n_gpu: int = 1, n_buckets: int = 4, sync_every_backward: bool = False,
                 defer_wgrads: bool = True, overlap_ctc: bool = False, comm_dtype: Optional[torch.dtype] = None, comm=None,
                 deterministic: bool = False):
        self.model = model
        # deterministic=True: the reference's set_seed asks cuDNN for deterministic kernels (helpers.py:93-104); here every
        # floating-point-atomic sum of the train step takes an ordered form - two runs from one state then agree bit for bit
        # (tests/test_hip_deterministic.py), at a price bench.py reports.  The setting lives in THIS step's js2t_ctx, bound to the
        # thread around the step's launches (`with self.ctx:`): another TrainStep in the same process keeps its own (round 5 wrote
        # a process-wide switch here - the last constructor won).
        from joeys2t_amd._lib import Context
        self.deterministic = bool(deterministic) or os.environ.get("JS2T_DETERMINISTIC", "0") == "1"
        self.ctx = Context(deterministic=int(self.deterministic))
        model.overlap_ctc = bool(overlap_ctc)
        self.rt = model.runtime
        self.store = self.rt.store
        self.normalization, self.batch_multiplier, self.n_gpu = normalization, batch_multiplier, n_gpu
        self.clip_grad_norm = clip_grad_norm
        self.optimizer = FlatAdamW(self.store, lr=learning_rate, betas=adam_betas, weight_decay=weight_decay)
        self.scheduler = None
        if scheduling == "warmupinversesquareroot":
            self.scheduler = WarmupInverseSquareRootScheduler(self.optimizer, peak_rate=learning_rate,
                                                              warmup=learning_rate_warmup, min_rate=learning_rate_min)
        elif scheduling is not None:
            raise NotImplementedError(f"scheduler {scheduling}")
        self.store.attach_grads(zero=True)
        self.store.auto_refresh = False  # js2t_adamw keeps the bf16 shadow in sync
        self.steps = 0
        self.micro = 0
        self.sync_every_backward = sync_every_backward
        # weight-gradient products are queued during backward and run grouped by layer type afterwards; under DDP the flat
        # gradient is exchanged range by range (the store's type ranges) while those products run
        self.rt.wgrad_queue = WgradQueue() if defer_wgrads else None
        # LayerNorm gamma / beta gradients go through per-XCD copies that are folded right after backward (ops.GradCopies);
        # only with the deferred products: the hook-driven exchange (no queue) may send a bucket while backward still runs
        self.rt.grad_copies = ops.GradCopies(self.store.device) if defer_wgrads else None
        # Weight gradients that ONE un-split product per micro-batch writes: the first micro-batch of an update overwrites them
        # (no read of the old gradient in its epilogue, no clearing pass in the update: 8 of 38 bytes per parameter), and on a
        # single GPU the last one leaves the sums of squares clip_grad_norm_ needs behind (no pass over them for the norm).
        # Candidates: nn.Linear weights that belong to one module only (a tied matrix collects two contributions).
        q = self.rt.wgrad_queue
        if q is not None and os.environ.get("JS2T_WGRAD_OVERWRITE", "1") != "0" and self.optimizer.update_ranges == [(0, self.store.total)]:
            from joeys2t_amd.runtime import RangeSet
            uses: Dict[int, int] = {}
            for m in model.modules():
                for p in m._parameters.values():
                    if p is not None:
                        uses[id(p)] = uses.get(id(p), 0) + 1
            spans = []
            for m in model.modules():
                if isinstance(m, torch.nn.Linear) and uses.get(id(m.weight), 0) == 1 and m.weight.requires_grad and id(m.weight) in self.store.offsets:
                    lo = self.store.offsets[id(m.weight)]
                    spans.append((lo, lo + m.weight.numel()))
            q.cand = RangeSet(spans)
            q.kept = self.optimizer.keep = RangeSet()  # grows with what first flushes are seen to overwrite
            q.grad_base = self.store.flat_grad
        self.reducer = None
        if use_ddp():
            # comm: a joeys2t_amd.comm.Communicator, or "cabi" (JS2T_COMM=cabi) to bootstrap one over the process group that is
            # up - the collectives then go through js2t_comm_* of the C boundary instead of torch.distributed.all_reduce
            if comm is None and os.environ.get("JS2T_COMM", "") == "cabi":
                comm = "cabi"
            if comm == "cabi":
                from joeys2t_amd.comm import Communicator
                comm = Communicator.from_process_group(self.store.device)
            self.reducer = FlatGradReducer(self.store, n_buckets=n_buckets, ranges=self.store.type_ranges if defer_wgrads else None,
                                           comm_dtype=comm_dtype, comm=comm)
        # leaves behind the encoder's output: what the first half of a cut backward pass accumulates into (see micro_step)
        self._late_leaves = [p for n, p in model.named_parameters() if n.startswith(("decoder.", "trg_embed.")) and p.requires_grad]
        # bucket bookkeeping by per-parameter notifications only without the queue (with it: exchange_and_flush)
        self.rt.on_grads_ready = self.reducer.params_ready if (self.reducer is not None and self.rt.wgrad_queue is None) else None
        # running statistics on the device: [loss, nll, ctc, n_correct, nseqs, ntokens]
        self.stats = torch.zeros(6, dtype=torch.float64, device=self.store.device)
        self._grad_seeds: Dict[float, torch.Tensor] = {}  # device constants seeding backward, by normalisation factor
        self.external_lr = False  # True: the caller delivers the step's learning rate to optimizer.lr_dev itself (graphed.py)

    def exchange_and_flush(self, plan=None):
        """Deferred weight-gradient products + gradient exchange of one optimizer step, overlapped: each range of the flat
        gradient goes to RCCL (side stream) as soon as the last product writing into it has been launched.  `plan`: a
        WgradQueue plan kept from a hipGraph capture (replayed steps), default: whatever the queue holds now."""
        q = self.rt.wgrad_queue
        with self.ctx:
            if plan is None:
                plan = q.take() if q is not None else []
            if self.reducer is None:
                WgradQueue.run(plan)
                return
            self.reducer.exchange_begin(plan)
            WgradQueue.run(plan, self.reducer.entries_done)
            self.reducer.finish()

    def micro_step(self, batch: Batch, sort: bool = True, update: bool = True, overlap: bool = True, flush: bool = True, cut_hook=None):
        """One micro-batch: forward, normalised loss, backward.  Returns the (device) normalised loss.
        `sort=False` skips batch.sort_by_src_length() (a host sync) for callers that sorted already;
        `update=False` leaves the optimizer step to the caller (hipGraph capture of forward+backward only);
        `overlap=False` disables the gradient exchange here (the caller runs exchange_and_flush() / reducer.reduce_all());
        `flush=False` leaves the deferred weight-gradient products queued (the caller keeps them as a plan);
        `cut_hook`: called between the two halves of the backward pass (cut at the encoder's output) INSTEAD of the partial exchange
        - for a caller that captures the halves as two hipGraphs and runs the exchange itself between their replays (bench.py)."""
        # this step's settings for every launch below; the backward passes run on THIS thread (autograd's device threads would not
        # see the binding)
        with self.ctx, torch.autograd.set_multithreading_enabled(False):
            return self._micro_step(batch, sort, update, overlap, flush, cut_hook)

    def _micro_step(self, batch: Batch, sort: bool, update: bool, overlap: bool, flush: bool, cut_hook):
        model = self.model
        model.train()
        self.rt.rng.begin_step()
        functional.reset_handover()
        if sort:
            batch.sort_by_src_length()
        last = (self.micro + 1) % self.batch_multiplier == 0
        q = self.rt.wgrad_queue
        if q is not None and q.cand is not None:
            q.first = self.micro % self.batch_multiplier == 0
            # the epilogues' sums are those of the LOCAL gradient: under data parallelism the norm is taken after the exchange
            collect = last and self.reducer is None and self.clip_grad_norm is not None and self.clip_grad_norm > 0
            if collect and self.optimizer.collector is None:
                from joeys2t_amd.builders import SumsqCollector
                self.optimizer.collector = SumsqCollector(self.store.flat_grad, self.store.total)
            q.collector = self.optimizer.collector if collect else None
        exchange = self.reducer is not None and overlap and (last or self.sync_every_backward)
        use_hooks = exchange and self.rt.wgrad_queue is None  # without the queue: bucket hooks fire during backward
        if use_hooks:
            self.reducer.begin(armed=True)
        functional.begin_memory_chain()
        try:
            total, nll, ctc, n_correct = model(return_type="loss", **vars(batch))
        except BaseException:
            functional.end_memory_chain(check=False)
            raise
        # batch.normalize() (batch.py:135-175) is a multiplication by a host constant: it seeds the backward pass instead of
        # running as a chain of scalar kernels (division forward + backward, then the same division for every statistic)
        inv_norm = self._inv_normalizer(batch)
        seed = self._grad_seeds.get(inv_norm)
        if seed is None:
            seed = torch.full((), inv_norm, dtype=torch.float32, device=total.device)
            if not torch.cuda.is_current_stream_capturing():  # made under capture it is filled by THAT graph's replays only: not shared
                self._grad_seeds[inv_norm] = seed
        # Data parallel: the backward pass in two halves, cut at the encoder's output.  Behind it (decoder, target embedding, both
        # output layers, the CTC branch) every gradient is complete after the first half: the decoder side's weight-gradient
        # products run there and their ranges of the flat gradient (ParamStore.late_ranges) travel while the encoder's backward
        # - three quarters of the pass - still runs (the reference's DistributedDataParallel reducer sends its buckets during
        # backward too, prediction.py:511-513).  The encoder output is the only tensor that connects the halves.
        enc_out = getattr(model, "_cut_tensor", None)
        model._cut_tensor = None
        can_cut = (EARLY_EXCHANGE and enc_out is not None and enc_out.grad_fn is not None and bool(self.store.late_ranges) and
                   self.rt.wgrad_queue is not None)
        cut = can_cut and (cut_hook is not None or (exchange and not use_hooks and self.reducer.armed_for_exchange()))
        try:
            if cut:
                torch.autograd.backward(total, grad_tensors=seed, inputs=[enc_out] + self._late_leaves)
                functional.end_memory_chain()  # every cross-attention block has run: the encoder-state gradient is whole
                if cut_hook is not None:
                    cut_hook()
                else:
                    plan = self.rt.wgrad_queue.take(final=False)
                    self.reducer.exchange_begin(plan, partial=True)
                    WgradQueue.run(plan, self.reducer.entries_done)
                g_enc, enc_out.grad = enc_out.grad, None
                enc_out.backward(g_enc)
            else:
                total.backward(gradient=seed)
        except BaseException:
            functional.end_memory_chain(check=False)
            raise
        if not cut:
            functional.end_memory_chain()
        if self.rt.grad_copies is not None:
            self.rt.grad_copies.fold()
        if use_hooks:
            self.reducer.finish()
        elif exchange:
            self.exchange_and_flush()
        elif flush:
            self.rt.flush_wgrads()
        with torch.no_grad():  # one launch: stats += [total, nll, ctc] * inv_norm, n_correct, nseqs, ntokens
            norm = ops.train_stats(self.stats, total.detach(), None if nll is None else nll.detach(), None if ctc is None else ctc.detach(),
                                   n_correct, inv_norm, batch.nseqs, batch.ntokens or 0)
        self.rt.rng.advance()
        self.micro += 1
        if last and update:
            self.update()
        return norm

    def _inv_normalizer(self, batch: Batch) -> float:
        """The factor batch.normalize(x, normalization, n_gpu, batch_multiplier) multiplies a 0-d loss by (batch.py:135-175:
        "sum" returns the tensor untouched, otherwise / normalizer, / n_gpu, / n_accumulation)."""
        if self.normalization == "sum":
            return 1.0
        f = 1.0 / {"batch": batch.nseqs, "tokens": batch.ntokens, "none": 1}[self.normalization]
        if self.n_gpu > 1:
            f /= self.n_gpu
        if self.batch_multiplier > 1:
            f /= self.batch_multiplier
        return f

    def update(self):
        """clip -> AdamW -> scheduler.step(steps) -> (grads cleared in the kernel) -> steps += 1."""
        with self.ctx:
            self.optimizer.clip_and_step(self.clip_grad_norm, zero_grad=True)
        if torch.cuda.is_current_stream_capturing():
            return  # host-side schedule bookkeeping happens per replay: after_replay()
        self.after_update()

    def after_update(self):
        if self.scheduler is not None:
            self.scheduler.step(self.steps)
            if not self.external_lr:
                self.optimizer.lr_dev.fill_(self.optimizer.param_groups[0]["lr"])
        self.steps += 1

    # ---- checkpoints in the reference's layout (training.py:149-190,220-285) ---------------------------------------------
    def state_for_checkpoint(self) -> Dict:
        """The reference's checkpoint layout (training.py:166-177), readable by ITS loader too: TrainStatistics.load_state_dict
        indexes all six stats keys (:819-826) and `train_iter_state` gets `.cpu()` called on it (:287) - a generator state
        tensor, here of a fresh torch.Generator (this driver owns no sampler; pass `train_iter_state` to save a real one)."""
        params = list(self.model.parameters())
        stats = dict(zip(["loss", "nll", "ctc", "n_correct", "nseqs", "ntokens"], self.stats.tolist()))  # local: no collective here
        return {"model_state": self.model.state_dict(), "optimizer_state": self.optimizer.torch_state_dict(params), "scaler_state": None,
                "scheduler_state": None if self.scheduler is None else {"step": self.scheduler._step, "rate": self.scheduler._rate},
                "train_iter_state": getattr(self, "train_iter_state", None) if getattr(self, "train_iter_state", None) is not None
                else torch.Generator().get_state(),
                "stats_state": {"epochs": getattr(self, "epochs", 1), "steps": self.steps, "total_tokens": int(stats.get("ntokens", 0)),
                                "total_correct": int(stats.get("n_correct", 0)), "best_ckpt_score": float("inf"), "best_ckpt_iter": 0}}

    def save_checkpoint(self, path):
        torch.save(self.state_for_checkpoint(), str(path))

    def init_from_checkpoint(self, path, reset_scheduler: bool = False, reset_optimizer: bool = False, map_location="cpu"):
        """Model, optimizer moments / step count and schedule position from a JoeyS2T checkpoint - the reference's own files
        load as they are (same parameter names and order)."""
        from joeys2t_amd.helpers import load_checkpoint
        ckpt = load_checkpoint(path, map_location)
        self.model.load_state_dict(ckpt["model_state"])
        if not reset_optimizer:
            self.optimizer.load_torch_state_dict(ckpt["optimizer_state"], list(self.model.parameters()))
        if not reset_scheduler and ckpt.get("scheduler_state") is not None and self.scheduler is not None:
            self.scheduler.load_state_dict(ckpt["scheduler_state"])
            self.optimizer.param_groups[0]["lr"] = ckpt["optimizer_state"]["param_groups"][0]["lr"] if not reset_optimizer else self.scheduler._rate
        self.optimizer.lr_dev.fill_(self.optimizer.param_groups[0]["lr"])
        self.steps = int(ckpt.get("stats_state", {}).get("steps", ckpt.get("steps", 0)))
        self.micro = 0

    def read_stats(self, reset: bool = True) -> Dict[str, float]:
        """One host sync (and, under DDP, one 6-element all-reduce) for everything the reference logs."""
        s = self.stats.clone()
        if use_ddp():
            torch.distributed.all_reduce(s)
        vals = s.tolist()
        if reset:
            self.stats.zero_()
        keys = ["loss", "nll", "ctc", "n_correct", "nseqs", "ntokens"]
        out = dict(zip(keys, vals))
        out["grad_norm"] = float(self.optimizer.norm_clip[0])
        out["lr"] = self.optimizer.param_groups[0]["lr"]
        return out
