#!/usr/bin/env python
"""bench.py — train-step throughput of the JoeyS2T hot path on MI355X.

One step = one pass of the whole hot path over one batch that is already resident in HBM:
  raw 16 kHz waveforms [32 x 240000] -> Kaldi fbank -> CMVN -> SpecAugment -> pad -> conv subsampler ->
  16-layer Transformer encoder / 8-layer decoder -> CTC + label-smoothed CE -> backward -> clip + AdamW
(configs/librispeech_100h.yaml shapes, bf16 compute, dropout 0.1, batch_multiplier 1 so that every step carries
its optimizer update).  Metric: input fbank frames per second over all GPUs (47,936 frames per GPU per step).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under torch.distributed.run (WORLD_SIZE set: how the driver starts it) the process is one of
the ranks; started plainly, `bench.py --gpus N` spawns its own N ranks (torch.distributed.run as a child process, before
this process has touched a GPU - the reference starts its ranks itself too, joeynmt/__main__.py:72-79) and exits with the
child's code.  Rank 0 prints the JSON line either way.

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel = bf16 MFMA GEMM, HIP-event timed on its
launch stream) and `cpu_baseline` (the CPU oracle restatement on a bounded sample) objects.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

LS100_MODEL = {
    "initializer": "xavier_uniform", "init_gain": 1.0, "bias_initializer": "zeros", "embed_initializer": "xavier_uniform",
    "embed_init_gain": 1.0, "tied_embeddings": False, "tied_softmax": False,
    "encoder": {"type": "transformer", "num_layers": 16, "num_heads": 4, "embeddings": {"embedding_dim": 80},
                "hidden_size": 512, "ff_size": 2048, "dropout": 0.1, "freeze": False, "subsample": True,
                "conv_kernel_sizes": [5, 5], "conv_channels": 512, "in_channels": 80, "layer_norm": "pre", "activation": "relu"},
    "decoder": {"type": "transformer", "num_layers": 8, "num_heads": 4,
                "embeddings": {"embedding_dim": 512, "scale": True, "dropout": 0.1}, "hidden_size": 512, "ff_size": 2048,
                "dropout": 0.1, "freeze": False, "layer_norm": "pre", "activation": "relu"},
}
VOCAB = 5000
SAMPLES = 240000  # 15.0 s @ 16 kHz  ->  1498 frames  ->  375 encoder positions
BATCH = 32
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def synth_waveforms(batch, samples, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(batch, samples, generator=g)).clamp_(-1.0, 1.0)


def synth_targets(batch, vocab, seed=1234, lo=40, hi=80):
    g = torch.Generator().manual_seed(seed + 1)
    lens = torch.randint(lo, hi + 1, (batch, ), generator=g)
    L = int(lens.max()) + 2
    trg = torch.full((batch, L), 1, dtype=torch.long)
    for b in range(batch):
        n = int(lens[b])
        trg[b, 0] = 2
        trg[b, 1:1 + n] = torch.randint(4, vocab, (n, ), generator=g)
        trg[b, 1 + n] = 3
    return trg, lens + 2


class GemmTimer:
    """HIP-event timing of every launch of one kernel family, on the stream the launches go to."""

    def __init__(self):
        self.records = []  # (key, flops, start_event, end_event)

    def wrap(self, key, flops, launch, nbytes=0):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        s.record(st)
        launch()
        e.record(st)
        self.records.append((key, flops, s, e, nbytes))

    def summary(self):
        """{kernel: [launches, flops, seconds, algorithmic operand + result bytes]}"""
        torch.cuda.synchronize()
        agg = {}
        for key, flops, s, e, nbytes in self.records:
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += s.elapsed_time(e) * 1e-3
            a[3] += nbytes
        return agg


def build_step(device, world, dtype=torch.bfloat16, seed=42, ragged=False, host_inputs=False, ddp=None):
    """`ddp`: run the data-parallel form of the step (graph-replayed forward + backward, graph-replayed weight-gradient
    groups interleaved with the RCCL exchange of the ranges they complete, graph-replayed update).  Default: world > 1;
    JS2T_BENCH_FORCE_DDP=1 takes that path on a single rank (a one-rank RCCL communicator) to rehearse it on a 1-GPU box."""
    ddp = (world > 1) if ddp is None else ddp
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.model import build_model
    from joeys2t_amd.tokenizers import SpeechProcessor
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    import copy
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = build_model(copy.deepcopy(LS100_MODEL), None, Vocabulary.synthetic(VOCAB))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.finalize(device, dtype, seed=seed)
    # gradient exchange of the scored line: fp32, the reference's all-reduce (torch DistributedDataParallel, prediction.py:508-515).
    # bf16 staging (half the bytes on xGMI, a 2^-9 rounding of gradients that come out of bf16 products) is timed as a SIDE figure
    # behind it (config.grad_exchange carries both); JS2T_COMM_DTYPE=bf16 makes it the headline's exchange instead.
    comm_dtype = None
    if ddp and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and os.environ.get("JS2T_COMM_DTYPE", "fp32") == "bf16":
        comm_dtype = torch.bfloat16
    step = TrainStep(model, learning_rate=2.0e-3, adam_betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=10.0,
                     learning_rate_warmup=10000, learning_rate_min=1.0e-6, normalization="batch", batch_multiplier=1,
                     n_gpu=world, comm_dtype=comm_dtype,
                     # second stream for the CTC branch - except in the gloo rehearsal, where several ranks time-slice ONE card
                     # and its cross-stream waits turn into a 13x slowdown (an artefact of that set-up, not of RCCL)
                     overlap_ctc=(world == 1 or torch.distributed.get_backend() == "nccl") and os.environ.get("JS2T_OVERLAP_CTC", "1") != "0")
    proc = SpeechProcessor(num_freq=80, min_length=10, max_length=6000,
                           specaugment=dict(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=100, time_mask_p=1.0),
                           cmvn=dict(norm_means=True, norm_vars=True, before=True))
    rank = int(os.environ.get("RANK", 0))
    trg, trg_len = synth_targets(BATCH, VOCAB, seed=1234 + rank)
    if ragged:
        # SURVEY 8(d)'s ragged variant: N_i ~ U{160000..272000} samples (10-17 s), pre-sorted by length (descending) so that
        # batch.sort_by_src_length() is the identity and the replayed graph sees the same order every step
        g = torch.Generator().manual_seed(4321 + rank)
        n_samples = sorted(torch.randint(160000, 272001, (BATCH, ), generator=g).tolist(), reverse=True)
        wave = synth_waveforms(BATCH, max(n_samples), seed=1234 + rank)
        for i, n in enumerate(n_samples):
            wave[i, n:] = 0.0
        wave = wave.to(device)
    else:
        wave = synth_waveforms(BATCH, SAMPLES, seed=1234 + rank).to(device)
        n_samples = [SAMPLES] * BATCH
    frames_list = [1 + (n - 400) // 160 for n in n_samples]
    frames = max(frames_list)
    step.optimizer.device_schedule = True
    # static per-step inputs: SpecAugment mask parameters are drawn on the host (np.random, reference order) and
    # copied into a fixed device buffer, so the same code runs eagerly or as a replayed hipGraph
    masks_host = torch.zeros((BATCH, 8), dtype=torch.int32).pin_memory()
    masks_dev = torch.zeros((BATCH, 8), dtype=torch.int32, device=device)
    state = {"batch": None}

    # --host-inputs: the waveforms arrive from (pinned) host memory every step, as a DataLoader would hand them over
    # (PCIe-inclusive rate, reported in DESIGN.md, never `value`).  The transfer of step i + 1 runs on a copy stream while
    # step i computes (datasets.PrefetchLoader's scheme): H2D into a staging buffer, then a 10 us device copy into the
    # graph's static input at the head of the step.  JS2T_NO_PREFETCH=1 puts the H2D copy in front of the step instead.
    wave_host = wave.cpu().pin_memory() if host_inputs else None
    prefetch = host_inputs and os.environ.get("JS2T_NO_PREFETCH", "0") != "1"
    if prefetch:
        stage_buf, copy_stream = torch.empty_like(wave), torch.cuda.Stream(device=device)
        h2d_done, stage_free = torch.cuda.Event(), torch.cuda.Event()

        def start_h2d():
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(stage_free)  # the previous step has read the staging buffer
                stage_buf.copy_(wave_host, non_blocking=True)
                h2d_done.record(copy_stream)

        stage_free.record(torch.cuda.current_stream())
        start_h2d()

    def pre_step():
        masks_host.copy_(torch.from_numpy(proc.draw_masks(frames_list)))
        masks_dev.copy_(masks_host, non_blocking=True)
        if prefetch:
            cur = torch.cuda.current_stream()
            cur.wait_event(h2d_done)
            wave.copy_(stage_buf, non_blocking=True)
            stage_free.record(cur)
            start_h2d()  # the next batch travels while this step computes
        elif wave_host is not None:
            wave.copy_(wave_host, non_blocking=True)

    def body(cut_hook=None):
        feats, lengths = proc.batch_from_waveforms(wave, n_samples, is_train=True, out_dtype=dtype, masks_dev=masks_dev)
        if state["batch"] is None:
            b = Batch(src=feats, src_length=torch.tensor(lengths, device=device), src_prompt_mask=None, trg=trg,
                      trg_length=trg_len, trg_prompt_mask=None, indices=torch.arange(BATCH), device=device, pad_index=1,
                      eos_index=3, is_train=True, task="S2T", n_gpu=1)
            b.sort_by_src_length()  # batch.sort_by_src_length() of training.py:555 (all lengths equal here)
            # the lengths as host integers (Batch keeps them when it is built from host tensors): a ragged batch then runs its
            # encoder on the live rows only (encoders.TransformerEncoder._packing; all lengths equal: nothing to drop)
            b.src_length_host = [int(v) for v in b.src_length.tolist()]
            state["batch"] = b
        state["batch"].src = feats
        # single GPU: the whole step incl. the update is ONE captured graph.  Data parallel: the same kernels in the same
        # order, cut where the RCCL calls go: forward + backward (the deferred weight-gradient products stay queued:
        # flush=False), then one piece per weight-gradient group, then the update (see capture()).
        return step.micro_step(state["batch"], sort=False, update=not ddp, overlap=False, flush=not ddp, cut_hook=cut_hook)

    def grad_only():
        """forward + backward + the deferred weight-gradient products of the bench batch, no update: the flat gradient stays"""
        pre_step()
        feats, lengths = proc.batch_from_waveforms(wave, n_samples, is_train=True, out_dtype=dtype, masks_dev=masks_dev)
        if state["batch"] is None:
            body()  # builds the batch (and takes a step)
        state["batch"].src = feats
        return step.micro_step(state["batch"], sort=False, update=False, overlap=False, flush=True)

    state["grad_only"] = grad_only

    def eager_step():
        pre_step()
        out = body()
        if ddp:
            step.exchange_and_flush()
            step.optimizer.clip_and_step(step.clip_grad_norm, zero_grad=True)
            step.after_update()
        return out

    graphs = {}
    if ddp:
        # Data parallel: the step cut where the RCCL calls go - two captured halves of forward + backward, one small graph per
        # completed range of the flat gradient, the update - lives in the library (graphed.GraphedDDPStep, tests/test_hip_ddp_graphed.py)
        from joeys2t_amd.graphed import GraphedDDPStep
        dd = GraphedDDPStep(step, body=lambda hook: body(cut_hook=hook), pre_step=pre_step,
                            exchange=os.environ.get("JS2T_BENCH_NO_EXCHANGE") != "1")  # =1, measurement only: the cuts without the collectives
        state["ddp_step"] = dd
        return dd.eager_step, dd.replay_step, dd.capture, step, sum(frames_list), (model, state)

    def capture():
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        pre_step()
        with torch.cuda.graph(g, capture_error_mode="global"):
            body()
        graphs["step"] = g
        # capturing executes nothing (and the graph's buffers hold no gradients yet): one real replayed step, which also puts
        # the host-side update counter / learning-rate schedule in line with the device's
        graph_step()

    def graph_step():
        pre_step()
        graphs["step"].replay()
        step.after_update()

    return eager_step, graph_step, capture, step, sum(frames_list), (model, state)


def varying_bench(device, steps, warmup, dtype=torch.bfloat16, seed=42, pool_utts=512, use_graphs=True):
    """Side figure: the train step over batches that change every step, as the reference's loader cuts them - utterances of
    10-17 s drawn by TokenBatchSampler (datasets.py:1249-1295 of the reference) from a shuffled synthetic corpus, handed to the
    GPU one batch ahead by PrefetchLoader (pinned rows -> HBM on a copy stream), run by graphed.GraphedTrainStep (one hipGraph
    per (B, frames / 64, target length / 8) bucket; first sight of a bucket = eager step + capture).  `warmup` steps fill the
    buckets; the timed steps include whatever new buckets still turn up.  value = UN-padded frames per second."""
    import copy
    from joeys2t_amd.datasets import PrefetchLoader, TokenBatchSampler
    from joeys2t_amd.graphed import GraphedTrainStep
    from joeys2t_amd.helpers_for_ddp import RandomSubsetSampler
    from joeys2t_amd.model import build_model
    from joeys2t_amd.tokenizers import SpeechProcessor
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = build_model(copy.deepcopy(LS100_MODEL), None, Vocabulary.synthetic(VOCAB))
    model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
    model.finalize(device, dtype, seed=seed)
    # JS2T_VARY_MULT=k: the config's accumulation over the varying batches too (k micro-batches of k different shapes per update)
    step = TrainStep(model, learning_rate=2.0e-3, adam_betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=10.0, learning_rate_warmup=10000,
                     learning_rate_min=1.0e-6, normalization="batch", batch_multiplier=int(os.environ.get("JS2T_VARY_MULT", 1)), n_gpu=1,
                     overlap_ctc=True)
    proc = SpeechProcessor(num_freq=80, min_length=10, max_length=6000,
                           specaugment=dict(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=100, time_mask_p=1.0),
                           cmvn=dict(norm_means=True, norm_vars=True, before=True))
    # bucket widths (measurement knobs): finer buckets pad less and need more graphs - affordable since they are captured ahead
    gstep = GraphedTrainStep(step, proc, compute_dtype=dtype, use_graphs=use_graphs, frame_bucket=int(os.environ.get("JS2T_VARY_FRAME_BUCKET", 64)),
                             row_bucket=int(os.environ.get("JS2T_VARY_ROW_BUCKET", 384)), max_graphs=256)
    # the corpus: pool_utts utterances of 10-17 s in pinned host memory (slices of one noise buffer), targets of 40-80 tokens
    g = torch.Generator().manual_seed(4321)
    n_samples = torch.randint(160000, 272001, (pool_utts, ), generator=g).tolist()
    noise = (0.1 * torch.randn(272000 + 4096 * 8, generator=g)).clamp_(-1.0, 1.0).pin_memory()
    starts = torch.randint(0, 4096 * 8, (pool_utts, ), generator=g).tolist()
    tlen = torch.randint(40, 81, (pool_utts, ), generator=g).tolist()
    targets = [torch.cat([torch.tensor([2]), torch.randint(4, VOCAB, (k, ), generator=g), torch.tensor([3])]) for k in tlen]
    frames = [1 + (n - 400) // 160 for n in n_samples]

    class Corpus:
        def __init__(self):
            self.indices, self.random_subset, self.seed = list(range(pool_utts)), -1, seed

        def __len__(self):
            return pool_utts

        def reset_indices(self):
            self.indices = list(range(pool_utts))

        def __getitem__(self, i):  # (index, source, target): the samplers only ask for lengths
            return i, range(frames[i]), range(tlen[i] + 2)

    sampler = TokenBatchSampler(RandomSubsetSampler(Corpus(), shuffle=True, generator=torch.Generator().manual_seed(seed)),
                                batch_size=BATCH * 1650, drop_last=False, seed=seed)

    def load(idx):
        L = max(tlen[i] for i in idx) + 2
        trg = torch.full((len(idx), L), 1, dtype=torch.long)
        for r, i in enumerate(idx):
            trg[r, :tlen[i] + 2] = targets[i]
        return {"wave": [noise[starts[i]:starts[i] + n_samples[i]] for i in idx], "n_samples": [n_samples[i] for i in idx], "trg": trg.numpy(),
                "trg_len": [tlen[i] + 2 for i in idx]}

    def batches():
        while True:  # epoch after epoch (the sampler re-shuffles)
            for b in sampler:
                yield b

    it = iter(PrefetchLoader(batches(), load, device))
    n_frames = 0

    marks = []  # (how, frames, start event, end event) of every timed step: the replayed ones are also reported on their own

    host = []  # (first launch of its graph?, host ms inside run()) per timed step
    def one(timed=False):
        nonlocal n_frames
        item = next(it)
        nf = sum(1 + (n - 400) // 160 for n in item["n_samples"])
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            bk = gstep.buckets.get(gstep.bucket_key(item["n_samples"], item["trg_len"])) if use_graphs else None
            first = bk is not None and bk.graph is not None and bk.replays == 0  # captured ahead, never launched: the launch uploads it
            h0 = time.perf_counter()
            e0.record()
        how = gstep.run(item["wave"], item["n_samples"], torch.from_numpy(item["trg"]), item["trg_len"])
        if timed:
            e1.record()
            marks.append((how, nf, e0, e1))
            host.append((first, (time.perf_counter() - h0) * 1e3))
        n_frames += nf
        return how

    for _ in range(step.batch_multiplier):
        one()  # the first update of the run: eager + capture of its own buckets
    # The loader knows its epoch: the samplers are deterministic given their seeds, so an identical second sampler lists the batches
    # to come and every bucket they fall into is captured BEFORE its first batch arrives (GraphedTrainStep.precapture: capturing
    # executes nothing).  JS2T_BENCH_NO_PRECAPTURE=1: first sight = eager step + capture inside the timed loop, as in round 4.
    n_pre = 0
    if use_graphs and os.environ.get("JS2T_BENCH_NO_PRECAPTURE", "0") != "1":
        peek = TokenBatchSampler(RandomSubsetSampler(Corpus(), shuffle=True, generator=torch.Generator().manual_seed(seed)),
                                 batch_size=BATCH * 1650, drop_last=False, seed=seed)
        seen, n_listed = set(), 0
        while n_listed < warmup + steps + 2:
            for idx in peek:
                seen.add(gstep.bucket_key([n_samples[i] for i in idx], [tlen[i] + 2 for i in idx]))
                n_listed += 1
                if n_listed >= warmup + steps + 2:
                    break
        for key in sorted(seen):
            n_pre += int(gstep.precapture(key))
    for _ in range(warmup - 1):
        one()
    torch.cuda.synchronize()
    gstep.read_stats()
    n_frames, before = 0, dict(gstep.counts)
    t0 = time.perf_counter()
    for _ in range(steps):
        one(timed=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = gstep.read_stats()
    rep = [(nf, e0.elapsed_time(e1)) for how, nf, e0, e1 in marks if how == "replay"]
    first_ms = sorted(ms for f, ms in host if f)
    later_ms = sorted(ms for f, ms in host if not f)
    rep_ms = sum(ms for _, ms in rep) / max(len(rep), 1)
    packed = sum(1 for k in gstep.buckets if k[3] > 0)
    return {"ms_per_step": round(dt / steps * 1e3, 3), "frames_per_s_unpadded": round(n_frames / dt, 1), "steps": steps, "warmup": warmup,
            # the replayed steps alone (GPU time between events around each): what a long run converges to once its buckets are captured
            "ms_per_replayed_step": round(rep_ms, 3), "frames_per_s_unpadded_replayed": round(sum(nf for nf, _ in rep) / max(rep_ms * len(rep), 1e-9) * 1e3, 1),
            "packed_encoder": {"buckets_packed": packed, "row_bucket": gstep.row_bucket,
                               "what": "encoder stack on the live sub-sampled positions only (js2t_pack_rows, js2t_attn_desc.seg); JS2T_PACKED_ENCODER=0: padded"},
            "utterances_per_batch": round(stats["nseqs"] / steps, 2), "buckets": len(gstep.buckets), "buckets_captured_ahead": n_pre,
            "timed_steps_replayed": gstep.counts["replay"] - before["replay"], "timed_steps_eager_plus_capture": gstep.counts["eager"] - before["eager"],
            # host time inside run() per step: a graph captured ahead is uploaded by its FIRST launch (once per bucket and process)
            "host_ms_in_run": {"first_launch_of_a_graph": {"steps": len(first_ms), "median": round(first_ms[len(first_ms) // 2], 2) if first_ms else None,
                                                           "sum": round(sum(first_ms), 1)},
                               "later_launches": {"steps": len(later_ms), "median": round(later_ms[len(later_ms) // 2], 2) if later_ms else None,
                                                  "max": round(later_ms[-1], 2) if later_ms else None}},
            "batch_multiplier": step.batch_multiplier, "updates": step.steps, "capture_errors": gstep.capture_errors[:2] or None,
            "loss": round(stats["loss"] / steps, 4), "launch": "hipGraph per (B, frames/64, target length/8, packed rows/384) bucket" if use_graphs else "eager",
            "what": "LS100 train step, a NEW batch every step: TokenBatchSampler over a shuffled corpus of 10-17 s utterances, PrefetchLoader "
                    "(pinned rows -> HBM one batch ahead), graphed.GraphedTrainStep"}


def train_step_flop(cfg, vocab, batch, t_sub, l_trg):
    """Algorithmic FLOP of one S2T train micro-batch (3 x forward: backward = input + weight gradients; the convention of
    roofline.conformer_train_step): per token and layer 2 x (projections + feed-forward) weights + attention 4 T d per side,
    sub-sampler 41 GFLOP per 32 x 15 s, both output layers."""
    e, dc = cfg["encoder"], cfg["decoder"]
    d, ff = e["hidden_size"], e["ff_size"]
    ne, nd = batch * t_sub, batch * l_trg
    enc = e["num_layers"] * ne * (2 * (4 * d * d + 2 * d * ff) + 4 * t_sub * d) + 41e9 * (batch / 32)
    dd, dff = dc["hidden_size"], dc["ff_size"]
    dec = dc["num_layers"] * (nd * (2 * (4 * dd * dd + 2 * dd * dd + 2 * dd * dff) + 4 * l_trg * dd + 4 * t_sub * dd) + ne * 2 * 2 * dd * dd)
    return 3.0 * (enc + dec + 2 * nd * dd * vocab + 2 * ne * d * vocab)


def config_faithful_bench(device, world, k=4, updates=5, dtype=torch.bfloat16, seed=42, merged=True, model_cfg=None, vocab=None,
                          loss=("crossentropy-ctc", 0.1, 0.3), what=None):
    """The train step AS THE CONFIG WRITES IT: `batch_multiplier: 4` (configs/librispeech_100h.yaml:85; loop at joeynmt/training.py:
    416-456) - four micro-batches of 32 x 15 s per optimizer update - through graphed.GraphedTrainStep: one capture per phase of the
    accumulation (first / middle / last micro-batch), and under a process group the last one cut where the collectives go (one
    exchange per update, on the summed gradient).  `merged`: the same update in ONE pass over the 128 utterances
    (batch_multiplier 1, batch 128: what 288 GB of HBM allow; identical gradient up to summation order, normalisation 'batch')."""
    import copy
    from joeys2t_amd.graphed import GraphedTrainStep
    from joeys2t_amd.model import build_model
    from joeys2t_amd.tokenizers import SpeechProcessor
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    rank = int(os.environ.get("RANK", 0))
    ddp = torch.distributed.is_initialized() and world > 1
    model_cfg = LS100_MODEL if model_cfg is None else model_cfg
    vocab = VOCAB if vocab is None else vocab

    def build(batch, mult):
        torch.manual_seed(seed)
        np.random.seed(seed)
        model = build_model(copy.deepcopy(model_cfg), None, Vocabulary.synthetic(vocab))
        model.loss_function = loss
        model.finalize(device, dtype, seed=seed)
        step = TrainStep(model, learning_rate=2.0e-3, adam_betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=10.0, learning_rate_warmup=10000,
                         learning_rate_min=1.0e-6, normalization="batch", batch_multiplier=mult, n_gpu=1,
                         overlap_ctc=(not ddp or torch.distributed.get_backend() == "nccl"))
        proc = SpeechProcessor(num_freq=80, min_length=10, max_length=6000,
                               specaugment=dict(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=100, time_mask_p=1.0),
                               cmvn=dict(norm_means=True, norm_vars=True, before=True))
        gs = GraphedTrainStep(step, proc, compute_dtype=dtype)
        wave = synth_waveforms(batch, SAMPLES, seed=1234 + rank).to(device)
        trg, trg_len = synth_targets(batch, vocab, seed=1234 + rank)
        return gs, step, (wave, [SAMPLES] * batch, trg.cpu(), trg_len.cpu())

    def timed(gs, item, mult, n_updates):
        for _ in range(3 * mult):  # first sights (one eager micro-batch + capture per phase), then replays
            gs.run(*item)
        if ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        before = dict(gs.counts)
        t0 = time.perf_counter()
        for _ in range(n_updates * mult):
            gs.run(*item)
        if ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if ddp:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, {k_: gs.counts[k_] - before.get(k_, 0) for k_ in ("eager", "replay")}

    frames_mb = BATCH * (1 + (SAMPLES - 400) // 160)
    gs, step, item = build(BATCH, k)
    dt, how = timed(gs, item, k, updates)
    key = next(iter(gs.buckets))  # (B, frame bucket, target bucket, packed rows): the shapes the kernels ran on
    flop = train_step_flop(model_cfg, vocab, BATCH, gs._sub_len(key[1]), key[2] - 1)
    ms_mb = dt / (updates * k) * 1e3
    res = {"batch_multiplier": k, "updates_timed": updates, "ms_per_micro_batch": round(ms_mb, 3),
           "ms_per_update": round(dt / updates * 1e3, 3), "frames_per_s": round(world * frames_mb * k * updates / dt, 1),
           "flop_per_micro_batch": flop, "achieved_tflops": round(flop / (ms_mb * 1e-3) / 1e12, 1),
           "frac_of_bf16_peak": round(flop / (ms_mb * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
           "micro_batches": how, "captures": {"buckets": len(gs.buckets), "phases": sorted(str(ph) for bk in gs.buckets.values() for ph in bk.graphs)},
           "capture_errors": gs.capture_errors[:2] or None,
           "launch": "graphed.GraphedTrainStep: a hipGraph per phase of the accumulation (first / middle / last micro-batch)" +
                     (", the last one cut where the collectives go" if ddp else ", the update inside the last one"),
           "what": what or f"configs/librispeech_100h.yaml as written: {k} micro-batches of 32 x 15 s per update (training.py:436-456)"}
    del gs, step, item
    torch.cuda.empty_cache()
    if merged and not ddp:
        try:
            gs, step, item = build(BATCH * k, 1)
            dt, how = timed(gs, item, 1, updates)
            res["same_update_in_one_pass"] = {"batch": BATCH * k, "batch_multiplier": 1, "ms_per_update": round(dt / updates * 1e3, 3),
                                              "frames_per_s": round(frames_mb * k * updates / dt, 1),
                                              "what": "the update's 128 utterances as ONE micro-batch (288 GB of HBM: nothing has to be accumulated in slices); "
                                                      "same gradient up to summation order - a side figure, not the config's procedure"}
            del gs, step, item
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            res["same_update_in_one_pass"] = {"error": repr(exc)[:300]}
    return res


def train_step_by_config(device):
    """The other BASELINE configs' train steps, as their files write them, on 32 x 15 s micro-batches:
    configs/mustc_st.yaml (12 + 6 layers, 8 heads of 64 - the head-size-64 attention kernels -, xavier_normal, ctc_weight 0.1,
    batch_multiplier 8) and configs/librispeech_960h.yaml (16 + 8 layers, V = 10000, ctc_weight 0.3, batch_multiplier 8), each through
    graphed.GraphedTrainStep; ms per micro-batch, frames/s and the fraction of the bf16 peak by train_step_flop's count."""
    import copy
    out = {}
    ls960 = copy.deepcopy(LS100_MODEL)  # librispeech_960h.yaml:108-137 is the LS100 stack; what differs is the vocabulary (:39) and the loop (:85)
    for name, kw in (("mustc_st", dict(model_cfg=MUSTC_MODEL, vocab=5000, loss=("crossentropy-ctc", 0.1, 0.1), k=8,
                                       what="configs/mustc_st.yaml:83-142 as written: 8 micro-batches of 32 x 15 s per update, 12 + 6 layers, 8 heads of 64")),
                     ("librispeech_960h", dict(model_cfg=ls960, vocab=10000, loss=("crossentropy-ctc", 0.1, 0.3), k=8,
                                               what="configs/librispeech_960h.yaml:39,85,98 as written: V = 10000, 8 micro-batches of 32 x 15 s per update"))):
        try:
            out[name] = config_faithful_bench(device, 1, updates=3, merged=False, **kw)
        except Exception as exc:  # noqa: BLE001 - side figures
            out[name] = {"error": repr(exc)[:300]}
        torch.cuda.empty_cache()
    return out


def measure_roofline(eager_step, model):
    """HIP-event timing of every GEMM launch of two eager train steps on ONE GPU (a spin kernel holds the GPU while the host
    enqueues the step, so the timed kernels run back to back) -> the `roofline` object of the JSON line."""
    from joeys2t_amd import ops
    timer = GemmTimer()
    # The eager pass is launch-bound on the host (~40 us of Python per kernel): an event pair around a launch
    # would also time the GPU waiting for the next packet.  A spin kernel in front of each step holds the GPU
    # while the host enqueues the whole step, so the timed kernels then run back to back from a full queue.
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    torch.cuda._sleep(10_000_000)
    s1.record()
    torch.cuda.synchronize()
    cycles_per_ms = 10_000_000 / max(s0.elapsed_time(s1), 1e-3)
    ops.GEMM_TIMER = timer
    overlap_ctc, model.overlap_ctc = model.overlap_ctc, False  # one stream: every launch is timed alone on the GPU
    for _ in range(2):
        torch.cuda._sleep(int(80 * cycles_per_ms))
        eager_step()
        torch.cuda.synchronize()
    model.overlap_ctc = overlap_ctc
    ops.GEMM_TIMER = None
    # an event pair with nothing between its records still reads a few microseconds (the two timestamp packets):
    # measured here the same way and taken off every timed launch
    torch.cuda._sleep(int(20 * cycles_per_ms))
    empty = GemmTimer()
    for _ in range(256):
        empty.wrap("empty", 0.0, lambda: None)
    pair_overhead = empty.summary()["empty"][2] / 256
    agg = timer.summary()
    for v in agg.values():
        v[2] = max(v[2] - v[0] * pair_overhead, 1e-9)
    # dominant kernel = the bf16 LDS-DMA MFMA GEMM family: the persistent 192x128 / panel kernels (forward and input-gradient
    # products of the encoder-sized layers), the 128/64-row tile kernel (everything else) and the grouped launches (the deferred
    # weight gradients: 256x128 or 128x128 tiles); per-kernel figures are listed beside it (labels are the host's: which kernel a
    # js2t_gemm / js2t_gemm_grouped call takes is the library's choice)
    fam = {k: v for k, v in agg.items() if k.startswith(("gemm_bf16_dma_", "gemm_bf16_p192_", "js2t_gemm_grouped"))}
    n = sum(v[0] for v in fam.values())
    flops = sum(v[1] for v in fam.values())
    secs = sum(v[2] for v in fam.values())
    achieved = flops / secs / 1e12
    algo_bytes = sum(v[3] for v in fam.values()) / n
    # HBM bytes per launch from the PMC passes of tools/profile_round.sh (FETCH_SIZE x2 + WRITE_SIZE): a committed
    # measurement, stamped with the kernel source it was taken on - null as soon as csrc/gemm.hip has changed since
    traffic, traffic_src = None, None
    tfile = ROOT / "profiles" / "gemm_traffic.json"
    if tfile.exists():
        tj = json.loads(tfile.read_text())
        import hashlib
        sha = hashlib.sha256((ROOT / "joeys2t_amd" / "csrc" / "gemm.hip").read_bytes()).hexdigest()[:16]
        traffic_src = {"file": "profiles/gemm_traffic.json", "measured_at_commit": tj.get("measured_at_commit"),
                       "measured_on": tj.get("measured_on"), "gemm_hip_sha16": tj.get("gemm_hip_sha16"),
                       "current_gemm_hip_sha16": sha, "stale": tj.get("gemm_hip_sha16") != sha}
        if not traffic_src["stale"]:
            traffic = tj.get("hbm_bytes_per_launch")
    key = ("gemm_bf16_p192_kernel<*> + gemm_bf16_p192s_kernel<*> + gemm_bf16_pan96_kernel<*> + gemm_bf16_dma_kernel<*> + "
           "gemm_bf16_wg256_kernel<*> + gemm_bf16_dma_grouped_kernel<*>")
    roofline = {"bound": "mfma", "kernel": key, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(algo_bytes),
                "flop_per_launch": round(flops / n), "launches_per_step": n // 2,
                "avg_launch_us": round(secs / n * 1e6, 2), "event_pair_overhead_us": round(pair_overhead * 1e6, 2),
                "all_gemm_kernels": {k: {"launches_per_step": v[0] // 2, "tflops": round(v[1] / v[2] / 1e12, 2),
                                         "ms_per_step": round(v[2] / 2 * 1e3, 3)} for k, v in agg.items()}}
    return roofline


def encoder_forward(model, batch, reps=20):
    """BASELINE.json's stated target is a fraction of the bf16 MFMA peak on the ENCODER FORWARD: time it alone (subsampler +
    16 layers + final LayerNorm, train mode / dropout on, hipGraph replay) and price it with SURVEY 8(d)'s count:
    7.06 MFLOP per token and layer x 16 layers x 12000 tokens = 1.355 TFLOP (+ 41 GFLOP of subsampler convolutions)."""
    model.train()
    kw = vars(batch)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model(return_type="encode", **kw)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            model(return_type="encode", **kw)
        g.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    flops = 16 * 12000 * 7.06e6 * (BATCH / 32) + 41e9 * (BATCH / 32)
    tf = flops / (ms * 1e-3) / 1e12
    return {"ms": round(ms, 3), "flop": flops, "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
            "what": "subsampler + 16 encoder layers + final LN, forward only, bf16, dropout on"}


PEAK_FP8_TFLOPS = 5000.0  # dense fp8 MFMA peak (MI355X_MICROARCH.md); the kernel uses the non-scaled K = 32 form, which issues at the bf16 rate


def conformer_fp8_forward(device, reps=10, modes=("fp8", "bf16")):
    """BASELINE.json configs[4]: Conformer-style encoder (relative-position attention + depthwise convolution module) on
    librispeech_960h shapes with fp8 MFMA - encoder forward on 32 x 15 s of synthetic features, train mode (dropout on),
    hipGraph replay; timed with e4m3 forward products (functional.FP8_FORWARD) and, beside it, in plain bf16.
    An EXTENSION: the reference's ConformerEncoder has no relative-position term and no fp8 path, so there is no parity
    target; the Conformer layers themselves are pinned by tests/golden/conformer.npz, the fp8 product and the bias by
    tests/test_hip_ops.py.  FLOP per token and layer: two feed-forward modules 2 x 4 d ff, attention 8 d^2 + 4 T' d,
    convolution module 2 d (2 d) + 2 d^2 + 2 x 31 d."""
    from joeys2t_amd import functional as Fn
    from joeys2t_amd.encoders import ConformerEncoder
    from joeys2t_amd.runtime import ParamStore, Runtime, install_runtime
    torch.manual_seed(42)
    d, ff, layers, heads, k = 512, 2048, 16, 4, 31
    enc = ConformerEncoder(hidden_size=d, ff_size=ff, num_layers=layers, num_heads=heads, dropout=0.1, emb_dropout=0.1, in_channels=80,
                           conv_channels=512, conv_kernel_sizes=[5, 5], depthwise_conv_kernel_size=k, alpha=1.0, layer_norm="pre",
                           rel_pos_clip=64)
    with torch.no_grad():
        for layer in enc.layers:
            layer.src_src_att.rel_pos_bias.normal_(0.0, 0.1)
    enc.to(device)
    rt = Runtime(device, torch.bfloat16)
    rt.store = ParamStore(enc, device)
    rt.rng.seed(42)
    install_runtime(enc, rt)
    rt.store.refresh(force=True)
    enc.train()
    frames = 1 + (SAMPLES - 400) // 160
    src = torch.randn(BATCH, frames, 80, device=device).bfloat16()
    lengths = torch.full((BATCH, ), frames, device=device)
    tp = ((frames - 1) // 2) // 2 + 1
    tokens = BATCH * tp
    flop = layers * tokens * (2 * 4 * d * ff + 8 * d * d + 4 * tp * d + 4 * d * d + 2 * d * d + 2 * k * d) + 2 * tokens * d * d + 41e9 * (BATCH / 32)
    out = {}
    for mode in modes:
        Fn.FP8_FORWARD = mode == "fp8"
        try:
            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        enc(src, lengths, None)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    enc(src, lengths, None)
                g.replay()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                s.record()
                for _ in range(reps):
                    g.replay()
                e.record()
                torch.cuda.synchronize()
            ms = s.elapsed_time(e) / reps
            out[mode] = {"ms": round(ms, 3), "achieved": round(flop / (ms * 1e-3) / 1e12, 1)}
        finally:
            Fn.FP8_FORWARD = False
    peak = PEAK_FP8_TFLOPS
    return {"what": "Conformer encoder forward (16 layers, d 512, ff 2048, 4 heads, depthwise k 31, rel-pos clip 64), 32 x 15 s, "
                    "dropout on; e4m3 forward products of the nn.Linear layers, everything else bf16",
            "flop": flop, "unit": "TFLOP/s", "fp8": out.get("fp8"), "bf16": out.get("bf16"), "peak": peak,
            "frac": round(out["fp8"]["achieved"] / peak, 4) if "fp8" in out else None,
            "parity": "extension: no reference target (joeynmt has neither rel-pos attention nor fp8)"}


PEAK_HBM_TBS = 8.0  # HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def measure_hbm_kernels(eager_step, step):
    """The HBM-bound kernels of the step against the 8 TB/s roofline (north_star: "achieved HBM GB/s ... against peak"): HIP events
    around every call of their Python entry points in two back-logged eager steps (the GEMM family's method: a spin kernel holds
    the GPU while the host enqueues), ALGORITHMIC bytes = each distinct tensor a call reads or writes, once (SURVEY 8(d):
    LayerNorm 2 passes over the operand per output, losses one read of the logits, CTC one read + one write, the front-end
    1.44 MB per 15 s utterance, the update 38 bytes per parameter minus what the kept gradients save)."""
    from joeys2t_amd import builders, ops
    from joeys2t_amd.tokenizers import SpeechProcessor
    records = []  # (name, bytes, start, end)

    def tensors_of(x, acc):
        if torch.is_tensor(x):
            if x.is_cuda and x.numel() > 0:
                acc[x.data_ptr()] = max(acc.get(x.data_ptr(), 0), x.numel() * x.element_size())
        elif isinstance(x, (list, tuple)):
            for y in x:
                tensors_of(y, acc)
        elif isinstance(x, dict):
            for y in x.values():
                tensors_of(y, acc)

    def timed(name, fn, fixed_bytes=None):
        def wrapper(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            s.record(st)
            out = fn(*a, **k)
            e.record(st)
            if fixed_bytes is None:
                acc = {}
                tensors_of(a, acc), tensors_of(k, acc), tensors_of(out, acc)
                nbytes = sum(acc.values())
            else:
                nbytes = fixed_bytes(*a, **k)
            records.append((name, nbytes, s, e))
            return out
        return wrapper

    opt, st = step.optimizer, step.store

    def update_bytes(*a, **k):
        kept = sum(hi - lo for lo, hi in (opt.keep.r if opt.keep is not None else []))
        n = st.total
        # read p, g, m, v; write p, m, v; bf16 shadow + transposed shadow; clear the gradients that are not overwritten;
        # the norm's pass over the gradients whose sums the weight-gradient epilogues did not leave behind
        lp = 2 * n if st.flat_lp is not None else 0
        lpt = 2 * sum(R * Cc for _, R, Cc, _ in st._tgroups) if st.flat_lp_t is not None else 0
        covered = sum(hi - lo for lo, hi in (opt.collector.covered if opt.collector is not None else []))
        return 28 * n + lp + lpt + 4 * (n - kept) + 4 * (n - covered)

    def frontend_bytes(self, wave, n_samples, *a, **k):  # one read of the waveform, one write of the f32 features
        return sum(4 * int(n) + 4 * 80 * (1 + (int(n) - 400) // 160) for n in n_samples)

    patches = [(ops, "layernorm_bwd", None), (ops, "layernorm_fwd", None), (ops, "xent_fwd", None), (ops, "xent_bwd", None),
               (ops, "row_lse", None), (ops, "ctc_alpha", None), (ops, "ctc_bwd", None), (ops, "embed_fwd", None),
               (SpeechProcessor, "batch_from_waveforms", frontend_bytes), (builders.FlatAdamW, "clip_and_step", update_bytes)]
    names = {"batch_from_waveforms": "front_end (fbank + cmvn_stats + feature_finalize)", "clip_and_step": "update (grad norm + AdamW + shadows + LayerNorm folds)",
             "ctc_alpha": "ctc_alpha_beta", "ctc_bwd": "ctc_grad"}
    saved = [(o, n, getattr(o, n)) for o, n, _ in patches]
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    torch.cuda._sleep(10_000_000)
    s1.record()
    torch.cuda.synchronize()
    cycles_per_ms = 10_000_000 / max(s0.elapsed_time(s1), 1e-3)
    model = step.model
    overlap_ctc, model.overlap_ctc = model.overlap_ctc, False  # one stream: every launch is timed alone on the GPU
    try:
        for (o, n, fb), (_, _, fn) in zip(patches, saved):
            setattr(o, n, timed(names.get(n, n), fn, fb))
        for _ in range(2):
            torch.cuda._sleep(int(80 * cycles_per_ms))
            eager_step()
            torch.cuda.synchronize()
    finally:
        for o, n, fn in saved:
            setattr(o, n, fn)
        model.overlap_ctc = overlap_ctc
    torch.cuda._sleep(int(20 * cycles_per_ms))
    empty = GemmTimer()
    for _ in range(256):
        empty.wrap("empty", 0.0, lambda: None)
    pair = empty.summary()["empty"][2] / 256
    torch.cuda.synchronize()
    agg = {}
    for name, nbytes, s, e in records:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += nbytes
        a[2] += max(s.elapsed_time(e) * 1e-3 - pair, 1e-9)
    out = {}
    for name, (n, nbytes, secs) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        tbs = nbytes / secs / 1e12
        out[name] = {"launches_per_step": n // 2, "algorithmic_MB_per_step": round(nbytes / 2 / 1e6, 1), "us_per_step": round(secs / 2 * 1e6, 1),
                     "achieved_TBps": round(tbs, 2), "frac": round(tbs / PEAK_HBM_TBS, 3)}
    return {"bound": "hbm", "peak": PEAK_HBM_TBS, "unit": "TB/s", "event_pair_overhead_us": round(pair * 1e6, 2), "kernels": out}


def conformer_train_step(device, reps=10, modes=("bf16", )):
    """BASELINE.json configs[4] as a TRAIN step: build_model(encoder.type: conformer, rel_pos_clip 64, depthwise kernel 31; 16 + 6
    layers, d 512, V 10000 as librispeech_960h) -> TrainStep (forward, CTC + label-smoothed CE, backward, clip + AdamW), 32 x 15 s
    of synthetic features, dropout on, hipGraph replay; in bf16 and with e4m3 forward products (functional.FP8_FORWARD; backward
    stays bf16).  EXTENSION: no reference parity target for the composition (model.py:417-421 refuses the encoder type); parity of
    the pieces and of the composed loss / gradients against the oracle: tests/test_hip_config5_train.py."""
    import copy
    from joeys2t_amd import functional as Fn
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.model import build_model
    from joeys2t_amd.training import TrainStep
    from joeys2t_amd.vocabulary import Vocabulary
    cfg = copy.deepcopy(LS100_MODEL)
    cfg["encoder"].update(type="conformer", depthwise_conv_kernel_size=31, rel_pos_clip=64)
    V = 10000
    frames = 1 + (SAMPLES - 400) // 160
    tp = ((frames - 1) // 2) // 2 + 1
    tokens = BATCH * tp
    d, ff, k, layers = 512, 2048, 31, cfg["encoder"]["num_layers"]
    trg, trg_len = synth_targets(BATCH, V, seed=99)
    L = trg.shape[1] - 1
    enc_fwd = layers * tokens * (2 * 4 * d * ff + 8 * d * d + 4 * tp * d + 4 * d * d + 2 * d * d + 2 * k * d) + 2 * tokens * d * d + 41e9 * (BATCH / 32)
    dec_fwd = cfg["decoder"]["num_layers"] * (BATCH * L * (8 * d * d + 4 * L * d + 4 * d * d + 4 * tp * d + 4 * d * ff) + 4 * d * d * tokens) + 2 * BATCH * L * d * V
    flop = 3 * (enc_fwd + dec_fwd + 2 * tokens * d * V)
    out = {}
    for mode in modes:
        Fn.FP8_FORWARD = mode == "fp8"
        try:
            torch.manual_seed(42)
            model = build_model(copy.deepcopy(cfg), None, Vocabulary.synthetic(V))
            model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
            with torch.no_grad():
                for layer in model.encoder.layers:
                    layer.src_src_att.rel_pos_bias.normal_(0.0, 0.1)
            model.finalize(device, torch.bfloat16, seed=42)
            # the rate and warm-up of the 1000-update soak runs (tools/conformer_soak.py, profiles/r05_conformer_soak_*.txt): the loss
            # falls steadily there.  The encoder has no final LayerNorm (encoders.py:376-445), so the CTC head starts on un-normalised
            # activations and the initial CTC term is ~30 x the uniform level: with the gradient clipped to norm 1 the first updates
            # are sign-like and two runs that differ in the last bit of one atomic sum drift apart within tens of updates - the spread
            # of same-seed mean losses round 4 reported.  js2t_set_deterministic now covers the extension kernels: same-seed runs agree
            # bit for bit (tests/test_hip_config5_train.py).  Reported: the loss in front of and behind the timed window, not a mean.
            step = TrainStep(model, learning_rate=1.0e-4, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=200,
                             normalization="batch", overlap_ctc=True)
            step.optimizer.device_schedule = True
            src = torch.randn(BATCH, frames, 80, device=device).bfloat16()
            batch = Batch(src=src, src_length=torch.full((BATCH, ), frames, device=device), src_prompt_mask=None, trg=trg, trg_length=trg_len,
                          trg_prompt_mask=None, indices=torch.arange(BATCH), device=device, pad_index=1, eos_index=3, is_train=True,
                          task="S2T", n_gpu=1)

            def body():
                return step.micro_step(batch, sort=False, update=True, overlap=False)

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            for _ in range(2):
                g.replay()
            step.read_stats(reset=True)
            g.replay()
            loss_first = step.read_stats()["loss"]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(reps):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / reps
            step.read_stats(reset=True)
            g.replay()
            loss_last = step.read_stats()["loss"]
            out[mode] = {"ms_per_step": round(ms, 3), "frames_per_s": round(BATCH * frames / (ms * 1e-3), 1),
                         "achieved": round(flop / (ms * 1e-3) / 1e12, 1), "loss_before_timed_window": round(loss_first, 3),
                         "loss_after_timed_window": round(loss_last, 3), "updates_between": reps + 1}
            del g, step, model
            torch.cuda.empty_cache()
        finally:
            Fn.FP8_FORWARD = False
    return {"what": "Conformer S2T train step (16 Conformer + 6 Transformer decoder layers, d 512, ff 2048, depthwise k 31, rel-pos clip 64, "
                    "V 10000, CTC 0.3 + label-smoothed CE), 32 x 15 s, dropout 0.1, clip + AdamW, hipGraph replay",
            "flop_per_step": flop, "unit": "TFLOP/s", "bf16": out.get("bf16"), "fp8": out.get("fp8"), "peak_bf16": PEAK_BF16_TFLOPS,
            "frac_bf16": round(out["bf16"]["achieved"] / PEAK_BF16_TFLOPS, 4) if "bf16" in out else None,
            "fp8_note": "e4m3 forward products are reported as INFERENCE-only (roofline.conformer_fp8_forward): as a train step they save less "
                        "than keeping both operand copies for the bf16 backward costs (round 4: 27.3 against 26.1 ms); "
                        "conformer_train_step(modes=('bf16', 'fp8')) still measures both, tools/conformer_soak.py trains both",
            "soak": "profiles/r05_conformer_soak_bf16.txt, _fp8.txt: 1000 updates at peak rate 1e-4 (200 warm-up): total loss 44.8 k -> 1.8 k; at 3e-4 the CTC term diverges after ~250 updates (profiles/r05_conformer_soak_bf16_lr3e-4_diverges.txt)",
            "parity": "extension: no reference target for the composition (joeynmt's build_model refuses the encoder type)"}


def host_cpu():
    """(threads this process may use, CPU model string, CPUs online) of the host the baseline is timed on"""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # a container's CPU share is a cgroup quota, not an affinity mask: more threads than that only fight each other
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                usable = max(1, min(usable, int(-(-float(quota) // period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return usable, model, os.cpu_count() or usable


def cpu_baseline(n_utts=BATCH, threads=None, warmup_steps=3, timed_steps=10, gpu_decode=None):
    """The CPU oracle (plain fp32 PyTorch/NumPy restatement of the reference path) timed on the host cores on the SAME
    workload as the GPU line - one batch of 32 utterances of 15 s through fbank -> ... -> loss -> backward -> clip -> AdamW
     - with every CPU this process may use (count and model string stated): 3 warm-up + 10 timed
    steps (BASELINE.md's plan; ~6 s each on the 16-thread share of the GPU box), cut to 1 + 2 on a host whose first step
    takes more than 10 s.  Plus (ii) of BASELINE.md section 3: beam-5 decode RTF of the oracle's search (full-prefix decoder
    pass, no KV cache, exactly the reference's algorithm) on the GPU leg's workload (cpu_decode_rtf)."""
    import copy
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    from oracle import s2t_oracle as O
    usable, cpu_model, online = host_cpu()
    threads = threads or usable
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    cfg = copy.deepcopy(LS100_MODEL)
    cfg["encoder"]["alpha"] = cfg["decoder"]["alpha"] = 1.0
    model = build_model(copy.deepcopy(LS100_MODEL), None, Vocabulary.synthetic(VOCAB))  # parameters only (CPU tensors)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "pe.pe" not in k) for k, v in model.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-3, betas=(0.9, 0.98), weight_decay=0.0)
    wave = synth_waveforms(n_utts, SAMPLES).numpy()
    trg, trg_len = synth_targets(n_utts, VOCAB)
    specials = dict(unk=0, pad=1, bos=2, eos=3)
    rs = np.random.RandomState(42)

    def step():
        feats = []
        for u in range(n_utts):
            f = O.cmvn(O.fbank(wave[u]))
            f = O.specaugment_apply(f, O.specaugment_params(f.shape[0], 80, rs, time_mask_t=100))
            feats.append(f.astype(np.float32))
        padded, lengths, _ = O.pad_features(feats)
        b = O.make_batch(torch.from_numpy(padded), torch.tensor(lengths), trg, trg_len, 1, 3)
        total, *_ = O.model_loss(sd, cfg, b, specials, 0.1, 0.3)
        opt.zero_grad()
        (total / n_utts).backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()

    t0 = time.perf_counter()
    step()  # first warm-up step
    warm = time.perf_counter() - t0
    print(f"[bench] cpu baseline: {threads} threads on {cpu_model}; first step {warm:.1f} s", file=sys.stderr, flush=True)
    if warm > 10.0:  # a slow or over-subscribed host: keep the sample bounded
        warmup_steps, timed_steps = 1, 2
    for _ in range(warmup_steps - 1):
        step()
    times = []
    for i in range(timed_steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu baseline: timed step {i + 1}/{timed_steps}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    dt = float(np.median(times))  # report the median
    frames = n_utts * (1 + (SAMPLES - 400) // 160)
    out = {"value": round(frames / dt, 1), "unit": "frames/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
           "cpus_online": online, "s_per_step": round(dt, 3), "s_per_step_min_max": [round(min(times), 3), round(max(times), 3)],
           "sample": f"{n_utts} utterances x 15 s (the GPU line's batch), LS100 model, full train step (fbank..AdamW), fp32, "
                     f"{warmup_steps} warm-up + {timed_steps} timed steps, median"}
    try:
        out["decode_beam5"] = cpu_decode_rtf()
    except Exception as exc:  # a side figure of a side figure
        out["decode_beam5"] = {"error": repr(exc)}
    return out


def cpu_decode_rtf(n_utts=BATCH, beam=5, alpha=1.0, max_len=100, budget_s=60.0):
    """beam-5 decode of the CPU oracle (search.py:345-825 restated: full-prefix decoder pass per
    step, encoder states tiled beam-fold) on mustc_st.yaml shapes - the GPU leg's workload: 32 utterances x 15 s, 100 steps.
    Time-boxed: a 2-utterance, 8-step probe prices an (utterance, step) pair first; if the full workload would take longer than
    `budget_s`, fewer utterances are decoded (all 100 steps kept) and `n_utts` says how many - bench.py then decodes the same
    sub-sample on the GPU as well (`decode_beam5.same_sample_as_cpu`), so the two RTFs always describe one workload."""
    import copy
    from joeys2t_amd.model import build_model
    from joeys2t_amd.vocabulary import Vocabulary
    from oracle import s2t_oracle as O
    torch.manual_seed(42)
    cfg = copy.deepcopy(MUSTC_MODEL)
    n, m = cfg["encoder"]["num_layers"], cfg["decoder"]["num_layers"]
    cfg["encoder"]["alpha"], cfg["decoder"]["alpha"] = 0.81 * (n**4 * m)**(1 / 16), (3 * m)**(1 / 4)
    model = build_model(copy.deepcopy(MUSTC_MODEL), None, Vocabulary.synthetic(VOCAB))
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    specials = dict(unk=0, pad=1, bos=2, eos=3)

    def decode(nu, steps):
        wave = synth_waveforms(nu, SAMPLES).numpy()
        t0 = time.perf_counter()
        with torch.no_grad():
            feats = [O.cmvn(O.fbank(wave[u])).astype(np.float32) for u in range(nu)]
            padded, lengths, _ = O.pad_features(feats)
            enc, mask, _ = O.encoder_forward(sd, cfg, torch.from_numpy(padded), torch.tensor(lengths))
            ids, _ = O.beam_search(sd, cfg, specials, enc, mask, beam, steps, alpha, n_best=1)
        return time.perf_counter() - t0, int(ids.shape[1])

    probe_s, _ = decode(2, 8)
    per_pair = probe_s / (2 * 8)
    # the full-prefix pass grows with the prefix: step t re-runs t + 1 positions, but the re-projected encoder states (375
    # positions per hypothesis and layer) dominate, so an (utterance, step) pair at step 100 costs ~1.3 x one at step 8
    est_full = 1.3 * per_pair * n_utts * max_len
    if est_full > budget_s:
        n_utts = max(1, min(n_utts, int(budget_s / (1.3 * per_pair * max_len))))
    print(f"[bench] cpu decode: probe {probe_s:.1f} s, full workload estimated {est_full:.0f} s -> decoding {n_utts} utterances", file=sys.stderr, flush=True)
    dt, steps = decode(n_utts, max_len)
    audio_s = n_utts * SAMPLES / 16000.0
    return {"rtf": round(dt / audio_s, 5), "wall_s": round(dt, 2), "audio_s": audio_s, "beam": beam, "steps": steps, "n_utts": n_utts,
            "sample": f"{n_utts} utterances x 15 s, beam {beam}, at most {max_len} steps" + (" (the GPU leg's workload)" if n_utts == BATCH else
                      f" (time-boxed to ~{budget_s:.0f} s; the GPU leg reports the same sub-sample under same_sample_as_cpu)"),
            "decoding": "reference algorithm: full-prefix decoder pass per step, no KV cache"}


MUSTC_MODEL = {
    "initializer": "xavier_normal", "init_gain": 1.0, "bias_initializer": "zeros", "embed_initializer": "xavier_normal",
    "embed_init_gain": 1.0, "tied_embeddings": False, "tied_softmax": False,
    "encoder": {"type": "transformer", "num_layers": 12, "num_heads": 8, "embeddings": {"embedding_dim": 80},
                "hidden_size": 512, "ff_size": 2048, "dropout": 0.1, "freeze": False, "subsample": True,
                "conv_kernel_sizes": [5, 5], "conv_channels": 512, "in_channels": 80, "layer_norm": "pre", "activation": "relu"},
    "decoder": {"type": "transformer", "num_layers": 6, "num_heads": 8,
                "embeddings": {"embedding_dim": 512, "scale": True, "dropout": 0.0, "freeze": False}, "hidden_size": 512,
                "ff_size": 2048, "dropout": 0.1, "freeze": False, "layer_norm": "pre", "activation": "relu"},
}


def decode_rtf(device, dtype=torch.bfloat16, beam=5, alpha=1.0, max_len=100, eos_scale=0.0, n_utts=BATCH, ctc_weight=0.0, return_ids=False):
    """Second half of BASELINE.json's metric: beam-5 decode real-time factor on configs/mustc_st.yaml shapes
    (12+6 layers, H=8, beam 5, alpha 1.0, max_output_length 100), 32 synthetic 15 s utterances resident in HBM.
    RTF = wall time of front-end + encode + beam search / seconds of audio."""
    import copy
    from joeys2t_amd.batch import Batch
    from joeys2t_amd.model import build_model
    from joeys2t_amd.search import search
    from joeys2t_amd.tokenizers import SpeechProcessor
    from joeys2t_amd.vocabulary import Vocabulary
    torch.manual_seed(42)
    model = build_model(copy.deepcopy(MUSTC_MODEL), None, Vocabulary.synthetic(VOCAB))
    if eos_scale > 0:
        # a random-init model never emits EOS: every hypothesis runs the full max_output_length, the best case for the replayed
        # step and the only case in which nothing finishes, nothing is compacted and the host never looks at the device.  A
        # larger EOS row in the output layer makes EOS win with a few per cent probability per step, so hypotheses finish at
        # scattered steps (geometric lengths) as they do for a trained model - the bookkeeping of search.py:640-700 then runs.
        with torch.no_grad():
            w = model.decoder.output_layer.weight
            w[3] = torch.nn.functional.normalize(torch.randn(w.shape[1]), dim=0) * eos_scale
    model.finalize(device, dtype).eval()
    proc = SpeechProcessor(num_freq=80, min_length=10, max_length=5000, cmvn=dict(norm_means=True, norm_vars=True, before=True))
    wave = synth_waveforms(n_utts, SAMPLES).to(device)

    def run():
        feats, lengths = proc.batch_from_waveforms(wave, [SAMPLES] * n_utts, is_train=False, out_dtype=dtype)
        b = Batch(src=feats, src_length=torch.tensor(lengths, device=device), src_prompt_mask=None, trg=None, trg_length=None,
                  trg_prompt_mask=None, indices=torch.arange(n_utts), device=device, pad_index=1, eos_index=3, is_train=False,
                  task="S2T", n_gpu=1)
        b.sort_by_src_length()
        extra = {"ctc_weight": ctc_weight, "ctc_candidates": 8} if ctc_weight > 0 else {}
        ids, _, _ = search(model, b, max_output_length=max_len, beam_size=beam, beam_alpha=alpha, n_best=1, **extra)
        return ids

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio_s = n_utts * SAMPLES / 16000.0
    # decoder FLOP of this decode with and without the key/value cache: per position and layer the projections
    # cost 8 d^2 (self) + 4 d^2 (cross q, o) + 4 d ff (FFN); attention 4 d per key; without a cache every step redoes the
    # whole prefix and re-projects the encoder states (4 d^2 per encoder position, hypothesis and layer)
    e, dcfg = MUSTC_MODEL["encoder"], MUSTC_MODEL["decoder"]
    d, ff, nl = dcfg["hidden_size"], dcfg["ff_size"], dcfg["num_layers"]
    S = ((1 + (SAMPLES - 400) // 160 - 1) // 2) // 2 + 1
    rows, steps = n_utts * beam, int(ids.shape[1])
    per_pos = nl * (12 * d * d + 4 * d * ff)
    cached = sum(rows * (per_pos + nl * 4 * d * (t + 1 + S) + 2 * d * VOCAB) for t in range(steps)) + n_utts * S * nl * 4 * d * d
    uncached = sum(rows * ((t + 1) * per_pos + nl * (4 * d * (t + 1) * (t + 2) // 2 + 4 * d * S * (t + 1)) + nl * S * 4 * d * d + 2 * d * VOCAB)
                   for t in range(steps))
    hyp_len = [int((row != 1).sum()) for row in np.asarray(ids)]  # non-pad tokens per best hypothesis (EOS cut by the search)
    if return_ids:
        return np.asarray(ids), dt
    return {"rtf": round(dt / audio_s, 6), "wall_s": round(dt, 3), "audio_s": audio_s, "beam": beam, "alpha": alpha,
            "n_utts": n_utts, "steps": steps, "steps_per_s": round(steps / dt, 1),
            "hyp_len_min_median_max": [int(np.min(hyp_len)), int(np.median(hyp_len)), int(np.max(hyp_len))], "decoder_tflop_kv_cached": round(cached / 1e12, 3),
            "decoder_tflop_full_prefix": round(uncached / 1e12, 3), "model": "mustc_st.yaml shapes, random init", "dtype": "bf16",
            "decoding": "KV-cached, hipGraph-replayed step"}


def decode_extras(device, beam=5, alpha=1.0, max_len=100):
    """(i) joint CTC / attention decoding (search(ctc_weight=0.3): f3, an extension - the reference returns `ctc_out`,
    model.py:162-166, and has no consumer) timed on the headline decode's workload; (ii) how often the bf16 decode returns the
    hypothesis of the fp32 mode (the mode whose ids the parity tests pin bit for bit) from the same weights: on a model whose
    hypotheses END (eos_scale 0.2: a random-init model otherwise never emits EOS), share of identical best hypotheses and the mean
    length of the common prefix."""
    out = {}
    audio_s = BATCH * SAMPLES / 16000.0
    try:
        dt_j = dt_a = float("inf")
        for _ in range(2):  # the better of two: a 0.1 s decode moves by 10 % from run to run
            ids_j, t = decode_rtf(device, beam=beam, alpha=alpha, max_len=max_len, ctc_weight=0.3, return_ids=True)
            dt_j = min(dt_j, t)
            ids_a, t = decode_rtf(device, beam=beam, alpha=alpha, max_len=max_len, return_ids=True)
            dt_a = min(dt_a, t)
        out["joint_ctc"] = {"ctc_weight": 0.3, "ctc_candidates": 8, "rtf": round(dt_j / audio_s, 6), "wall_s": round(dt_j, 3), "steps": int(ids_j.shape[1]),
                            "attention_only_wall_s": round(dt_a, 3), "ms_per_step_added": round((dt_j - dt_a) / max(1, int(ids_j.shape[1])) * 1e3, 3),
                            "what": "js2t_beam_pick + js2t_ctc_prefix_step per step on top of the KV-cached beam step, MuST-C shapes, 32 x 15 s, bf16"}
    except Exception as exc:  # noqa: BLE001
        out["joint_ctc"] = {"error": repr(exc)[:300]}
    try:
        cmp = {}
        for label, eos_scale in (("hypotheses_that_end", 0.2), ("free_running_100_tokens", 0.0)):
            ids = {}
            for name, dt_ in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
                ids[name], _ = decode_rtf(device, dtype=dt_, beam=beam, alpha=alpha, max_len=max_len, eos_scale=eos_scale, return_ids=True)
            same, prefix, lens = 0, [], []
            for ra, rb in zip(ids["bf16"], ids["fp32"]):
                ra, rb = [int(v) for v in ra if v != 1], [int(v) for v in rb if v != 1]
                same += int(ra == rb)
                n = 0
                while n < min(len(ra), len(rb)) and ra[n] == rb[n]:
                    n += 1
                prefix.append(n)
                lens.append(len(rb))
            cmp[label] = {"n_utts": len(prefix), "identical_hypotheses": same, "share_identical": round(same / max(1, len(prefix)), 3),
                          "mean_common_prefix_tokens": round(float(np.mean(prefix)), 2), "mean_fp32_length": round(float(np.mean(lens)), 2)}
        cmp["what"] = ("best beam-5 hypothesis per utterance, bf16 (the timed mode) against fp32 (the mode of the bit-exact id tests) from one random-init "
                       "MuST-C-size model: with its EOS row scaled so that hypotheses end, and free-running for 100 tokens (near-uniform next-token "
                       "distributions: every step is a near-tie, the hardest case for equality; a trained model's margins are wider)")
        out["bf16_vs_fp32"] = cmp
    except Exception as exc:  # noqa: BLE001
        out["bf16_vs_fp32"] = {"error": repr(exc)[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--no-decode", action="store_true", help="skip the beam-5 decode RTF measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side figures (encoder-forward and Conformer fp8 timings): a kernel trace of the train step alone")
    ap.add_argument("--ragged", action="store_true", help="utterances of 10-17 s instead of 32 x 15 s (value counts un-padded frames)")
    ap.add_argument("--host-inputs", action="store_true", help="copy the waveforms from pinned host memory every step (PCIe-inclusive rate; not the headline value)")
    ap.add_argument("--varying", action="store_true", help="side figure only: a new ragged batch every step (TokenBatchSampler + PrefetchLoader + one hipGraph per shape bucket)")
    args = ap.parse_args()

    # JS2T_BENCH_BACKEND=gloo + several ranks on one card rehearses the N > 1 code path on a single-GPU box
    # (gradient exchange through gloo instead of RCCL); never set by the driver
    backend = os.environ.get("JS2T_BENCH_BACKEND", "nccl")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, backend))  # this process never touches a GPU
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; measuring {world} ranks", file=sys.stderr)
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_ddp = world == 1 and os.environ.get("JS2T_BENCH_FORCE_DDP", "0") == "1"
    if force_ddp:
        os.environ["JS2T_DDP_SINGLE"] = "1"  # the one-rank communicator really issues its all-reduces
    roofline_pre = None
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {"device_id": device} if backend == "nccl" else {}
        import datetime
        # a rank that dies inside a side leg must cost the others minutes, not the default half hour, before they give up on it
        kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("JS2T_BENCH_PG_TIMEOUT", 300)))
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, **kw)
    n_ranks_seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    if args.varying:
        if world != 1:
            sys.exit("bench.py --varying: single GPU")
        res = varying_bench(device, args.steps if args.steps != 20 else 100, args.warmup if args.warmup != 5 else 60, use_graphs=not args.no_graph)
        print(json.dumps({"metric": "audio frames/sec (train step, varying batches)", "value": res["frames_per_s_unpadded"], "unit": "frames/s",
                          "n_gpus": 1, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "configs/librispeech_100h.yaml ASR train step, a new batch every step"}, "varying": res}), flush=True)
        return
    from joeys2t_amd import ops
    eager_step, graph_step, capture, step, frames_per_step, (model, state) = build_step(
        device, world, ragged=args.ragged, host_inputs=args.host_inputs, ddp=(world > 1 or force_ddp))
    use_graph = not args.no_graph
    one_step = eager_step
    capture_error = None
    if use_graph:
        # Data parallel: a capture that fails on this node must not cost the whole line - every rank then times the same kernels
        # launched eagerly, and says so (GraphedDDPStep.try_capture settles the outcome over the process group)
        dd = state.get("ddp_step")
        if dd is not None:
            capture_error = dd.try_capture()
        else:
            capture()
        if capture_error is None:
            one_step = graph_step
        else:
            use_graph = False
            print(f"bench.py: rank {rank}: graph capture failed, timing the eager step instead: {capture_error}", file=sys.stderr, flush=True)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    step.read_stats(reset=True)  # statistics of the timed steps only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = step.read_stats()
    # how much of a step the HOST spends inside the launch of its graph(s): with the GPU idle (synchronised before each launch) the
    # call returns when the ~900 nodes are enqueued.  A step cannot be shorter than this on this runtime.
    launch_host_ms = None
    if use_graph:
        # N > 1: a step is collectives too - EVERY rank runs these eight steps (never a subset of the ranks); the figure is rank 0's:
        # what its host spends launching the step's ~15 graphs and handing the ranges to RCCL between them, with the GPU idle
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            one_step()
            ts.append((time.perf_counter() - h0) * 1e3)
        torch.cuda.synchronize()
        launch_host_ms = round(sorted(ts)[len(ts) // 2], 3)
        step.read_stats(reset=True)

    exchange_times = None
    faithful = None
    if (world > 1 or force_ddp) and step.reducer is not None:
        headline_dtype = "bf16" if step.reducer.comm_dtype == torch.bfloat16 else "fp32"
        exchange_times = {headline_dtype: round(elapsed / args.steps * 1e3, 3)}
        # the other exchange as a side figure: collectives are never captured, so the reducer's staging dtype is a host-side switch
        # between two replays - every rank flips it at the same point.  (RCCL only: gloo has no bf16 staging path.)
        if torch.distributed.get_backend() == "nccl" and os.environ.get("JS2T_BENCH_ONE_EXCHANGE", "0") != "1":
            try:
                other = None if headline_dtype == "bf16" else torch.bfloat16
                keep = step.reducer.comm_dtype
                step.reducer.comm_dtype = other
                for _ in range(2):
                    one_step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                barrier()
                e2 = time.perf_counter() - t1
                if world > 1:
                    t = torch.tensor([e2], dtype=torch.float64, device=device)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    e2 = float(t.item())
                exchange_times["fp32" if other is None else "bf16"] = round(e2 / args.steps * 1e3, 3)
                step.reducer.comm_dtype = keep
                step.read_stats(reset=True)
            except Exception as exc:  # noqa: BLE001 - a side figure
                exchange_times["error"] = repr(exc)[:200]
        if world > 1 and not args.no_extras:
            # the config as written (batch_multiplier 4) through the composed graph driver, over the same process group
            try:
                faithful = config_faithful_bench(device, world, updates=3, merged=False)
            except Exception as exc:  # noqa: BLE001
                faithful = {"error": repr(exc)[:300]}
    if world > 1:
        # everything that needs the other ranks is done: the group is dissolved HERE, so that rank 0's single-GPU side measurements
        # (the roofline of the N = 1 step's kernels: the dominant kernel family does not change with N) keep no rank waiting in a
        # rendezvous or a collective - round 5 took them before init_process_group, with N - 1 ranks idle at the store
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        if rank == 0 and not args.no_roofline:
            try:
                es, _, _, st1, _, (m1, st_state) = build_step(device, 1, ddp=False)
                roofline_pre = measure_roofline(es, m1)
                roofline_pre["measured"] = "rank 0, single-GPU step, after the timed data-parallel steps (process group dissolved)"
                if not args.no_extras:
                    roofline_pre["encoder_forward"] = encoder_forward(m1, st_state["batch"])
                del es, st1, m1, st_state
                torch.cuda.empty_cache()
            except Exception as exc:  # never lose the N > 1 line over its side figure
                roofline_pre = {"error": repr(exc)[:300]}
    varying = None
    if rank == 0 and world == 1 and not args.no_roofline and not args.no_extras and use_graph:
        try:  # side figure: a new batch every step through sampler + loader + one graph per shape bucket
            import gc
            # This leg runs right behind the headline, in front of the roofline / extension legs: behind their allocations and graphs the
            # host side of a varying step (loader + bucket bookkeeping + graph launch, config.graph_launch_host_ms) ran ~1 ms slower
            # and the line turned host-bound (12.8 ms per step against 11.9 ms on the GPU; stand-alone `--varying`: 11.8 against 11.8).
            # A full collection or a hipMalloc inside the 100 timed steps would be a stall of this process's history, not of the path timed.
            gc.collect()
            torch.cuda.empty_cache()
            gc.freeze()
            try:
                varying = varying_bench(device, 100, 250)
            finally:
                gc.unfreeze()
        except Exception as exc:
            varying = {"error": repr(exc)}
    roofline = None
    if rank == 0 and world == 1 and not args.no_roofline:  # N = 1: on the step that was just timed
        roofline = measure_roofline(eager_step, model)
        try:
            roofline["hbm_kernels"] = measure_hbm_kernels(eager_step, step)
        except Exception as exc:  # a side table: never lose the headline line over it
            roofline["hbm_kernels"] = {"error": repr(exc)}
    elif rank == 0 and roofline_pre is not None:  # N > 1: measured on rank 0's GPU before the process group formed (see above)
        roofline = roofline_pre
    if rank == 0 and world == 1 and not force_ddp and not args.no_extras and use_graph:
        try:
            faithful = config_faithful_bench(device, 1)
        except Exception as exc:  # noqa: BLE001 - a side figure
            faithful = {"error": repr(exc)[:300]}
    by_config = None
    if rank == 0 and world == 1 and not force_ddp and not args.no_extras and use_graph and not args.no_roofline:
        by_config = train_step_by_config(device)
    if roofline is not None and not args.no_extras and world == 1:
        roofline["encoder_forward"] = encoder_forward(model, state["batch"])
        try:
            roofline["conformer_fp8_forward"] = conformer_fp8_forward(device)
        except Exception as exc:  # a side figure of an extension: never lose the headline line over it
            roofline["conformer_fp8_forward"] = {"error": repr(exc)}
        try:
            roofline["conformer_train_step"] = conformer_train_step(device)
        except Exception as exc:
            roofline["conformer_train_step"] = {"error": repr(exc)}

    fp32_mode = None
    if rank == 0 and world == 1 and roofline is not None and not args.no_extras:
        try:
            fp32_mode = fp32_parity_mode(device)
        except Exception as exc:
            fp32_mode = {"error": repr(exc)[:300]}
    decode = None
    if rank == 0 and world == 1 and not args.no_decode:
        try:
            decode = decode_rtf(device)
            # the same decode with a model whose lower beams end in EOS at every step while the best one goes on: the
            # host-side bookkeeping of finished hypotheses (search.py:683-717) then runs in all 100 steps
            fin = decode_rtf(device, eos_scale=0.2)
            decode["with_hypotheses_finishing_every_step"] = {k: fin[k] for k in ("rtf", "wall_s", "steps", "hyp_len_min_median_max")}
            if not args.no_extras:
                decode.update(decode_extras(device))
        except Exception as exc:
            decode = {"error": repr(exc)}
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline()
            except Exception as exc:  # the baseline is a reported side figure; never lose the GPU line over it
                cpu = {"error": repr(exc)}
            # the two decode RTFs describe one workload: if the CPU leg had to be cut to fewer utterances, decode those on the GPU too
            cd = cpu.get("decode_beam5") if isinstance(cpu, dict) else None
            if isinstance(cd, dict) and isinstance(decode, dict) and "rtf" in decode and cd.get("n_utts", BATCH) != BATCH:
                try:
                    same = decode_rtf(device, n_utts=cd["n_utts"])
                    decode["same_sample_as_cpu"] = {k: same[k] for k in ("rtf", "wall_s", "n_utts", "steps")}
                except Exception as exc:
                    decode["same_sample_as_cpu"] = {"error": repr(exc)}
        value = world * frames_per_step * args.steps / elapsed
        out = {
            "metric": "audio frames/sec (train step, 80-mel, 15 s utt, bs32 per GPU)", "value": round(value, 1),
            "unit": "frames/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs/librispeech_100h.yaml ASR train step on synthetic 16 kHz waveforms",
                       "global_batch": BATCH * world, "frames_per_utt": frames_per_step // BATCH, "encoder_len": ((int(state["batch"].src.shape[1]) - 1) // 2) // 2 + 1,
                       "lengths": "ragged 10-17 s, un-padded frames counted" if args.ragged else "fixed 15 s",
                       "vocab": VOCAB, "batch_multiplier": 1, "dropout": 0.1, "parallelism": f"dp{world}",
                       "graph_launch_host_ms": launch_host_ms,
                       "launch": ("hipGraph replay" if world == 1 and not force_ddp else
                                  "hipGraph replay in pieces (fwd + decoder-side bwd | decoder-side weight-gradient groups | encoder bwd | weight-gradient groups up to each completed gradient range | update) around the RCCL calls") if use_graph else "eager",
                       "backend": backend if n_ranks_seen > 1 or force_ddp else None,
                       "capture_error": capture_error,
                       "grad_exchange": None if not (n_ranks_seen > 1 or force_ddp) else {
                           "headline": "bf16 staging, fp32 accumulation in the flat gradient" if step.reducer is not None and step.reducer.comm_dtype == torch.bfloat16 else "fp32 (the reference's all-reduce)",
                           "ms_per_step": exchange_times},
                       "loss": round(stats["loss"] / max(1, args.steps), 4)},
            "precision": "bf16 products, fp32 accumulation / master weights / statistics; parity with the fp32 reference at bf16 tolerance "
                         "(tests/test_hip_config_width.py), at 1e-4 in the fp32 mode below",
            "fp32_parity_mode": fp32_mode,
            "config_faithful": faithful, "train_step_by_config": by_config,
            "roofline": roofline, "cpu_baseline": cpu, "decode_beam5": decode, "varying_batches": varying,
        }
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def fp32_parity_mode(device, n_steps=3):
    """The SAME train step in the fp32 parity mode (compute_dtype float32: f32 MFMA products, materialised attention - the mode whose
    outputs meet the reference at 1e-4, tests/test_hip_model.py) timed beside the bf16 headline, and how far the bf16 step's numbers
    are from it on the bench batch: relative L2 of the decoder logits (eval mode) and of the flat gradient (train mode, same dropout
    decisions: the masks are a function of seed, step, call site, row and column, not of the dtype).  The CPU baseline beside the
    headline is fp32 arithmetic (configs/librispeech_100h.yaml:6): this is the same-arithmetic GPU figure."""
    res = {}
    grads, logits = {}, {}
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        es, _, _, st, frames, (m, state) = build_step(device, 1, dtype=dt, ddp=False)
        st.optimizer.param_groups[0]["lr"] = 0.0  # the comparison wants both models ON the initial weights: steps that move nothing
        st.scheduler = None
        st.optimizer.lr_dev.fill_(0.0)
        st.optimizer.device_schedule = True
        state["grad_only"]()  # first call builds the batch (one zero-rate update) ...
        st.store.flat_grad.zero_()
        if st.optimizer.keep is not None:
            for lo, hi in list(st.optimizer.keep.r):
                st.optimizer.keep.remove(lo, hi)
        st.micro = 0
        st.rt.rng.state[1:2].zero_()  # ... and both dtypes then draw the masks of step 0
        state["grad_only"]()
        torch.cuda.synchronize()
        grads[name] = st.store.flat_grad.detach().float().clone()
        m.eval()
        with torch.no_grad():
            out, _, _ = m._encode_decode(**vars(state["batch"]))
        logits[name] = out.detach().float().clone()
        m.train()
        if name == "fp32":
            for _ in range(1):
                es()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                es()
            torch.cuda.synchronize()
            dt_s = (time.perf_counter() - t0) / n_steps
            res["ms_per_step"] = round(dt_s * 1e3, 2)
            res["frames_per_s"] = round(frames / dt_s, 1)
            res["launch"] = "eager (the step's GPU time exceeds its launch time in this mode)"
        del es, st, m, state
        torch.cuda.empty_cache()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    res["bf16_vs_fp32"] = {"logits_rel_l2": round(rel(logits["bf16"], logits["fp32"]), 5),
                           "flat_gradient_rel_l2": round(rel(grads["bf16"], grads["fp32"]), 5),
                           "flat_gradient_cosine": round(float(torch.dot(grads["bf16"], grads["fp32"]) / (grads["bf16"].norm() * grads["fp32"].norm())), 6),
                           "grad_norm_bf16": round(float(grads["bf16"].norm()), 4), "grad_norm_fp32": round(float(grads["fp32"].norm()), 4)}
    res["note"] = "fp32 mode: what the 1e-4 parity tests run; the headline value is the bf16 mode (BASELINE.json config: bf16)"
    return res


def self_launch(n, backend):
    """`python bench.py --gpus N` from a plain shell: start the N ranks as a child torch.distributed.run and hand its exit
    code back.  Nothing in this process has initialised the GPU (device_count() does not), and nothing is exec'ed."""
    import socket
    import subprocess
    if backend == "nccl" and torch.cuda.device_count() < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs, this node shows {torch.cuda.device_count()} "
              "(JS2T_BENCH_BACKEND=gloo rehearses the path with several ranks on one card)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__":
    main()
