"""Soak run of BASELINE.json configs[4] as bench.py builds it (Conformer encoder 16 x d 512 with relative-position attention,
6-layer decoder, V 10000, 32 x 15 s of synthetic features, dropout 0.1): N updates on a small set of fixed batches, loss per update
-> a curve under profiles/.  EXTENSION without a reference target (see bench.py conformer_train_step).
usage (GPU box): python tools/conformer_soak.py [steps] [lr] [bf16|fp8] [deterministic 0|1] > profiles/r05_conformer_soak_<mode>.txt"""
import copy
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from joeys2t_amd import functional as Fn  # noqa: E402
from joeys2t_amd.batch import Batch  # noqa: E402
from joeys2t_amd.model import build_model  # noqa: E402
from joeys2t_amd.training import TrainStep  # noqa: E402
from joeys2t_amd.vocabulary import Vocabulary  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
det = len(sys.argv) > 4 and sys.argv[4] == "1"
warmup = int(sys.argv[5]) if len(sys.argv) > 5 else 200
dev = torch.device("cuda:0")
Fn.FP8_FORWARD = mode == "fp8"
cfg = copy.deepcopy(bench.LS100_MODEL)
cfg["encoder"].update(type="conformer", depthwise_conv_kernel_size=31, rel_pos_clip=64)
V, B = 10000, bench.BATCH
frames = 1 + (bench.SAMPLES - 400) // 160
torch.manual_seed(42)
model = build_model(copy.deepcopy(cfg), None, Vocabulary.synthetic(V))
model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
with torch.no_grad():
    for layer in model.encoder.layers:
        layer.src_src_att.rel_pos_bias.normal_(0.0, 0.1)
model.finalize(dev, torch.bfloat16, seed=42)
step = TrainStep(model, learning_rate=lr, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=warmup, normalization="batch",
                 overlap_ctc=True, deterministic=det)
batches = []
for s in range(4):  # four fixed batches in turn: a model that trains memorises them
    trg, trg_len = bench.synth_targets(B, V, seed=99 + s)
    src = torch.randn(B, frames, 80, generator=torch.Generator().manual_seed(7 + s)).to(dev).bfloat16()
    batches.append(Batch(src=src, src_length=torch.full((B, ), frames, device=dev), src_prompt_mask=None, trg=trg, trg_length=trg_len,
                         trg_prompt_mask=None, indices=torch.arange(B), device=dev, pad_index=1, eos_index=3, is_train=True, task="S2T", n_gpu=1))
print(f"# config 5 soak: mode {mode}, peak lr {lr}, warm-up {warmup} updates (inverse square root after), clip 1.0, deterministic {int(det)}, "
      f"{steps} updates over 4 fixed batches of {B} x {frames} frames; loss = batch-normalised total (0.7 CE + 0.3 CTC), per update")
print("# update loss nll ctc grad_norm lr")
for i in range(steps):
    step.micro_step(batches[i % 4], sort=False)
    if i < 20 or i % 10 == 9:
        s = step.read_stats()
        print(f"{i + 1} {s['loss']:.4f} {s['nll']:.4f} {s['ctc']:.4f} {s['grad_norm']:.4f} {s['lr']:.3e}", flush=True)
    else:
        step.read_stats()
