"""Which Python lines of the config-5 train step launch device-to-device copies and stray element-wise kernels: one eager step under
torch.profiler (with_stack), device copies / adds grouped by the innermost joeys2t_amd frame.  usage: python tools/conformer_copy_probe.py"""
import collections
import copy
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from joeys2t_amd.batch import Batch  # noqa: E402
from joeys2t_amd.model import build_model  # noqa: E402
from joeys2t_amd.training import TrainStep  # noqa: E402
from joeys2t_amd.vocabulary import Vocabulary  # noqa: E402

dev = torch.device("cuda:0")
cfg = copy.deepcopy(bench.LS100_MODEL)
cfg["encoder"].update(type="conformer", depthwise_conv_kernel_size=31, rel_pos_clip=64)
V = 10000
frames = 1 + (bench.SAMPLES - 400) // 160
trg, trg_len = bench.synth_targets(bench.BATCH, V, seed=99)
torch.manual_seed(42)
model = build_model(copy.deepcopy(cfg), None, Vocabulary.synthetic(V))
model.loss_function = ("crossentropy-ctc", 0.1, 0.3)
model.finalize(dev, torch.bfloat16, seed=42)
step = TrainStep(model, learning_rate=1e-4, adam_betas=(0.9, 0.98), clip_grad_norm=1.0, learning_rate_warmup=200, normalization="batch", overlap_ctc=True)
step.optimizer.device_schedule = True
src = torch.randn(bench.BATCH, frames, 80, device=dev).bfloat16()
batch = Batch(src=src, src_length=torch.full((bench.BATCH, ), frames, device=dev), src_prompt_mask=None, trg=trg, trg_length=trg_len,
              trg_prompt_mask=None, indices=torch.arange(bench.BATCH), device=dev, pad_index=1, eos_index=3, is_train=True, task="S2T", n_gpu=1)
for _ in range(2):
    step.micro_step(batch, sort=False, update=True, overlap=False)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.micro_step(batch, sort=False, update=True, overlap=False)
    torch.cuda.synchronize()
by = collections.Counter()
tm = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_"):
        frame = next((f for f in (ev.stack or []) if "joeys2t_amd" in f or "bench.py" in f), "?")
        key = (ev.name, frame.split("/")[-1][:90])
        by[key] += 1
        tm[key] += ev.device_time_total
for key, n in sorted(by.items(), key=lambda kv: -tm[kv[0]])[:40]:
    print(f"{key[0]:18s} x{n:4d} {tm[key]:9.0f} us  {key[1]}")
