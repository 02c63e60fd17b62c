"""The update tail alone (clip + AdamW + derived weights) on the LS100 model of bench.py: fused kernel against the separate passes.
usage: python tools/update_bench.py [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from joeys2t_amd import builders  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
eager_step, graph_step, capture, step, frames, _ = bench.build_step(dev, 1)
eager_step()
eager_step()  # folds and transposed shadows exist now
torch.cuda.synchronize()
opt = step.optimizer
for fused in (False, True, False, True):
    builders.FUSED_UPDATE = fused
    for _ in range(3):
        opt.clip_and_step(step.clip_grad_norm, zero_grad=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        opt.clip_and_step(step.clip_grad_norm, zero_grad=True)
    e.record()
    torch.cuda.synchronize()
    print(f"fused={fused}: {s.elapsed_time(e) / reps * 1e3:.1f} us per update (clip + AdamW + shadows + folds)", flush=True)
