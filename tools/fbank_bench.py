"""Kernel time of the fbank front-end on the bench batch (32 x 15 s): usage python tools/fbank_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd.helpers_for_audio import get_extractor  # noqa: E402

dev = torch.device("cuda:0")
wave = (0.1 * torch.randn(32, 240000)).clamp_(-1, 1).to(dev)
ex = get_extractor(dev)
n = [240000] * 32
for _ in range(3):
    ex.batch(wave, n)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ex.batch(wave, n)
e.record()
torch.cuda.synchronize()
print(f"fbank 32 x 15 s: {s.elapsed_time(e) * 50:.1f} us per batch (incl. host-side offsets)")
