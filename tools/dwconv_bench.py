"""Depthwise convolution over the batch index (the Conformer module's quirk, transformer_layers.py:410-475): forward, input and
weight gradient at config-5 shapes.   usage: python tools/dwconv_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for L, N, C, K in [(32, 375, 512, 31), (32, 375, 512, 15), (8, 375, 512, 31), (64, 200, 512, 31)]:
    x = torch.randn(L, N, C, device=dev).bfloat16()
    dy = torch.randn(L, N, C, device=dev).bfloat16()
    w = torch.randn(C, K, device=dev)
    b = torch.zeros(C, device=dev)
    fwd = t(lambda: ops.dwconv_outer_fwd(x, w, b))
    bwd_dx = t(lambda: ops.dwconv_outer_bwd(dy, x, w, need_dx=True, dw_out=None)) if False else None
    both = t(lambda: ops.dwconv_outer_bwd(dy, x, w))
    print(f"L={L} N={N} C={C} K={K}: forward {fwd:7.1f} us, backward (dx + dw) {both:7.1f} us", flush=True)
