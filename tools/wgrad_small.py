"""Decoder-sized grouped weight gradients (K = 2592 tokens, 6 layers): time vs split-K factor.
usage: python tools/wgrad_small.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd.functional import wgrad_split  # noqa: E402

dev = torch.device("cuda:0")
T, L = 2592, 6
for rows, cols in ((2048, 512), (512, 2048), (1536, 512), (512, 512), (1024, 512)):
    line = f"dW[{rows},{cols}] x{L} K={T} (chosen split {wgrad_split(rows, cols, T, L)}):"
    for split in (1, 2, 3, 4, 6, 8):
        dzs = [(torch.randn(T, rows, device=dev) * (torch.rand(T, rows, device=dev) > 0.5)).bfloat16() for _ in range(L)]
        xs = [(torch.randn(T, cols, device=dev) * (torch.rand(T, cols, device=dev) > 0.5)).bfloat16() for _ in range(L)]
        Cs = [torch.zeros(rows, cols, device=dev) for _ in range(L)]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(12):
            if i == 2:
                s.record()
            ops.gemm_grouped(dzs, xs, Cs, M=rows, N=cols, K=T, lda=rows, ldb=cols, ldc=cols, split_k=split, beta=0.0 if split > 1 else 1.0)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 10
        line += f"  s{split}: {us:6.1f}us"
    print(line, flush=True)
