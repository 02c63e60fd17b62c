"""Grouped weight-gradient launch (16 layers) with padded operand row strides: which strides camp on L2 channels?
usage: python tools/wgrad_pad.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T, L = 12000, 16
for rows, cols in ((2048, 512), (512, 2048), (1536, 512), (512, 512)):
    for pa, pb in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32)):
        dzs = [torch.randn(T, rows + pa, device=dev).bfloat16() for _ in range(L)]
        xs = [torch.randn(T, cols + pb, device=dev).bfloat16() for _ in range(L)]
        Cs = [torch.zeros(rows, cols, device=dev) for _ in range(L)]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(7):
            if i == 2:
                s.record()
            ops.gemm_grouped(dzs, xs, Cs, M=rows, N=cols, K=T, lda=rows + pa, ldb=cols + pb, ldc=cols, split_k=1, beta=1.0)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 5
        print(f"dW[{rows},{cols}] x{L}  pad dY={pa:3d} X={pb:3d}: {us:8.1f} us {2.0 * rows * cols * T * L / us / 1e6:7.1f} TF", flush=True)
