"""Does the row stride of the k-contiguous A operand matter (power-of-two strides vs padded rows)?
usage: python tools/gemm_nn_pad.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (M, N, K) in ((12000, 512, 2048), (12000, 2048, 512), (12000, 512, 512), (12000, 1536, 512)):
    for pad in (0, 64, 128):
        A = torch.randn(M, K + pad, device=dev).bfloat16()
        B = torch.randn(N, K, device=dev).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            ops.gemm(A, B, C, M=M, N=N, K=K, lda=K + pad, ldb=K, ldc=N)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.gemm(A, B, C, M=M, N=N, K=K, lda=K + pad, ldb=K, ldc=N)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 20
        print(f"M={M} N={N} K={K} lda=K+{pad:3d}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
