"""Decoder-sized GEMMs (M = 2592): kernel durations under rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (M, N, K, tb) in [(2592, 512, 512, 0), (2592, 512, 2048, 0), (2592, 512, 512, 1), (2592, 512, 2048, 1), (2592, 512, 1536, 1),
                      (2592, 2048, 512, 0), (2592, 1536, 512, 0)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(K, N, device=dev) if tb else torch.randn(N, K, device=dev)).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=B.shape[1], ldc=N, trans_b=bool(tb))
    torch.cuda.synchronize()
    s.record()
    for _ in range(100):
        ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=B.shape[1], ldc=N, trans_b=bool(tb))
    e.record()
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K} tb={tb}: {s.elapsed_time(e) * 10:.1f} us/launch (back-to-back incl. launch gaps)")
