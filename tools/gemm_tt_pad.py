"""Does the row stride of reduction-major operands matter (L2 / HBM channel camping)?  C = A[K,M]^T B[K,N] with padded
leading dimensions.  usage: python tools/gemm_tt_pad.py M N K"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
for pad in (0, 8, 64, 128, 192):
    A = torch.randn(K, M + pad, device=dev).bfloat16()
    B = torch.randn(K, N + pad, device=dev).bfloat16()
    C = torch.zeros(M, N, device=dev, dtype=torch.float32)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(12):
        if i == 2:
            s.record()
        ops.gemm(A, B, C, M=M, N=N, K=K, lda=M + pad, ldb=N + pad, ldc=N, trans_a=True, trans_b=True, split_k=1)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 10
    print(f"TT M={M} N={N} K={K} pad={pad:4d}: {us:8.1f} us, {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
