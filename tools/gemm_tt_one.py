"""Run the weight-gradient form of js2t_gemm (C[M,N] = A[K,M]^T B[K,N], f32 C) for counter passes.
usage: python tools/gemm_tt_one.py M N K [split_k] [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
split = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
A = torch.randn(K, M, device=dev).bfloat16()
B = torch.randn(K, N, device=dev).bfloat16()
C = torch.zeros(M, N, device=dev, dtype=torch.float32)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(reps + 2):
    if i == 2:
        s.record()
    ops.gemm(A, B, C, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=True, split_k=split)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / reps
print(f"TT M={M} N={N} K={K} split={split}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.1f} TF")
