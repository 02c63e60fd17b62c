"""Run js2t_gemm on one NN shape (for counter passes).  usage: python tools/gemm_one.py M N K [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
import os  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

if "JS2T_P192" in os.environ:
    lib().js2t_gemm_p192_mode(int(os.environ["JS2T_P192"]))
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).bfloat16()
B = torch.randn(N, K, device=dev).bfloat16()
C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
torch.cuda.synchronize()
