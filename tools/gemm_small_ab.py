"""Decoder-sized products (M = 2592): the 64-row LDS-DMA tile kernel (default) against the persistent 192x128 kernel forced on
(js2t_gemm_p192_mode(1)), interleaved in one process.  usage: python tools/gemm_small_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2592
for (N, K, kw) in [(512, 512, {}), (512, 512, dict(bias=True)), (1536, 512, dict(bias=True)), (512, 1536, {}), (2048, 512, dict(bias=True, act="relu")),
                   (512, 2048, dict(bias=True)), (2048, 512, {}), (512, 2048, {})]:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    k2 = dict(kw)
    if k2.pop("bias", False):
        k2["bias"] = torch.randn(N, device=dev)
    res = {}
    for rnd in range(3):
        for mode in (-1, 1):
            lib().js2t_gemm_p192_mode(mode)
            for _ in range(3):
                ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **k2)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **k2)
            e.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(s.elapsed_time(e) * 20)
    a, b = min(res[-1]), min(res[1])
    print(f"M={M} N={N:5d} K={K:5d} {str(sorted(kw)):24s} default {a:6.1f} us | p192 forced {b:6.1f} us  x{a / b:4.2f}", flush=True)
lib().js2t_gemm_p192_mode(-1)
