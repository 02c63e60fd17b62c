"""A measured ceiling for the encoder layer's products on the persistent kernels: the same launches with
the tile epilogue compiled out (-DJS2T_GEMM_NOEPI: the operands stream through the LDS ring, every MFMA is issued, nothing is
stored) against the shipped kernels.  What is left is the feed (L2 -> LDS at ~29 B/clk/CU) + MFMA issue + tile switch - the
time no epilogue work, however cheap, can get under at this tile shape.
usage (GPU box): python tools/k512_ceiling.py            # shipped library
                 JS2T_HIPCC_EXTRA=-DJS2T_GEMM_NOEPI python -c "from joeys2t_amd import _build; _build.build_library(force=True)" && python tools/k512_ceiling.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

import ctypes as C  # noqa: E402
import os  # noqa: E402

from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
T = 12000
if os.environ.get("K512_RING"):  # force one form of the persistent kernel for every product (4 = loader / consumer waves)
    lib().js2t_gemm_p192_ring(C.c_int(int(os.environ["K512_RING"])))
    print("forced js2t_gemm_p192_ring(%s)" % os.environ["K512_RING"])


def timed(run, reps=200):
    for _ in range(5):
        run()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            run()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best


tot = 0.0
for name, N, K, kw in [("QKV", 1536, 512, dict(bias=True)), ("out-proj", 512, 512, dict(bias=True, drop=True, res=True)),
                       ("FFN1", 2048, 512, dict(bias=True, relu=True, drop=True)), ("FFN2", 512, 2048, dict(bias=True, drop=True, res=True))]:
    A = torch.randn(T, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    C = torch.zeros(T, N, device=dev, dtype=torch.bfloat16)
    extra = {}
    if kw.get("bias"):
        extra["bias"] = torch.randn(N, device=dev)
    if kw.get("relu"):
        extra["act"] = "relu"
    if kw.get("drop"):
        extra.update(dropout_p=0.1, rng=ops.dropout_rng(dev), rng_stream=3)
    if kw.get("res"):
        extra.update(residual=torch.randn(T, N, device=dev).bfloat16(), ldr=N, res_scale=1.0)
    us = timed(lambda: ops.gemm(A, B, C, M=T, N=N, K=K, lda=K, ldb=K, ldc=N, **extra))
    tot += us
    print(f"{name:9s} 12000 x {N} x {K}: {us:6.1f} us  {2e-6 * T * N * K / us:6.0f} TFLOP/s", flush=True)
att = 4 * 32 * 4 * 375 * 375 * 128 * 1e-6  # attention forward MFLOP of a layer (not run here)
print(f"sum of the four products: {tot:.1f} us per encoder layer = {2e-6 * T * (1536 * 512 + 512 * 512 + 2 * 2048 * 512) / tot:.0f} TFLOP/s")
