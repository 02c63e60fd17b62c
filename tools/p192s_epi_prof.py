"""Where the consumer waves of the loader/consumer GEMM spend a tile, with the real epilogues of an encoder layer's four products
(gemm.hip built with -DJS2T_P192S_DBG=8: cycle counters only).  Block 0, consumer wave 0.
usage (GPU box): JS2T_HIPCC_EXTRA=-DJS2T_P192S_DBG=8 python -c "from joeys2t_amd import _build; _build.build_library(force=True)" && python tools/p192s_epi_prof.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
T = 12000
lib().js2t_gemm_p192_ring(ctypes.c_int(4))
for name, N, K, kw in [("plain QKV", 1536, 512, {}), ("QKV", 1536, 512, dict(bias=True)), ("out-proj", 512, 512, dict(bias=True, drop=True, res=True)),
                       ("plain FFN1", 2048, 512, {}), ("FFN1", 2048, 512, dict(bias=True, relu=True, drop=True)),
                       ("FFN2", 512, 2048, dict(bias=True, drop=True, res=True))]:
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
    run = lambda: ops.gemm(A, B, C, M=T, N=N, K=K, lda=K, ldb=K, ldc=N, **extra)  # noqa: E731
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        run()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 50
    buf = (ctypes.c_ulonglong * 8)()
    lib().js2t_debug_p192s_prof(buf)
    tot, bar, lw, lb, li, ns, epi = [buf[i] for i in (0, 1, 2, 3, 4, 5, 6)]
    ntile = max(ns // (K // 64), 1)
    print(f"{name:11s} N={N} K={K}: {us:6.1f} us | block 0 wave 0: {tot} ticks, {ntile} tiles of {K // 64} stages; per tile: total {tot / ntile:.0f}, "
          f"epilogue {epi / ntile:.0f}, lgkm+barrier wait {bar / ntile:.0f}; loader per stage: vmcnt wait {lw / max(ns, 1):.0f}, barrier {lb / max(ns, 1):.0f}, "
          f"issue {li / max(ns, 1):.0f}", flush=True)
