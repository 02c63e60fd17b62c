"""Where a stage of the loader/consumer GEMM (js2t_gemm_p192_ring(4)) goes: the same launches with instrumented builds
(JS2T_LIB=...; gemm.hip compiled with -DJS2T_P192S_DBG=1: no requests, =2: no MFMAs, =3: neither).  Results are garbage
in those builds - timing only.   usage: JS2T_LIB=joeys2t_amd/build/libdbg1.so python tools/p192s_parts.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
T = 12000
lib().js2t_gemm_p192_mode(1)
lib().js2t_gemm_p192_ring(int(os.environ.get("RING", "4")))
for name, M, N, K in [("N=512 K=2048", T, 512, 2048), ("N=512 K=1536", T, 512, 1536), ("N=512 K=512", T, 512, 512),
                      ("N=2048 K=512", T, 2048, 512), ("N=2048 K=2048", T, 2048, 2048)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)  # noqa: E731
    best = 1e9
    for _ in range(4):
        run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            run()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / 30)
    tiles = -(-M // 192) * (N // 128)
    stages = -(-tiles // 256) * (K // 64)
    if hasattr(lib(), "js2t_debug_p192s_prof"):
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        lib().js2t_debug_p192s_prof(buf)
        tot, bar, lw, lb, li, ns, epi = [buf[i] for i in (0, 1, 2, 3, 4, 5, 6)]
        print(f"    block 0: consumer ticks {tot} ({tot / max(best * 1e3, 1):.2f} per ns), per stage {tot / max(ns, 1):.0f} of which "
              f"lgkm+barrier {bar / max(ns, 1):.0f}, epilogue share {epi / max(ns, 1):.0f}; loader per stage: vmcnt wait {lw / max(ns, 1):.0f}, "
              f"barrier {lb / max(ns, 1):.0f}, issue {li / max(ns, 1):.0f}")
    print(f"{os.environ.get('JS2T_LIB', 'product')[-12:]:12s} {name:14s} {best:7.1f} us  {2.0 * M * N * K / best / 1e6:6.0f} TF  "
          f"{best * 1e3 / stages:6.0f} ns per stage (incl. epilogue share)", flush=True)
