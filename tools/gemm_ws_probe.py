"""Weight-stationary K = 512 product (csrc/gemm_ws.hip) against js2t_gemm's kernel choice on the encoder's shapes.
usage: python tools/gemm_ws_probe.py [reps]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = lib()


def ws(x, w, out, bias, relu):
    check(L.js2t_debug_gemm_ws512(ops._p(x), C.c_int64(x.stride(0)), ops._p(w), C.c_int64(w.stride(0)), ops._p(out), C.c_int64(out.stride(0)),
                                  ops._p(bias), int(relu), int(x.shape[0]), int(w.shape[0]), ops._stream()), "ws512")


def timeit(fn):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


torch.manual_seed(0)
for name, M, N, relu in (("ffn1 relu", 12000, 2048, True), ("qkv", 12000, 1536, False), ("out", 12000, 512, False), ("ragged rows", 11991, 2048, True),
                         ("few rows", 70, 512, False), ("decoder ffn1", 2592, 2048, True)):
    x = torch.randn(M, 512, device=dev).bfloat16()
    w = (torch.randn(N, 512, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    a, c = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
    ops.gemm(x, w, a, M=M, N=N, K=512, lda=512, ldb=512, ldc=N, bias=b, act="relu" if relu else None)
    ws(x, w, c, b, relu)
    torch.cuda.synchronize()
    diff = (a.float() - c.float()).abs().max().item()
    t0 = timeit(lambda: ops.gemm(x, w, a, M=M, N=N, K=512, lda=512, ldb=512, ldc=N, bias=b, act="relu" if relu else None))
    t1 = timeit(lambda: ws(x, w, c, b, relu))
    fl = 2.0 * M * N * 512
    print(f"{name:14s} M={M:6d} N={N:5d}  js2t_gemm {t0:7.1f} us {fl / t0 / 1e6:7.1f} TF   weight-stationary {t1:7.1f} us {fl / t1 / 1e6:7.1f} TF   max|diff| {diff:g}")
