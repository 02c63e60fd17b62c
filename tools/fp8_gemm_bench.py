"""e4m3 products of the Conformer / LS shapes on the persistent kernels: loader / consumer form (ring 4: the scaled K = 128
instruction unless built with -DJS2T_FP8_NO_SCALED) and the two-blocks-per-CU form (ring 2: K = 32 instruction), bf16 beside
them.  usage: python tools/fp8_gemm_bench.py [reps]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")


def t(fn):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for (M, N, K, what) in [(12000, 2048, 512, "FFN1"), (12000, 1536, 512, "QKV"), (12000, 512, 2048, "FFN2"), (12000, 512, 512, "out-proj")]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    w8, ws = ops.quantize_fp8(W)
    a8, sc = ops.quantize_fp8(A, mul=ws)
    ref = None
    row = [f"{what:9s} {M}x{N}x{K}:"]
    for ring in (4, 2):
        lib().js2t_gemm_p192_ring(C.c_int(ring))
        lib().js2t_gemm_p192_mode(C.c_int(1))
        us8 = t(lambda: ops.gemm(a8, w8, Cc, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, alpha_dev=sc))
        out8 = Cc.float().clone()
        usb = t(lambda: ops.gemm(A, W, Cc, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias))
        if ref is None:
            ref = out8
        same = torch.equal(ref, out8)
        row.append(f"ring {ring}: e4m3 {us8:6.1f} us ({2e-6 * M * N * K / us8:6.0f} TFLOP/s)  bf16 {usb:6.1f} us  {'=' if same else 'DIFFERS from ring 4'}")
    lib().js2t_gemm_p192_ring(C.c_int(-1))
    lib().js2t_gemm_p192_mode(C.c_int(-1))
    err = (ref - (A.float() @ W.float().t() + bias)).norm() / (A.float() @ W.float().t() + bias).norm()
    row.append(f"rel. L2 vs bf16-operand fp32 math {err.item():.3f}")
    print("  ".join(row), flush=True)
