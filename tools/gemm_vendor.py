"""Practical-ceiling probe: js2t_gemm beside the vendor library (torch.matmul -> hipBLASLt) on the train-step shapes.
Measurement aid only; nothing in the product path calls the vendor library.  usage: python tools/gemm_vendor.py [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

import os  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

if "JS2T_P192" in os.environ:
    lib().js2t_gemm_p192_mode(int(os.environ["JS2T_P192"]))
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def timeit(run):
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def bench(name, M, N, K, ta=False, tb=False):
    A = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    B = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    C = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    ours = timeit(lambda: ops.gemm(A, B, C, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, trans_a=ta, trans_b=tb))
    Am = A.t() if ta else A
    Bm = B if tb else B.t()
    vend = timeit(lambda: torch.matmul(Am, Bm, out=C))
    f = 2.0 * M * N * K / 1e6
    print(f"{name:22s} M={M:6d} N={N:5d} K={K:6d} {'T' if ta else 'N'}{'T' if tb else 'N'}  ours {ours:7.1f} us {f / ours:7.1f} TF   vendor {vend:7.1f} us {f / vend:7.1f} TF", flush=True)


T = 12000
bench("ffn1 fwd", T, 2048, 512)
bench("ffn2 fwd", T, 512, 2048)
bench("qkv fwd", T, 1536, 512)
bench("out fwd", T, 512, 512)
bench("qkv dgrad", T, 512, 1536)
bench("ctc proj", T, 5000, 512)
bench("dec out", 2592, 512, 512)
bench("dec ffn1", 2592, 2048, 512)
bench("dec ffn2", 2592, 512, 2048)
bench("wgrad ffn1 (TT)", 2048, 512, T, ta=True, tb=True)
bench("wgrad out (TT)", 512, 512, T, ta=True, tb=True)
bench("big square", 8192, 8192, 8192)
bench("4k square", 4096, 4096, 4096)
