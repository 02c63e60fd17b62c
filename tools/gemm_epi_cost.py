"""What the epilogue terms cost on the persistent GEMM: the same product with and without dropout / residual (LS100 shapes).
usage: python tools/gemm_epi_cost.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T = 12000


def timed(run, reps=50):
    for _ in range(5):
        run()
    best = 1e9
    for _ in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            run()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best


for name, N, K in [("ffn1", 2048, 512), ("out-proj", 512, 512), ("ffn2", 512, 2048)]:
    A = torch.randn(T, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.zeros(T, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    res = torch.randn(T, N, device=dev).bfloat16()
    rng = ops.dropout_rng(dev)
    base = dict(M=T, N=N, K=K, lda=K, ldb=K, ldc=N)
    variants = {"plain": {}, "bias": dict(bias=bias), "bias+relu": dict(bias=bias, act="relu"),
                "bias+drop": dict(bias=bias, dropout_p=0.1, rng=rng, rng_stream=3),
                "bias+relu+drop": dict(bias=bias, act="relu", dropout_p=0.1, rng=rng, rng_stream=3),
                "bias+res": dict(bias=bias, residual=res, ldr=N, res_scale=1.0),
                "bias+drop+res": dict(bias=bias, dropout_p=0.1, rng=rng, rng_stream=3, residual=res, ldr=N, res_scale=1.0)}
    out = []
    for vn, kw in variants.items():
        out.append(f"{vn} {timed(lambda: ops.gemm(A, B, C, **base, **kw)):.1f}")
    print(f"{name:9s} N={N} K={K}: " + " | ".join(out), flush=True)
