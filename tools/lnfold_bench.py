"""Micro-benchmark of the LayerNorm-fold epilogues of js2t_gemm against the plain ones, LS100 shapes (run on the GPU box).
usage: python tools/lnfold_bench.py [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def bench(name, M, N, K, act=None, drop=0.0, res=False, ln=False, stats=False):
    A = torch.randn((M, K), device=dev).bfloat16()
    B = (torch.randn((N, K), device=dev) / K**0.5).bfloat16()
    C = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    kw = {"bias": torch.randn(N, device=dev)}
    if act:
        kw["act"] = act
    if drop > 0:
        kw.update(dropout_p=drop, rng=ops.dropout_rng(dev), rng_stream=3)
    if res:
        kw.update(residual=torch.randn(M, N, device=dev).bfloat16(), ldr=N, res_scale=1.0)
    if ln:
        g8 = A.float().view(M, 8, 64)
        kw["ln"] = (torch.stack([g8.sum(2), (g8 * g8).sum(2)], dim=2).contiguous(), 1e-6, torch.empty(M, device=dev), torch.empty(M, device=dev))
    if stats:
        kw["rs_partial"] = torch.empty((M, 8, 2), device=dev)

    def run():
        ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)

    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)


for T in (12000, ):
    bench("qkv", T, 1536, 512)
    bench("qkv + ln", T, 1536, 512, ln=True)
    bench("ffn1 relu drop", T, 2048, 512, act="relu", drop=0.1)
    bench("ffn1 relu drop + ln", T, 2048, 512, act="relu", drop=0.1, ln=True)
    bench("out drop res", T, 512, 512, drop=0.1, res=True)
    bench("out drop res + stats", T, 512, 512, drop=0.1, res=True, stats=True)
    bench("ffn2 drop res", T, 512, 2048, drop=0.1, res=True)
    bench("ffn2 drop res + stats", T, 512, 2048, drop=0.1, res=True, stats=True)
for T in (2592, ):
    bench("dec qkv", T, 1536, 512)
    bench("dec qkv + ln", T, 1536, 512, ln=True)
    bench("dec ffn1 relu drop", T, 2048, 512, act="relu", drop=0.1)
    bench("dec ffn1 relu drop + ln", T, 2048, 512, act="relu", drop=0.1, ln=True)
    bench("dec q", T, 512, 512)
    bench("dec q + ln", T, 512, 512, ln=True)
    bench("dec out drop res", T, 512, 512, drop=0.1, res=True)
    bench("dec out drop res + stats", T, 512, 512, drop=0.1, res=True, stats=True)
    bench("dec ffn2 drop res", T, 512, 2048, drop=0.1, res=True)
    bench("dec ffn2 drop res + stats", T, 512, 2048, drop=0.1, res=True, stats=True)
