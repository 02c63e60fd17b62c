"""Micro-benchmark of js2t_gemm on the shapes of the LS100 train step (run on the GPU box).
usage: python tools/gemm_bench.py [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def bench(name, M, N, K, ta=False, tb=False, batch=1, split=1, bias=False, act=None, drop=0.0, res=False, gate=False,
          out=torch.bfloat16):
    A = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    B = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    C = torch.zeros((M, N), device=dev, dtype=torch.float32 if split > 1 else out)
    kw = {}
    if bias:
        kw["bias"] = torch.randn(N, device=dev)
    if act:
        kw["act"] = act
    if drop > 0:
        kw.update(dropout_p=drop, rng=ops.dropout_rng(dev), rng_stream=3)
    if res:
        kw.update(residual=torch.randn(M, N, device=dev).to(out), ldr=N, res_scale=1.0)
    if gate:
        kw.update(gate=torch.randn(M, N, device=dev).to(out), ldg=N, gate_scale=1.1)

    def run():
        ops.gemm(A, B, C, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, trans_a=ta, trans_b=tb, split_k=split, **kw)

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
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} {'T' if ta else 'N'}{'T' if tb else 'N'} split={split:2d} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)


T = 12000
bench("ffn1 fwd plain", T, 2048, 512)
bench("ffn1 fwd bias", T, 2048, 512, bias=True)
bench("ffn1 fwd bias+relu", T, 2048, 512, bias=True, act="relu")
bench("ffn1 fwd bias+relu+drop", T, 2048, 512, bias=True, act="relu", drop=0.1)
bench("ffn1 fwd plain f32out", T, 2048, 512, out=torch.float32)
bench("ffn2 fwd plain", T, 512, 2048)
bench("ffn2 fwd bias+drop+res", T, 512, 2048, bias=True, drop=0.1, res=True)
bench("qkv fwd", T, 1536, 512, bias=True)
bench("out fwd", T, 512, 512, bias=True, drop=0.1, res=True)
bench("ctc proj", T, 5000, 512)
bench("dgrad ffn2 (NN) plain", T, 2048, 512, tb=True)
bench("dgrad ffn2 (NN) gate", T, 2048, 512, tb=True, gate=True)
bench("dgrad ffn1 (NN)", T, 512, 2048, tb=True)
bench("wgrad ffn1 (TT) split4", 2048, 512, T, ta=True, tb=True, split=4)
bench("wgrad ffn1 (TT) split8", 2048, 512, T, ta=True, tb=True, split=8)
bench("wgrad ffn2 (TT) split4", 512, 2048, T, ta=True, tb=True, split=4)
bench("wgrad out (TT) split16", 512, 512, T, ta=True, tb=True, split=16)
bench("wgrad out (TT) split8", 512, 512, T, ta=True, tb=True, split=8)
bench("big square", 8192, 8192, 8192)
bench("big NN", 8192, 8192, 8192, tb=True)
bench("4k square", 4096, 4096, 4096)
