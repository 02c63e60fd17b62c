"""Feasibility probe: an encoder-forward-like chain (LN, QKV, flash attention, out-proj + residual, LN, FFN1, FFN2 + residual;
16 layers) on ONE stream over 12000 tokens against TWO streams over 6000 tokens each (two half batches), both replayed
from a hipGraph.  Prints ms per pass.  usage: python tools/two_stream_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
d, ff, H, T = 512, 2048, 4, 375
rng = ops.dropout_rng(dev)
W = {k: (torch.randn(s, device=dev) * 0.03).bfloat16() for k, s in
     dict(qkv=(3 * d, d), o=(d, d), f1=(ff, d), f2=(d, ff)).items()}
bias = {k: torch.zeros(n, device=dev) for k, n in dict(qkv=3 * d, o=d, f1=ff, f2=d).items()}
g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)


def layer(x, B):
    M = B * T
    h, _, _ = ops.layernorm_fwd(x, g, b, 1e-6)
    qkv = torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16)
    ops.gemm(h, W["qkv"], qkv, M=M, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d, bias=bias["qkv"])
    ctx, _ = ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, d // H, None, 0.1, rng, 3)
    y = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    ops.gemm(ctx, W["o"], y, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, bias=bias["o"], dropout_p=0.1, rng=rng, rng_stream=4, residual=x, ldr=d)
    h, _, _ = ops.layernorm_fwd(y, g, b, 1e-6)
    z = torch.empty(M, ff, device=dev, dtype=torch.bfloat16)
    ops.gemm(h, W["f1"], z, M=M, N=ff, K=d, lda=d, ldb=d, ldc=ff, bias=bias["f1"], act="relu", dropout_p=0.1, rng=rng, rng_stream=5)
    out = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    ops.gemm(z, W["f2"], out, M=M, N=d, K=ff, lda=ff, ldb=ff, ldc=d, bias=bias["f2"], dropout_p=0.1, rng=rng, rng_stream=6, residual=y, ldr=d)
    return out


def chain(x, B, layers=16):
    for _ in range(layers):
        x = layer(x, B)
    return x


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def capture(body):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    return gr


x32 = torch.randn(32 * T, d, device=dev).bfloat16()
xa, xb = x32[:16 * T].contiguous(), x32[16 * T:].contiguous()
s2 = torch.cuda.Stream()


def one():
    chain(x32, 32)


def two():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        chain(xb, 16)
    chain(xa, 16)
    cur.wait_stream(s2)


def two_serial():
    chain(xa, 16)
    chain(xb, 16)


for name, body in (("1 stream x 32 utt", one), ("2 streams x 16 utt", two), ("1 stream, 2 x 16 utt back to back", two_serial)):
    gr = capture(body)
    print(f"{name:36s} {timed(gr.replay):8.3f} ms", flush=True)
