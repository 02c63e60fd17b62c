"""Feasibility probe: a decoder-forward-like chain (8 layers: LN, QKV, self-attention, out-proj, LN, q-proj, cross-attention over
a [12000, 1024] key/value memory, out-proj, LN, FFN1, FFN2) for 32 utterances x 81 target positions on ONE stream against
two half batches on TWO streams, both replayed from a hipGraph.  usage: python tools/two_stream_decoder_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
d, ff, H, L, S = 512, 2048, 4, 81, 375
rng = ops.dropout_rng(dev)
W = {k: (torch.randn(s, device=dev) * 0.03).bfloat16() for k, s in
     dict(qkv=(3 * d, d), o=(d, d), q=(d, d), kv=(2 * d, d), o2=(d, d), f1=(ff, d), f2=(d, ff)).items()}
bias = {k: torch.zeros(v.shape[0], device=dev) for k, v in W.items()}
g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)


def lin(x, k, M, **kw):
    N, K = W[k].shape
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(x, W[k], y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias[k], **kw)
    return y


def layer(x, mem, B):
    M = B * L
    h, _, _ = ops.layernorm_fwd(x, g, b, 1e-6)
    qkv = lin(h, "qkv", M)
    ctx, _ = ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, L, L, d // H, None, 0.1, rng, 3)
    y = lin(ctx, "o", M, dropout_p=0.1, rng=rng, rng_stream=4, residual=x, ldr=d)
    h, _, _ = ops.layernorm_fwd(y, g, b, 1e-6)
    q = lin(h, "q", M)
    kv = lin(mem, "kv", B * S)
    ctx, _ = ops.flash_attn_fwd(q, 0, kv, 0, kv, d, B, H, L, S, d // H, None, 0.1, rng, 5)
    y2 = lin(ctx, "o2", M, dropout_p=0.1, rng=rng, rng_stream=6, residual=y, ldr=d)
    h, _, _ = ops.layernorm_fwd(y2, g, b, 1e-6)
    z = lin(h, "f1", M, act="relu", dropout_p=0.1, rng=rng, rng_stream=7)
    return lin(z, "f2", M, dropout_p=0.1, rng=rng, rng_stream=8, residual=y2, ldr=d)


def chain(x, mem, B, layers=8):
    for _ in range(layers):
        x = layer(x, mem, B)
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


x32 = torch.randn(32 * L, d, device=dev).bfloat16()
m32 = torch.randn(32 * S, d, device=dev).bfloat16()
streams = [torch.cuda.Stream() for _ in range(3)]


def split(n):
    per = 32 // n
    xs = [x32[i * per * L:(i + 1) * per * L].contiguous() for i in range(n)]
    ms = [m32[i * per * S:(i + 1) * per * S].contiguous() for i in range(n)]

    def body():
        cur = torch.cuda.current_stream()
        for i in range(1, n):
            streams[i - 1].wait_stream(cur)
            with torch.cuda.stream(streams[i - 1]):
                chain(xs[i], ms[i], per)
        chain(xs[0], ms[0], per)
        for i in range(1, n):
            cur.wait_stream(streams[i - 1])
    return body


print(f"1 stream  x 32 utt {timed(capture(lambda: chain(x32, m32, 32)).replay):8.3f} ms", flush=True)
for n in (2, 4):
    print(f"{n} streams x {32 // n:2d} utt {timed(capture(split(n)).replay):8.3f} ms", flush=True)
