"""A/B of the persistent 192x128 GEMM's variants on the LS100 train-step shapes, interleaved in one process:
ring 3 = one block per CU, two stages in flight; ring 2 = two blocks per CU, one stage in flight each; 4 = one block of
four consumer + four loader waves per CU (3-slot ring).
usage: python tools/p192_ring_ab.py [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
T = 12000
SHAPES = [("qkv fwd bias", T, 1536, 512, dict(bias=True)), ("out-proj fwd bias+drop+res", T, 512, 512, dict(bias=True, drop=0.1, res=True)),
          ("ffn1 fwd bias+relu+drop", T, 2048, 512, dict(bias=True, act="relu", drop=0.1)),
          ("ffn2 fwd bias+drop+res", T, 512, 2048, dict(bias=True, drop=0.1, res=True)), ("dx qkv plain", T, 512, 1536, {}),
          ("dx out plain", T, 512, 512, {}), ("dx ffn2 gate", T, 2048, 512, dict(gate=True)), ("dx ffn1 plain", T, 512, 2048, {}),
          ("kv proj memory", T, 1024, 512, dict(bias=True)), ("ctc proj", T, 5000, 512, {})]


def make(M, N, K, bias=False, act=None, drop=0.0, res=False, gate=False):
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    kw = {}
    if bias:
        kw["bias"] = torch.randn(N, device=dev)
    if act:
        kw["act"] = act
    if drop > 0:
        kw.update(dropout_p=drop, rng=ops.dropout_rng(dev), rng_stream=3)
    if res:
        kw.update(residual=torch.randn(M, N, device=dev).bfloat16(), ldr=N, res_scale=1.0)
    if gate:
        kw.update(gate=torch.randn(M, N, device=dev).bfloat16(), ldg=N, gate_scale=1.1)
    return lambda: ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw), C


def timed(run):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


lib().js2t_gemm_p192_mode(1)
for name, M, N, K, kw in SHAPES:
    run, C = make(M, N, K, **kw)
    res, outs = {2: [], 3: [], 4: []}, {}
    for rnd in range(4):
        for ring in (3, 2, 4):
            lib().js2t_gemm_p192_ring(ring)
            run()
            torch.cuda.synchronize()
            if rnd == 0:
                outs[ring] = C.clone()
            res[ring].append(timed(run))
    same = torch.equal(outs[2], outs[3]) and torch.equal(outs[4], outs[3])
    fl = 2.0 * M * N * K
    a, b, c = min(res[3]), min(res[2]), min(res[4])
    print(f"{name:30s} M={M} N={N:5d} K={K:5d}  ring3 {a:7.1f} us {fl / a / 1e6:6.0f} TF | ring2 {b:7.1f} us {fl / b / 1e6:6.0f} TF  "
          f"x{a / b:5.2f} | split {c:7.1f} us {fl / c / 1e6:6.0f} TF x{a / c:5.2f}  identical={same}", flush=True)
