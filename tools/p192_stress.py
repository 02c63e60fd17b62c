"""Race screen for the persistent 192x128 GEMM: many random shapes (partial row / column tiles, partial K stages, several
tiles per block) against the vendor library's result, each shape launched several times back to back.  A stage read
before its DMA landed shows up as a tile that is off by far more than bf16 rounding.
usage: python tools/p192_stress.py [iterations]"""
import random
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
random.seed(1)
lib().js2t_gemm_p192_mode(1)
worst = 0.0
for it in range(iters):
    M = random.choice([192, 500, 1000, 2592, 7000, 12000, 20000, random.randint(1, 30000)])
    N = random.choice([128, 256, 512, 1000, 1536, 2048, 8 * random.randint(16, 300)])
    K = random.choice([192, 256, 400, 512, 1280, 2048, 8 * random.randint(24, 400)])
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).bfloat16()
    ref = (A.float() @ B.float().t())
    kws = [dict(), dict(bias=bias), dict(residual=res, ldr=N, res_scale=1.0, bias=bias), dict(gate=res, ldg=N, gate_scale=1.0)]
    kw = kws[it % 4]
    want = ref + (bias if "bias" in kw else 0)
    if "residual" in kw:
        want = want + res.float()
    if "gate" in kw:
        want = torch.where(res.float() > 0, want, torch.zeros_like(want))
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for rep in range(3):
        C.fill_(float("nan"))
        ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)
        err = (C.float() - want).abs().max().item()
        tol = 0.02 * (K ** 0.5) + 0.05 * want.abs().max().item() / 8
        worst = max(worst, err / tol)
        if not (err <= tol):
            print(f"MISMATCH it={it} rep={rep} M={M} N={N} K={K} kw={list(kw)} err={err} tol={tol}", flush=True)
            sys.exit(1)
    if it % 25 == 0:
        print(f"it {it}: M={M} N={N} K={K} ok, worst err/tol so far {worst:.3f}", flush=True)
print("no mismatch in", iters, "shapes x 3 launches; worst err/tol", round(worst, 3))
