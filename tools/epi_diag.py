"""Round-6 diagnostic: the persistent 192x128 kernel's fused epilogues against the register-staged kernel on one shape, per variant:
where do they differ (rows / columns / values)?  usage: [JS2T_LIB=...] python tools/epi_diag.py [M N K]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (7000, 1536, 256)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


A = rnd(M, K, seed=1).bfloat16().to(dev)
B = rnd(N, K, seed=2).bfloat16().to(dev)
bias = rnd(N, seed=3).to(dev)
res = rnd(M, N, seed=4).bfloat16().to(dev)
rng = ops.dropout_rng(dev)
variants = {"relu+drop": dict(bias=bias, act="relu", dropout_p=0.1, rng=rng, rng_stream=7),
            "drop+res": dict(bias=bias, dropout_p=0.2, rng=rng, rng_stream=9, residual=res, ldr=N, res_scale=0.7),
            "gate": dict(gate=res, ldg=N, gate_scale=1.3, alpha=0.5)}
for name, kw in variants.items():
    for rep in range(3):
        out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(A, B, out1, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)
        lib().js2t_gemm_force_regstage(1)
        try:
            ops.gemm(A, B, out2, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)
        finally:
            lib().js2t_gemm_force_regstage(0)
        torch.cuda.synchronize()
        d = (out1.float() - out2.float()).abs()
        bad = (d > 2e-2 + 2e-2 * out2.float().abs()).nonzero()
        print(f"{name} rep {rep}: {bad.shape[0]} beyond tolerance; zero-pattern differences {int(((out1 == 0) != (out2 == 0)).sum())}")
        for r, c in bad[:12].tolist():
            print(f"   ({r},{c}) p192 {out1[r, c].item():.4f} regstage {out2[r, c].item():.4f}")
