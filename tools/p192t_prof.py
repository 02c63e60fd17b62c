"""Cycle breakdown of a K step of the reduction-major persistent GEMM (instrumented build, see tools/p192_prof.py).
usage: python tools/p192t_prof.py rows cols tokens members"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

rows, cols, T, L = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
lib().js2t_gemm_p192_mode(1)
dzs = [torch.randn(T, rows, device=dev).bfloat16() for _ in range(L)]
xs = [torch.randn(T, cols, device=dev).bfloat16() for _ in range(L)]
Cs = [torch.zeros(rows, cols, device=dev) for _ in range(L)]
for _ in range(2):
    ops.gemm_grouped(dzs, xs, Cs, M=rows, N=cols, K=T, lda=rows, ldb=cols, ldc=cols, split_k=1, beta=1.0)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
fn = lib().js2t_debug_p192_prof
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
tiles = -(-rows // 192) * (cols // 128) * L
per_block = -(-tiles // min(tiles, 256))
steps = per_block * -(-T // 64)
names = ["first half: MFMA + reads + 5 requests", "vmcnt wait", "lgkmcnt wait", "barrier", "issue_begin", "second half", "epilogues (whole block)"]
for i, n in enumerate(names):
    den = 1 if i == 6 else steps
    print(f"{n:40s} {out[i] / den:11.1f} cycles")
print(f"per step {sum(out[:6]) / steps:9.1f}   (MFMA-bound: 768)")
