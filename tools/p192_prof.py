"""Where a K step of the persistent 192x128 GEMM spends its cycles (wave 0 of block 0, s_memtime deltas).
Needs the instrumented build:  JS2T_HIPCC_EXTRA=-DJS2T_P192_PROF python -m joeys2t_amd._build --force
usage: python tools/p192_prof.py M N K"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "plain"  # plain | gate | res
dev = torch.device("cuda:0")
lib().js2t_gemm_p192_mode(1)
A = torch.randn(M, K, device=dev).bfloat16()
B = torch.randn(N, K, device=dev).bfloat16()
C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
kw = {}
if mode == "gate":
    kw = dict(gate=(torch.randn(M, N, device=dev) * (torch.rand(M, N, device=dev) > 0.5)).bfloat16(), ldg=N, gate_scale=1.1)
elif mode == "res":
    kw = dict(residual=torch.randn(M, N, device=dev).bfloat16(), ldr=N, res_scale=1.0, bias=torch.randn(N, device=dev),
              dropout_p=0.1, rng=ops.dropout_rng(dev), rng_stream=3)
fn2 = lib().js2t_debug_p192_prof2
fn2.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
for it in range(3):
    if it == 2:
        torch.cuda.synchronize()
        fn2(None, 1)
    ops.gemm(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
fn = lib().js2t_debug_p192_prof
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
tiles = -(-M // 192) * -(-N // 128)
per_block = -(-tiles // min(tiles, 256))
steps = per_block * (K // 64)
names = ["top: reads(F1)+MFMA(F0) issue", "vmcnt wait", "lgkmcnt wait", "barrier", "DMA issue", "reads(F0)+MFMA(F1) issue", "epilogue (per tile)"]
tot = 0
for i, n in enumerate(names):
    den = per_block if i == 6 else steps
    print(f"{n:34s} {out[i] / den:9.1f} cycles")
    tot += out[i]
print(f"total per step {tot / steps:9.1f}   (MFMA-bound: 768)")
out2 = (ctypes.c_ulonglong * 8)()
assert fn2(out2, 0) == 0
enames = ["issue rows of block 0", "wait block 0", "compute + store block 0", "wait block 1", "compute + store block 1", "wait block 2", "compute + store block 2"]
print("epilogue of wave 0 (cycles per tile):")
for i, n in enumerate(enames):
    print(f"  {n:28s} {out2[i] / per_block:9.1f}")
