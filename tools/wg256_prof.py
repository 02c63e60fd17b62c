"""Tick breakdown of a K stage of the 256x128 grouped weight-gradient kernel (wave 0 of block 0).
Needs the instrumented build:  JS2T_HIPCC_EXTRA=-DJS2T_P192_PROF python -m joeys2t_amd._build --force
usage: python tools/wg256_prof.py rows cols tokens members"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

rows, cols, T, L = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
lib().js2t_gemm_wg256_mode(1)
dzs = [torch.randn(T, rows, device=dev).bfloat16() for _ in range(L)]
xs = [torch.randn(T, cols, device=dev).bfloat16() for _ in range(L)]
Cs = [torch.zeros(rows, cols, device=dev) for _ in range(L)]
for _ in range(3):
    ops.gemm_grouped(dzs, xs, Cs, M=rows, N=cols, K=T, lda=rows, ldb=cols, ldc=cols, split_k=1, beta=0.0)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
fn = lib().js2t_debug_p192_prof
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
steps = -(-T // 64)
names = ["multiplying wave 0: first half: 16 MFMA + 16 reads", "  lgkmcnt wait", "  barrier", "  second half: 16 MFMA + 16 reads",
         "requesting wave 8: vmcnt wait (stage s + 1)", "  barrier", "  12 requests"]
print(f"{L} x dW[{rows},{cols}] over {T}: {(rows // 256) * (cols // 128) * L} tiles")
for i, n in enumerate(names):
    print(f"{n:52s} {out[i] / steps:9.1f} ticks")
print(f"per stage {sum(out[:4]) / steps:9.1f} / {sum(out[4:7]) / steps:9.1f} ticks (s_memtime, 100 MHz: x 24 = core clocks at 2.4 GHz; MFMA-bound: 1024 clocks per SIMD)")
