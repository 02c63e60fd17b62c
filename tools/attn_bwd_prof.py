"""Tick breakdown of a query tile of the dK/dV pass of flash_bwd (wave 0 of block 0; instrumented build -DJS2T_ATTN_PROF,
JS2T_LIB pointing at it).  usage: python tools/attn_bwd_prof.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
B, H, T, dh = 32, 4, 375, 128
d = H * dh
qkv = torch.randn(B * T, 3 * d, device=dev).bfloat16()
mask = torch.ones(B, 1, T, dtype=torch.bool, device=dev)
rng = ops.dropout_rng(dev)
out, lse = ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, dh, mask, 0.1, rng, 5)
go = torch.randn(B * T, d, device=dev).bfloat16()
dpart = (go.float() * out.float()).view(B * T, d // 64, 64).sum(-1).contiguous()
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.flash_attn_bwd(go, out, lse, qkv, 2 * d, qkv, 0, qkv, d, dqkv, 2 * d, dqkv, 0, dqkv, d, B, H, T, T, dh, mask, 0.1, rng, 5,
                       delta_partial=dpart)
torch.cuda.synchronize()
res = (ctypes.c_ulonglong * 8)()
fn = lib().js2t_debug_attn_prof
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(res) == 0
tiles = -(-T // 64)
names = ["per-query scalars to LDS", "vmcnt wait (prefetch landed?)", "barrier", "request next Q / dO (8 pieces) + scalars",
         "S, dP (64 MFMA) + exp / dropout / dS + pack", "dV, dK (64 MFMA + transposing reads)"]
for i, n in enumerate(names):
    print(f"{n:48s} {res[i] / tiles:9.1f} ticks / tile")
print(f"per tile {sum(res[:6]) / tiles:9.1f}   (MFMA-bound: 2048 for the SIMD's two waves)")
