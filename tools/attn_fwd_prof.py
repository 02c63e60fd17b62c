"""Cycle breakdown of a key tile of flash_fwd (wave 0 of block 0; instrumented build -DJS2T_ATTN_PROF).
usage: python tools/attn_fwd_prof.py"""
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
for _ in range(3):
    ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, dh, mask, 0.1, rng, 5)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
fn = lib().js2t_debug_attn_prof
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
tiles = -(-T // 64)
names = ["vmcnt wait (prefetch landed?)", "barrier", "request next K/V", "K reads + QK^T", "softmax + dropout + pack", "V reads + PV"]
for i, n in enumerate(names):
    print(f"{n:32s} {out[i] / tiles:9.1f} cycles / tile")
print(f"per tile {sum(out[:6]) / tiles:9.1f}   (MFMA-bound: 512)")
print(f"prologue (query fragments, key mask, first request) {out[6]:9d} cycles, epilogue (stores) {out[7]:9d} cycles, "
      f"{tiles} tiles {sum(out[:6]):9d} cycles")
