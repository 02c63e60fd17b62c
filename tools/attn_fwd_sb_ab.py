import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import torch
from joeys2t_amd import ops
from joeys2t_amd._lib import lib
dev = torch.device("cuda:0")
B, H, T, dh = 32, 4, 375, 128
d = H * dh
qkv = torch.randn(B * T, 3 * d, device=dev).bfloat16()
mask = torch.ones(B, 1, T, dtype=torch.bool, device=dev)
rng = ops.dropout_rng(dev)
def timed(fn, reps=200):
    for _ in range(10): fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best
for sb in (-1, 0, 1, -1, 1):
    lib().js2t_debug_attn_fwd_sb(C.c_int(sb))
    t = timed(lambda: ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, dh, mask, 0.1, rng, 5))
    print(f"fwd sb={sb}: {t:.1f} us")
lib().js2t_debug_attn_fwd_sb(C.c_int(-1))
