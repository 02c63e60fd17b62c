"""How much of the attention backward is tail: time of the two passes against the batch size (blocks = B * 4 heads * 6 tiles; 512 resident)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, T, dh = 4, 375, 128
d = H * dh
for B in (16, 21, 24, 28, 32, 36, 40, 42, 48, 64):
    qkv = torch.randn(B * T, 3 * d, device=dev).bfloat16()
    dout = torch.randn(B * T, d, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    mask = torch.ones(B, 1, T, dtype=torch.bool, device=dev)
    rng = ops.dropout_rng(dev)
    out, lse = ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, dh, mask, 0.1, rng, 5)

    def fwd():
        ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, dh, mask, 0.1, rng, 5)

    def bwd():
        ops.flash_attn_bwd(dout, out, lse, qkv, 2 * d, qkv, 0, qkv, d, dqkv, 2 * d, dqkv, 0, dqkv, d, B, H, T, T, dh, mask, 0.1, rng, 5)

    res = []
    for fn in (fwd, bwd):
        for _ in range(5):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(200):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 200 * 1e3)
    print(f"B={B:3d} blocks per pass {B * H * 6:5d}   fwd {res[0]:6.1f} us ({res[0] / B:5.2f} / utt)   bwd {res[1]:6.1f} us ({res[1] / B:5.2f} / utt)")
