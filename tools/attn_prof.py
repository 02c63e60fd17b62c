"""One fused-attention forward + backward at the encoder self-attention shape (for rocprofv3 --pmc runs)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, T, dh = 32, 4, 375, 128
d = H * dh
qkv = torch.randn(B * T, 3 * d, device=dev).bfloat16()
dout = torch.randn(B * T, d, device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
mask = torch.ones(B, 1, T, dtype=torch.bool, device=dev)
rng = ops.dropout_rng(dev)
for _ in range(3):
    out, lse = ops.flash_attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, B, H, T, T, dh, mask, 0.1, rng, 5)
    ops.flash_attn_bwd(dout, out, lse, qkv, 2 * d, qkv, 0, qkv, d, dqkv, 2 * d, dqkv, 0, dqkv, d, B, H, T, T, dh, mask, 0.1, rng, 5)
torch.cuda.synchronize()
