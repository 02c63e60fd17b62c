"""LayerNorm forward: plain against the e4m3-emitting variant (run on the GPU box)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows, D = 12000, 512
x = torch.randn(rows, D, device=dev).bfloat16()
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
st = ops.new_fp8_state(x)


def t(fn, reps=200):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


print("plain            ", round(t(lambda: ops.layernorm_fwd(x, g, b, 1e-6)), 2), "us")
print("fp8 + bf16 output", round(t(lambda: ops.layernorm_fwd_fp8(x, g, b, 1e-6, st, want_y=True)), 2), "us")
print("fp8 only         ", round(t(lambda: ops.layernorm_fwd_fp8(x, g, b, 1e-6, st, want_y=False)), 2), "us")
