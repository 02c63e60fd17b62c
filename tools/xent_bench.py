"""Label-smoothed cross-entropy kernels at the LS100 decoder's size (2592 x 5000 f32 logits, bf16 gradient): isolated timing.
usage: python tools/xent_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
R, V = 2592, 5000
x = torch.randn(R, V, device=dev)
trg = torch.randint(4, V, (R, ), device=dev)
g = torch.ones((), device=dev)


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best


loss_rows, correct, lse = ops.xent_fwd(x, trg, 1, 0.1)
tf = timed(lambda: ops.xent_fwd(x, trg, 1, 0.1))
tb = timed(lambda: ops.xent_bwd(x, trg, lse, g, 1.0, 1, 0.1, out_dtype=torch.bfloat16))
tb32 = timed(lambda: ops.xent_bwd(x, trg, lse, g, 1.0, 1, 0.1))
print(f"xent_fwd {tf:6.1f} us ({R * V * 4 / tf / 1e6:5.2f} TB/s)   xent_bwd -> bf16 {tb:6.1f} us ({R * V * 6 / tb / 1e6:5.2f} TB/s)   "
      f"xent_bwd -> f32 {tb32:6.1f} us ({R * V * 8 / tb32 / 1e6:5.2f} TB/s)")
