"""Round 6: js2t_ctc_prefix_step at the decode's size (32 utterances x beam 5, 8 candidates, T' = 375, V = 5000): the thread-per-pair
kernel of round 5 against the block-per-hypothesis kernel with LDS-staged operands, and the other launches a joint CTC / attention step
adds (js2t_beam_pick, the selection, the gathers of the winners' variables).  usage: python tools/ctc_prefix_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(12)
B, k, T, V, C = 32, 5, 375, 5000, 8
blank, eos = 2, 3
logp = torch.log_softmax(torch.randn(B, T, V, generator=g) * 2.0, -1).to(dev)
in_len = torch.full((B, ), T, dtype=torch.int64, device=dev)
rows = B * k
r_prev = ops.ctc_prefix_init(logp, in_len, k, blank)
psi_prev = torch.zeros(rows, device=dev)
last = torch.full((rows, ), blank, dtype=torch.int64, device=dev)
cand = torch.randint(4, V, (rows, C), generator=g).to(dev)
cand_lp = torch.log_softmax(torch.randn(rows, C, generator=g), -1).to(dev)
logits = torch.randn(rows, V, generator=g).to(dev)


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n_out in (1, 40):
    for mode, name in ((1, "thread per pair (round 5)"), (0, "block per hypothesis, LDS-staged")):
        lib().js2t_debug_ctc_prefix_thread_per_pair(mode)
        us = timed(lambda: ops.ctc_prefix_step(logp, in_len, r_prev, last, cand, cand_lp, psi_prev, n_out, k, blank, eos, 0.3))
        print(f"ctc_prefix_step n_out {n_out:3d} {name:36s} {us:8.1f} us", flush=True)
lib().js2t_debug_ctc_prefix_thread_per_pair(0)
print(f"beam_pick {timed(lambda: ops.beam_pick(logits, C, [1, 2])):8.1f} us")
local = torch.randn(rows, C, device=dev)
print(f"beam_step over beam x candidates {timed(lambda: ops.beam_step(local, psi_prev, B, k, [], 1.0, normalized=True)):8.1f} us")
r_new = torch.randn(rows, C, T, 2, device=dev)
idx = torch.randint(0, rows * C, (rows, ), device=dev)
print(f"gather of the winners' variables {timed(lambda: r_new.view(rows * C, T, 2).index_select(0, idx)):8.1f} us")
