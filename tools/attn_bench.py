"""Micro-benchmark of the fused attention kernels on the LS100 train-step shapes (run on the GPU box).
usage: python tools/attn_bench.py [reps]   (profile with rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

import os
from joeys2t_amd._lib import lib  # noqa: E402
if "ATTN_FWD_SB" in os.environ:  # 0 / 1: force the double- / single-buffered forward kernel (default: by grid size)
    lib().js2t_debug_attn_fwd_sb(int(os.environ["ATTN_FWD_SB"]))
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
P_DROP = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1


def bench(name, B, H, Tq, Tk, p=0.1, causal=False):
    dh = 128
    d = H * dh
    qkv = torch.randn(B * Tq, 3 * d, device=dev).bfloat16()
    kv = qkv if Tk == Tq else torch.randn(B * Tk, 3 * d, device=dev).bfloat16()
    dout = torch.randn(B * Tq, d, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    dkv = dqkv if Tk == Tq else torch.empty_like(kv)
    mask = torch.ones(B, Tq if causal else 1, Tk, dtype=torch.bool, device=dev)
    if causal:
        mask = torch.tril(mask)
    elif os.environ.get("ATTN_NO_MASK"):  # how much the key-mask staging costs: no mask at all
        mask = None
    rng = ops.dropout_rng(dev)

    def fwd():
        return ops.flash_attn_fwd(qkv, 2 * d, kv, 0, kv, d, B, H, Tq, Tk, dh, mask, p, rng, 5)

    out, lse = fwd()

    def bwd():
        ops.flash_attn_bwd(dout, out, lse, qkv, 2 * d, kv, 0, kv, d, dqkv, 2 * d, dkv, 0, dkv, d, B, H, Tq, Tk, dh, mask, p, rng, 5)

    # delta = rowsum(dO * O) handed over as partial sums (in the step: from the output projection's input-gradient epilogue): both
    # backward passes as ONE grid
    part = (dout.float() * out.float()).view(B * Tq, d // 64, 64).sum(-1).contiguous()

    def bwd1():
        ops.flash_attn_bwd(dout, out, lse, qkv, 2 * d, kv, 0, kv, d, dqkv, 2 * d, dkv, 0, dkv, d, B, H, Tq, Tk, dh, mask, p, rng, 5,
                           delta_partial=part)

    for fn, nm, gemms in ((fwd, "fwd", 2), (bwd, "bwd (two launches)", 7), (bwd1, "bwd (one grid)", 7)):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / reps
        fl = gemms * 2.0 * B * H * Tq * Tk * dh
        print(f"{name:24s} {nm:18s} B={B} H={H} Tq={Tq} Tk={Tk} {us:8.1f} us {fl / us / 1e6:7.1f} TF", flush=True)


bench("encoder self", 32, 4, 375, 375, p=P_DROP)
bench("decoder self (causal)", 32, 4, 81, 81, p=P_DROP, causal=True)
bench("decoder cross", 32, 4, 81, 375, p=P_DROP)
