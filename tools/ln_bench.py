"""LayerNorm backward: cost of the parameter-gradient atomics and of the fused dropout output (12000 x 512 bf16).
usage: python tools/ln_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows, D = 12000, 512
x = torch.randn(rows, D, device=dev).bfloat16()
dy = torch.randn(rows, D, device=dev).bfloat16()
add = torch.randn(rows, D, device=dev).bfloat16()
gamma = torch.ones(D, device=dev)
_, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros_like(gamma), 1e-6)
big = torch.zeros(64 * 2048 + 2048, device=dev)  # room for the JS2T_LN_SPREAD experiment (copies 2048 floats apart)
dg, db = big[:D], big[1024:1024 + D]
rng = ops.dropout_rng(dev)


def t(fn, n=200):
    for _ in range(10):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


print("dx only                      %.2f us" % t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, need_param_grads=False, add=add)))
print("dx + atomics onto gradients  %.2f us" % t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, add=add, grad_out=(dg, db))))
print("dx + partial slab + reduce   %.2f us" % t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, add=add)))
print("dx + atomics + dropout copy  %.2f us" % t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, add=add, grad_out=(dg, db), drop=(0.1, rng, 3))))
