"""Would one product for all decoder layers' cross-attention K|V pay?  LS100 shapes: encoder states [12000, 512], six layers.
forward: 6 x (12000 x 1024 x 512, bias) against 1 x (12000 x 6144 x 512, bias);
backward (input gradient): 6 x (12000 x 512 x 1024, running sum added in the epilogue) against 1 x (12000 x 512 x 6144).
usage: python tools/kv_group_probe.py [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M, d, L = 12000, 512, 6
torch.manual_seed(0)
mem = torch.randn(M, d, device=dev).bfloat16()
w = (torch.randn(L * 2 * d, d, device=dev) * 0.04).bfloat16()
w_t = w.t().contiguous()  # [d, L*2d]
b = torch.randn(L * 2 * d, device=dev)
kv6 = [torch.empty(M, 2 * d, device=dev, dtype=torch.bfloat16) for _ in range(L)]
kv1 = torch.empty(M, L * 2 * d, device=dev, dtype=torch.bfloat16)
dkv = (torch.randn(M, L * 2 * d, device=dev) * 0.1).bfloat16()
dkv6 = [dkv[:, i * 2 * d:(i + 1) * 2 * d].contiguous() for i in range(L)]
dm = [torch.empty(M, d, device=dev, dtype=torch.bfloat16) for _ in range(L)]
dm1 = torch.empty(M, d, device=dev, dtype=torch.bfloat16)


def fwd6():
    for i in range(L):
        ops.gemm(mem, w[i * 2 * d:(i + 1) * 2 * d], kv6[i], M=M, N=2 * d, K=d, lda=d, ldb=d, ldc=2 * d, bias=b[i * 2 * d:(i + 1) * 2 * d])


def fwd1():
    ops.gemm(mem, w, kv1, M=M, N=L * 2 * d, K=d, lda=d, ldb=d, ldc=L * 2 * d, bias=b)


def bwd6():
    for i in range(L):
        wt = w_t[:, i * 2 * d:(i + 1) * 2 * d]
        extra = {} if i == 0 else dict(residual=dm[i - 1], ldr=d, res_scale=1.0)
        ops.gemm(dkv6[i], wt, dm[i], M=M, N=d, K=2 * d, lda=2 * d, ldb=w_t.stride(0), ldc=d, **extra)


def bwd6_strided():  # the same six, reading their operand out of the shared [M, L*2d] buffer
    for i in range(L):
        wt = w_t[:, i * 2 * d:(i + 1) * 2 * d]
        extra = {} if i == 0 else dict(residual=dm[i - 1], ldr=d, res_scale=1.0)
        ops.gemm(dkv[:, i * 2 * d:(i + 1) * 2 * d], wt, dm[i], M=M, N=d, K=2 * d, lda=L * 2 * d, ldb=w_t.stride(0), ldc=d, **extra)


def bwd1():
    ops.gemm(dkv, w_t, dm1, M=M, N=d, K=L * 2 * d, lda=L * 2 * d, ldb=L * 2 * d, ldc=d)


for name, fn in (("forward, six products", fwd6), ("forward, one product", fwd1), ("input gradient, six chained", bwd6),
                 ("input gradient, six chained (strided operand)", bwd6_strided), ("input gradient, one product", bwd1)):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"{name:48s} {s.elapsed_time(e) / reps * 1e3:8.1f} us")
a = torch.cat([k.float() for k in kv6], dim=1)
print("forward max |diff|", (a - kv1.float()).abs().max().item())
print("input gradient rel diff", ((dm[-1].float() - dm1.float()).norm() / dm1.float().norm()).item())
