"""256x128 three-slot-ring kernel for grouped weight gradients (gemm_bf16_wg256_kernel) against the two-stage 128x128 kernel:
bit-identity of products, row sums (to atomics' order) and tile sums of squares, then interleaved timing on the train step's groups.
usage (GPU box): python tools/wg256_bench.py"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")


def operands(n, M, N, K, seed, sparse=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    As, Bs = [], []
    for _ in range(n):
        a = torch.randn(K, M, generator=g)
        if sparse:  # a ReLU-gated gradient: half of it zero
            a = a * (torch.rand(K, M, generator=g) > 0.5)
        As.append(a.bfloat16().to(dev))
        Bs.append(torch.randn(K, N, generator=g).bfloat16().to(dev))
    return As, Bs


def run(mode, As, Bs, M, N, K, beta, rowsum, sumsq, c0, split_k=1):
    lib().js2t_gemm_wg256_mode(C.c_int(mode))
    n = len(As)
    Cs = [c.clone() for c in c0]
    rs = [torch.zeros(M, device=dev) for _ in range(n)] if rowsum else None
    ss = torch.full((ops.grouped_blocks(M, N, n),), float("nan"), device=dev) if sumsq else None
    ops.gemm_grouped(As, Bs, Cs, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, beta=beta, alpha=0.5, a_rowsums=rs, sumsq_partial=ss, split_k=split_k)
    torch.cuda.synchronize()
    lib().js2t_gemm_wg256_mode(C.c_int(-1))
    return Cs, rs, ss


ok = True
for (n, M, N, K, beta, rowsum, sumsq) in [(3, 512, 256, 1000, 0.0, True, True), (2, 256, 128, 200, 1.0, False, False), (5, 768, 384, 4130, 1.0, True, True),
                                          (16, 2048, 512, 12000, 0.0, True, True), (16, 512, 2048, 12000, 1.0, True, False)]:
    As, Bs = operands(n, M, N, K, 7 + M)
    c0 = [torch.randn(M, N, device=dev) for _ in range(n)]
    C1, r1, s1 = run(1, As, Bs, M, N, K, beta, rowsum, sumsq, c0)
    C0, r0, s0 = run(0, As, Bs, M, N, K, beta, rowsum, sumsq, c0)
    same = all(torch.equal(a, b) for a, b in zip(C1, C0))
    ref = 0.5 * (As[0].float().T @ Bs[0].float()) + beta * c0[0]
    err = ((C1[0] - ref).norm() / ref.norm()).item()
    rs_ok = True
    if rowsum:
        rs_ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-3) for a, b in zip(r1, r0)) and torch.allclose(r1[0], As[0].float().sum(0), rtol=1e-4, atol=1e-2)
    ss_ok = True
    if sumsq:
        tot1, tot0 = s1.double().sum().item(), s0.double().sum().item()
        want = sum((c.double() ** 2).sum().item() for c in C1)
        ss_ok = abs(tot1 - want) <= 1e-5 * want and abs(tot0 - want) <= 1e-5 * want and bool(torch.isfinite(s1).all())
    print(f"check {n:2d} x dW[{M},{N}] over {K}: beta={beta} identical={same} vs_f32 {err:.2e} rowsum={rs_ok} sumsq={ss_ok}", flush=True)
    ok &= same and err < 1e-5 and rs_ok and ss_ok
for (n, M, N, K, sk) in [(16, 512, 512, 12000, 2), (8, 1024, 512, 12000, 2), (3, 256, 256, 2592, 4), (2, 512, 128, 1100, 3)]:
    As, Bs = operands(n, M, N, K, 11 + M)
    c0 = [torch.zeros(M, N, device=dev) for _ in range(n)]
    C1, r1, _ = run(1, As, Bs, M, N, K, 0.0, True, False, c0, split_k=sk)
    C0, r0, _ = run(0, As, Bs, M, N, K, 0.0, True, False, c0, split_k=sk)
    same = all(torch.equal(a, b) for a, b in zip(C1, C0))
    close = all(torch.allclose(a, b, rtol=1e-5, atol=1e-3) for a, b in zip(C1, C0))
    ref = 0.5 * (As[-1].float().T @ Bs[-1].float())
    err = ((C1[-1] - ref).norm() / ref.norm()).item()
    rs_ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-3) for a, b in zip(r1, r0)) and torch.allclose(r1[0], As[0].float().sum(0), rtol=1e-4, atol=1e-2)
    print(f"check {n:2d} x dW[{M},{N}] over {K} in {sk} slices: identical={same} close={close} vs_f32 {err:.2e} rowsum={rs_ok}", flush=True)
    ok &= close and err < 1e-5 and rs_ok and (same or sk > 2)
print("ALL OK" if ok else "MISMATCH", flush=True)


def timed(fn, reps=20):
    for _ in range(3):
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


K = 12000
for name, n, M, N, sparse in [("FFN1 16 x dW[2048,512]", 16, 2048, 512, True), ("FFN2 16 x dW[512,2048]", 16, 512, 2048, False),
                              ("QKV 16 x dW[1536,512]", 16, 1536, 512, False), ("out 16 x dW[512,512]", 16, 512, 512, False),
                              ("mem K|V 8 x dW[1024,512]", 8, 1024, 512, False)]:
    As, Bs = operands(n, M, N, K, 3, sparse)
    Cs = [torch.zeros(M, N, device=dev) for _ in range(n)]
    rs = [torch.zeros(M, device=dev) for _ in range(n)]
    ss = torch.zeros(ops.grouped_blocks(M, N, n), device=dev)
    fl = 2e-6 * n * M * N * K
    for sk in ((1,) if n * (M // 256) * (N // 128) > 128 else (1, 2)):
        us = {}
        kw = dict(a_rowsums=rs, sumsq_partial=ss) if sk == 1 else dict(a_rowsums=rs, split_k=sk)
        for rnd in range(2):
            for mode in (0, 1):
                lib().js2t_gemm_wg256_mode(C.c_int(mode))
                t = timed(lambda: ops.gemm_grouped(As, Bs, Cs, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, **kw))
                us[mode] = min(us.get(mode, 1e9), t)
        lib().js2t_gemm_wg256_mode(C.c_int(-1))
        print(f"{name:26s} split {sk}: 128x128 {us[0]:7.1f} us {fl / us[0]:5.0f} TF/s | 256x128 {us[1]:7.1f} us {fl / us[1]:5.0f} TF/s | x{us[0] / us[1]:.2f}",
              flush=True)
