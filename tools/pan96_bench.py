"""Panel-resident kernel (gemm_panel.hip) against the persistent 192x128 kernels on the encoder layer's K = 512 products:
bit-identity first (also on ragged shapes), then interleaved timing of both forms.
usage (GPU box): python tools/pan96_bench.py [--quick]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from joeys2t_amd import ops  # noqa: E402
from joeys2t_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
quick = "--quick" in sys.argv


def epilogue(kind, M, N, K, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    kw = {}
    if kind in ("bias", "qkv", "ffn1", "ffn1_eval", "ffn1_nofold"):
        kw["bias"] = torch.randn(N, generator=g).to(dev)
    if kind in ("ffn1", "ffn1_eval", "ffn1_nofold"):
        kw["act"] = "relu"
    if kind in ("ffn1", "ffn1_nofold"):
        kw.update(dropout_p=0.1, rng=ops.dropout_rng(dev), rng_stream=5)
    if kind in ("qkv", "ffn1", "ffn1_eval"):
        part = (torch.randn(M, 8, 2, generator=g).abs() * 30.0 + 40.0).to(dev)  # sums / sums of squares of a plausible row
        part[:, :, 0] *= 0.01
        kw["ln"] = (part, 1e-6, torch.zeros(M, device=dev), torch.zeros(M, device=dev))
    if kind == "gate":
        kw.update(gate=torch.randn(M, N, generator=g).bfloat16().to(dev), ldg=N, gate_scale=1.0 / 0.9)
    return kw


def run(A, B, Cc, M, N, K, kw):
    ops.gemm(A, B, Cc, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)


def check(M, N, K, kind, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = torch.randn(M, K, generator=g).bfloat16().to(dev)
    B = (torch.randn(N, K, generator=g) / K**0.5).bfloat16().to(dev)
    kw = epilogue(kind, M, N, K, seed)
    outs, stats = [], []
    for mode in (1, 0):
        lib().js2t_gemm_panel_mode(C.c_int(mode))
        lib().js2t_gemm_p192_mode(C.c_int(1))
        Cc = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        if "ln" in kw:
            kw["ln"][2].zero_(), kw["ln"][3].zero_()
        run(A, B, Cc, M, N, K, kw)
        torch.cuda.synchronize()
        outs.append(Cc)
        if "ln" in kw:
            stats.append((kw["ln"][2].clone(), kw["ln"][3].clone()))
    lib().js2t_gemm_panel_mode(C.c_int(-1))
    lib().js2t_gemm_p192_mode(C.c_int(-1))
    same = torch.equal(outs[0], outs[1])
    fin = bool(torch.isfinite(outs[0].float()).all())
    st = all(torch.equal(a, b) for a, b in zip(stats[0], stats[1])) if stats else True
    if not same:
        bad = (outs[0] != outs[1]).nonzero()
        print(f"   first mismatch at {bad[0].tolist()} of {bad.shape[0]}: {outs[0][tuple(bad[0])].item()} vs {outs[1][tuple(bad[0])].item()}; "
              f"rows {bad[:, 0].min().item()}..{bad[:, 0].max().item()} cols {bad[:, 1].min().item()}..{bad[:, 1].max().item()}")
    print(f"check {M:6d} x {N:5d} x {K:4d} {kind:12s}: identical={same} finite={fin} ln_stats={st}", flush=True)
    return same and fin and st


def timed(fn, reps=100):
    for _ in range(5):
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


ok = True
cases = [(12000, 1536, 512, "qkv"), (12000, 2048, 512, "ffn1"), (12000, 2048, 512, "gate"), (12000, 1536, 512, "plain"),
         (11975, 1000, 256, "bias"), (4111, 2048, 512, "ffn1_eval"), (8200, 200, 384, "gate"), (12000, 6144, 512, "bias"),
         (9000, 5000, 512, "bias"), (3000, 2048, 128, "ffn1_nofold"), (97, 136, 256, "plain"), (12000, 512, 512, "bias")]
if "--time-only" not in sys.argv:
    for i, (M, N, K, kind) in enumerate(cases if not quick else cases[:4]):
        ok &= check(M, N, K, kind, i)
    print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)

T = 12000
tot = {0: 0.0, 1: 0.0}
for name, N, K, kind in [("QKV", 1536, 512, "qkv"), ("FFN1", 2048, 512, "ffn1"), ("dFFN2", 2048, 512, "gate"), ("mem K|V", 6144, 512, "bias"),
                         ("out-proj", 512, 512, "bias"), ("CTC proj", 5000, 512, "bias")]:
    A = torch.randn(T, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    Cc = torch.zeros(T, N, device=dev, dtype=torch.bfloat16)
    kw = epilogue(kind, T, N, K)
    us = {}
    for rnd in range(2):
        for mode in (0, 1):
            lib().js2t_gemm_panel_mode(C.c_int(mode))
            t = timed(lambda: run(A, B, Cc, T, N, K, kw))
            us[mode] = min(us.get(mode, 1e9), t)
    lib().js2t_gemm_panel_mode(C.c_int(-1))
    fl = 2e-6 * T * N * K
    if name in ("QKV", "FFN1", "dFFN2"):
        tot[0] += us[0]
        tot[1] += us[1]
    print(f"{name:9s} {T} x {N} x {K} {kind:6s}: persistent {us[0]:6.1f} us {fl / us[0]:5.0f} TF/s | panel {us[1]:6.1f} us {fl / us[1]:5.0f} TF/s | x{us[0] / us[1]:.2f}",
          flush=True)
print(f"QKV + FFN1 + dFFN2 per layer: persistent {tot[0]:.1f} us, panel {tot[1]:.1f} us")
