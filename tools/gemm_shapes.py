"""In-situ per-shape GEMM table of the bench train step: HIP-event time, TFLOP/s and HBM-roofline time per shape."""
import collections
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from joeys2t_amd import ops  # noqa: E402


class ShapeTimer(bench.GemmTimer):
    pass


def main():
    device = torch.device("cuda", 0)
    eager_step = bench.build_step(device, 1)[0]
    for _ in range(3):
        eager_step()
    timer = ShapeTimer()
    orig = ops.gemm

    def gemm(A, B, C_out, **kw):
        tag = (int(bool(kw.get("trans_a"))), int(bool(kw.get("trans_b"))), int(kw.get("split_k", 1)), kw["M"], kw["N"], kw["K"],
               kw.get("batch", 1), "b" if kw.get("bias") is not None else "-", kw.get("act") or "-",
               "d" if kw.get("dropout_p", 0) > 0 else "-", "r" if kw.get("residual") is not None else "-",
               "c" if kw.get("conv") is not None else "-", str(C_out.dtype)[6:])
        timer.tag = tag
        return orig(A, B, C_out, **kw)

    class T:
        def wrap(self, key, flops, launch, nbytes=0):
            tag = timer.tag if "grouped" not in key else (1, 1, 0, 0, 0, 0, 0, "-", "grp", "-", "-", "-", key[-12:])
            timer.wrap(tag, flops, launch)

    ops.gemm = gemm
    import joeys2t_amd.functional as F
    ops.GEMM_TIMER = T()
    n = 3
    for _ in range(n):
        eager_step()
    ops.GEMM_TIMER = None
    agg = timer.summary()
    tot = 0.0
    rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
    print(f"{'ta tb sk     M     N     K  bat epi':44s} {'n/step':>6s} {'us':>8s} {'TF':>7s} {'ms/step':>8s} {'hbm_us':>7s}")
    for tag, (cnt, flops, secs, _) in rows:
        ta, tb, sk, M, N, K, bat, b, act, d, r, c, cd = tag
        es = 4 if cd == "float32" else 2
        byts = bat * (M * K * 2 + N * K * 2 + M * N * es)
        tot += secs / n
        print(f"{ta:2d} {tb:2d} {sk:2d} {M:6d} {N:5d} {K:5d} {bat:4d} {b}{act[:4]:4s}{d}{r}{c} {cd[:4]:5s} {cnt / n:6.1f} {secs / cnt * 1e6:8.1f} "
              f"{flops / secs / 1e12:7.1f} {secs / n * 1e3:8.3f} {byts / 6.3e12 * 1e6:7.1f}")
    print("total GEMM ms/step", tot * 1e3)


if __name__ == "__main__":
    main()
