"""Decode timing when hypotheses finish at scattered steps (bench.decode_rtf(eos_scale=...)) with a host-side profile.
usage: python tools/eos_probe.py [eos_scale]"""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.17
for s in (0.0, 0.18, 0.2):
    r = bench.decode_rtf(dev, eos_scale=s)
    print(s, {k: r[k] for k in ("rtf", "wall_s", "steps", "hyp_len_min_median_max")}, flush=True)
pr = cProfile.Profile()
pr.enable()
r = bench.decode_rtf(dev, eos_scale=sc)
pr.disable()
print(sc, r)
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
