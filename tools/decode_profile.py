"""Where the beam-5 decode wall time goes (MuST-C shapes): cProfile of one search() call after a warm-up."""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

dev = torch.device("cuda:0")
bench.decode_rtf(dev)  # warm-up + library load
pr = cProfile.Profile()
pr.enable()
out = bench.decode_rtf(dev)
pr.disable()
print(out)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
