"""Beam-5 decode of the bench (MuST-C shapes) alone, for rocprofv3."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

print(bench.decode_rtf(torch.device("cuda:0")))
