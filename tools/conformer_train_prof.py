"""The Conformer (config 5) train step alone, for rocprofv3 --kernel-trace --stats: bench.conformer_train_step in bf16.
usage: python tools/conformer_train_prof.py [bf16|fp8]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
print(bench.conformer_train_step(torch.device("cuda:0"), reps=10, modes=(mode, )))
