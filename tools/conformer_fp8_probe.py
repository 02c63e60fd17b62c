"""Conformer encoder forward in e4m3 / bf16 (bench.conformer_fp8_forward); `python tools/conformer_fp8_probe.py fp8` times one mode
only (for a kernel trace of that mode alone)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

modes = tuple(sys.argv[1:]) or ("fp8", "bf16")
print(json.dumps(bench.conformer_fp8_forward(torch.device("cuda:0"), modes=modes), indent=1))
