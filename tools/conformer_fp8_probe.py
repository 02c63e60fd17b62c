import sys, json; from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
print(json.dumps(bench.conformer_fp8_forward(torch.device("cuda:0")), indent=1))
