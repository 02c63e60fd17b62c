"""HBM bytes per launch of the bf16 LDS-DMA GEMM family from the two PMC passes of tools/profile_round.sh.
bytes = FETCH_SIZE [KB] * 1024 * 2 (gfx950 counts 128-byte fabric reads as 64, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE [KB] * 1024
usage: python tools/gemm_traffic.py <prof dir> <out json>"""
import csv
import datetime
import hashlib
import json
import subprocess
import sys
from pathlib import Path

d, out = sys.argv[1], sys.argv[2]


def load(name):
    return {r["kernel"]: (int(r["launches"]), float(r[list(r.keys())[2]])) for r in csv.DictReader(open(f"{d}/{name}"))}


fetch, write = load("pmc_fetch_by_kernel.csv"), load("pmc_write_by_kernel.csv")
fam = [k for k in fetch if k.startswith(("gemm_bf16_dma_", "gemm_bf16_p192_", "gemm_bf16_p192s_", "gemm_bf16_pan96_", "gemm_bf16_wg256_", "gemm_bf16_w256_"))]
n = sum(fetch[k][0] for k in fam)
fb = sum(fetch[k][1] for k in fam) * 1024 * 2
wb = sum(write[k][1] for k in fam if k in write) * 1024
res = {"kernel_family": "gemm_bf16_p192_kernel<*> + gemm_bf16_p192s_kernel<*> + gemm_bf16_pan96_kernel<*> + gemm_bf16_dma_kernel<*> + gemm_bf16_wg256_kernel<*> + "
                        "gemm_bf16_dma_grouped_kernel<*>", "launches_measured": n,
       "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n, "hbm_bytes_per_launch": round((fb + wb) / n),
       "per_kernel": {k: {"launches": fetch[k][0], "fetch_bytes_per_launch": fetch[k][1] * 2048 / fetch[k][0],
                          "write_bytes_per_launch": (write[k][1] * 1024 / write[k][0]) if k in write else None} for k in fam},
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over an eager 3-step bench run; FETCH_SIZE doubled"}
# provenance: bench.py reports `traffic` only while csrc/gemm.hip is byte-identical to the source this was measured on
root = Path(__file__).resolve().parent.parent
res["gemm_hip_sha16"] = hashlib.sha256((root / "joeys2t_amd" / "csrc" / "gemm.hip").read_bytes()).hexdigest()[:16]
res["measured_on"] = datetime.date.today().isoformat()
try:
    res["measured_at_commit"] = subprocess.run(["git", "-C", str(root), "rev-parse", "--short", "HEAD"], capture_output=True,
                                               text=True).stdout.strip() or None
except OSError:
    res["measured_at_commit"] = None  # the GPU box has no .git: stamped by the caller after copying into profiles/
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("launches_measured", "fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}))
