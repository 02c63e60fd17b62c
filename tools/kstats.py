"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel time per bench step."""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    if "copyBuffer" in r["Name"]:
        continue
    tot += float(r["TotalDurationNs"])
print(f"total kernel ms/step (excl. copies): {tot / 1e6 / steps:.3f}")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:78]
    print(f"{name:78s} n/step={float(r['Calls']) / steps:7.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / steps:8.3f} avg_us={float(r['AverageNs']) / 1e3:8.2f}")
