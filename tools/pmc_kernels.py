"""Per-kernel means of all counters in a rocprofv3 --pmc counter_collection CSV."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    if pat not in name:
        continue
    a = agg[name][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, s) in sorted(cs.items()):
        print(f"   {c:32s} n={n:5d} mean={s / n:16.1f}")
