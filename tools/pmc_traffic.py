"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: launches, sum and mean of one counter."""
import csv
import glob
import re
import sys

d, counter = sys.argv[1], sys.argv[2]
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != counter:
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
print(f"kernel,launches,{counter}_sum,{counter}_mean")
for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{n},{s:.1f},{s / n:.3f}')
