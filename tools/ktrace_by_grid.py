"""Durations of one kernel in a rocprofv3 --kernel-trace CSV grouped by its grid size (which shapes are the slow ones).
usage: python tools/ktrace_by_grid.py <dir> <kernel-name-substring>"""
import collections
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        by[(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g, v in sorted(by.items()):
    v.sort()
    print(f"blocks {g:6d}: calls {len(v):6d}  median {v[len(v) // 2]:8.1f} us  mean {sum(v) / len(v):8.1f}  total {sum(v) / 1e3:8.2f} ms")
