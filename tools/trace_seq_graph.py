"""Kernel sequence of the LAST hipGraph replay of the train step in a rocprofv3 kernel trace: only the kernels that are
not ours (torch element-wise kernels, runtime copies), each with its two neighbours, plus the totals by name."""
import csv
import glob
import sys
from collections import Counter

d = sys.argv[1]
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "fbank" in r["Kernel_Name"]]
seq = rows[starts[-1]:]


def nm(r):
    return r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]


def us(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


t0 = int(seq[0]["Start_Timestamp"])
tot = Counter()
for i, r in enumerate(seq):
    n = nm(r)
    foreign = n.startswith("void at::") or "rocclr" in n or n.startswith("at::")
    if foreign:
        tot[n[:60]] += us(r)
        prev, nxt = (nm(seq[i - 1]) if i else "-"), (nm(seq[i + 1]) if i + 1 < len(seq) else "-")
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {us(r):6.1f} {n[:58]:58s} grid={r['Grid_Size_X']:>8s} | after {prev[:34]:34s} | before {nxt[:34]}")
print("step span us:", (int(seq[-1]["End_Timestamp"]) - t0) / 1e3, "kernels:", len(seq))
for k, v in tot.most_common():
    print(f"{v:8.1f} us  {k}")
