"""One replayed train step from a rocprofv3 kernel trace: every kernel with start offset, duration and the idle gap in front of it
(per stream overlap ignored: a negative gap = it started while the previous one ran).  usage: trace_step.py <dir> [out.txt]"""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "fbank" in r["Kernel_Name"]]
seq = rows[starts[-2]:starts[-1]]


def nm(r):
    return r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:64]


t0 = int(seq[0]["Start_Timestamp"])
end_prev = t0
busy = 0
lines = []
gaps = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end_prev) / 1e3
    if gap > 0:
        gaps += gap
    lines.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:6.1f}  {nm(r)}  grid={r['Grid_Size_X']}")
    end_prev = max(end_prev, e)
span = (end_prev - t0) / 1e3
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print(f"# step span {span:.1f} us, {len(seq)} kernels, idle {gaps:.1f} us", file=out)
print("\n".join(lines), file=out)
