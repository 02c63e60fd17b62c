"""Idle time between consecutive kernels of the replayed train step: python tools/gap_stats.py <rocprofv3 dir> (kernel-trace csv)."""
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))))
# the last adamw launches delimit steps
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
a, b = ad[-3], ad[-2]  # one full replayed step
seg = rows[a + 1:b + 1]
span = seg[-1][1] - seg[0][0]
busy_end, gaps, busy = seg[0][0], [], 0
for s, e, n, q in seg:
    if s > busy_end:
        gaps.append((s - busy_end, n))
        busy_end_prev = busy_end
    busy_end = max(busy_end, e)
tot_gap = sum(g for g, _ in gaps)
print(f"kernels {len(seg)}  span {span / 1e6:.3f} ms  idle (no kernel running) {tot_gap / 1e6:.3f} ms in {len(gaps)} gaps, median {sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3:.2f} us")
big = sorted(gaps, reverse=True)[:8]
for g, n in big:
    print(f"  {g / 1e3:7.2f} us before {n[:70]}")
