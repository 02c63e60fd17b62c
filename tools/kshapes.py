"""Per-(kernel, grid) durations from a rocprofv3 kernel trace: maps GEMM launches back to shapes."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
for (name, gx, gy, gz), v in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    v.sort()
    print(f"{name:45s} grid=({gx},{gy},{gz}) n={len(v):4d} med={v[len(v)//2]:8.1f}us min={v[0]:8.1f} total_ms={sum(v)/1e3:8.2f}")
