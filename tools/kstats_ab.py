"""Two rocprofv3 --kernel-trace --stats directories side by side, per bench step: which kernels the difference sits in.
The number of steps a run made is read from the calls of a once-per-step kernel (the fused update, adamw_items_kernel).
usage: python tools/kstats_ab.py DIR_A DIR_B [rows]"""
import csv
import glob
import re
import sys


def load(d):
    f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    steps = sum(int(r["Calls"]) for r in rows if "adamw_items_kernel" in r["Name"])
    out = {}
    for r in rows:
        name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)[:60]
        ms, n = out.get(name, (0.0, 0.0))
        out[name] = (ms + float(r["TotalDurationNs"]) / 1e6 / steps, n + float(r["Calls"]) / steps)
    return out, steps


a, sa = load(sys.argv[1])
b, sb = load(sys.argv[2])
print(f"steps: A {sa}  B {sb};   kernel ms/step: A {sum(v[0] for v in a.values()):.3f}  B {sum(v[0] for v in b.values()):.3f}")
names = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[0] - b.get(k, (0, 0))[0]))
for k in names[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    ma, na = a.get(k, (0.0, 0.0))
    mb, nb = b.get(k, (0.0, 0.0))
    print(f"{k:60s} A {ma:7.3f} ms ({na:6.1f}/step)  B {mb:7.3f} ms ({nb:6.1f}/step)  B-A {mb - ma:+7.3f}")
