"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file (CPU-side: no GPU needed).

usage: python tools/isa_mix.py file.s <kernel-name-substring> [min_block_instructions]
Prints, per label-delimited block, the counts of MFMA / VALU / transcendental / SALU / LDS / VMEM / waitcnt instructions,
so that the VALU-per-MFMA ratio of a loop body can be read off before a PMC run confirms it."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z][^\s:]*:", l) and name in l.split(":")[0])
    blocks, cur, label = [], Counter(), "entry"
    ops = Counter()
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append((label, cur))
            cur, label = Counter(), s.split(":")[0]
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        cur[classify(op)] += 1
        ops[op] += 1
    blocks.append((label, cur))
    print(f"{'block':>14} {'mfma':>5} {'valu':>5} {'trans':>5} {'salu':>5} {'lds':>4} {'vmem':>4} {'wait':>4} {'bar':>3}  valu+trans/mfma")
    for label, c in blocks:
        n = sum(c.values())
        if n < min_n:
            continue
        v = c["valu"] + c["trans"]
        ratio = f"{v / c['mfma']:.2f}" if c["mfma"] else "-"
        print(f"{label[-14:]:>14} {c['mfma']:5d} {c['valu']:5d} {c['trans']:5d} {c['salu']:5d} {c['lds']:4d} {c['vmem']:4d} {c['wait']:4d} {c['barrier']:3d}  {ratio}")
    if "--ops" in sys.argv:
        for op, n in ops.most_common(40):
            print(f"  {op:32s} {n}")


if __name__ == "__main__":
    main()
