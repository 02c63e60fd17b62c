"""Definition-level similarity against the reference tree (development check, container only: reads /root/reference).
For every class / function in joeys2t_amd/*.py that has a same-named definition in joeynmt/*.py: share of the repo
definition's characters (docstrings dropped, ast.unparse-normalised) lying in >= 25-character blocks common with it."""
import ast
import difflib
import sys
from pathlib import Path


def defs(path):
    out = {}
    tree = ast.parse(Path(path).read_text())
    for node in ast.walk(tree):
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)):
            for sub in ast.walk(node):
                if isinstance(sub, (ast.ClassDef, ast.FunctionDef, ast.Module)) and sub.body and isinstance(sub.body[0], ast.Expr) and \
                        isinstance(getattr(sub.body[0], "value", None), ast.Constant) and isinstance(sub.body[0].value.value, str):
                    sub.body = sub.body[1:] or [ast.Pass()]
            out.setdefault(node.name, []).append(ast.unparse(node))
    return out


def share(a, b, k=25):
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    return sum(m.size for m in sm.get_matching_blocks() if m.size >= k) / max(1, len(a))


ref = {}
for p in Path("/root/reference/joeynmt").glob("*.py"):
    for name, srcs in defs(p).items():
        ref.setdefault(name, []).extend(srcs)
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
rows = []
for p in sorted(Path("joeys2t_amd").glob("*.py")):
    for name, srcs in defs(p).items():
        for s in srcs:
            if name in ref and len(s) > 200:
                rows.append((max(share(s, r) for r in ref[name]), p.name, name, len(s)))
for sh, f, n, L in sorted(rows, reverse=True):
    if sh >= thr:
        print(f"{sh:5.2f}  {f}:{n}  ({L} chars)")
