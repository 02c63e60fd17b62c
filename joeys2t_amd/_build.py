"""Builds libjoeys2t_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

The library is the product; there is no CPU fallback.  `build_library()` is what
`__graft_entry__.build()` calls; the GPU box only uses the prebuilt .so shipped with the snapshot.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libjoeys2t_hip.so"
OBJ_DIR = PKG_DIR / "build"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-Wno-inline-asm"]
FLAGS += os.environ.get("JS2T_HIPCC_EXTRA", "").split()  # e.g. -DJS2T_P192_PROF for tools/p192_prof.py (use with --force)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: cannot build libjoeys2t_hip.so")
    return exe


def _sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _stale(obj: Path, src: Path, headers) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, *headers])


def build_library(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.hpp")) + list((PKG_DIR.parent / "include").glob("*.h"))
    jobs = []
    objs = []
    for src in _sources():
        obj = OBJ_DIR / (src.name + ".o")
        objs.append(obj)
        if force or _stale(obj, src, headers):
            cmd = [hipcc, *FLAGS, "-x", "hip", "-c", str(src), "-o", str(obj)]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not LIB_PATH.exists():
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB_PATH), *map(str, objs)])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in os.sys.argv, verbose=True))
