"""CPU: the C-ABI library builds, loads and exports every symbol that include/joeys2t_hip.h declares;
the product path refuses CPU tensors instead of falling back."""
import ctypes

import pytest
import torch

from joeys2t_amd import _lib, ops


def test_library_exports_every_declared_symbol():
    names = _lib.declared_symbols()
    assert "js2t_gemm" in names and "js2t_ctc_bwd" in names and len(names) >= 25
    handle = _lib.lib()
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert handle.js2t_abi_version() >= 1


def test_gemm_desc_matches_header_layout():
    # field order / count of the ctypes mirror follows the header text
    import re
    text = _lib.HEADER_PATH.read_text()
    body = text[text.index("typedef struct js2t_gemm_desc {"):text.index("} js2t_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = re.sub(r"^(const\s+)?[a-z0-9_]+(\s*\*)?\s+", "", decl)
        fields += [n.strip().lstrip("*") for n in names.split(",")]
    assert fields == [f[0] for f in _lib.GemmDesc._fields_]


def test_no_cpu_fallback():
    x = torch.zeros(4, 8)
    with pytest.raises(ops.Js2tError):
        ops.glu_fwd(x)
    with pytest.raises(ops.Js2tError):
        ops.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-6)


def test_comm_entry_points_reject_bad_arguments():
    """js2t_comm_* (the RCCL communicator behind the boundary): argument checks only - no GPU, no RCCL call."""
    from joeys2t_amd import comm
    L = comm._bind(_lib.lib())
    assert L.js2t_comm_unique_id_bytes() == 128
    assert L.js2t_comm_unique_id(None, 128) != 0 and b"128" in L.js2t_last_error()
    h = ctypes.c_void_p()
    assert L.js2t_comm_init(ctypes.byref(h), b"\0" * 128, 64, 1, 0, 0) != 0       # an id of the wrong size
    assert L.js2t_comm_init(ctypes.byref(h), b"\0" * 128, 128, 2, 2, 0) != 0      # rank outside the world
    assert h.value is None
    assert L.js2t_comm_allreduce_async(None, None, 0, 0, 0, None) != 0
    assert L.js2t_comm_wait(None, None, 0) != 0
    assert L.js2t_comm_stream(None) is None
    assert L.js2t_comm_destroy(None) == 0
    with pytest.raises(_lib.Js2tError):
        comm.Communicator(0, 1, torch.device("cpu"), exchange_id=lambda b: b)


def test_range_set():
    """runtime.RangeSet: the bookkeeping of gradient pieces the update leaves un-cleared (host logic)."""
    from joeys2t_amd.runtime import RangeSet
    r = RangeSet([(0, 10), (10, 20), (40, 50), (45, 60)])
    assert r.r == [(0, 20), (40, 60)] and r.contains(5, 15) and not r.contains(15, 45) and r.contains(40, 60)
    assert r.minus([(5, 8), (18, 45)]) == [(0, 5), (8, 18), (45, 60)]
    assert r.minus([]) == [(0, 20), (40, 60)] and r.minus([(0, 100)]) == []
    v = r.version
    r.remove(0, 20)
    assert r.r == [(40, 60)] and r.version == v + 1 and bool(r)
    r.remove(30, 70)
    assert not r
    r.add(8, 16)
    r.add(16, 24)
    assert r.r == [(8, 24)] and r.contains(10, 20)
