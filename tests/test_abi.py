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


def test_attn_desc_matches_header_layout():
    """the ctypes mirror of js2t_attn_desc ends with the round-4 / round-5 fields, in the header's order"""
    import re
    from joeys2t_amd._lib import HEADER_PATH, AttnDesc
    text = HEADER_PATH.read_text()
    body = text[text.index("typedef struct js2t_attn_desc"):text.index("} js2t_attn_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [m.group(1) for m in re.finditer(r"(\w+)\s*(?:;|,)", body)]
    mirror = [f[0] for f in AttnDesc._fields_]
    assert [n for n in names if n in mirror] == mirror, (names, mirror)
    assert mirror[-5:] == ["delta_partial", "delta_groups", "seg", "seg_rows", "seg_keys"]


def test_packed_rows_host_logic():
    """ops.PackedRows.from_lengths: row offsets of a ragged batch from HOST lengths (no device involved), rounding of the row
    count, rejection of lengths outside 1..T; Batch keeps the host lengths in step with sort_by_src_length."""
    import torch
    from joeys2t_amd import ops
    from joeys2t_amd._lib import Js2tError
    pk = ops.PackedRows.from_lengths([5, 3, 7, 1], 7, "cpu", round_to=1)
    assert pk.seg.tolist() == [0, 5, 8, 15, 16] and (pk.B, pk.T, pk.rows) == (4, 7, 16) and pk.seg.dtype == torch.int32
    assert ops.PackedRows.from_lengths([5, 3, 7, 1], 7, "cpu", round_to=64).rows == 64
    assert ops.PackedRows.from_lengths([64], 64, "cpu", round_to=64).rows == 64
    for bad in ([0, 3], [8, 3], []):
        with pytest.raises(Js2tError):
            ops.PackedRows.from_lengths(bad, 7, "cpu")
    with pytest.raises(Js2tError):
        ops.PackedRows(torch.zeros(3, dtype=torch.int64), 2, 4, 8)
    from joeys2t_amd.batch import Batch
    b = Batch(src=torch.zeros(3, 9, 4), src_length=torch.tensor([4, 9, 6]), src_prompt_mask=None, trg=None, trg_length=None,
              trg_prompt_mask=None, indices=torch.arange(3), device=torch.device("cpu"), pad_index=1, eos_index=3, is_train=False,
              task="S2T", n_gpu=1)
    assert b.src_length_host == [4, 9, 6]
    b.sort_by_src_length()
    assert b.src_length_host == [9, 6, 4] == b.src_length.tolist()


def test_sub_sampled_lengths_on_the_host():
    """what the graphed driver and the packed encoder compute on the host equals Conv1dSubsampler.get_out_seq_lens_tensor
    (reference encoders.py:348-352)"""
    import torch
    from joeys2t_amd.encoders import Conv1dSubsampler
    for ks in ([5, 5], [3, 3], [3, 5, 3]):
        sub = Conv1dSubsampler(80, 64, 32, ks)
        lens = torch.arange(1, 400)
        want = sub.get_out_seq_lens_tensor(lens).tolist()
        assert [sub.out_len(int(n)) for n in lens] == want


def test_context_settings_are_per_caller_not_per_process():
    """js2t_ctx (include/joeys2t_hip.h): settings bound to the calling thread around a caller's launches - no GPU needed to check the
    plumbing: nesting, per-thread binding, the process-wide setters as overrides.  (The numerical side - two train steps in one
    process, one of them deterministic - is tests/test_hip_deterministic.py::test_two_steps_in_one_process_one_deterministic.)"""
    import threading
    from joeys2t_amd import _lib
    from joeys2t_amd._lib import Context, lib
    a, b = Context(deterministic=1, gemm_p192_ring=2), Context()
    assert a.get("deterministic") == 1 and b.get("deterministic") == -1 and _lib.effective("deterministic") == 0
    with a:
        assert _lib.effective("deterministic") == 1 and _lib.effective("gemm_p192_ring") == 2
        with b:  # nested: the inner binding counts, the outer one comes back
            assert _lib.effective("deterministic") == 0 and _lib.effective("gemm_p192_ring") == -1
        assert _lib.effective("deterministic") == 1
        seen = []
        t = threading.Thread(target=lambda: seen.append(_lib.effective("deterministic")))  # another thread: nothing bound there
        t.start()
        t.join()
        assert seen == [0]
        lib().js2t_gemm_p192_ring(4)  # the test override wins over the context ...
        try:
            assert _lib.effective("gemm_p192_ring") == 4
        finally:
            lib().js2t_gemm_p192_ring(-1)
        assert _lib.effective("gemm_p192_ring") == 2  # ... and gives way again
    assert _lib.effective("deterministic") == 0 and _lib.effective("gemm_p192_ring") == -1
