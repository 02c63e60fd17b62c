"""ctypes binding of libjoeys2t_hip.so (the C ABI declared in include/joeys2t_hip.h).

There is no fallback: if the shared library is missing or a call is rejected, we raise.
"""
import ctypes as C
import os
import re
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("JS2T_LIB", PKG_DIR / "libjoeys2t_hip.so"))  # JS2T_LIB: instrumented builds of tools/
HEADER_PATH = PKG_DIR.parent / "include" / "joeys2t_hip.h"

F32, BF16 = 0, 1
ACT_CODES = {None: 0, "none": 0, "relu": 1, "gelu": 2, "swish": 3, "tanh": 4, "hardswish": 5}


class Js2tError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    """Mirror of js2t_gemm_desc (include/joeys2t_hip.h)."""
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("batch", C.c_int32), ("batch_inner", C.c_int32),
        ("dtype_ab", C.c_int32), ("dtype_c", C.c_int32),
        ("trans_a", C.c_int32), ("trans_b", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64), ("a_stride_o", C.c_int64), ("a_stride_i", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64), ("b_stride_o", C.c_int64), ("b_stride_i", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_stride_o", C.c_int64), ("c_stride_i", C.c_int64),
        ("alpha", C.c_float),
        ("alpha_dev", C.c_void_p),
        ("bias", C.c_void_p),
        ("act", C.c_int32),
        ("preact", C.c_void_p),
        ("dropout_p", C.c_float),
        ("rng_state", C.c_void_p),
        ("rng_stream", C.c_uint32),
        ("residual", C.c_void_p), ("ldr", C.c_int64), ("res_scale", C.c_float),
        ("gate", C.c_void_p), ("ldg", C.c_int64), ("gate_scale", C.c_float),
        ("beta", C.c_float),
        ("conv", C.c_int32), ("conv_tin", C.c_int32), ("conv_tout", C.c_int32), ("conv_c", C.c_int32),
        ("conv_stride", C.c_int32), ("conv_pad", C.c_int32),
        ("split_k", C.c_int32),
        ("a_rowsum", C.c_void_p),
        ("ln_partial", C.c_void_p), ("ln_eps", C.c_float), ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p), ("rs_partial", C.c_void_p),
        ("c8", C.c_void_p), ("ldc8", C.c_int64), ("c8_state", C.c_void_p), ("c8_mul", C.c_void_p), ("c8_scale_out", C.c_void_p),
        ("fp8_state", C.c_void_p),
        ("dot_src", C.c_void_p), ("ld_dot", C.c_int64), ("dot_partial", C.c_void_p),
        ("sumsq_partial", C.c_void_p),
    ]


class AttnDesc(C.Structure):
    """Mirror of js2t_attn_desc (include/joeys2t_hip.h)."""
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("d_o", C.c_void_p),
        ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("lse", C.c_void_p), ("delta", C.c_void_p),
        ("mask", C.c_void_p),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64), ("ld_do", C.c_int64),
        ("ld_dq", C.c_int64), ("ld_dk", C.c_int64), ("ld_dv", C.c_int64), ("mask_sb", C.c_int64), ("mask_sq", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32), ("head_dim", C.c_int32),
        ("scale", C.c_float), ("dropout_p", C.c_float),
        ("rng_state", C.c_void_p), ("rng_stream", C.c_uint32),
        ("rel_R", C.c_int32), ("rel_bias", C.c_void_p), ("d_rel_bias", C.c_void_p),
        ("delta_partial", C.c_void_p), ("delta_groups", C.c_int32), ("seg", C.c_void_p), ("seg_rows", C.c_int64),
        ("seg_keys", C.c_int32),
    ]


def declared_symbols(header: Path = HEADER_PATH):
    """Names of every function declared in the public header."""
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(js2t_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the library was not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise Js2tError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        # torch first: its wheel carries its own libamdhip64, and a process that has loaded /opt/rocm's copy through this library
        # BEFORE torch ends up with two HIP runtimes - the kernels' one then reports "no ROCm-capable device" (seen with
        # __graft_entry__.build() followed by smoke() in one process).  Loaded after torch, the library binds to torch's runtime:
        # one runtime, one set of streams and allocations.
        import torch  # noqa: F401
        _lib = C.CDLL(str(LIB_PATH))
        _lib.js2t_last_error.restype = C.c_char_p
        _lib.js2t_colsum_partial_rows.restype = C.c_int64
        _lib.js2t_colsum_partial_rows.argtypes = [C.c_int64]
        _lib.js2t_sumsq_partials.restype = C.c_int64
        _lib.js2t_sumsq_partials.argtypes = [C.c_int64]
        _lib.js2t_cmvn_stats_workspace.restype = C.c_int64
        _lib.js2t_gemm_grouped_blocks.restype = C.c_int64
        _lib.js2t_gemm_grouped_blocks.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        _lib.js2t_ctx_create.restype = C.c_void_p
        _lib.js2t_ctx_destroy.argtypes = [C.c_void_p]
        _lib.js2t_ctx_set.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        _lib.js2t_ctx_get.argtypes = [C.c_void_p, C.c_int32]
        _lib.js2t_ctx_bind.restype = C.c_void_p
        _lib.js2t_ctx_bind.argtypes = [C.c_void_p]
        if "JS2T_P192" in os.environ:  # kernel-selection override for A/B measurements (see js2t_gemm_p192_mode)
            _lib.js2t_gemm_p192_mode(int(os.environ["JS2T_P192"]))
        if "JS2T_WG256" in os.environ:  # 0 / 1 / -1: the 256x128 kernel of the grouped weight gradients (js2t_gemm_wg256_mode)
            _lib.js2t_gemm_wg256_mode(int(os.environ["JS2T_WG256"]))
        if "JS2T_PANEL" in os.environ:  # 0 / 1 / -1: the panel-resident kernel (js2t_gemm_panel_mode)
            _lib.js2t_gemm_panel_mode(int(os.environ["JS2T_PANEL"]))
        if "JS2T_P192_RING" in os.environ:  # 2 = two blocks per CU with a two-slot ring (js2t_gemm_p192_ring)
            _lib.js2t_gemm_p192_ring(int(os.environ["JS2T_P192_RING"]))
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().js2t_last_error().decode("utf-8", "replace")
        raise Js2tError(f"libjoeys2t_hip: {what} failed (rc={rc}): {msg}")


CTX_KEYS = {"deterministic": 0, "gemm_p192_mode": 1, "gemm_p192_ring": 2, "gemm_wg256_mode": 3, "gemm_panel_mode": 4}


class Context:
    """A js2t_ctx (include/joeys2t_hip.h): the settings one caller's launches run with - deterministic mode, kernel-selection rules -
    bound to the calling thread for the length of a `with` block (the previous binding comes back at its end; blocks nest).

        ctx = Context(deterministic=True)
        with ctx:
            ...  # every library call made by this thread in here sees deterministic = 1

    Autograd's backward runs the library's kernels on worker threads unless told otherwise: TrainStep runs its backward passes under
    torch.autograd.set_multithreading_enabled(False), i.e. on the thread that holds the binding."""

    def __init__(self, **settings):
        self._h = lib().js2t_ctx_create()
        if not self._h:
            raise Js2tError("js2t_ctx_create failed")
        self._prev = []
        for k, v in settings.items():
            self.set(k, v)

    def set(self, key: str, value) -> None:
        check(lib().js2t_ctx_set(self._h, CTX_KEYS[key], int(value)), "js2t_ctx_set")

    def get(self, key: str) -> int:
        return int(lib().js2t_ctx_get(self._h, CTX_KEYS[key]))

    def __enter__(self):
        self._prev.append(lib().js2t_ctx_bind(self._h))
        return self

    def __exit__(self, *exc):
        lib().js2t_ctx_bind(self._prev.pop())
        return False

    def __del__(self):
        try:
            if self._h and _lib is not None:
                _lib.js2t_ctx_destroy(self._h)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
        self._h = None


def effective(key: str) -> int:
    """the value of a context key the calling thread's next launch would see (process-wide test override > bound context > default)"""
    return int(lib().js2t_ctx_effective(CTX_KEYS[key]))
