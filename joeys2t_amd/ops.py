"""Thin Python launchers for the C ABI in include/joeys2t_hip.h.

Each function allocates its outputs with torch (device memory / stream plumbing only), passes raw
device pointers to libjoeys2t_hip.so and enqueues on torch's current HIP stream.  Nothing here computes
on the host and nothing falls back to PyTorch math: a CPU tensor or a missing library raises.
"""
import ctypes as C
import math
from typing import Optional, Sequence

import torch

from joeys2t_amd._lib import ACT_CODES, BF16, F32, AttnDesc, GemmDesc, Js2tError, check, lib

FP8 = 2  # JS2T_FP8_E4M3: OCP e4m3fn bytes (js2t_gemm operands only)
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float8_e4m3fn: FP8}


def dt_code(t_or_dtype) -> int:
    dtype = t_or_dtype.dtype if torch.is_tensor(t_or_dtype) else t_or_dtype
    try:
        return _DT[dtype]
    except KeyError:
        raise Js2tError(f"unsupported dtype {dtype}: the HIP path computes in float32 or bfloat16") from None


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Js2tError(
                "joeys2t_amd ops run on the GPU through libjoeys2t_hip.so only; got a CPU tensor "
                "(there is no CPU fallback in the product path)"
            )


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------------------- RNG state
class DropoutRng:
    """Device-resident {seed, offset} pair read by every dropout-bearing kernel, plus a host-side
    call-site counter.  `advance()` bumps the offset with a device op so a captured hipGraph draws fresh
    masks on every replay; call-site ids restart at every step so forward and backward agree."""

    def __init__(self, device, seed: int = 42):
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        self._site = 0

    def begin_step(self):
        self._site = 0

    def advance(self):
        self.state[1:2].add_(1)

    def next_site(self) -> int:
        self._site += 1
        return self._site

    def seed(self, seed: int):
        self.state[0] = seed
        self.state[1] = 0


_rngs = {}


def dropout_rng(device) -> DropoutRng:
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _rngs:
        _rngs[key] = DropoutRng(device)
    return _rngs[key]


# ----------------------------------------------------------------------------------------- GEMM
GEMM_TIMER = None  # bench.py installs an object with .wrap(key, flops, launch) to HIP-event-time every GEMM launch


def _takes_p192(d) -> bool:
    """Mirror of the library's default choice of the persistent 192x128 kernel (gemm.hip p192_eligible), used only to
    label the launches bench.py times with the kernel rocprofv3 will show for them."""
    if d.trans_a or d.trans_b or d.conv or d.split_k > 1 or d.batch != 1 or d.dtype_c != BF16:
        return False
    if (d.N & 7) or d.N < 128 or (d.K & 7) or d.K < 192 or d.preact or d.beta != 0.0 or d.act not in (0, 1) or (d.residual and d.gate):
        return False
    if d.bias and (d.bias & 15):
        return False
    if ((d.C or 0) & 15) or (d.ldc & 7) or (d.residual and ((d.ldr & 7) or (d.residual & 15))) or (d.gate and ((d.ldg & 7) or (d.gate & 15))):
        return False
    return -(-d.M // 192) * -(-d.N // 128) >= 200


LN_FOLD_WIDTH = 512  # row length the LayerNorm fold of js2t_gemm is built for: eight 64-column groups


def row_partials(M: int, device) -> torch.Tensor:
    """f32 [M, 8, 2] for js2t_gemm's rs_partial (producer) / ln_partial (consumer): per 64-column group {sum, sum of squares}."""
    return torch.empty((M, LN_FOLD_WIDTH // 64, 2), dtype=torch.float32, device=device)


def gemm(A, B, C_out, *, M, N, K, lda, ldb, ldc, trans_a=False, trans_b=False, batch=1, batch_inner=1,
         a_strides=(0, 0), b_strides=(0, 0), c_strides=(0, 0), a_off=0, b_off=0, c_off=0, alpha=1.0,
         alpha_dev=None, bias=None, act=None, preact=None, dropout_p=0.0, rng: Optional[DropoutRng] = None,
         rng_stream=0, residual=None, ldr=0, res_scale=1.0, gate=None, ldg=0, gate_scale=1.0, beta=0.0,
         conv=None, split_k=1, a_rowsum=None, ln=None, rs_partial=None, fp8_state=None, c8=None, dot=None):
    """C = epilogue(alpha * op(A) op(B)^T) — see js2t_gemm in the header.  Offsets are in elements.
    dot = (src bf16 [M, >= N], partial f32 [M, N // 64]): partial[m, g] = sum over the 64-column group of bf16(C[m, c]) * src[m, c].
    ln = (partial f32[M,8,2], eps, mean_out f32[M] | None, rstd_out f32[M] | None): LayerNorm folded into the product (B = the
    centred, gamma-scaled weight of ParamStore.fold); rs_partial f32[M,8,2]: the stored rows' partial sums are written to it."""
    _dev(A, B, C_out, bias, preact, residual, gate, alpha_dev, a_rowsum, rs_partial)
    if C_out is None and c8 is None:
        raise Js2tError("gemm: no output")
    if A.dtype != B.dtype:
        raise Js2tError(f"gemm: A/B dtype mismatch {A.dtype} vs {B.dtype}")
    c_dtype = torch.bfloat16 if C_out is None else C_out.dtype  # e4m3 products may write ONLY their e4m3 second output (c8)
    d = GemmDesc()
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.batch, d.batch_inner = int(batch), int(batch_inner)
    d.dtype_ab, d.dtype_c = dt_code(A), dt_code(c_dtype)
    d.trans_a, d.trans_b = int(bool(trans_a)), int(bool(trans_b))
    esa, esc = A.element_size(), (2 if C_out is None else C_out.element_size())
    d.A = A.data_ptr() + a_off * esa
    d.lda, d.a_stride_o, d.a_stride_i = int(lda), int(a_strides[0]), int(a_strides[1])
    d.B = B.data_ptr() + b_off * esa
    d.ldb, d.b_stride_o, d.b_stride_i = int(ldb), int(b_strides[0]), int(b_strides[1])
    d.C = None if C_out is None else C_out.data_ptr() + c_off * esc
    d.ldc, d.c_stride_o, d.c_stride_i = int(ldc), int(c_strides[0]), int(c_strides[1])
    d.alpha = float(alpha)
    d.alpha_dev = None if alpha_dev is None else alpha_dev.data_ptr()
    if bias is not None:
        if bias.dtype != torch.float32:
            raise Js2tError("gemm: bias must be float32")
        d.bias = bias.data_ptr()
    d.act = ACT_CODES[act]
    if preact is not None:
        if preact.dtype != c_dtype:
            raise Js2tError("gemm: preact dtype must match C")
        d.preact = preact.data_ptr() + c_off * esc
    d.dropout_p = float(dropout_p)
    if dropout_p > 0.0:
        if rng is None:
            raise Js2tError("gemm: dropout needs a DropoutRng")
        d.rng_state = rng.state.data_ptr()
        d.rng_stream = int(rng_stream)
    if residual is not None:
        if residual.dtype != c_dtype:
            raise Js2tError("gemm: residual dtype must match C")
        d.residual, d.ldr, d.res_scale = residual.data_ptr(), int(ldr), float(res_scale)
    if gate is not None:
        if gate.dtype != c_dtype:
            raise Js2tError("gemm: gate dtype must match C")
        d.gate, d.ldg, d.gate_scale = gate.data_ptr(), int(ldg), float(gate_scale)
    d.beta = float(beta)
    if conv is not None:
        d.conv = 1
        d.conv_tin, d.conv_tout, d.conv_c, d.conv_stride, d.conv_pad = (int(v) for v in conv)
    d.split_k = int(split_k)
    d.a_rowsum = None if a_rowsum is None else a_rowsum.data_ptr()
    if ln is not None:
        part, eps, mean_out, rstd_out = ln
        _dev(part, mean_out, rstd_out)
        if part.dtype != torch.float32 or part.numel() != M * 16 or not part.is_contiguous():
            raise Js2tError("gemm: ln needs partial sums f32[M, 8, 2]")
        d.ln_partial, d.ln_eps = part.data_ptr(), float(eps)
        if mean_out is not None:
            if min(mean_out.numel(), rstd_out.numel()) < M or mean_out.dtype != torch.float32 or rstd_out.dtype != torch.float32:
                raise Js2tError("gemm: ln mean / rstd outputs must be f32[M]")
            d.ln_mean, d.ln_rstd = mean_out.data_ptr(), rstd_out.data_ptr()
    if fp8_state is not None:
        _dev(fp8_state)
        d.fp8_state = fp8_state.data_ptr()
    if c8 is not None:  # (e4m3 out [M, N], delayed-scale state f32[4], mul f32[1] | None, scale_out f32[1] | None)
        out8, st8, mul8, sc8 = c8
        _dev(out8, st8, mul8, sc8)
        if out8.dtype != torch.float8_e4m3fn or out8.shape != (M, N) or not out8.is_contiguous():
            raise Js2tError("gemm: c8 output must be a contiguous float8_e4m3fn [M, N]")
        d.c8, d.ldc8, d.c8_state = out8.data_ptr(), N, st8.data_ptr()
        d.c8_mul = None if mul8 is None else mul8.data_ptr()
        d.c8_scale_out = None if sc8 is None else sc8.data_ptr()
    if dot is not None:
        src, part = dot
        _dev(src, part)
        if (src.dtype != torch.bfloat16 or src.dim() != 2 or src.shape[0] < M or src.stride(1) != 1 or part.dtype != torch.float32 or
                part.numel() != M * (N // 64) or not part.is_contiguous() or N % 64):
            raise Js2tError("gemm: dot = (bf16 [M, >= N] rows, contiguous f32 [M, N // 64])")
        d.dot_src, d.ld_dot, d.dot_partial = src.data_ptr(), int(src.stride(0)), part.data_ptr()
    if rs_partial is not None:
        if rs_partial.dtype != torch.float32 or rs_partial.numel() != M * 16 or not rs_partial.is_contiguous():
            raise Js2tError("gemm: rs_partial must be contiguous f32[M, 8, 2]")
        d.rs_partial = rs_partial.data_ptr()
    if GEMM_TIMER is not None:
        if d.dtype_ab == FP8:
            key = "gemm_fp8_p192_kernel<0,0,0>"
        elif d.dtype_ab == BF16:
            fam = "gemm_bf16_kernel" if d.conv else ("gemm_bf16_p192_kernel" if _takes_p192(d) else "gemm_bf16_dma_kernel")
            key = f"{fam}<{int(d.trans_a)},{int(d.trans_b)},{int(d.split_k > 1)}>"
        else:
            key = "gemm_generic_kernel"
        GEMM_TIMER.wrap(key, 2.0 * d.M * d.N * d.K * d.batch, lambda: check(lib().js2t_gemm(C.byref(d), _stream()), "js2t_gemm"),
                        nbytes=d.batch * (esa * (d.M * d.K + d.N * d.K) + esc * d.M * d.N))
    else:
        check(lib().js2t_gemm(C.byref(d), _stream()), "js2t_gemm")
    return C_out


def gemm_grouped(As, Bs, Cs, *, M, N, K, lda, ldb, ldc, split_k=1, beta=0.0, alpha=1.0, a_rowsums=None, sumsq_partial=None):
    """len(As) independent products C_i = alpha * A_i^T B_i (+ beta C_i) of one shape in one launch (js2t_gemm_grouped):
    bf16 [K, M] / [K, N] operands (trans_a = trans_b = 1), f32 or bf16 C_i, optional f32[M] row-sum targets.
    sumsq_partial: f32[grouped_blocks(M, N, n)] - every block leaves the sum of squares of the values it stored there."""
    n = len(As)
    if not (n == len(Bs) == len(Cs)) or (a_rowsums is not None and len(a_rowsums) != n):
        raise Js2tError("gemm_grouped: list lengths differ")
    if n == 0:
        return
    _dev(*As, *Bs, *Cs, *(a_rowsums or ()))
    if any(t.dtype != torch.bfloat16 for t in (*As, *Bs)) or any(c.dtype != Cs[0].dtype for c in Cs):
        raise Js2tError("gemm_grouped: bf16 operands and one output dtype required")
    d = GemmDesc()
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.batch, d.batch_inner = 1, 1
    d.dtype_ab, d.dtype_c = BF16, dt_code(Cs[0])
    d.trans_a, d.trans_b = 1, 1
    d.lda, d.ldb, d.ldc = int(lda), int(ldb), int(ldc)
    d.alpha, d.beta, d.split_k = float(alpha), float(beta), int(split_k)
    d.res_scale = d.gate_scale = 1.0
    if sumsq_partial is not None:
        _dev(sumsq_partial)
        if sumsq_partial.dtype != torch.float32 or sumsq_partial.numel() < grouped_blocks(d.M, d.N, n):
            raise Js2tError("gemm_grouped: sumsq_partial must be f32[grouped_blocks(M, N, count)]")
        d.sumsq_partial = sumsq_partial.data_ptr()
    arr = C.c_void_p * n
    pa, pb, pc = arr(*[t.data_ptr() for t in As]), arr(*[t.data_ptr() for t in Bs]), arr(*[t.data_ptr() for t in Cs])
    pr = None if a_rowsums is None else arr(*[t.data_ptr() for t in a_rowsums])
    if GEMM_TIMER is not None:
        # (which kernel a launch takes is the library's choice: the 256x128 one for >= 160 such tiles without split-K, else 128x128)
        GEMM_TIMER.wrap(f"js2t_gemm_grouped<split_k {'>' if d.split_k > 1 else '='} 1> (gemm_bf16_wg256_kernel | gemm_bf16_dma_grouped_kernel)",
                        2.0 * d.M * d.N * d.K * n,
                        lambda: check(lib().js2t_gemm_grouped(C.byref(d), n, pa, pb, pc, pr, _stream()), "js2t_gemm_grouped"),
                        nbytes=n * (2 * (d.M * d.K + d.N * d.K) + Cs[0].element_size() * d.M * d.N * (2 if d.beta else 1)))
    else:
        check(lib().js2t_gemm_grouped(C.byref(d), n, pa, pb, pc, pr, _stream()), "js2t_gemm_grouped")


def grouped_blocks(M: int, N: int, count: int) -> int:
    """blocks (= sumsq_partial entries) of an un-split gemm_grouped launch"""
    return int(lib().js2t_gemm_grouped_blocks(int(M), int(N), int(count)))


# ----------------------------------------------------------------------------------------- element-wise
def cast(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(src, out)
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(lib().js2t_cast(_p(src), dt_code(src), _p(out), dt_code(out), C.c_int64(src.numel()), _stream()), "js2t_cast")
    return out


def axpby(x, a: float, y=None, b: float = 0.0):
    _dev(x, y)
    out = torch.empty_like(x)
    check(lib().js2t_axpby(_p(x), C.c_float(a), _p(y), C.c_float(b), _p(out), C.c_int64(x.numel()), dt_code(x), _stream()),
          "js2t_axpby")
    return out


def transpose_groups(src, dst, table, n_groups: int, total_tiles: int):
    _dev(src, dst, table)
    check(lib().js2t_transpose_groups(_p(src), _p(dst), _p(table), int(n_groups), C.c_int64(total_tiles), _stream()), "js2t_transpose_groups")


def glu_fwd(x: torch.Tensor, T: Optional[int] = None, valid_t: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x[rows, 2C] -> [rows, C]; with (T, valid_t): rows = batch * T and positions t >= *valid_t (device int64 scalar) come out 0."""
    _dev(x, valid_t)
    rows, c2 = x.shape[0], x.shape[1]
    y = torch.empty((rows, c2 // 2), dtype=x.dtype, device=x.device)
    check(lib().js2t_glu_fwd_crop(_p(x), _p(y), C.c_int64(rows), C.c_int64(c2 // 2), C.c_int64(T if valid_t is not None else max(rows, 1)),
                                  _p(valid_t), dt_code(x), _stream()), "js2t_glu_fwd")
    return y


def glu_bwd(x: torch.Tensor, dy: torch.Tensor, T: Optional[int] = None, valid_t: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(x, dy, valid_t)
    dx = torch.empty_like(x)
    check(lib().js2t_glu_bwd_crop(_p(x), _p(dy), _p(dx), C.c_int64(x.shape[0]), C.c_int64(x.shape[1] // 2),
                                  C.c_int64(T if valid_t is not None else max(x.shape[0], 1)), _p(valid_t), dt_code(x), _stream()), "js2t_glu_bwd")
    return dx


def add_pe_dropout(x, pe, extra, p, rng: Optional[DropoutRng], site: int):
    _dev(x, pe, extra)
    B, T, D = x.shape
    y = torch.empty_like(x)
    check(lib().js2t_add_pe_dropout(_p(x), _p(pe), _p(extra), _p(y), C.c_int64(B), C.c_int64(T), C.c_int64(D), dt_code(x),
                                    C.c_float(p), _p(rng.state) if p > 0 else None, C.c_uint32(site), _stream()),
          "js2t_add_pe_dropout")
    return y


def dropout_bwd(dy, p, rng: DropoutRng, site: int):
    _dev(dy)
    dy = dy.contiguous()
    cols = dy.shape[-1]
    rows = dy.numel() // cols
    dx = torch.empty_like(dy)
    check(lib().js2t_dropout_bwd(_p(dy), _p(dx), C.c_int64(rows), C.c_int64(cols), dt_code(dy), C.c_float(p),
                                 _p(rng.state), C.c_uint32(site), _stream()), "js2t_dropout_bwd")
    return dx


def act_bwd(dh, z, act: str, scale: float = 1.0):
    _dev(dh, z)
    dz = torch.empty_like(dh)
    check(lib().js2t_act_bwd(_p(dh), _p(z), _p(dz), C.c_int64(dh.numel()), ACT_CODES[act], dt_code(dh), C.c_float(scale),
                             _stream()), "js2t_act_bwd")
    return dz


def embed_fwd(ids, table, scale: float, out_dtype):
    _dev(ids, table)
    ids = ids.contiguous()
    D, V = table.shape[1], table.shape[0]
    out = torch.empty((*ids.shape, D), dtype=out_dtype, device=table.device)
    check(lib().js2t_embed_fwd(_p(ids), _p(table), dt_code(table), _p(out), dt_code(out), C.c_int64(ids.numel()),
                               C.c_int64(D), C.c_int64(V), C.c_float(scale), _stream()), "js2t_embed_fwd")
    return out


def embed_bwd(ids, dout, vocab: int, scale: float, pad_idx: int, out: Optional[torch.Tensor] = None):
    _dev(ids, dout, out)
    ids, dout = ids.contiguous(), dout.contiguous()
    D = dout.shape[-1]
    dtable = out if out is not None else torch.zeros((vocab, D), dtype=torch.float32, device=dout.device)
    check(lib().js2t_embed_bwd(_p(ids), _p(dout), dt_code(dout), _p(dtable), C.c_int64(ids.numel()), C.c_int64(D),
                               C.c_int64(vocab), C.c_float(scale), C.c_int64(-1 if pad_idx is None else pad_idx), _stream()),
          "js2t_embed_bwd")
    return dtable


def colsum(x2d: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    _dev(x2d, out)
    rows, cols = x2d.shape
    if out is None:
        out = torch.empty((cols,), dtype=torch.float32, device=x2d.device)
        accumulate = False
    nparts = lib().js2t_colsum_partial_rows(rows)
    partial = torch.empty((max(nparts, 1) * cols,), dtype=torch.float32, device=x2d.device)
    check(lib().js2t_colsum(_p(x2d), dt_code(x2d), _p(out), _p(partial), C.c_int64(rows), C.c_int64(cols), int(accumulate),
                            _stream()), "js2t_colsum")
    return out


def conv_weight_pack(w: torch.Tensor, dtype) -> torch.Tensor:
    _dev(w)
    cout, cin, k = w.shape
    wp = torch.empty((cout, k * cin), dtype=dtype, device=w.device)
    check(lib().js2t_conv_weight_pack(_p(w.contiguous()), _p(wp), dt_code(wp), C.c_int64(cout), C.c_int64(cin), C.c_int64(k),
                                      _stream()), "js2t_conv_weight_pack")
    return wp


def conv_weight_unpack_grad(dwp_t: torch.Tensor, cout: int, cin: int, k: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[K*Cin, Cout] wgrad -> torch Conv1d layout; `out` given = accumulate into it (in-place gradient accumulation)."""
    _dev(dwp_t, out)
    acc = out is not None
    dw = out if acc else torch.empty((cout, cin, k), dtype=torch.float32, device=dwp_t.device)
    check(lib().js2t_conv_weight_unpack_grad(_p(dwp_t), _p(dw), C.c_int64(cout), C.c_int64(cin), C.c_int64(k), int(acc),
                                             _stream()), "js2t_conv_weight_unpack_grad")
    return dw


def im2col(x3d, K, stride, pad, tout):
    """[B, T, C] -> [B*tout, K*C] (js2t_im2col): rows of the strided convolution's A operand."""
    _dev(x3d)
    B, tin, Cc = x3d.shape
    x3d = x3d.contiguous()
    col = torch.empty((B * tout, K * Cc), dtype=x3d.dtype, device=x3d.device)
    check(lib().js2t_im2col(_p(x3d), _p(col), C.c_int64(B), C.c_int64(tin), C.c_int64(tout), C.c_int64(Cc), C.c_int64(K),
                            C.c_int64(stride), C.c_int64(pad), dt_code(x3d), _stream()), "js2t_im2col")
    return col


def col2im(dcol, B, tin, tout, Cc, K, stride, pad):
    _dev(dcol)
    dx = torch.empty((B, tin, Cc), dtype=dcol.dtype, device=dcol.device)
    check(lib().js2t_col2im(_p(dcol), _p(dx), C.c_int64(B), C.c_int64(tin), C.c_int64(tout), C.c_int64(Cc), C.c_int64(K),
                            C.c_int64(stride), C.c_int64(pad), dt_code(dcol), _stream()), "js2t_col2im")
    return dx


def subsample_lengths_mask(lengths: torch.Tensor, t_out: int, kernel_sizes: Sequence[int]):
    _dev(lengths)
    B = lengths.shape[0]
    lengths = lengths.to(torch.int64).contiguous()
    out_len = torch.empty((B,), dtype=torch.int64, device=lengths.device)
    mask = torch.empty((B, 1, t_out), dtype=torch.bool, device=lengths.device)
    ks = (C.c_int32 * len(kernel_sizes))(*kernel_sizes)
    check(lib().js2t_subsample_lengths_mask(_p(lengths), _p(out_len), _p(mask), C.c_int64(B), C.c_int64(t_out), ks,
                                            C.c_int32(len(kernel_sizes)), _stream()), "js2t_subsample_lengths_mask")
    return out_len, mask


# ----------------------------------------------------------------------------------------- layer norm
def layernorm_fwd(x, gamma, beta, eps: float):
    _dev(x, gamma, beta)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(lib().js2t_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), C.c_int64(rows), C.c_int64(D),
                                   C.c_float(eps), dt_code(x), _stream()), "js2t_layernorm_fwd")
    return y, mean, rstd


def layernorm_fwd_fp8(x, gamma, beta, eps: float, state: torch.Tensor, mul: Optional[torch.Tensor] = None, want_y: bool = True):
    """LayerNorm whose result also (want_y=False: only) comes out as e4m3 with the delayed scale of `state` (new_fp8_state):
    -> (y | None, mean, rstd, y8 float8_e4m3fn, scale f32[1] = S [* mul]) - js2t_layernorm_fwd_fp8."""
    _dev(x, gamma, beta, state, mul)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x) if want_y else None
    y8 = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    scale = torch.empty((1, ), dtype=torch.float32, device=x.device)
    check(lib().js2t_layernorm_fwd_fp8(_p(x), _p(gamma), _p(beta), _p(y), _p(y8), _p(state), _p(mul), _p(scale), _p(mean), _p(rstd),
                                       C.c_int64(rows), C.c_int64(D), C.c_float(eps), dt_code(x), _stream()), "js2t_layernorm_fwd_fp8")
    return y, mean, rstd, y8, scale


class GradCopies:
    """Workspace for parameter-gradient sums that many blocks would otherwise add onto the same few addresses (LayerNorm
    gamma / beta): `copies` partial copies per destination, folded into the real gradients once per step (js2t_fold_copies).
    Destinations are registered on first use (before any hipGraph capture: warm-up steps); capacity is fixed."""
    COPIES, MAX_D, MAX_SLOTS = 8, 1024, 256

    def __init__(self, device):
        self.stride = 2 * self.MAX_D  # floats per copy: [gamma | beta]
        self.ws = torch.zeros(self.MAX_SLOTS * self.COPIES * self.stride, dtype=torch.float32, device=device)
        self.slots = {}   # (dgamma ptr, dbeta ptr) -> slot
        # the fold kernel's table lives at ONE address for the object's lifetime (a captured hipGraph keeps that pointer):
        # allocated at full capacity, rows are written in place, the launch passes the number of rows in use
        self.table = torch.zeros((2 * self.MAX_SLOTS, 3), dtype=torch.int64, device=device)
        self.n_rows = 0
        self.captured_rows = None  # rows in use when a capture last saw fold(): later registrations would be missed by it

    def lookup(self, dg: torch.Tensor, db: torch.Tensor):
        """-> (ws view for gamma, ws view for beta) of the slot registered for these gradient views, or None."""
        D = dg.numel()
        if D > self.MAX_D or db.numel() != D or dg.dtype != torch.float32:
            return None
        key = (dg.data_ptr(), db.data_ptr())
        slot = self.slots.get(key)
        if slot is None:
            if torch.cuda.is_current_stream_capturing():
                raise Js2tError("GradCopies: a LayerNorm gradient slot was first seen during hipGraph capture - run one eager "
                                "step first (the captured fold launch would never sum this slot's copies)")
            if len(self.slots) >= self.MAX_SLOTS:
                return None
            slot = len(self.slots)
            self.slots[key] = slot
            base = slot * self.COPIES * self.stride
            rows = torch.tensor([[base, dg.data_ptr(), D], [base + self.MAX_D, db.data_ptr(), D]], dtype=torch.int64)
            self.table[self.n_rows:self.n_rows + 2].copy_(rows)
            self.n_rows += 2
        base = slot * self.COPIES * self.stride
        return self.ws[base:base + D], self.ws[base + self.MAX_D:base + self.MAX_D + D]

    def invalidate(self):
        """Forget every registration (the flat gradient buffer was re-allocated: the stored destination addresses are stale)."""
        self.slots.clear()
        self.n_rows = 0
        self.ws.zero_()

    def fold(self):
        if self.n_rows:
            if torch.cuda.is_current_stream_capturing():
                self.captured_rows = self.n_rows
            elif self.captured_rows is not None and self.n_rows != self.captured_rows:
                raise Js2tError("GradCopies: slots were registered after a hipGraph captured fold(); re-capture the step")
            check(lib().js2t_fold_copies(_p(self.ws), _p(self.table), C.c_int32(self.n_rows), C.c_int32(self.COPIES),
                                         C.c_int64(self.stride), _stream()), "js2t_fold_copies")


def layernorm_bwd(dy, x, gamma, mean, rstd, need_param_grads=True, add=None, add_scale=1.0, grad_out=None, drop=None,
                  copies: Optional[GradCopies] = None, n_out=None, beta=None):
    """grad_out = (dgamma_buf, dbeta_buf): accumulate the parameter gradients into these buffers in place.
    drop = (p, rng, site): also return dropout_bwd(dx, p, rng, site) as a fourth value (js2t_layernorm_bwd_dropout).
    copies: with grad_out, accumulate into that workspace's copies of (dgamma_buf, dbeta_buf) instead (folded by its owner).
    n_out (with beta): also write the LayerNorm's forward result xhat * gamma + beta there (js2t_layernorm_bwd_fused)."""
    _dev(dy, x, gamma, mean, rstd, add, n_out, beta)
    if n_out is not None and (beta is None or n_out.shape != x.shape or n_out.dtype != x.dtype or not n_out.is_contiguous()):
        raise Js2tError("layernorm_bwd: n_out must look like x and needs beta")
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty_like(x)
    dgamma = dbeta = partial = None
    if need_param_grads:
        if grad_out is not None:
            dgamma, dbeta = grad_out
        else:
            dgamma = torch.empty((D,), dtype=torch.float32, device=x.device)
            dbeta = torch.empty((D,), dtype=torch.float32, device=x.device)
        nparts = (rows + 15) // 16  # >= one partial per row block of the vectorised backward kernel (16..64 rows)
        partial = torch.empty((2 * max(nparts, 1) * D,), dtype=torch.float32, device=x.device)
    acc_g, acc_b, stride = dgamma, dbeta, 0
    if grad_out is not None and need_param_grads and copies is not None:
        views = copies.lookup(dgamma, dbeta)
        ws = copies
        copies = 1
        if views is not None:
            acc_g, acc_b = views
            copies, stride = ws.COPIES, ws.stride
    else:
        copies = 1
    p_drop, rng, site = drop if drop is not None else (0.0, None, 0)
    dxd = torch.empty_like(x) if drop is not None else None
    check(lib().js2t_layernorm_bwd_fused(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(add), C.c_float(add_scale), _p(acc_g),
                                         _p(acc_b), _p(partial), int(grad_out is not None and need_param_grads), C.c_int64(rows),
                                         C.c_int64(D), dt_code(x), _p(dxd), C.c_float(p_drop), None if rng is None else _p(rng.state),
                                         C.c_uint32(site & 0xFFFFFFFF), C.c_int32(copies), C.c_int64(stride), _p(beta) if n_out is not None else None,
                                         _p(n_out), _stream()),
          "js2t_layernorm_bwd_fused")
    if drop is None:
        return dx, dgamma, dbeta
    return dx, dgamma, dbeta, dxd


def fold_ln_weights(table: torch.Tensor, n_entries: int, max_rows: int):
    """(Re)derive the centred, gamma-scaled bf16 weights and the beta-absorbing biases of every LayerNorm fold
    (js2t_fold_ln_weights; table int64[n, 8] on the device)."""
    _dev(table)
    check(lib().js2t_fold_ln_weights(_p(table), C.c_int32(n_entries), C.c_int32(max_rows), _stream()), "js2t_fold_ln_weights")


def layernorm_bwd_supports_dropout(x) -> bool:
    """Shapes the fused second output of layernorm_bwd(drop=...) is available for (the vectorised kernel)."""
    D = x.shape[-1]
    return D % 8 == 0 and D <= 2048 and x.data_ptr() % 16 == 0 and x.is_contiguous()


# ----------------------------------------------------------------------------------------- Conformer convolution module
def dwconv_outer_fwd(x3d: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Depthwise convolution along the OUTER index of x[L, N, C]; w f32[C, K] (js2t_dwconv_outer_fwd)."""
    _dev(x3d, w, bias)
    L, N, Cc = x3d.shape
    x3d = x3d.contiguous()
    y = torch.empty_like(x3d)
    check(lib().js2t_dwconv_outer_fwd(_p(x3d), _p(w), _p(bias), _p(y), C.c_int64(L), C.c_int64(N), C.c_int64(Cc), int(w.shape[1]),
                                      dt_code(x3d), _stream()), "js2t_dwconv_outer_fwd")
    return y


def dwconv_outer_bwd(dy3d, x3d, w, need_dx=True, dw_out: Optional[torch.Tensor] = None):
    """-> (dx | None, dw f32[C, K]); dw_out: accumulate the weight gradient into this buffer instead."""
    _dev(dy3d, x3d, w, dw_out)
    L, N, Cc = x3d.shape
    dy3d = dy3d.contiguous()
    dx = torch.empty_like(x3d) if need_dx else None
    dw = dw_out if dw_out is not None else torch.zeros((Cc, w.shape[1]), dtype=torch.float32, device=x3d.device)
    check(lib().js2t_dwconv_outer_bwd(_p(dy3d), _p(x3d), _p(w), _p(dx), _p(dw), C.c_int64(L), C.c_int64(N), C.c_int64(Cc),
                                      int(w.shape[1]), dt_code(x3d), _stream()), "js2t_dwconv_outer_bwd")
    return dx, (None if dw_out is not None else dw)


def bn_act_fwd(x2d, gamma, beta, running_mean, running_var, eps: float, momentum: float, train: bool, act: Optional[str]):
    """BatchNorm over the rows of x[rows, C] + activation -> (y, mean, invstd); running statistics updated in place when training."""
    _dev(x2d, gamma, beta, running_mean, running_var)
    rows, Cc = x2d.shape
    x2d = x2d.contiguous()
    y = torch.empty_like(x2d)
    mean = torch.empty((Cc,), dtype=torch.float32, device=x2d.device)
    invstd = torch.empty_like(mean)
    ws = torch.empty((2 * Cc,), dtype=torch.float32, device=x2d.device)
    check(lib().js2t_bn_act_fwd(_p(x2d), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(mean), _p(invstd), _p(y), _p(ws),
                                C.c_int64(rows), C.c_int64(Cc), C.c_float(eps), C.c_float(momentum), int(train), ACT_CODES[act],
                                dt_code(x2d), _stream()), "js2t_bn_act_fwd")
    return y, mean, invstd


def bn_act_bwd(dy2d, x2d, gamma, beta, mean, invstd, train: bool, act: Optional[str], grad_out=None):
    """-> (dx, dgamma, dbeta); grad_out = (dgamma_buf, dbeta_buf) accumulates in place (returned grads are then None)."""
    _dev(dy2d, x2d, gamma, beta, mean, invstd)
    rows, Cc = x2d.shape
    dy2d = dy2d.contiguous()
    dx = torch.empty_like(x2d)
    if grad_out is not None:
        dg, db = grad_out
    else:
        dg = torch.zeros((Cc,), dtype=torch.float32, device=x2d.device)
        db = torch.zeros_like(dg)
    ws = torch.empty((2 * Cc,), dtype=torch.float32, device=x2d.device)
    check(lib().js2t_bn_act_bwd(_p(dy2d), _p(x2d), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(dx), _p(dg), _p(db), _p(ws),
                                C.c_int64(rows), C.c_int64(Cc), int(train), ACT_CODES[act], dt_code(x2d), _stream()), "js2t_bn_act_bwd")
    return (dx, None, None) if grad_out is not None else (dx, dg, db)


# ----------------------------------------------------------------------------------------- softmax
def softmax_fwd(S, mask, B, H, Tq, Tk, ld, p, rng: Optional[DropoutRng], site: int):
    _dev(S, mask)
    P = torch.empty_like(S)
    Pd = torch.empty_like(S) if p > 0 else P
    mask_sb = mask_sq = 0
    if mask is not None:
        if mask.dtype != torch.bool or mask.dim() != 3 or mask.shape[2] != Tk or not mask.is_contiguous():
            raise Js2tError(f"softmax: mask must be contiguous bool [B, 1|Tq, Tk], got {tuple(mask.shape)} {mask.dtype}")
        mask_sq = 0 if mask.shape[1] == 1 else Tk
        if mask.shape[1] not in (1, Tq):
            raise Js2tError(f"softmax: mask query dim {mask.shape[1]} != 1 or {Tq}")
        mask_sb = 0 if mask.shape[0] == 1 else mask.shape[1] * Tk
        if mask.shape[0] not in (1, B):
            raise Js2tError(f"softmax: mask batch dim {mask.shape[0]} != 1 or {B}")
    check(lib().js2t_softmax_fwd(_p(S), _p(mask), _p(P), _p(Pd), C.c_int64(B), C.c_int64(H), C.c_int64(Tq), C.c_int64(Tk),
                                 C.c_int64(ld), C.c_int64(mask_sb), C.c_int64(mask_sq), dt_code(S), C.c_float(p),
                                 _p(rng.state) if p > 0 else None, C.c_uint32(site), _stream()), "js2t_softmax_fwd")
    return P, Pd


def softmax_bwd(P, dPd, Z, Tq, Tk, ld, p, rng: Optional[DropoutRng], site: int):
    _dev(P, dPd)
    dS = torch.empty_like(P)
    check(lib().js2t_softmax_bwd(_p(P), _p(dPd), _p(dS), C.c_int64(Z), C.c_int64(Tq), C.c_int64(Tk), C.c_int64(ld),
                                 dt_code(P), C.c_float(p), _p(rng.state) if p > 0 else None, C.c_uint32(site), _stream()),
          "js2t_softmax_bwd")
    return dS


def attn_head_mean(P, B, H, Tq, Tk, ld):
    _dev(P)
    out = torch.empty((B, Tq, Tk), dtype=torch.float32, device=P.device)
    check(lib().js2t_attn_head_mean(_p(P), _p(out), C.c_int64(B), C.c_int64(H), C.c_int64(Tq), C.c_int64(Tk), C.c_int64(ld),
                                    dt_code(P), _stream()), "js2t_attn_head_mean")
    return out


def rel_bias_add(S, rel_bias, B, H, Tq, Tk, ld):
    """S[b,h,q,k] += rel_bias[h, clamp(k - q)] in place (js2t_rel_bias_add; the materialised attention path)."""
    _dev(S, rel_bias)
    R = (rel_bias.shape[1] - 1) // 2
    check(lib().js2t_rel_bias_add(_p(S), _p(rel_bias), C.c_int64(B), C.c_int64(H), C.c_int64(Tq), C.c_int64(Tk), C.c_int64(ld),
                                  C.c_int32(R), dt_code(S), _stream()), "js2t_rel_bias_add")
    return S


def rel_bias_grad(dS, d_rel_bias, B, H, Tq, Tk, ld):
    """d_rel_bias += the per-(head, clipped distance) sums of dS (js2t_rel_bias_grad)."""
    _dev(dS, d_rel_bias)
    R = (d_rel_bias.shape[1] - 1) // 2
    check(lib().js2t_rel_bias_grad(_p(dS), _p(d_rel_bias), C.c_int64(B), C.c_int64(H), C.c_int64(Tq), C.c_int64(Tk), C.c_int64(ld),
                                   C.c_int32(R), dt_code(dS), _stream()), "js2t_rel_bias_grad")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------------------- vocabulary rows / losses
def row_lse(x2d: torch.Tensor, want_argmax: bool = False):
    _dev(x2d)
    rows, V = x2d.shape
    lse = torch.empty((rows,), dtype=torch.float32, device=x2d.device)
    am = torch.empty((rows,), dtype=torch.int64, device=x2d.device) if want_argmax else None
    check(lib().js2t_row_lse(_p(x2d), _p(lse), _p(am), C.c_int64(rows), C.c_int64(V), dt_code(x2d), _stream()), "js2t_row_lse")
    return lse, am


def ctc_collapse(best: torch.Tensor, in_len: torch.Tensor, blank: int, pad: int):
    """best i64[B,T] (frame-wise arg-max labels) -> (ids i64[B,T] pad-filled, lengths i64[B]): repeats merged, blanks dropped."""
    _dev(best, in_len)
    B, T = best.shape
    best = best.contiguous()
    in_len = in_len.to(torch.int64).contiguous()
    out = torch.empty((B, T), dtype=torch.int64, device=best.device)
    n = torch.empty((B,), dtype=torch.int64, device=best.device)
    check(lib().js2t_ctc_collapse(_p(best), _p(in_len), _p(out), _p(n), C.c_int64(B), C.c_int64(T), C.c_int64(blank), C.c_int64(pad),
                                  _stream()), "js2t_ctc_collapse")
    return out, n


def log_softmax(x: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
    _dev(x)
    x = x.contiguous()
    V = x.shape[-1]
    rows = x.numel() // V
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib().js2t_log_softmax(_p(x), dt_code(x), _p(y), dt_code(y), C.c_int64(rows), C.c_int64(V), _stream()),
          "js2t_log_softmax")
    return y


def train_stats(stats6, total, nll, ctc, n_correct, inv_norm: float, nseqs: float, ntokens: float):
    """stats6 += [total, nll, ctc] * inv_norm, n_correct, nseqs, ntokens; returns total * inv_norm (0-d f32)."""
    _dev(stats6, total, nll, ctc, n_correct)
    assert stats6.dtype == torch.float64 and stats6.numel() == 6 and total.dtype == torch.float32
    assert n_correct is None or n_correct.dtype == torch.int64
    norm = torch.empty((), dtype=torch.float32, device=total.device)
    check(lib().js2t_train_stats(_p(stats6), _p(total), _p(nll), _p(ctc), _p(n_correct), C.c_double(inv_norm), C.c_double(nseqs),
                                 C.c_double(ntokens), _p(norm), _stream()), "js2t_train_stats")
    return norm


class LinComb2Fn(torch.autograd.Function):
    """a*x + b*y of two 0-d (or same-shaped) f32 tensors as ONE launch (loss interpolation, loss.py:164); the backward is two."""

    @staticmethod
    def forward(ctx, x, y, a: float, b: float):
        ctx.ab = (a, b)
        return axpby(x, a, y, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        g = g.contiguous()
        return axpby(g, a), axpby(g, b), None, None


def sum_f32(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(lib().js2t_sum_f32(_p(x), C.c_int64(x.numel()), _p(out), _stream()), "js2t_sum_f32")
    return out


def xent_fwd(logits2d, trg1d, pad_idx: int, smoothing: float):
    _dev(logits2d, trg1d)
    rows, V = logits2d.shape
    dev = logits2d.device
    loss_rows = torch.empty((rows,), dtype=torch.float32, device=dev)
    correct_rows = torch.empty((rows,), dtype=torch.float32, device=dev)
    lse = torch.empty((rows,), dtype=torch.float32, device=dev)
    check(lib().js2t_xent_fwd(_p(logits2d), dt_code(logits2d), _p(trg1d), _p(loss_rows), _p(correct_rows), _p(lse),
                              C.c_int64(rows), C.c_int64(V), C.c_int64(pad_idx), C.c_float(smoothing), _stream()),
          "js2t_xent_fwd")
    return loss_rows, correct_rows, lse


def xent_bwd(logits2d, trg1d, lse, g_dev, scale: float, pad_idx: int, smoothing: float, out_dtype=None):
    """out_dtype: torch.bfloat16 for f32 logits -> the gradient in bf16 (js2t_xent_bwd_as); default the logits' dtype."""
    _dev(logits2d, trg1d, lse, g_dev)
    rows, V = logits2d.shape
    d = torch.empty(logits2d.shape, dtype=out_dtype or logits2d.dtype, device=logits2d.device)
    check(lib().js2t_xent_bwd_as(_p(logits2d), dt_code(logits2d), _p(trg1d), _p(lse), _p(g_dev), C.c_float(scale), _p(d), dt_code(d),
                                 C.c_int64(rows), C.c_int64(V), C.c_int64(pad_idx), C.c_float(smoothing), _stream()),
          "js2t_xent_bwd_as")
    return d


def _ctc_geometry(logits, pack):
    """(B, T, V, row offsets pointer): [B, T, V] logits, or - with `pack` (PackedRows) - packed [pack.rows, V] ones."""
    if pack is None:
        B, T, V = logits.shape
        return B, T, V, None
    if logits.dim() != 2 or logits.shape[0] != pack.rows or not logits.is_contiguous():
        raise Js2tError(f"CTC over packed rows: contiguous [{pack.rows}, V] logits expected, got {tuple(logits.shape)}")
    _dev(pack.seg)
    return pack.B, pack.T, logits.shape[1], C.c_void_p(pack.seg.data_ptr())


def ctc_alpha(logits3d, lse, targets, in_len, tgt_len, blank: int, zero_infinity: bool, with_beta: bool = False, pack: "PackedRows" = None):
    """Forward CTC recursion; with_beta=True also runs the (independent) backward recursion in the same launch and
    returns it as the 4th value for ctc_bwd(beta=...).  pack: the logits / lse are packed rows (js2t_ctc_alpha row_offsets)."""
    _dev(logits3d, lse, targets, in_len, tgt_len)
    B, T, V, roff = _ctc_geometry(logits3d, pack)
    Lmax = targets.shape[1]
    dev = logits3d.device
    alpha = torch.empty((B, T, 2 * Lmax + 1), dtype=torch.float32, device=dev)
    beta = torch.empty_like(alpha) if with_beta else None
    nll = torch.empty((B,), dtype=torch.float32, device=dev)
    loss_rows = torch.empty((B,), dtype=torch.float32, device=dev)
    check(lib().js2t_ctc_alpha(_p(logits3d), dt_code(logits3d), _p(lse), _p(targets), _p(in_len), _p(tgt_len), _p(alpha), _p(beta),
                               _p(nll), _p(loss_rows), C.c_int64(B), C.c_int64(T), C.c_int64(V), C.c_int64(Lmax),
                               C.c_int64(blank), int(zero_infinity), roff, _stream()), "js2t_ctc_alpha")
    return alpha, nll, loss_rows, beta


def ctc_bwd(logits3d, lse, targets, in_len, tgt_len, alpha, nll, g_dev, scale: float, blank: int, zero_infinity: bool, beta=None,
            pack: "PackedRows" = None):
    _dev(logits3d, lse, targets, in_len, tgt_len, alpha, nll, g_dev, beta)
    B, T, V, roff = _ctc_geometry(logits3d, pack)
    Lmax = targets.shape[1]
    ready = beta is not None
    if beta is None:
        beta = torch.empty_like(alpha)
    d = torch.empty_like(logits3d)
    check(lib().js2t_ctc_bwd(_p(logits3d), dt_code(logits3d), _p(lse), _p(targets), _p(in_len), _p(tgt_len), _p(alpha),
                             _p(beta), _p(nll), _p(g_dev), C.c_float(scale), _p(d), C.c_int64(B), C.c_int64(T), C.c_int64(V),
                             C.c_int64(Lmax), C.c_int64(blank), int(zero_infinity), int(ready), roff,
                             C.c_int64(0 if pack is None else pack.rows), _stream()), "js2t_ctc_bwd")
    return d


def attn_decode(q2d, k_view, v_view, ldkv: int, idx, idx_ld: int, Tmax: int, length: int, key_mask, H: int, dh: int, len_dev=None,
                group: int = 1):
    """Single-query attention over cached keys / values (js2t_attn_decode).  k_view / v_view: tensors whose data_ptr is the
    first key / value element (row 0, position 0, head 0); ldkv: elements between consecutive positions."""
    _dev(q2d, k_view, v_view, idx, key_mask, len_dev)
    rows = q2d.shape[0]
    out = torch.empty((rows, H * dh), dtype=q2d.dtype, device=q2d.device)
    check(lib().js2t_attn_decode(_p(q2d), C.c_int64(q2d.stride(0)), _p(k_view), _p(v_view), C.c_int64(ldkv), _p(idx), int(idx_ld), int(Tmax),
                                 int(length), _p(len_dev), _p(key_mask), _p(out), C.c_int64(out.stride(0)), int(rows), int(H), int(dh),
                                 C.c_float(1.0 / math.sqrt(dh)), int(group), dt_code(q2d), _stream()), "js2t_attn_decode")
    return out


# ----------------------------------------------------------------------------------------- beam search
# ----------------------------------------------------------------------------------------- fp8 (e4m3) forward mode
WEIGHT_VERSION = 0  # bumped whenever parameters change (optimizer step, load_state_dict): invalidates cached e4m3 weights


def quantize_fp8(x: torch.Tensor, mul: Optional[torch.Tensor] = None):
    """Per-tensor dynamic e4m3 quantisation without a host sync: amax on the device (js2t_absmax), y = e4m3(x * 448 / amax)
    (js2t_quantize_fp8).  Returns (y float8_e4m3fn like x, scale f32[1] = amax / 448 [* mul]); the fp8 GEMM takes `scale`
    as alpha_dev, so passing the weight's scale as `mul` folds both dequantisation factors into one device scalar."""
    _dev(x, mul)
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous():
        raise Js2tError("quantize_fp8: contiguous float32 / bfloat16 input")
    y = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    amax = torch.empty((1, ), dtype=torch.float32, device=x.device)
    scale = torch.empty((1, ), dtype=torch.float32, device=x.device)
    check(lib().js2t_absmax(_p(x), dt_code(x), C.c_int64(x.numel()), _p(amax), _stream()), "js2t_absmax")
    check(lib().js2t_quantize_fp8(_p(x), dt_code(x), _p(y), C.c_int64(x.numel()), _p(amax), _p(mul), _p(scale), _stream()),
          "js2t_quantize_fp8")
    return y, scale


def quantize_fp8_delayed(x: torch.Tensor, state: torch.Tensor, mul: Optional[torch.Tensor] = None):
    """One-pass e4m3 quantisation with the scale of an earlier call (js2t_quantize_fp8_delayed); `state` f32[4] on the device
    lives with the call site (new_fp8_state()).  Returns (y, scale f32[1])."""
    _dev(x, state, mul)
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous():
        raise Js2tError("quantize_fp8_delayed: contiguous float32 / bfloat16 input")
    y = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty((1, ), dtype=torch.float32, device=x.device)
    check(lib().js2t_quantize_fp8_delayed(_p(x), dt_code(x), _p(y), C.c_int64(x.numel()), _p(state), _p(mul), _p(scale), _stream()),
          "js2t_quantize_fp8_delayed")
    return y, scale


def new_fp8_state(x: torch.Tensor) -> torch.Tensor:
    """Calibrate a delayed-scaling state on x: {max|x| / 448, 0, 0, 0}."""
    _dev(x)
    amax = torch.empty((1, ), dtype=torch.float32, device=x.device)
    check(lib().js2t_absmax(_p(x), dt_code(x), C.c_int64(x.numel()), _p(amax), _stream()), "js2t_absmax")
    state = torch.zeros((4, ), dtype=torch.float32, device=x.device)
    state[0:1] = amax / 448.0
    return state


def rep_penalty(log_probs: torch.Tensor, tokens: torch.Tensor, penalty: float):
    """penalize_repetition (search.py:972-1001) in place on f32 [rows, V]; tokens int64 [rows, L] with ids in [0, V)."""
    _dev(log_probs, tokens)
    if log_probs.dtype != torch.float32 or not log_probs.is_contiguous() or tokens.dtype != torch.int64:
        raise Js2tError("rep_penalty: log_probs must be contiguous float32, tokens int64")
    tokens = tokens.contiguous()
    rows, V = log_probs.shape
    if tokens.shape[0] != rows:
        raise Js2tError(f"rep_penalty: {tokens.shape[0]} token rows for {rows} score rows")
    check(lib().js2t_rep_penalty(_p(log_probs), _p(tokens), C.c_int64(rows), C.c_int64(V), C.c_int64(tokens.shape[1]),
                                 C.c_float(penalty), _stream()), "js2t_rep_penalty")
    return log_probs


def logp_set(log_probs: torch.Tensor, rows, cols, value: float):
    """log_probs[rows[i], cols[i]] = value (host index lists or device int64 tensors), in place."""
    _dev(log_probs)
    if log_probs.dtype != torch.float32 or not log_probs.is_contiguous():
        raise Js2tError("logp_set: log_probs must be contiguous float32")
    dev = log_probs.device
    rows = torch.as_tensor(rows, dtype=torch.int64).to(dev).contiguous()
    cols = torch.as_tensor(cols, dtype=torch.int64).to(dev).contiguous()
    if rows.numel() != cols.numel():
        raise Js2tError("logp_set: rows and cols differ in length")
    check(lib().js2t_logp_set(_p(log_probs), _p(rows), _p(cols), C.c_int64(rows.numel()), C.c_int64(log_probs.shape[1]),
                              C.c_float(value), _stream()), "js2t_logp_set")
    return log_probs


def beam_step(logits2d: torch.Tensor, beam_log_probs: torch.Tensor, n_batch: int, beam: int, forbid_ids, length_penalty: float,
              normalized: bool = False):
    """Fused log-softmax + masks + beam score + length penalty + top-k (js2t_beam_step).  Returns
    (scores [n_batch, beam], flat ids [n_batch, beam] int64, row lse [n_batch*beam]).
    normalized=True: the rows already are (edited) log-probabilities (js2t_beam_step_logp)."""
    _dev(logits2d, beam_log_probs)
    if logits2d.dtype != torch.float32 or not logits2d.is_contiguous():
        raise Js2tError("beam_step: logits must be contiguous float32")
    V = logits2d.shape[1]
    dev = logits2d.device
    scores = torch.empty((n_batch, beam), dtype=torch.float32, device=dev)
    ids = torch.empty((n_batch, beam), dtype=torch.int64, device=dev)
    lse = torch.empty((n_batch * beam, ), dtype=torch.float32, device=dev)
    fb = (C.c_int32 * max(1, len(forbid_ids)))(*forbid_ids)
    blp = beam_log_probs.contiguous().float()
    fn = lib().js2t_beam_step_logp if normalized else lib().js2t_beam_step
    check(fn(_p(logits2d), _p(blp), _p(scores), _p(ids), _p(lse), C.c_int64(n_batch), C.c_int32(beam),
             C.c_int64(V), fb, C.c_int32(len(forbid_ids)), C.c_float(length_penalty), _stream()),
          "js2t_beam_step")
    return scores, ids, lse


def beam_pick(logits2d: torch.Tensor, n_pick: int, forbid_ids, row_scores: Optional[torch.Tensor] = None):
    """For every row the n_pick (<= 8) best tokens by log-softmax (forbidden ids masked) + row_scores[row] (js2t_beam_pick): the
    candidate pre-selection of joint CTC / attention decoding.  Returns (scores [rows, n_pick], ids [rows, n_pick] int64, lse [rows])."""
    _dev(logits2d, row_scores)
    if logits2d.dtype != torch.float32 or not logits2d.is_contiguous():
        raise Js2tError("beam_pick: logits must be contiguous float32")
    rows, V = logits2d.shape
    dev = logits2d.device
    if row_scores is None:
        row_scores = torch.zeros((rows, ), dtype=torch.float32, device=dev)
    scores = torch.empty((rows, n_pick), dtype=torch.float32, device=dev)
    ids = torch.empty((rows, n_pick), dtype=torch.int64, device=dev)
    lse = torch.empty((rows, ), dtype=torch.float32, device=dev)
    fb = (C.c_int32 * max(1, len(forbid_ids)))(*forbid_ids)
    check(lib().js2t_beam_pick(_p(logits2d), _p(row_scores.contiguous().float()), _p(scores), _p(ids), _p(lse), C.c_int64(rows), C.c_int32(n_pick),
                               C.c_int64(V), fb, C.c_int32(len(forbid_ids)), _stream()), "js2t_beam_pick")
    return scores, ids, lse


CTC_LOG0 = -1.0e30  # log 0 of the CTC prefix variables (js2t_ctc_prefix_step carries no infinities)


def ctc_prefix_init(ctc_log_probs: torch.Tensor, in_len: torch.Tensor, beam: int, blank: int) -> torch.Tensor:
    """Forward variables of the EMPTY prefix for every hypothesis slot: [B * beam, T, 2] = (log 0, running sum of the blank's
    log-probabilities inside the utterance's frames)."""
    B, T, _ = ctc_log_probs.shape
    rb = torch.cumsum(ctc_log_probs[:, :, blank].float(), dim=1)
    live = torch.arange(T, device=ctc_log_probs.device).unsqueeze(0) < in_len.view(-1, 1)
    r = torch.full((B, T, 2), CTC_LOG0, dtype=torch.float32, device=ctc_log_probs.device)
    r[:, :, 1] = torch.where(live, rb, torch.full_like(rb, CTC_LOG0))
    return r.repeat_interleave(beam, dim=0).contiguous()


def ctc_prefix_step(ctc_log_probs, in_len, r_prev, last_tok, cand, cand_lp, psi_prev, n_out: int, beam: int, blank: int, eos: int,
                    weight: float):
    """One extension step of the CTC prefix score for every (hypothesis, candidate) pair (js2t_ctc_prefix_step; Watanabe et al. 2017,
    Algorithm 2).  Returns (local [rows, C] step scores, psi [rows, C], r_new [rows, C, T, 2])."""
    _dev(ctc_log_probs, in_len, r_prev, last_tok, cand, cand_lp, psi_prev)
    B, T, V = ctc_log_probs.shape
    rows, Cn = cand.shape
    if (ctc_log_probs.dtype != torch.float32 or not ctc_log_probs.is_contiguous() or r_prev.shape != (rows, T, 2) or not r_prev.is_contiguous()
            or rows != B * beam or cand.dtype != torch.int64 or in_len.dtype != torch.int64 or last_tok.dtype != torch.int64):
        raise Js2tError("ctc_prefix_step: ctc_log_probs f32 [B, T, V], r_prev f32 [B * beam, T, 2], int64 cand / in_len / last_tok")
    dev = cand.device
    local = torch.empty((rows, Cn), dtype=torch.float32, device=dev)
    psi = torch.empty((rows, Cn), dtype=torch.float32, device=dev)
    r_new = torch.empty((rows, Cn, T, 2), dtype=torch.float32, device=dev)
    check(lib().js2t_ctc_prefix_step(_p(ctc_log_probs), _p(in_len.contiguous()), _p(r_prev), _p(last_tok.contiguous()), _p(cand.contiguous()),
                                     _p(cand_lp.contiguous().float()), _p(psi_prev.contiguous().float()), _p(local), _p(psi), _p(r_new),
                                     C.c_int64(rows), C.c_int32(beam), C.c_int32(Cn), C.c_int32(T), C.c_int64(V), C.c_int32(n_out),
                                     C.c_int32(blank), C.c_int32(eos), C.c_float(weight), _stream()), "js2t_ctc_prefix_step")
    return local, psi, r_new


# ----------------------------------------------------------------------------------------- fused attention
def _mask_strides(mask, B, Tq, Tk):
    if mask is None:
        return 0, 0
    if mask.dtype != torch.bool or mask.dim() != 3 or mask.shape[2] != Tk or not mask.is_contiguous():
        raise Js2tError(f"attention mask must be contiguous bool [B|1, 1|Tq, Tk], got {tuple(mask.shape)} {mask.dtype}")
    if mask.shape[1] not in (1, Tq) or mask.shape[0] not in (1, B):
        raise Js2tError(f"attention mask shape {tuple(mask.shape)} does not broadcast to [{B},{Tq},{Tk}]")
    return (0 if mask.shape[0] == 1 else mask.shape[1] * Tk), (0 if mask.shape[1] == 1 else Tk)


def flash_supported(q_t, k_t, v_t, dh: int) -> bool:
    """Fused attention kernel constraints: bf16, head size 128 or 64, 16-byte aligned rows."""
    ok = q_t.dtype == torch.bfloat16 and dh in (128, 64) and k_t.shape[0] >= 1
    for t in (q_t, k_t, v_t):
        ok = ok and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
    return ok


def _attn_desc(q_t, q_off, k_t, k_off, v_t, v_off, B, H, Tq, Tk, dh, mask, p, rng, site, rel_bias=None):
    d = AttnDesc()
    if rel_bias is not None:
        if rel_bias.dtype != torch.float32 or rel_bias.dim() != 2 or rel_bias.shape[0] != H or rel_bias.shape[1] % 2 != 1 or \
                not rel_bias.is_contiguous():
            raise Js2tError(f"relative-position bias must be contiguous float32 [H, 2R+1], got {tuple(rel_bias.shape)} {rel_bias.dtype}")
        d.rel_bias, d.rel_R = rel_bias.data_ptr(), (rel_bias.shape[1] - 1) // 2
    es = q_t.element_size()
    d.q, d.k, d.v = q_t.data_ptr() + q_off * es, k_t.data_ptr() + k_off * es, v_t.data_ptr() + v_off * es
    d.ldq, d.ldk, d.ldv = q_t.stride(0), k_t.stride(0), v_t.stride(0)
    d.B, d.H, d.Tq, d.Tk, d.head_dim = B, H, Tq, Tk, dh
    d.scale, d.dropout_p = 1.0 / math.sqrt(dh), float(p)
    if mask is not None:
        d.mask = mask.data_ptr()
        d.mask_sb, d.mask_sq = _mask_strides(mask, B, Tq, Tk)
    if p > 0:
        d.rng_state, d.rng_stream = rng.state.data_ptr(), int(site)
    return d


class PackedRows:
    """Where the utterances of a ragged batch live once their dead positions are dropped (js2t_pack_rows): entry b owns rows
    seg[b] .. seg[b+1] of a [rows, C] buffer; B entries of at most T positions; rows >= seg[B] (rounded up by the caller, the
    tail is kept zero)."""
    __slots__ = ("seg", "B", "T", "rows")

    def __init__(self, seg: torch.Tensor, B: int, T: int, rows: int):
        if seg.dtype != torch.int32 or seg.numel() != B + 1 or not seg.is_contiguous():
            raise Js2tError("PackedRows: seg must be a contiguous int32 [B + 1] tensor")
        self.seg, self.B, self.T, self.rows = seg, int(B), int(T), int(rows)

    @staticmethod
    def from_lengths(lengths, T: int, device, round_to: int = 1) -> "PackedRows":
        """lengths: HOST integers (no device sync), each 1..T."""
        lens = [int(v) for v in lengths]
        if not lens or min(lens) < 1 or max(lens) > T:
            raise Js2tError(f"PackedRows: lengths must lie in 1..{T}")
        off = [0]
        for n in lens:
            off.append(off[-1] + n)
        rows = -(-off[-1] // round_to) * round_to
        seg = torch.tensor(off, dtype=torch.int32).pin_memory().to(device, non_blocking=True) if torch.device(device).type == "cuda" \
            else torch.tensor(off, dtype=torch.int32)
        return PackedRows(seg, len(lens), T, rows)


def pack_rows(x2d, pk: PackedRows):
    """[B*T, C] -> [pk.rows, C]: the live rows of every entry, back to back; the tail zeroed."""
    _dev(x2d, pk.seg)
    if x2d.dim() != 2 or x2d.shape[0] != pk.B * pk.T or not x2d.is_contiguous() or (x2d.shape[1] * x2d.element_size()) % 16:
        raise Js2tError(f"pack_rows: contiguous [{pk.B * pk.T}, C] rows of a multiple of 16 bytes, got {tuple(x2d.shape)}")
    out = torch.empty((pk.rows, x2d.shape[1]), dtype=x2d.dtype, device=x2d.device)
    check(lib().js2t_pack_rows(C.c_void_p(x2d.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(pk.seg.data_ptr()), pk.B, pk.T,
                               C.c_int64(pk.rows), C.c_int64(x2d.shape[1] * x2d.element_size()), 1, _stream()), "js2t_pack_rows")
    return out


def unpack_rows(xp, pk: PackedRows):
    """[pk.rows, C] -> [B*T, C]: every entry at b*T again, zeros behind its length."""
    _dev(xp, pk.seg)
    if xp.dim() != 2 or xp.shape[0] != pk.rows or not xp.is_contiguous() or (xp.shape[1] * xp.element_size()) % 16:
        raise Js2tError(f"unpack_rows: contiguous [{pk.rows}, C] rows of a multiple of 16 bytes, got {tuple(xp.shape)}")
    out = torch.empty((pk.B * pk.T, xp.shape[1]), dtype=xp.dtype, device=xp.device)
    check(lib().js2t_pack_rows(C.c_void_p(xp.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(pk.seg.data_ptr()), pk.B, pk.T,
                               C.c_int64(pk.rows), C.c_int64(xp.shape[1] * xp.element_size()), 0, _stream()), "js2t_pack_rows")
    return out


def zero_tail_rows(buf, pk: PackedRows):
    """rows seg[B] .. pk.rows of a contiguous [pk.rows, C] buffer zeroed (what the packed attention kernels do not write)."""
    _dev(buf, pk.seg)
    if buf.dim() != 2 or buf.shape[0] != pk.rows or not buf.is_contiguous() or (buf.shape[1] * buf.element_size()) % 16:
        raise Js2tError(f"zero_tail_rows: contiguous [{pk.rows}, C] rows of a multiple of 16 bytes, got {tuple(buf.shape)}")
    check(lib().js2t_pack_rows(C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), C.c_void_p(pk.seg.data_ptr()), pk.B, pk.T,
                               C.c_int64(pk.rows), C.c_int64(buf.shape[1] * buf.element_size()), 2, _stream()), "js2t_pack_rows")


def _seg_check(pk, q_t, B, Tq, Tk, k_t=None):
    """k_t given: cross-attention over packed KEYS (js2t_attn_desc.seg_keys) - the queries stay [B * Tq]."""
    if k_t is not None:
        if pk.B != B or pk.T != Tk or k_t.shape[0] != pk.rows or q_t.shape[0] != B * Tq:
            raise Js2tError(f"flash attention over packed keys: {pk.B} entries of <= {pk.T} keys in {pk.rows} rows do not match "
                            f"B={B} Tq={Tq} Tk={Tk} key rows={k_t.shape[0]} query rows={q_t.shape[0]}")
    elif pk.B != B or pk.T != Tq or Tq != Tk or q_t.shape[0] != pk.rows:
        raise Js2tError(f"flash attention over packed rows: {pk.B} entries of <= {pk.T} positions in {pk.rows} rows do not match "
                        f"B={B} Tq={Tq} Tk={Tk} rows={q_t.shape[0]}")
    _dev(pk.seg)


def flash_attn_fwd(q_t, q_off, k_t, k_off, v_t, v_off, B, H, Tq, Tk, dh, mask, p, rng, site, rel_bias=None, seg: PackedRows = None,
                   seg_keys: bool = False):
    """seg: self-attention over packed rows (q_t / k_t / v_t are [seg.rows, ...]; B, Tq == Tk = the padded geometry); with seg_keys
    cross-attention over packed KEYS (k_t / v_t are [seg.rows, ...], q_t stays [B * Tq, ...]; Tk = the longest entry)."""
    _dev(q_t, k_t, v_t, mask, rel_bias)
    rows = B * Tq
    if seg is not None:
        _seg_check(seg, q_t, B, Tq, Tk, k_t if seg_keys else None)
        rows = B * Tq if seg_keys else seg.rows
    out = torch.empty((rows, H * dh), dtype=q_t.dtype, device=q_t.device)
    lse = torch.empty((B * H, Tq), dtype=torch.float32, device=q_t.device)
    d = _attn_desc(q_t, q_off, k_t, k_off, v_t, v_off, B, H, Tq, Tk, dh, mask, p, rng, site, rel_bias)
    d.o, d.ldo, d.lse = out.data_ptr(), out.stride(0), lse.data_ptr()
    if seg is not None:
        d.seg, d.seg_rows, d.seg_keys = seg.seg.data_ptr(), seg.rows, int(bool(seg_keys))  # rows no entry owns are zeroed by the kernel
    check(lib().js2t_flash_attn_fwd(C.byref(d), _stream()), "js2t_flash_attn_fwd")
    return out, lse


def flash_attn_bwd(dout, out, lse, q_t, q_off, k_t, k_off, v_t, v_off, dq_t, dq_off, dk_t, dk_off, dv_t, dv_off, B, H, Tq, Tk,
                   dh, mask, p, rng, site, rel_bias=None, d_rel_bias=None, delta_partial=None, seg: PackedRows = None, seg_keys: bool = False):
    """d_rel_bias (f32, shape of rel_bias): the bias gradient is ADDED into it (atomics): zero it unless accumulating.
    seg: packed rows, as in flash_attn_fwd; rows of dq / dk / dv no entry owns are zeroed by the kernels (js2t_attn_desc.seg_rows).
    delta_partial f32 [B*Tq, H*dh // 64]: rowsum(dout * out) as partial sums per 64-column group (gemm(dot=...) of the product that
    made dout) - the two passes then run as one grid and `out` is not read."""
    _dev(dout, out, lse, q_t, k_t, v_t, dq_t, dk_t, dv_t, mask, rel_bias, d_rel_bias, delta_partial)
    d = _attn_desc(q_t, q_off, k_t, k_off, v_t, v_off, B, H, Tq, Tk, dh, mask, p, rng, site, rel_bias)
    nrows = B * Tq
    if seg is not None:
        _seg_check(seg, q_t, B, Tq, Tk, k_t if seg_keys else None)
        d.seg, d.seg_rows, d.seg_keys = seg.seg.data_ptr(), seg.rows, int(bool(seg_keys))
        if not seg_keys:
            nrows = seg.rows
    if d_rel_bias is not None:
        if rel_bias is None or d_rel_bias.shape != rel_bias.shape or d_rel_bias.dtype != torch.float32 or not d_rel_bias.is_contiguous():
            raise Js2tError("d_rel_bias must be a contiguous float32 tensor shaped like rel_bias")
        d.d_rel_bias = d_rel_bias.data_ptr()
    es = q_t.element_size()
    if delta_partial is not None:
        if delta_partial.dtype != torch.float32 or delta_partial.numel() != nrows * (H * dh // 64) or not delta_partial.is_contiguous():
            raise Js2tError("flash_attn_bwd: delta_partial must be contiguous f32 [B*Tq, H*dh // 64]")
        d.delta_partial, d.delta_groups = delta_partial.data_ptr(), H * dh // 64
        d.o, d.ldo, d.lse = out.data_ptr(), out.stride(0), lse.data_ptr()
    else:
        delta = torch.empty_like(lse)
        d.o, d.ldo, d.lse, d.delta = out.data_ptr(), out.stride(0), lse.data_ptr(), delta.data_ptr()
    d.d_o, d.ld_do = dout.data_ptr(), dout.stride(0)
    d.dq, d.ld_dq = dq_t.data_ptr() + dq_off * es, dq_t.stride(0)
    d.dk, d.ld_dk = dk_t.data_ptr() + dk_off * es, dk_t.stride(0)
    d.dv, d.ld_dv = dv_t.data_ptr() + dv_off * es, dv_t.stride(0)
    check(lib().js2t_flash_attn_bwd(C.byref(d), _stream()), "js2t_flash_attn_bwd")
