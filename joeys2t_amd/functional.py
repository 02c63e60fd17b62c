"""Differentiable building blocks of the JoeyS2T hot path, each backed by HIP kernels only.

Granularity: one torch.autograd.Function per *residual block* (self-attention, cross-attention,
feed-forward) plus a few leaf functions (conv+GLU, positional encoding, embedding, layer norm, linear).
Inside a block, forward and backward are explicit sequences of C-ABI launches (ops.py) so that every
FLOP of the path — including residual-gradient sums and dropout masks — runs in our kernels; torch's
autograd engine only routes gradients between blocks and into `param.grad`.

Reference semantics reproduced (file:line in the reference repo):
  * MultiHeadedAttention.forward      transformer_layers.py:49-115  (k,v,q order; q scaled by 1/sqrt(dh))
  * PositionwiseFeedForward.forward   transformer_layers.py:159-168
  * TransformerEncoderLayer.forward   transformer_layers.py:267-289
  * TransformerDecoderLayer.forward   transformer_layers.py:348-407
  * Conv1dSubsampler.forward          encoders.py:354-373
"""
import math
from typing import List, Optional

import os

import torch

from joeys2t_amd import ops
from joeys2t_amd.ops import DropoutRng

LN_EPS = 1e-6  # every nn.LayerNorm on the path uses eps=1e-6 (transformer_layers.py:146,248,339-340)


# ------------------------------------------------------------------------------------------------
# plain (non-autograd) helpers: forward / backward of the primitive pieces
# ------------------------------------------------------------------------------------------------
def _row_major_2d(t: torch.Tensor) -> torch.Tensor:
    if t.dim() != 2 or t.stride(1) != 1:
        raise ops.Js2tError(f"expected a row-major 2-D tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    return t


# fp8 forward mode (BASELINE config 5, "fp8 MFMA"; an extension - the reference computes in fp32 / fp16 only): forward products
# of nn.Linear layers on e4m3 operands with per-tensor scales (weights quantised once per optimizer step, activations with a
# delayed scale that adapts on the device); backward keeps the bf16 activations and weights it has anyway.
# Which products: those whose INPUT is quantised by the kernel that produces it - a LayerNorm in front of the Linear writes
# e4m3 next to (no_grad: instead of) bf16 while the normalised row is in registers (js2t_layernorm_fwd_fp8): q/k/v projections,
# first feed-forward layers.  A separate quantisation pass per input (round 2: every eligible Linear, FP8_SEPARATE_PASS) costs
# more than the e4m3 product saves: 8.8 ms against 6.6 ms in bf16 on the Conformer encoder of bench.py.
FP8_FORWARD = os.environ.get("JS2T_FP8_FORWARD", "0") == "1"
FP8_SEPARATE_PASS = os.environ.get("JS2T_FP8_SEPARATE_PASS", "0") == "1"  # also Linears whose input needs a pass of its own
_FP8_WEIGHTS = {}  # (data_ptr, shape) -> (ops.WEIGHT_VERSION, e4m3 weight, scale f32[1])
_FP8_STATES = {}   # (weight data_ptr, shape) -> delayed-scaling state of the activations that meet this weight
FP8_DELAYED = os.environ.get("JS2T_FP8_DELAYED", "1") != "0"  # one quantisation pass with the previous call's scale


def _fp8_weight(w):
    key = (w.data_ptr(), tuple(w.shape))
    hit = _FP8_WEIGHTS.get(key)
    if hit is None or hit[0] != ops.WEIGHT_VERSION:
        w8, ws = ops.quantize_fp8(w)
        hit = _FP8_WEIGHTS[key] = (ops.WEIGHT_VERSION, w8, ws)
    return hit[1], hit[2]


def _fp8_eligible(x2d, w, out_dtype, preact, alpha) -> bool:
    M, K = x2d.shape
    N = w.shape[0]
    return (FP8_FORWARD and x2d.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and (out_dtype in (None, torch.bfloat16)) and
            preact is None and alpha == 1.0 and x2d.is_contiguous() and w.is_contiguous() and K % 16 == 0 and K >= 128 and
            N % 8 == 0 and N >= 128 and M >= 1)


def _fp8_state(w, calibrate_on):
    """delayed-scaling state of the activations that meet weight `w`; created (calibrated on `calibrate_on`) at first use"""
    key = (w.data_ptr(), tuple(w.shape))
    st = _FP8_STATES.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise ops.Js2tError("fp8 forward: run one eager step before capturing (activation scales are calibrated on first use)")
        st = _FP8_STATES[key] = ops.new_fp8_state(calibrate_on)
    return st


def ln_fwd_fp8(x2d, gamma, beta, w, want_y: bool):
    """LayerNorm in front of the Linear with weight `w`, emitting the e4m3 operand of that product: -> (n | None, mean, rstd, x8)
    with x8 = (e4m3 tensor, scale f32[1] = s_x * s_w) for linear_fwd(x8=...)."""
    w8, ws = _fp8_weight(w)
    key = (w.data_ptr(), tuple(w.shape))
    if key not in _FP8_STATES:  # calibrate on this call's own LayerNorm output (eager, once)
        _fp8_state(w, ops.layernorm_fwd(x2d, gamma, beta, LN_EPS)[0])
    n, mean, rstd, y8, sc = ops.layernorm_fwd_fp8(x2d, gamma, beta, LN_EPS, _FP8_STATES[key], mul=ws, want_y=want_y)
    return n, mean, rstd, (y8, sc, _FP8_STATES[key])


def linear_fwd(x2d, w, b, *, act=None, dropout_p=0.0, rng=None, site=0, residual=None, res_scale=1.0,
               out_dtype=None, preact=None, alpha=1.0, ln=None, rs_partial=None, x8=None, y8_for=None, want_y=True):
    """y[M,N] = epilogue(x2d[M,K] @ w[N,K]^T + b) — one js2t_gemm launch.  ln / rs_partial: see ops.gemm (LayerNorm fold).
    x8 = (e4m3 x, scale, state): the input already quantised by its producer (ln_fwd_fp8 or an e4m3 product's second output):
    e4m3 product, x2d only supplies the shape.  y8_for = weight of the FOLLOWING Linear (with x8 only): the result is also
    written as e4m3 for that product -> returns (y | None when not want_y, x8 triple for the next linear_fwd)."""
    M, K = (x8[0].shape if x8 is not None else x2d.shape)
    _row_major_2d(w)
    N = w.shape[0]
    if x8 is not None:
        y = torch.empty((M, N), dtype=torch.bfloat16, device=w.device) if (want_y or y8_for is None) else None
        w8, _ = _fp8_weight(w)
        c8 = nxt = None
        if y8_for is not None:
            _, ws_next = _fp8_weight(y8_for)
            st_next = _FP8_STATES[(y8_for.data_ptr(), tuple(y8_for.shape))]  # calibrated by the caller on a first bf16 pass
            out8 = torch.empty((M, N), dtype=torch.float8_e4m3fn, device=w.device)
            sc_next = torch.empty((1, ), dtype=torch.float32, device=w.device)
            c8, nxt = (out8, st_next, ws_next, sc_next), (out8, sc_next, st_next)
        ops.gemm(x8[0], w8, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, act=act, dropout_p=dropout_p, rng=rng, rng_stream=site,
                 residual=residual, ldr=0 if residual is None else residual.stride(0), res_scale=res_scale, alpha_dev=x8[1],
                 fp8_state=x8[2], c8=c8)
        return y if y8_for is None else (y, nxt)
    _row_major_2d(x2d)
    y = torch.empty((M, N), dtype=out_dtype or x2d.dtype, device=x2d.device)
    if (ln is None and rs_partial is None and FP8_SEPARATE_PASS and _fp8_eligible(x2d, w, out_dtype, preact, alpha) and
            act in (None, "relu")):
        w8, ws = _fp8_weight(w)
        if FP8_DELAYED:
            x8q, sc = ops.quantize_fp8_delayed(x2d, _fp8_state(w, x2d), mul=ws)
        else:
            x8q, sc = ops.quantize_fp8(x2d, mul=ws)  # sc = s_x * s_w on the device: the GEMM's alpha_dev
        ops.gemm(x8q, w8, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, act=act, dropout_p=dropout_p, rng=rng, rng_stream=site,
                 residual=residual, ldr=0 if residual is None else residual.stride(0), res_scale=res_scale, alpha_dev=sc)
        return y
    ops.gemm(x2d, w, y, M=M, N=N, K=K, lda=x2d.stride(0), ldb=w.stride(0), ldc=N, bias=b, act=act, preact=preact,
             dropout_p=dropout_p, rng=rng, rng_stream=site, residual=residual,
             ldr=0 if residual is None else residual.stride(0), res_scale=res_scale, alpha=alpha, ln=ln, rs_partial=rs_partial)
    return y


def linear_bwd(dz2d, x2d, w, *, need_dx=True, need_dw=True, need_db=True, gate=None, gate_scale=1.0, dx_dtype=None,
               dw_out=None, db_out=None, queue=None, w_t=None, dx_add=None, dot_src=None):
    """Gradients of y = x w^T + b given dz = dL/dy.  Returns (dx, dW_f32, db_f32).
    dw_out / db_out: fp32 gradient buffers (views of the flat gradient store) to ACCUMULATE into; the corresponding
    return value is then None (nothing left for autograd to add)."""
    _row_major_2d(dz2d), _row_major_2d(x2d), _row_major_2d(w)
    M, N = dz2d.shape
    K = x2d.shape[1]
    dx = dw = db = None
    if need_dx:
        # dx[M,K] = dz[M,N] . W[N,K]   (B(n'=k, red=n) lives at W[n*ldw + k] -> trans_b)
        dx = torch.empty((M, K), dtype=dx_dtype or x2d.dtype, device=x2d.device)
        extra = {}
        if dx_add is not None:  # dx = dz . W + dx_add in the product's epilogue (a gradient that accumulates over several users of x)
            assert gate is None and dx_add.shape == dx.shape and dx_add.dtype == dx.dtype
            extra = dict(residual=dx_add, ldr=dx_add.stride(0), res_scale=1.0)
        if w_t is not None and w_t.dtype == dz2d.dtype:
            # W^T [K, N] is at hand (ParamStore.view_t): both operands k-contiguous -> plain product, register-direct epilogue
            if (dot_src is not None and FUSE_ATTN_DELTA and gate is None and dx_add is None and K % 128 == 0 and dx.dtype == torch.bfloat16 and
                    dot_src.dtype == torch.bfloat16 and dot_src.shape == dx.shape and dot_src.stride(1) == 1 and dot_src.stride(0) % 8 == 0 and
                    dot_src.data_ptr() % 16 == 0 and _mfma_operand(dz2d) and _mfma_operand(w_t) and N % 8 == 0):
                # dx is an attention block's dO and dot_src its saved output O: the epilogue leaves rowsum(dO * O) per 64-column
                # group behind (LINEAR_BWD_DOT: read by the caller) - the attention backward then needs no pass of its own for it
                part = torch.empty((M, K // 64), dtype=torch.float32, device=dx.device)
                extra = dict(extra, dot=(dot_src, part))
                LINEAR_BWD_DOT[0] = part
            ops.gemm(dz2d, w_t, dx, M=M, N=K, K=N, lda=dz2d.stride(0), ldb=w_t.stride(0), ldc=K, gate=gate,
                     ldg=0 if gate is None else gate.stride(0), gate_scale=gate_scale, **extra)
        else:
            ops.gemm(dz2d, w, dx, M=M, N=K, K=N, lda=dz2d.stride(0), ldb=w.stride(0), ldc=K, trans_b=True, gate=gate,
                     ldg=0 if gate is None else gate.stride(0), gate_scale=gate_scale, **extra)
    # the bias gradient (column sums of dz) rides on the weight-gradient product when that runs on the bf16 LDS-DMA
    # kernel: its dz tiles are already in LDS, so no separate pass over dz is needed
    fuse_db = need_dw and need_db and _mfma_operand(dz2d) and _mfma_operand(x2d)
    if (queue is not None and need_dw and dw_out is not None and (not need_db or (fuse_db and db_out is not None))
            and _mfma_operand(dz2d) and _mfma_operand(x2d)):
        # deferred: runs with the same product of the other layers as one grouped launch (runtime.WgradQueue)
        queue.add(dz2d, x2d, dw_out, db_out if need_db else None)
        need_dw = need_db = False
    if need_db and db_out is None:
        db = torch.zeros((N, ), dtype=torch.float32, device=x2d.device) if fuse_db else None
    rs = (db_out if db_out is not None else db) if fuse_db else None
    if need_dw:
        # dW[N,K] = dz^T[N,M] . x[M,K]  (both operands reduction-major -> trans_a, trans_b), f32 output
        sk = wgrad_split(N, K, M)
        if dw_out is not None:
            # split-K atomics add onto whatever is there; a single-slice GEMM accumulates through beta = 1
            ops.gemm(dz2d, x2d, dw_out, M=N, N=K, K=M, lda=dz2d.stride(0), ldb=x2d.stride(0), ldc=K, trans_a=True,
                     trans_b=True, split_k=sk, beta=0.0 if sk > 1 else 1.0, a_rowsum=rs)
        else:
            dw = (torch.zeros if sk > 1 else torch.empty)((N, K), dtype=torch.float32, device=x2d.device)
            ops.gemm(dz2d, x2d, dw, M=N, N=K, K=M, lda=dz2d.stride(0), ldb=x2d.stride(0), ldc=K, trans_a=True, trans_b=True,
                     split_k=sk, a_rowsum=rs)
    if need_db and not fuse_db:
        dzc = dz2d if dz2d.is_contiguous() else dz2d.contiguous()
        if db_out is not None:
            ops.colsum(dzc, out=db_out, accumulate=True)
        else:
            db = ops.colsum(dzc)
    if db_out is not None:
        db = None
    return dx, dw, db


# rowsum(dO * O) of an attention block from its output projection's input-gradient product (linear_bwd(dot_src=O)): the partial
# sums of the last such call (None: the product could not carry it), taken by the block's backward right after
FUSE_ATTN_DELTA = os.environ.get("JS2T_FUSE_ATTN_DELTA", "1") != "0"
LINEAR_BWD_DOT = [None]


def _mfma_operand(t: torch.Tensor) -> bool:
    """True when js2t_gemm takes `t` on the bf16 LDS-DMA path (16-byte aligned rows)."""
    return t.dtype == torch.bfloat16 and t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0


_WGRAD_BLOCKS = int(os.environ.get("JS2T_WGRAD_BLOCKS", "512"))


def wgrad_split(rows: int, cols: int, red: int, count: int = 1) -> int:
    """Split-K factor for `count` weight-gradient GEMMs [rows, cols] = sum over `red` tokens launched together: one
    output has only rows*cols/128^2 tiles (16..64 for the 512/2048-wide layers) against 256 CUs, so the token
    dimension is cut until up to two blocks per CU exist."""
    tiles = ((rows + 127) // 128) * ((cols + 127) // 128) * count
    nk = (red + 63) // 64
    # at most two blocks per CU, rounded DOWN: a slice more costs another pass of f32 atomics over the whole output
    # (6 x dW[2048,512] over 2592 tokens: 46.7 us unsplit, 76.2 us in two slices - tools/wgrad_small.py; whole train
    # step, same box: 14.44 ms rounding up, 14.31 rounding down, 14.33-14.36 with 320-384 blocks as the target)
    return max(1, min(_WGRAD_BLOCKS // tiles, nk // 4 if nk >= 8 else 1, 32))


class AttnShape:
    __slots__ = ("B", "Tq", "Tk", "H", "dh", "ld", "seg", "seg_keys")

    def __init__(self, B, Tq, Tk, H, dh, seg=None, seg_keys=False):
        self.B, self.Tq, self.Tk, self.H, self.dh = B, Tq, Tk, H, dh
        self.seg = seg  # ops.PackedRows: self-attention over the packed rows of a ragged batch (fused kernels only)
        self.seg_keys = bool(seg_keys)  # seg describes the KEY side only: cross-attention over packed encoder states
        self.ld = ops.round_up(Tk, 8)  # score row stride: 16-byte rows for the bf16 GEMMs


USE_FLASH = True  # fused attention kernels when the shapes allow (bf16, head size 128); tests flip this to compare


def attn_fwd(q_t, q_off, k_t, k_off, v_t, v_off, shp: AttnShape, mask, p, rng, site, need_probs=False, rel_bias=None):
    """softmax(mask(q k^T / sqrt(dh))) v for all heads; q_t/k_t/v_t are [B*T, ld*] row-major 2-D buffers whose
    head h lives at column off + h*dh.  Returns (ctx[B*Tq, H*dh], P, Pd) on the materialised path and
    (ctx, None, lse) on the fused path (P is None)."""
    B, Tq, Tk, H, dh, ld = shp.B, shp.Tq, shp.Tk, shp.H, shp.dh, shp.ld
    dev, dt = q_t.device, q_t.dtype
    Z = B * H
    if USE_FLASH and not need_probs and ops.flash_supported(q_t, k_t, v_t, dh):
        # fused kernel: scores / probabilities stay on chip; the forward keeps (out, lse) for backward
        out, lse = ops.flash_attn_fwd(q_t, q_off, k_t, k_off, v_t, v_off, B, H, Tq, Tk, dh, mask, p, rng, site, rel_bias, seg=shp.seg,
                                      seg_keys=shp.seg_keys)
        return out, None, lse
    if shp.seg is not None:
        raise ops.Js2tError("attention over packed rows needs the fused kernels (bf16, head size 64 / 128): the encoder packs only then")
    S = torch.empty((Z, Tq, ld), dtype=dt, device=dev)
    ops.gemm(q_t, k_t, S, M=Tq, N=Tk, K=dh, lda=q_t.stride(0), ldb=k_t.stride(0), ldc=ld, batch=Z, batch_inner=H,
             a_strides=(Tq * q_t.stride(0), dh), b_strides=(Tk * k_t.stride(0), dh), c_strides=(H * Tq * ld, Tq * ld),
             a_off=q_off, b_off=k_off, alpha=1.0 / math.sqrt(dh))
    if rel_bias is not None:  # materialised path (fp32 compute / head sizes the fused kernels do not take)
        _check_rel_bias(rel_bias, H)
        ops.rel_bias_add(S, rel_bias, B, H, Tq, Tk, ld)
    P, Pd = ops.softmax_fwd(S, mask, B, H, Tq, Tk, ld, p, rng, site)
    ctx = torch.empty((B * Tq, H * dh), dtype=dt, device=dev)
    ops.gemm(Pd, v_t, ctx, M=Tq, N=dh, K=Tk, lda=ld, ldb=v_t.stride(0), ldc=H * dh, trans_b=True, batch=Z, batch_inner=H,
             a_strides=(H * Tq * ld, Tq * ld), b_strides=(Tk * v_t.stride(0), dh), c_strides=(Tq * H * dh, dh), b_off=v_off)
    return ctx, P, Pd


def _check_rel_bias(t, H):
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[0] != H or t.shape[1] % 2 != 1 or not t.is_contiguous():
        raise ops.Js2tError(f"relative-position bias must be contiguous float32 [H, 2R+1], got {tuple(t.shape)} {t.dtype}")


def attn_bwd(dctx, q_t, q_off, k_t, k_off, v_t, v_off, dq_t, dq_off, dk_t, dk_off, dv_t, dv_off, shp: AttnShape, P, Pd, p,
             rng, site, ctx_out=None, mask=None, rel_bias=None, d_rel_bias=None, delta_partial=None):
    """Writes dq/dk/dv into the given (row-major 2-D) gradient buffers at the given column offsets.
    Fused path: P is None, Pd carries the forward's row log-sum-exp and ctx_out its output."""
    B, Tq, Tk, H, dh, ld = shp.B, shp.Tq, shp.Tk, shp.H, shp.dh, shp.ld
    if P is None:
        ops.flash_attn_bwd(dctx, ctx_out, Pd, q_t, q_off, k_t, k_off, v_t, v_off, dq_t, dq_off, dk_t, dk_off, dv_t, dv_off,
                           B, H, Tq, Tk, dh, mask, p, rng, site, rel_bias, d_rel_bias, delta_partial=delta_partial, seg=shp.seg,
                           seg_keys=shp.seg_keys)
        return
    Z = B * H
    dev, dt = dctx.device, dctx.dtype
    sP = (H * Tq * ld, Tq * ld)
    scale = 1.0 / math.sqrt(dh)
    # dPd[q,k] = dctx[q,:] . v[k,:]
    dPd = torch.empty((Z, Tq, ld), dtype=dt, device=dev)
    ops.gemm(dctx, v_t, dPd, M=Tq, N=Tk, K=dh, lda=dctx.stride(0), ldb=v_t.stride(0), ldc=ld, batch=Z, batch_inner=H,
             a_strides=(Tq * dctx.stride(0), dh), b_strides=(Tk * v_t.stride(0), dh), c_strides=sP, b_off=v_off)
    dS = ops.softmax_bwd(P, dPd, Z, Tq, Tk, ld, p, rng, site)
    if d_rel_bias is not None:
        _check_rel_bias(d_rel_bias, H)
        ops.rel_bias_grad(dS, d_rel_bias, B, Tq=Tq, Tk=Tk, H=H, ld=ld)
    # dv[k,:] = sum_q Pd[q,k] dctx[q,:]
    ops.gemm(Pd, dctx, dv_t, M=Tk, N=dh, K=Tq, lda=ld, ldb=dctx.stride(0), ldc=dv_t.stride(0), trans_a=True, trans_b=True,
             batch=Z, batch_inner=H, a_strides=sP, b_strides=(Tq * dctx.stride(0), dh),
             c_strides=(Tk * dv_t.stride(0), dh), c_off=dv_off)
    # dq[q,:] = scale * sum_k dS[q,k] k[k,:]
    ops.gemm(dS, k_t, dq_t, M=Tq, N=dh, K=Tk, lda=ld, ldb=k_t.stride(0), ldc=dq_t.stride(0), trans_b=True, batch=Z,
             batch_inner=H, a_strides=sP, b_strides=(Tk * k_t.stride(0), dh), c_strides=(Tq * dq_t.stride(0), dh),
             b_off=k_off, c_off=dq_off, alpha=scale)
    # dk[k,:] = scale * sum_q dS[q,k] q[q,:]
    ops.gemm(dS, q_t, dk_t, M=Tk, N=dh, K=Tq, lda=ld, ldb=q_t.stride(0), ldc=dk_t.stride(0), trans_a=True, trans_b=True,
             batch=Z, batch_inner=H, a_strides=sP, b_strides=(Tq * q_t.stride(0), dh),
             c_strides=(Tk * dk_t.stride(0), dh), b_off=q_off, c_off=dk_off, alpha=scale)


def _split_rows(t: Optional[torch.Tensor], sizes: List[int]):
    if t is None:
        return [None] * len(sizes)
    return list(torch.split(t, sizes, dim=0))


# ------------------------------------------------------------------------------------------------
# block descriptors: what a residual block needs to know (weights are passed as tensors)
# ------------------------------------------------------------------------------------------------
class BlockCfg:
    """Static configuration of one residual block (no tensors)."""

    def __init__(self, *, kind, num_heads=1, alpha=1.0, ln_mode="pre", act="relu", training=True, attn_dropout=0.0,
                 out_dropout=0.0, need_weights=False):
        assert ln_mode in ("pre", "post", "none")
        self.kind = kind  # "self" | "cross" | "ffn"
        self.H = num_heads
        # p_in: dropout on attention probabilities (self/cross) or on the activated hidden layer (ffn);
        # p_out: dropout on the block's last projection, before the residual add.
        self.p_in = float(attn_dropout) if training else 0.0
        self.p_out = float(out_dropout) if training else 0.0
        self.alpha = float(alpha)
        self.ln_mode = ln_mode
        self.act = act
        self.need_weights = need_weights

    @property
    def any_dropout(self) -> bool:
        return self.p_in > 0 or self.p_out > 0


def _ln_fwd(x2d, gamma, beta):
    return ops.layernorm_fwd(x2d, gamma, beta, LN_EPS)


# Hand-over of a fused product between two ResidualBlockFn nodes of a pre-LN stack.  Block i writes y = alpha x +
# drop(core(LN(x))); block i+1 normalises y first, and its LayerNorm backward produces exactly the gradient block i's
# backward then multiplies with its output-dropout mask.  Forward: block i leaves (p_out, site, rng) under y's address;
# block i+1 picks it up.  Backward: block i+1 lets its LayerNorm backward emit the masked copy as well and leaves it under
# the address of the gradient it returns; block i takes it only if the gradient it receives is that very tensor (no
# fan-out accumulated in between) AND (p, site, rng) are its own - anything else falls back to js2t_dropout_bwd.
# Entries hold their tensors, so an address cannot be re-used while it is a key; begin_step() clears both tables.
_DROP_HINT = {}
_DROP_READY = {}
# LayerNorm fold (bf16 pre-LN stacks of width 512): the block that WRITES a residual-stream tensor y lets its last product's
# epilogue also write the partial row sums / sums of squares of y (js2t_gemm rs_partial) and leaves them under y's address;
# the next block, whose first act would be LayerNorm(y), takes them and runs its first product on y itself with gamma-scaled,
# row-centred weights, finishing the normalisation with one multiplication per output element (js2t_gemm ln_partial;
# runtime.ParamStore.fold keeps the derived weights).  No kernel reads a row just to normalise it; the normalised activations,
# which the deferred weight gradient still wants, are re-materialised by the block's LayerNorm BACKWARD while it has x, mean
# and rstd in registers anyway.  Only for products the persistent 192x128 kernel takes (encoder-sized row counts).
_LN_STATS = {}
LN_FOLD = os.environ.get("JS2T_LN_FOLD", "1") != "0"  # tests flip this to compare with the standalone LayerNorm kernel
FUSE_LN_DROPOUT_BWD = os.environ.get("JS2T_LN_DROPOUT_HANDOVER", "1") != "0"  # tests flip this to compare with the separate kernel


# Gradient of the encoder states ("memory"): every decoder layer's cross-attention block contributes dkv . W_kv, and
# autograd would add the six [B*S, d] tensors one launch at a time.  Instead the blocks chain them: forward counts the
# blocks that took a given memory tensor, each block's backward adds the running sum in its product's epilogue
# (linear_bwd(dx_add=...)) and returns None - except the last one to run, which hands the sum to autograd.
# Only between begin_memory_chain() and end_memory_chain() - i.e. inside TrainStep.micro_step(), which runs exactly one
# forward and one backward in between; end_memory_chain() fails loudly if a block that was counted never ran (its share
# and everything chained behind it would be lost).  Anywhere else autograd adds the tensors as usual.
_MEM_USERS = {}  # (data_ptr, shape) -> number of cross blocks whose backward is still to come
_MEM_ACC = {}    # (data_ptr, shape) -> running sum
CHAIN_MEMORY_GRADS = os.environ.get("JS2T_CHAIN_MEMORY_GRADS", "1") != "0"
_CHAIN_ACTIVE = False
# One step further (MemoryKVFn): the K | V projections of the encoder states for ALL decoder layers are one product
# [B*S, d] x [L*2d, d]^T, each cross block reads its 2d columns of the result in place, and in backward writes its dK | dV into
# the same columns of ONE gradient buffer - the last block to run hands that buffer to autograd, and the encoder-state
# gradient is one product over K = L*2d instead of L chained ones (LS100: forward 113 -> 90 us, backward 110 -> 60 us).  Under
# the same bookkeeping, and with gradients only inside the chain (elsewhere the blocks project for themselves).
GROUP_MEMORY_KV = os.environ.get("JS2T_GROUP_MEMORY_KV", "1") != "0"
_KV_USERS = {}   # data_ptr of the grouped projections -> number of cross blocks whose backward is still to come
_KV_GRAD = {}    # data_ptr -> the shared gradient buffer


LOGIT_GRAD_DTYPE = {}   # data_ptr of f32 logits -> compute dtype of the projection that produced them (LinearFn -> loss)
LOGIT_GRAD_READY = {}   # data_ptr of the placeholder gradient autograd carries -> (the gradient in that dtype, the placeholder)


def reset_handover():
    LOGIT_GRAD_DTYPE.clear()
    LOGIT_GRAD_READY.clear()
    _DROP_HINT.clear()
    _DROP_READY.clear()
    _LN_STATS.clear()


def _fold_operand(t: torch.Tensor) -> bool:
    return t.dtype == torch.bfloat16 and t.is_contiguous() and t.data_ptr() % 16 == 0 and t.shape[1] % 8 == 0


def fold_shapes_ok(M: int, N: int, K: int) -> bool:
    """Mirror of the library's shape preconditions for ln_partial / rs_partial (any row count: the persistent 192x128 kernel
    or, below its tile threshold, the 64 / 128-row tile kernel)."""
    return N % 128 == 0 and K % 8 == 0 and K >= 64 and M >= 1


def begin_memory_chain():
    global _CHAIN_ACTIVE
    _MEM_USERS.clear()
    _MEM_ACC.clear()
    _KV_USERS.clear()
    _KV_GRAD.clear()
    _CHAIN_ACTIVE = CHAIN_MEMORY_GRADS


def chain_active() -> bool:
    return _CHAIN_ACTIVE


def end_memory_chain(check: bool = True):
    global _CHAIN_ACTIVE
    _CHAIN_ACTIVE = False
    left = sum(v for v in _MEM_USERS.values() if v != 0) + sum(v for v in _KV_USERS.values() if v != 0)
    _MEM_USERS.clear()
    _MEM_ACC.clear()
    _KV_USERS.clear()
    _KV_GRAD.clear()
    if check and left:
        raise RuntimeError(f"encoder-state gradient chain incomplete: {left} cross-attention backward pass(es) never ran, their "
                           "share of the gradient is lost; set JS2T_CHAIN_MEMORY_GRADS=0")


class PackRowsFn(torch.autograd.Function):
    """[B, T, C] -> [1, rows, C]: the live positions of every utterance back to back (ops.pack_rows); backward scatters the
    gradient back and leaves zeros behind every length."""

    @staticmethod
    def forward(ctx, x, pk):
        ctx.pk, ctx.shape = pk, tuple(x.shape)
        return ops.pack_rows(x.reshape(-1, x.shape[-1]), pk).unsqueeze(0)

    @staticmethod
    def backward(ctx, dy):
        return ops.unpack_rows(dy.reshape(-1, dy.shape[-1]).contiguous(), ctx.pk).view(ctx.shape), None


class UnpackRowsFn(torch.autograd.Function):
    """[1, rows, C] -> [B, T, C] with zeros behind every length; backward gathers the live rows of the gradient."""

    @staticmethod
    def forward(ctx, xp, pk):
        ctx.pk = pk
        return ops.unpack_rows(xp.reshape(-1, xp.shape[-1]), pk).view(pk.B, pk.T, xp.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        return ops.pack_rows(dy.reshape(-1, dy.shape[-1]).contiguous(), ctx.pk).unsqueeze(0), None


class ResidualBlockFn(torch.autograd.Function):
    """One residual block:  pre-LN:  y = drop(core(LN(x))) + alpha*x ;  post-LN: y = LN(drop(core(x)) + alpha*x).

    forward(ctx, cfg, rng, x, memory, mask, wts, *params)
      x       [B,T,d] activations (compute dtype)
      memory  [B,S,d] encoder states for kind == "cross", else None
      mask    bool [B,1|T,Tk] or None
      wts     dict of compute-dtype weight tensors (possibly fused views):
                self : w_in [3d,d] (k;v;q), b_in [3d], w_out [d,d], b_out [d]
                cross: w_q [d,d], b_q, w_kv [2d,d], b_kv, w_out, b_out
                ffn  : w1 [ff,d], b1, w2 [d,ff], b2
              plus ln_g, ln_b (f32)
      params  the leaf nn.Parameters in the order `param_order(kind)`; gradients are returned in this order.
    For cfg.ln_mode == "none" the ln_g/ln_b entries and the two trailing params are omitted.
    Returns y (and head-averaged attention weights for kind == "cross" when cfg.need_weights).
    """

    @staticmethod
    def forward(ctx, cfg: BlockCfg, rng: Optional[DropoutRng], x, memory, mask, wts, *params):
        B, T, d = x.shape
        x2 = x.reshape(B * T, d)
        p_in, p_out = cfg.p_in, cfg.p_out
        sites = [rng.next_site() if rng is not None else 0 for _ in range(2)]
        saved = {}
        # will backward run?  (grad mode is always off inside Function.forward: torch.is_grad_enabled() cannot tell)
        need_bwd = any(ctx.needs_input_grad)
        hint = _DROP_HINT.pop(x.data_ptr(), None)
        ctx.prev_drop = None
        if (hint is not None and FUSE_LN_DROPOUT_BWD and cfg.ln_mode == "pre" and hint[3].shape == x.shape and
                ops.layernorm_bwd_supports_dropout(x2)):
            ctx.prev_drop = hint[:3]  # (p, site, rng) of the block that produced x
        x8_first = x8_second = None  # fp8 forward mode: the products' inputs as e4m3, written by the kernels that produce them
        lnf = None   # (partial sums of x's rows, eps, mean out, rstd out) when this block's LayerNorm is folded into its first product
        fold = wts.get("fold") if (LN_FOLD and cfg.ln_mode == "pre" and not FP8_FORWARD) else None
        sink0 = wts.get("sink") or {}
        if (fold is not None and d == ops.LN_FOLD_WIDTH and _fold_operand(x2) and fold_shapes_ok(B * T, fold.w.shape[0], d) and
                (not torch.is_grad_enabled() or ("_wq" in sink0 and "ln_g" in sink0))):
            hint = _LN_STATS.pop(x.data_ptr(), None)
            # the statistics belong to the tensor the producing block returned, unmodified since (views of one storage share a
            # version counter, an in-place edit in between bumps it): anything else goes through the LayerNorm kernel as before
            if hint is not None and hint[1].shape == x.shape and hint[1].dtype == x.dtype and hint[2] == x._version == hint[1]._version:
                mean = torch.empty((B * T, ), dtype=torch.float32, device=x.device)
                rstd = torch.empty_like(mean)
                lnf = (hint[0], LN_EPS, mean, rstd)
        if lnf is not None:
            n = x2  # the product reads the raw rows; what it computes is LN(x) W^T + b
            saved.update(mean=mean, rstd=rstd, folded=True)
            w_first = {"self": "w_in", "cross": "w_q", "ffn": "w1"}[cfg.kind]
            b_first = {"self": "b_in", "cross": "b_q", "ffn": "b1"}[cfg.kind]
            wts = dict(wts)
            wts["_" + w_first], wts["_" + b_first] = fold.w, fold.bias
        elif cfg.ln_mode == "pre":
            w0 = wts[{"self": "w_in", "cross": "w_q", "ffn": "w1"}[cfg.kind]]
            if (FP8_FORWARD and (cfg.kind != "ffn" or cfg.act in (None, "relu")) and _fp8_eligible(x2, w0, None, None, 1.0) and
                    ops.layernorm_bwd_supports_dropout(x2)):
                # the LayerNorm writes the e4m3 operand of the block's first product itself (and bf16 only if backward wants it)
                n, mean, rstd, x8_first = ln_fwd_fp8(x2, wts["ln_g"], wts["ln_b"], w0, want_y=need_bwd)
            else:
                n, mean, rstd = _ln_fwd(x2, wts["ln_g"], wts["ln_b"])
            saved.update(mean=mean, rstd=rstd)
        else:
            n = x2

        def first(name):  # weight / bias of the block's first product: the gamma-scaled pair when folded
            return wts.get("_" + name, wts[name]) if lnf is not None else wts[name]

        att_w = None
        if cfg.kind == "self":
            H, dh = cfg.H, d // cfg.H
            qkv = linear_fwd(n, first("w_in"), first("b_in"), ln=lnf, x8=x8_first)  # columns: [k | v | q]
            pk = wts.get("pack")  # ops.PackedRows: x is [1, pk.rows, d], the live rows of a ragged batch back to back
            if pk is not None:
                if B != 1 or T != pk.rows or mask is None or mask.shape[0] != pk.B or mask.shape[-1] != pk.T:
                    raise ops.Js2tError(f"self-attention over packed rows: x {tuple(x.shape)} / mask do not match {pk.B} x {pk.T} in {pk.rows} rows")
                shp = AttnShape(pk.B, pk.T, pk.T, H, dh, seg=pk)
            else:
                shp = AttnShape(B, T, T, H, dh)
            c, P, Pd = attn_fwd(qkv, 2 * d, qkv, 0, qkv, d, shp, mask, p_in, rng, sites[0], rel_bias=wts.get("rel_bias"))
            saved.update(qkv=qkv, P=P, Pd=Pd, shp=shp)
        elif cfg.kind == "cross":
            H, dh = cfg.H, d // cfg.H
            q = linear_fwd(n, first("w_q"), first("b_q"), ln=lnf, x8=x8_first)
            kv_off = wts.get("kv_off")
            if kv_off is None:
                S = memory.shape[1]
                m2 = memory.reshape(B * S, memory.shape[2])
                kv = linear_fwd(m2, wts["w_kv"], wts["b_kv"])  # columns: [k | v]
                kv_off = 0
            else:  # `memory` IS the projections of all layers (MemoryKVFn) [B*S, L*2d]: this layer's [k | v] start at column kv_off
                kv, m2 = memory, None
                mem_pack = wts.get("mem_pack")  # ops.PackedRows: the projections of the PACKED encoder states, [rows, L*2d]
                S = kv.shape[0] // B if mem_pack is None else mem_pack.T
                rows = B * S if mem_pack is None else mem_pack.rows
                if kv.dim() != 2 or kv.shape[0] != rows or kv_off + 2 * d > kv.shape[1] or (mem_pack is not None and mem_pack.B != B):
                    raise ops.Js2tError(f"cross block: grouped projections {tuple(kv.shape)} do not hold [k | v] at column {kv_off}")
                if kv.requires_grad:
                    if not _CHAIN_ACTIVE:
                        raise ops.Js2tError("cross block: the grouped K | V projections carry gradients only inside TrainStep.micro_step")
                    key = kv.data_ptr()
                    _KV_USERS[key] = _KV_USERS.get(key, 0) + 1
                    saved["kv_key"] = key
            mem_pack = wts.get("mem_pack") if wts.get("kv_off") is not None else None
            shp = AttnShape(B, T, S, H, dh, seg=mem_pack, seg_keys=mem_pack is not None)
            c, P, Pd = attn_fwd(q, 0, kv, kv_off, kv, kv_off + d, shp, mask, p_in, rng, sites[0], need_probs=cfg.need_weights)
            if cfg.need_weights:
                att_w = ops.attn_head_mean(P, B, H, T, S, shp.ld)
            saved.update(q=q, kv=kv, kv_off=kv_off, m2=m2, P=P, Pd=Pd, shp=shp)
            if m2 is not None and _CHAIN_ACTIVE and memory.requires_grad and _mfma_operand(m2):
                key = (m2.data_ptr(), tuple(m2.shape))
                _MEM_USERS[key] = _MEM_USERS.get(key, 0) + 1
                saved["mem_key"] = key
        else:  # ffn
            pre = None
            if cfg.act != "relu" and cfg.act is not None:
                pre = torch.empty((B * T, wts["w1"].shape[0]), dtype=x.dtype, device=x.device)
            if lnf is not None and pre is not None:
                raise ops.Js2tError("LayerNorm fold: ReLU / no activation only")  # (the caller does not offer a fold otherwise)
            x8_second = None
            w2 = wts["w2"]
            if (x8_first is not None and cfg.act in (None, "relu") and w2.dtype == torch.bfloat16 and w2.is_contiguous() and
                    w2.shape[1] % 16 == 0 and w2.shape[1] >= 128 and w2.shape[0] % 8 == 0 and w2.shape[0] >= 128):
                # e4m3 chain: this product also writes the e4m3 operand of the second feed-forward product (K = ff: where the
                # halved operand bytes matter most); the bf16 copy only if backward will want it
                if (w2.data_ptr(), tuple(w2.shape)) not in _FP8_STATES:  # first use: calibrate the hidden layer's scale on a bf16 pass
                    c = linear_fwd(n, wts["w1"], wts["b1"], act=cfg.act, dropout_p=p_in, rng=rng, site=sites[0], x8=x8_first)
                    _fp8_state(w2, c)
                c, x8_second = linear_fwd(n, wts["w1"], wts["b1"], act=cfg.act, dropout_p=p_in, rng=rng, site=sites[0], x8=x8_first,
                                          y8_for=w2, want_y=need_bwd)
            else:
                c = linear_fwd(n, first("w1"), first("b1"), act=cfg.act, dropout_p=p_in, rng=rng, site=sites[0], preact=pre, ln=lnf, x8=x8_first)
            saved.update(pre=pre)
        w_last, b_last = (wts["w2"], wts["b2"]) if cfg.kind == "ffn" else (wts["w_out"], wts["b_out"])
        # the row statistics of what this block writes, for a following block that folds its LayerNorm (see _LN_STATS)
        out_stats = None
        if (LN_FOLD and cfg.ln_mode == "pre" and cfg.alpha != 0.0 and not FP8_FORWARD and wts.get("emit_stats") and
                b_last is not None and _fold_operand(c) and _fold_operand(x2) and w_last.dtype == torch.bfloat16 and
                d == ops.LN_FOLD_WIDTH and fold_shapes_ok(B * T, d, c.shape[1])):
            out_stats = ops.row_partials(B * T, x.device)
        u = linear_fwd(c, w_last, b_last, dropout_p=p_out, rng=rng, site=sites[1],
                       residual=x2 if cfg.alpha != 0.0 else None, res_scale=cfg.alpha, rs_partial=out_stats,
                       x8=x8_second if cfg.kind == "ffn" else None)
        if cfg.ln_mode != "post":
            y = u
        else:
            y, mean, rstd = _ln_fwd(u, wts["ln_g"], wts["ln_b"])
            saved.update(mean=mean, rstd=rstd, u=u)
        ctx.cfg, ctx.rng, ctx.sites, ctx.wts, ctx.saved = cfg, rng, sites, wts, saved
        ctx.params = params
        ctx.x2, ctx.n, ctx.c = x2, (None if lnf is not None else n), c
        ctx.mask = mask
        ctx.shape = (B, T, d)
        ctx.mem_shape = None if memory is None else tuple(memory.shape)
        ctx.nparams = len(params)
        y = y.view(B, T, d)
        if out_stats is not None:
            if len(_LN_STATS) > 64:
                _LN_STATS.clear()
            _LN_STATS[y.data_ptr()] = (out_stats, y, y._version)
        if p_out > 0 and rng is not None and cfg.ln_mode != "post" and FUSE_LN_DROPOUT_BWD and x.requires_grad:
            if len(_DROP_HINT) > 64:  # forward passes without a training step around them: do not pile up activations
                _DROP_HINT.clear()
            _DROP_HINT[y.data_ptr()] = (p_out, sites[1], rng, y)
        if cfg.kind == "cross":
            if att_w is not None:
                ctx.mark_non_differentiable(att_w)
            return y, att_w
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        cfg, rng, sites, wts, sv = ctx.cfg, ctx.rng, ctx.sites, ctx.wts, ctx.saved
        B, T, d = ctx.shape
        p, p_out = cfg.p_in, cfg.p_out
        x2, n, c = ctx.x2, ctx.n, ctx.c
        n_re = None
        if sv.get("folded"):
            # the forward never wrote LN(x): the queued weight-gradient product gets a buffer that this block's LayerNorm
            # backward (below) fills; without the queue the product runs right away and needs it now
            sink_b = wts.get("sink") or {}
            if sink_b.get("_wq") is not None and ops.layernorm_bwd_supports_dropout(x2):
                n = n_re = torch.empty_like(x2)
            else:
                n = _ln_fwd(x2, wts["ln_g"], wts["ln_b"])[0]
        dy2 = dy.reshape(B * T, d)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        g = {}
        sink = wts.get("sink") or {}
        ln_sink = (sink["ln_g"], sink["ln_b"]) if "ln_g" in sink else None
        ln_copies = sink.get("_copies") if ln_sink is not None else None

        wq = sink.get("_wq")

        def sk(name):
            return sink.get(name)

        if cfg.ln_mode != "post":
            du = dy2
        else:
            du, g["ln_g"], g["ln_b"] = ops.layernorm_bwd(dy2, sv["u"], wts["ln_g"], sv["mean"], sv["rstd"], grad_out=ln_sink,
                                                         copies=ln_copies)
        dz_o = du
        if p_out > 0:
            ready = _DROP_READY.pop(dy2.data_ptr(), None)
            if (ready is not None and cfg.ln_mode != "post" and ready[1] is not None and ready[1].data_ptr() == dy2.data_ptr() and
                    ready[1].shape == dy2.shape and ready[2] == (p_out, sites[1]) and ready[3] is rng):
                dz_o = ready[0]
            else:
                dz_o = ops.dropout_bwd(du, p_out, rng, sites[1])
        dmem = None
        if cfg.kind == "ffn":
            relu = cfg.act == "relu"
            # dh = dz_o . W2, gated by the saved post-dropout activations for ReLU (sign carries both masks)
            dh, g["w2"], g["b2"] = linear_bwd(dz_o, c, wts["w2"], gate=c if relu else None,
                                              gate_scale=1.0 / (1.0 - p) if relu else 1.0, dw_out=sk("w2"), db_out=sk("b2"), queue=wq, w_t=wts.get("w2_t"))
            if relu or cfg.act is None:
                dz1 = dh
                if cfg.act is None and p > 0:
                    dz1 = ops.dropout_bwd(dh, p, rng, sites[0])
            else:
                dz1 = ops.dropout_bwd(dh, p, rng, sites[0]) if p > 0 else dh
                dz1 = ops.act_bwd(dz1, sv["pre"], cfg.act)
            dn, g["w1"], g["b1"] = linear_bwd(dz1, n, wts["w1"], dw_out=sk("w1"), db_out=sk("b1"), queue=wq, w_t=wts.get("w1_t"))
        elif cfg.kind == "self":
            fused = sv["P"] is None  # fused attention kernels: delta can ride on this product's epilogue
            LINEAR_BWD_DOT[0] = None
            dc, g["w_out"], g["b_out"] = linear_bwd(dz_o, c, wts["w_out"], dw_out=sk("w_out"), db_out=sk("b_out"), queue=wq, w_t=wts.get("w_out_t"),
                                                    dot_src=c if fused else None)
            dpart, LINEAR_BWD_DOT[0] = LINEAR_BWD_DOT[0], None
            qkv = sv["qkv"]
            dqkv = torch.empty_like(qkv)  # (packed rows: the kernels zero the rows no utterance owns - the next product reads them)
            rel, d_rel = wts.get("rel_bias"), None
            if rel is not None:
                d_rel = sk("rel_bias")  # the parameter's slice of the flat gradient: the kernel adds into it
                if d_rel is None:
                    d_rel = g["rel_bias"] = torch.zeros_like(rel)
            attn_bwd(dc, qkv, 2 * d, qkv, 0, qkv, d, dqkv, 2 * d, dqkv, 0, dqkv, d, sv["shp"], sv["P"], sv["Pd"], p, rng,
                     sites[0], ctx_out=c, mask=ctx.mask, rel_bias=rel, d_rel_bias=d_rel, delta_partial=dpart)
            dn, g["w_in"], g["b_in"] = linear_bwd(dqkv, n, wts["w_in"], dw_out=sk("w_in"), db_out=sk("b_in"), queue=wq, w_t=wts.get("w_in_t"))
        else:  # cross
            fused = sv["P"] is None
            LINEAR_BWD_DOT[0] = None
            dc, g["w_out"], g["b_out"] = linear_bwd(dz_o, c, wts["w_out"], dw_out=sk("w_out"), db_out=sk("b_out"), queue=wq, w_t=wts.get("w_out_t"),
                                                    dot_src=c if fused else None)
            dpart, LINEAR_BWD_DOT[0] = LINEAR_BWD_DOT[0], None
            q, kv, off = sv["q"], sv["kv"], sv["kv_off"]
            dq = torch.empty_like(q)
            if sv["m2"] is None:  # grouped projections: dK | dV go into this layer's columns of the gradient all layers share
                kkey = sv.get("kv_key")
                dkv = None
                if kkey is not None:
                    dkv = _KV_GRAD.get(kkey)
                    if dkv is None:
                        dkv = _KV_GRAD[kkey] = torch.empty_like(kv)
                else:  # nothing behind the projections wants a gradient: a scratch the kernel can write to
                    dkv = torch.empty_like(kv)
                attn_bwd(dc, q, 0, kv, off, kv, off + d, dq, 0, dkv, off, dkv, off + d, sv["shp"], sv["P"], sv["Pd"], p, rng, sites[0],
                         ctx_out=c, mask=ctx.mask, delta_partial=dpart)
                g["w_kv"] = g["b_kv"] = None  # MemoryKVFn's
                if kkey is not None:
                    _KV_USERS[kkey] -= 1
                    if _KV_USERS[kkey] == 0:  # every layer has written its columns: autograd gets the buffer, once
                        dmem = _KV_GRAD.pop(kkey)
            else:
                dkv = torch.empty_like(kv)
                attn_bwd(dc, q, 0, kv, 0, kv, d, dq, 0, dkv, 0, dkv, d, sv["shp"], sv["P"], sv["Pd"], p, rng, sites[0],
                         ctx_out=c, mask=ctx.mask, delta_partial=dpart)
                key = sv.get("mem_key") if ctx.needs_input_grad[3] else None
                dmem2, g["w_kv"], g["b_kv"] = linear_bwd(dkv, sv["m2"], wts["w_kv"], need_dx=ctx.needs_input_grad[3],
                                                         dw_out=sk("w_kv"), db_out=sk("b_kv"), queue=wq, w_t=wts.get("w_kv_t"),
                                                         dx_add=None if key is None else _MEM_ACC.get(key))
                if key is not None:
                    _MEM_USERS[key] -= 1
                    if _MEM_USERS[key] > 0:  # more cross blocks to come: keep the sum, nothing for autograd yet
                        _MEM_ACC[key] = dmem2
                        dmem2 = None
                    else:
                        _MEM_ACC.pop(key, None)
                dmem = None if dmem2 is None else dmem2.view(ctx.mem_shape)
            dn, g["w_q"], g["b_q"] = linear_bwd(dq, n, wts["w_q"], dw_out=sk("w_q"), db_out=sk("b_q"), queue=wq, w_t=wts.get("w_q_t"))
        if cfg.ln_mode == "pre":
            if ctx.prev_drop is not None:
                pp, psite, prng = ctx.prev_drop
                dx2, g["ln_g"], g["ln_b"], dxd = ops.layernorm_bwd(dn, x2, wts["ln_g"], sv["mean"], sv["rstd"], add=du, add_scale=cfg.alpha,
                                                                   grad_out=ln_sink, drop=(pp, prng, psite), copies=ln_copies,
                                                                   n_out=n_re, beta=wts["ln_b"])
                _DROP_READY[dx2.data_ptr()] = (dxd, dx2, (pp, psite), prng)
            else:
                dx2, g["ln_g"], g["ln_b"] = ops.layernorm_bwd(dn, x2, wts["ln_g"], sv["mean"], sv["rstd"], add=du,
                                                              add_scale=cfg.alpha, grad_out=ln_sink, copies=ln_copies,
                                                              n_out=n_re, beta=wts["ln_b"])
        elif cfg.alpha != 0.0:
            dx2 = ops.axpby(dn, 1.0, du, cfg.alpha)
        else:
            dx2 = dn
        if sink:
            # every parameter gradient of this block was accumulated in place: nothing goes back through autograd
            notify = wts.get("notify")
            if notify is not None:
                skip = wts.get("notify_skip") or ()
                notify([p_ for p_ in ctx.params if id(p_) not in skip])
            grads = [None] * ctx.nparams
        else:
            grads = _route_param_grads(cfg.kind, g, d, cfg.ln_mode != "none")
            if "rel_bias" in wts:  # the relative-position table rides last in the parameter list (MultiHeadedAttention.run_block)
                grads.append(g.get("rel_bias"))
        assert len(grads) == ctx.nparams
        return (None, None, dx2.view(B, T, d), dmem, None, None, *grads)


class MemoryKVFn(torch.autograd.Function):
    """K | V projections of the encoder states for the cross-attention of ALL decoder layers as one product (the reference runs
    k_layer / v_layer per layer, transformer_layers.py:66-68 under :383): kv[B*S, L*2d] = memory @ [k_0; v_0; k_1; ...]^T + b.

    forward(ctx, memory [B,S,d], wts, *params):  wts = w_kv [L*2d, d], b_kv f32[L*2d], w_kv_t (optional), sink, notify;
    params = the 4L leaf parameters, weights first ([k_0.w, v_0.w, ...], then the biases in the same order)."""

    @staticmethod
    def forward(ctx, memory, wts, *params):
        B, S, d = memory.shape
        m2 = memory.reshape(B * S, d)
        kv = linear_fwd(m2, wts["w_kv"], wts["b_kv"])
        ctx.m2, ctx.wts, ctx.mem_shape, ctx.params = m2, wts, tuple(memory.shape), params
        return kv

    @staticmethod
    def backward(ctx, dkv):
        wts, params = ctx.wts, ctx.params
        sink = wts.get("sink") or {}
        nw = len(params) // 2
        need_w = any(p.requires_grad for p in params)
        dmem2, dw, db = linear_bwd(dkv, ctx.m2, wts["w_kv"], need_dx=ctx.needs_input_grad[0], need_dw=need_w, need_db=need_w,
                                   dw_out=sink.get("w_kv"), db_out=sink.get("b_kv"), queue=sink.get("_wq"), w_t=wts.get("w_kv_t"))
        if sink:
            notify = wts.get("notify")
            if notify is not None:
                notify(list(params))
            grads = [None] * len(params)
        else:
            grads = _split_rows(dw, [p.shape[0] for p in params[:nw]]) + _split_rows(db, [p.shape[0] for p in params[nw:]])
        return (None if dmem2 is None else dmem2.view(ctx.mem_shape), None, *grads)


def param_order(kind: str) -> List[str]:
    """Order in which a block's leaf parameters are passed to / returned from ResidualBlockFn."""
    if kind == "self":
        return ["k.w", "v.w", "q.w", "k.b", "v.b", "q.b", "o.w", "o.b", "ln.g", "ln.b"]
    if kind == "cross":
        return ["k.w", "v.w", "q.w", "k.b", "v.b", "q.b", "o.w", "o.b", "ln.g", "ln.b"]
    return ["w1", "b1", "w2", "b2", "ln.g", "ln.b"]


def _route_param_grads(kind, g, d, has_ln):
    ln = [g["ln_g"], g["ln_b"]] if has_ln else []
    if kind == "self":
        wk, wv, wq = _split_rows(g["w_in"], [d, d, d])
        bk, bv, bq = _split_rows(g["b_in"], [d, d, d])
        return [wk, wv, wq, bk, bv, bq, g["w_out"], g["b_out"], *ln]
    if kind == "cross":
        wk, wv = _split_rows(g["w_kv"], [d, d])
        bk, bv = _split_rows(g["b_kv"], [d, d])
        return [wk, wv, g["w_q"], bk, bv, g["b_q"], g["w_out"], g["b_out"], *ln]
    return [g["w1"], g["b1"], g["w2"], g["b2"], *ln]


# ------------------------------------------------------------------------------------------------
# leaf functions
# ------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(d, eps=1e-6) — final encoder/decoder norm (encoders.py:281-282, decoders.py:617-618)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, sink=None, notify=None):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        y, mean, rstd = ops.layernorm_fwd(x2, gamma, beta, LN_EPS)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shape, ctx.sink, ctx.notify, ctx.leaves = shape, sink, notify, (gamma, beta)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(x2.shape)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        sink = ctx.sink  # (dgamma view, dbeta view[, ops.GradCopies])
        dx, dg, db = ops.layernorm_bwd(dy2, x2, gamma, mean, rstd, grad_out=None if sink is None else sink[:2],
                                       copies=sink[2] if sink is not None and len(sink) > 2 else None)
        if ctx.sink is not None:
            if ctx.notify is not None:
                ctx.notify(ctx.leaves)
            dg = db = None
        return dx.view(ctx.shape), dg, db, None, None


class CutFn(torch.autograd.Function):
    """Identity whose output is where TrainStep cuts the backward pass in two (training.py: autograd.backward(..., inputs=[y]) for
    the half behind it, y.backward(dy) for the half in front).  The engine RUNS the node a captured non-leaf tensor came out of
    in both halves - harmless for this one; a node with side effects there (a LayerNorm adding its parameter gradients to the flat
    store) would add them twice."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        return dy


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) — vocabulary / CTC projections (decoders.py:620-623), Conformer input linear."""

    @staticmethod
    def forward(ctx, x, w_compute, weight, bias, out_dtype, sink=None, notify=None):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        y = linear_fwd(x2, w_compute, bias, out_dtype=out_dtype)
        ctx.x2, ctx.w = x2, w_compute
        ctx.has_bias = bias is not None
        ctx.shape = shape
        ctx.sink, ctx.notify, ctx.leaves = sink, notify, (weight, bias)
        if y.dtype != x2.dtype and any(ctx.needs_input_grad):
            # f32 logits out of a bf16 product: tell the loss that consumes them which type this node's backward multiplies in
            if len(LOGIT_GRAD_DTYPE) > 16:
                LOGIT_GRAD_DTYPE.clear()
            LOGIT_GRAD_DTYPE[y.data_ptr()] = x2.dtype
            ctx.y_ptr = y.data_ptr()
        return y.view(*shape[:-1], w_compute.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.x2, ctx.w
        ready = LOGIT_GRAD_READY.pop(ctx.y_ptr, None) if getattr(ctx, "y_ptr", None) is not None else None
        if ready is not None:
            d_lp, ph = ready
            is_placeholder = dy.data_ptr() == ph.data_ptr() and dy.shape == ph.shape and all(st == 0 for st in dy.stride())
            if is_placeholder and d_lp.dtype == x2.dtype:
                dy = d_lp.view(dy.shape)  # the loss wrote its gradient in this node's compute type (loss._XentFn.backward)
            else:
                # the loss handed its gradient over on the side and gave autograd a NaN placeholder; what arrived here is not that
                # placeholder (a second consumer of the logits, a copy or a cast in between): its NaNs must not reach dX and dW
                raise ops.Js2tError("LinearFn.backward: the loss handed its logit gradient over in the projection's compute type, but autograd "
                                    "delivered a different tensor - the logits have another consumer or a copy sits between projection and loss")
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if dy2.dtype != x2.dtype:
            dy2 = ops.cast(dy2, x2.dtype)
        sink = ctx.sink or {}
        dx, dw, db = linear_bwd(dy2, x2, w, need_dx=ctx.needs_input_grad[0], need_dw=ctx.needs_input_grad[2],
                                need_db=ctx.has_bias and ctx.needs_input_grad[3], dw_out=sink.get("w"), db_out=sink.get("b"),
                                queue=sink.get("_wq"), w_t=sink.get("_w_t"))
        if sink and ctx.notify is not None:
            ctx.notify(ctx.leaves)
        return (None if dx is None else dx.view(ctx.shape)), None, dw, db, None, None, None


class AddPeDropoutFn(torch.autograd.Function):
    """dropout(x + pe[:T] (+ extra)) — transformer_layers.py:204-213 + encoders.py:273-276 / decoders.py:599-602."""

    @staticmethod
    def forward(ctx, x, pe, extra, p, rng):
        site = rng.next_site() if (rng is not None and p > 0) else 0
        y = ops.add_pe_dropout(x.contiguous(), pe, None if extra is None else extra.contiguous(), p, rng, site)
        ctx.p, ctx.rng, ctx.site = p, rng, site
        ctx.has_extra = extra is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = ops.dropout_bwd(dy, ctx.p, ctx.rng, ctx.site) if ctx.p > 0 else dy
        return dx, None, (dx if ctx.has_extra else None), None, None


class EmbedFn(torch.autograd.Function):
    """lut(ids) * sqrt(d) — Embeddings.forward (embeddings.py:55-64)."""

    @staticmethod
    def forward(ctx, ids, table, scale, pad_idx, out_dtype, sink=None, notify=None):
        ctx.ids, ctx.scale, ctx.pad_idx, ctx.vocab = ids, scale, pad_idx, table.shape[0]
        ctx.sink, ctx.notify, ctx.leaf = sink, notify, table
        return ops.embed_fwd(ids, table, scale, out_dtype)

    @staticmethod
    def backward(ctx, dout):
        dt = ops.embed_bwd(ctx.ids, dout, ctx.vocab, ctx.scale, ctx.pad_idx, out=ctx.sink)
        if ctx.sink is not None:
            if ctx.notify is not None:
                ctx.notify((ctx.leaf, ))
            dt = None
        return None, dt, None, None, None, None, None


class AxpbyFn(torch.autograd.Function):
    """a*x + b*y (the Conformer layer's half-step residuals, reference transformer_layers.py:535-537,553,560-561)."""

    @staticmethod
    def forward(ctx, x, a, y, b):
        ctx.ab = (a, b)
        return ops.axpby(x.contiguous(), a, y.contiguous(), b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        g = g.contiguous()

        def part(c, wanted):  # a factor of one hands the gradient through as it is (autograd does not write into it): no pass, no copy
            return None if not wanted else (g if c == 1.0 else ops.axpby(g, c))

        return part(a, ctx.needs_input_grad[0]), None, part(b, ctx.needs_input_grad[2]), None


class ConvModuleFn(torch.autograd.Function):
    """Conformer ConvolutionModule after its LayerNorm (reference transformer_layers.py:458-475) on [B, T, C] activations:
    pointwise Linear C -> 2C, GLU, depthwise convolution and BatchNorm over the batch index (the reference convolves the
    transposed tensor, see include/joeys2t_hip.h), Hardswish, pointwise Linear C -> C, dropout.  One autograd node;
    parameter gradients come back through autograd (this module is off the LS100 hot path)."""

    @staticmethod
    def forward(ctx, x, w1, b1, wd, bd, gamma, beta, running_mean, running_var, w2, b2, p, rng, training, compute_dtype,
                w1_lp=None, w2_lp=None, sink=None, notify=None):
        """w1_lp / w2_lp: the two pointwise weights in the compute dtype when the flat store keeps such a shadow (stable
        addresses: the e4m3 copies of the fp8 forward mode are cached by address); otherwise they are cast here."""
        B, T, Cc = x.shape
        x2 = x.reshape(B * T, Cc)
        if x2.dtype != compute_dtype:
            x2 = ops.cast(x2, compute_dtype)
        w1c = w1.reshape(w1.shape[0], Cc)
        w2c = w2.reshape(w2.shape[0], -1)
        if compute_dtype != torch.float32:
            w1c = w1_lp.reshape(w1.shape[0], Cc) if w1_lp is not None else ops.cast(w1c, compute_dtype)
            w2c = w2_lp.reshape(w2.shape[0], -1) if w2_lp is not None else ops.cast(w2c, compute_dtype)
        h = linear_fwd(x2, w1c, b1)                                   # [B*T, 2C]
        u = ops.glu_fwd(h)                                            # [B*T, C]
        wd2 = wd.reshape(wd.shape[0], wd.shape[-1]).contiguous()
        v = ops.dwconv_outer_fwd(u.view(B, T, -1), wd2, bd)           # conv along B
        v2 = v.view(B * T, -1)
        z, mean, invstd = ops.bn_act_fwd(v2, gamma, beta, running_mean, running_var, 1e-5, 0.1, training, "hardswish")
        site = rng.next_site() if (training and p > 0) else 0
        y = linear_fwd(z, w2c, b2, dropout_p=p if training else 0.0, rng=rng, site=site)
        ctx.saved = (x2, w1c, h, u, wd2, v2, gamma, beta, mean, invstd, z, w2c)
        ctx.cfg = (B, T, Cc, p if training else 0.0, rng, site, training, x.dtype)
        ctx.sink, ctx.notify, ctx.leaves = sink, notify, (w1, b1, w2, b2)
        return y.view(B, T, -1).to(x.dtype) if y.dtype != x.dtype else y.view(B, T, -1)

    @staticmethod
    def backward(ctx, dy):
        x2, w1c, h, u, wd2, v2, gamma, beta, mean, invstd, z, w2c = ctx.saved
        B, T, Cc, p, rng, site, training, in_dtype = ctx.cfg
        dy2 = dy.reshape(B * T, -1)
        if dy2.dtype != z.dtype:
            dy2 = ops.cast(dy2.contiguous(), z.dtype)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if p > 0:
            dy2 = ops.dropout_bwd(dy2, p, rng, site)
        sink = ctx.sink or {}

        def sv(name, like):  # the parameter's slice of the flat gradient, shaped like the 2-D weight the product sees
            t = sink.get(name)
            return None if t is None else t.view(like.shape)

        dz, dw2, db2 = linear_bwd(dy2, z, w2c, dw_out=sv("w2", w2c), db_out=sink.get("b2"), queue=sink.get("_wq"))
        dv, dgamma, dbeta = ops.bn_act_bwd(dz, v2, gamma, beta, mean, invstd, training, "hardswish")
        du, dwd = ops.dwconv_outer_bwd(dv.view(B, T, -1), u.view(B, T, -1), wd2)
        dbd = ops.colsum(dv)
        dh = ops.glu_bwd(h, du.view(B * T, -1))
        dx, dw1, db1 = linear_bwd(dh, x2, w1c, dw_out=sv("w1", w1c), db_out=sink.get("b1"), queue=sink.get("_wq"))
        if sink and ctx.notify is not None:
            ctx.notify(ctx.leaves)
        dx = dx.view(B, T, Cc)
        if dx.dtype != in_dtype:
            dx = dx.to(in_dtype)
        return (dx, None if dw1 is None else dw1.view(dw1.shape[0], Cc, 1), db1, dwd.view(dwd.shape[0], 1, -1), dbd, dgamma, dbeta, None, None,
                None if dw2 is None else dw2.view(dw2.shape[0], -1, 1), db2, None, None, None, None, None, None, None, None)


def conv_out_len(t_in: int, k: int, stride: int = 2) -> int:
    """Conv1d(k, stride=2, padding=k//2) output length (encoders.py:339-345)."""
    return (t_in + 2 * (k // 2) - (k - 1) - 1) // stride + 1


class Conv1dGluFn(torch.autograd.Function):
    """GLU(Conv1d(k, stride 2, pad k//2)(x)) on [B,T,C] activations — one layer of Conv1dSubsampler
    (encoders.py:362-368) as an implicit-im2col MFMA GEMM + GLU epilogue kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, compute_dtype, sink=None, notify=None, valid_t=None):
        """valid_t (device int64 scalar, optional): output positions >= *valid_t are zeroed, as if the tensor ended there (a
        batch padded to a bucket length in front of the NEXT convolution, see ops.glu_fwd)."""
        ctx.sink, ctx.notify, ctx.leaves = sink, notify, (weight, bias)
        ctx.valid_t = valid_t
        B, T, Cin = x.shape
        Cout, _, K = weight.shape
        stride, pad = 2, K // 2
        Tout = conv_out_len(T, K, stride)
        x = x.contiguous()
        wp = ops.conv_weight_pack(weight, compute_dtype)  # [Cout, K*Cin]
        M = B * Tout
        pre = torch.empty((M, Cout), dtype=compute_dtype, device=x.device)
        conv = (T, Tout, Cin, stride, pad)
        vec = 8 if x.dtype == torch.bfloat16 else 4
        if Cin % vec == 0:
            # the taps written out once ([M, K*Cin], kept for the weight gradient): both products then take the LDS-DMA
            # kernels instead of the register-staged kernel that gathers the taps itself (bf16 LS100 layers: 170 -> 80 us
            # forward, 297 -> 80 us weight gradient)
            col = ops.im2col(x, K, stride, pad, Tout)
            ops.gemm(col, wp, pre, M=M, N=Cout, K=K * Cin, lda=K * Cin, ldb=K * Cin, ldc=Cout, bias=bias)
        else:
            col = None
            ops.gemm(x, wp, pre, M=M, N=Cout, K=K * Cin, lda=stride * Cin, ldb=K * Cin, ldc=Cout, bias=bias, conv=conv)
        y = ops.glu_fwd(pre, Tout, valid_t)
        ctx.x, ctx.wp, ctx.pre, ctx.conv, ctx.col = (x if col is None else None), wp, pre, conv, col
        ctx.dims = (B, T, Cin, Cout, K, Tout)
        return y.view(B, Tout, Cout // 2)

    @staticmethod
    def backward(ctx, dy):
        B, T, Cin, Cout, K, Tout = ctx.dims
        M = B * Tout
        x, wp, pre, conv = ctx.x, ctx.wp, ctx.pre, ctx.conv
        dy2 = dy.reshape(M, Cout // 2)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dpre = ops.glu_bwd(pre, dy2, Tout, ctx.valid_t)
        sink = ctx.sink or {}
        db = ops.colsum(dpre, out=sink.get("b"), accumulate=True) if sink else ops.colsum(dpre)
        # dWp^T[K*Cin, Cout] = im2col(x)^T[K*Cin, M] . dpre[M, Cout]
        sk = wgrad_split(K * Cin, Cout, M)
        dwp_t = (torch.zeros if sk > 1 else torch.empty)((K * Cin, Cout), dtype=torch.float32, device=dpre.device)
        if ctx.col is not None:
            ops.gemm(ctx.col, dpre, dwp_t, M=K * Cin, N=Cout, K=M, lda=K * Cin, ldb=Cout, ldc=Cout, trans_a=True, trans_b=True, split_k=sk)
        else:
            ops.gemm(x, dpre, dwp_t, M=K * Cin, N=Cout, K=M, lda=conv[3] * Cin, ldb=Cout, ldc=Cout, trans_a=True, trans_b=True,
                     conv=conv, split_k=sk)
        dw = ops.conv_weight_unpack_grad(dwp_t, Cout, Cin, K, out=sink.get("w"))
        if sink:
            if ctx.notify is not None:
                ctx.notify(ctx.leaves)
            dw = db = None
        dx = None
        if ctx.needs_input_grad[0]:
            dcol = torch.empty((M, K * Cin), dtype=pre.dtype, device=pre.device)
            ops.gemm(dpre, wp, dcol, M=M, N=K * Cin, K=Cout, lda=Cout, ldb=K * Cin, ldc=K * Cin, trans_b=True)
            dx = ops.col2im(dcol, B, T, Tout, Cin, K, conv[3], conv[4])
        return dx, dw, db, None, None, None, None
