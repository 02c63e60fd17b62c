"""Transformer layers with the reference's module / parameter names, computed by HIP kernels.

Mirrors joeynmt/transformer_layers.py (reference): MultiHeadedAttention (:17-115), PositionwiseFeedForward
(:118-168), PositionalEncoding (:171-213), TransformerEncoderLayer (:216-289), TransformerDecoderLayer
(:292-407).  The nn.Linear / nn.LayerNorm sub-modules are kept only as *parameter containers* so that
`state_dict()` keys match the reference checkpoints (`...src_src_att.k_layer.weight`, `...pwff_layer.0.weight`,
...); their torch forward is never called.  All math runs in joeys2t_amd.functional (libjoeys2t_hip.so).
"""
import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from joeys2t_amd import functional as Fn
from joeys2t_amd.builders import build_activation
from joeys2t_amd.runtime import runtime_of


def _rng(rt, x: Tensor):
    if rt.device is None:
        from joeys2t_amd.ops import dropout_rng
        return dropout_rng(x.device)
    return rt.rng


def _layer_norm(rt, ln: nn.LayerNorm, x: Tensor) -> Tensor:
    """A stand-alone LayerNorm whose parameter gradients go straight into the flat store (as encoders.TransformerEncoder's final one)"""
    sk = rt.sinks({"g": [ln.weight], "b": [ln.bias]})
    return Fn.LayerNormFn.apply(x, ln.weight, ln.bias, None if sk is None else (sk["g"], sk["b"], sk.get("_copies")), rt.grads_ready)


class MultiHeadedAttention(nn.Module):
    """Multi-head attention; argument order of forward is (k, v, q, mask) as in the reference (:49-56)."""

    def __init__(self, num_heads: int, size: int, dropout: float = 0.1, rel_pos_clip: Optional[int] = None) -> None:
        """rel_pos_clip (extension, BASELINE config 5 "rel-pos attn"; not in the reference): a learned bias per head and
        clipped relative distance, rel_pos_bias[h, clamp(j - i, -R, R) + R], added to the scaled self-attention scores
        inside the fused attention kernels (bf16 compute only).  Zero-initialised: the module then equals the reference's."""
        super().__init__()
        assert size % num_heads == 0
        self.rel_pos_bias = None
        if rel_pos_clip:
            self.rel_pos_bias = nn.Parameter(torch.zeros(num_heads, 2 * int(rel_pos_clip) + 1))
        self.head_size = size // num_heads
        self.model_size = size
        self.num_heads = num_heads
        self.k_layer = nn.Linear(size, num_heads * self.head_size)
        self.v_layer = nn.Linear(size, num_heads * self.head_size)
        self.q_layer = nn.Linear(size, num_heads * self.head_size)
        self.output_layer = nn.Linear(size, size)
        self.softmax = nn.Softmax(dim=-1)  # kept for repr/state parity; unused
        self.dropout = nn.Dropout(dropout)

    def fuse_groups(self):
        return [[self.k_layer.weight, self.v_layer.weight, self.q_layer.weight],
                [self.k_layer.bias, self.v_layer.bias, self.q_layer.bias]]

    # parameters in functional.param_order("self"/"cross") order, without the layer-norm pair
    def _attn_params(self):
        return [self.k_layer.weight, self.v_layer.weight, self.q_layer.weight, self.k_layer.bias, self.v_layer.bias,
                self.q_layer.bias, self.output_layer.weight, self.output_layer.bias]

    def _weights(self, rt, kind: str) -> dict:
        k, v, q, o = self.k_layer, self.v_layer, self.q_layer, self.output_layer
        w = {"w_out": rt.weight([o.weight]), "b_out": rt.bias([o.bias]), "w_out_t": rt.weight_t([o.weight])}
        if kind == "self":
            w["w_in"] = rt.weight([k.weight, v.weight, q.weight])
            w["b_in"] = rt.bias([k.bias, v.bias, q.bias])
            w["w_in_t"] = rt.weight_t([k.weight, v.weight, q.weight])
        else:
            w["w_kv"] = rt.weight([k.weight, v.weight])
            w["b_kv"] = rt.bias([k.bias, v.bias])
            w["w_q"] = rt.weight([q.weight])
            w["b_q"] = rt.bias([q.bias])
            w["w_kv_t"] = rt.weight_t([k.weight, v.weight])
            w["w_q_t"] = rt.weight_t([q.weight])
        return w

    def run_block(self, x: Tensor, memory: Optional[Tensor], mask: Optional[Tensor], *, ln: Optional[nn.LayerNorm],
                  ln_mode: str, alpha: float, out_dropout: float, need_weights: bool = False,
                  memory_kv: Optional[Tuple[Tensor, int]] = None, pack=None, mem_pack=None):
        """[LN] -> attention -> output projection (+dropout) + alpha*x [-> LN] as one fused autograd node.
        `memory_kv` = (projections [B*S, L*2d] of the encoder states for all decoder layers, this layer's first column): the
        block reads its keys and values from there instead of projecting `memory` itself (functional.MemoryKVFn)."""
        rt = runtime_of(self)
        kind = "self" if memory is None else "cross"
        x = rt.act_in(x)
        if memory_kv is not None:
            memory = memory_kv[0]
        elif memory is not None:
            memory = rt.act_in(memory)
        cfg = Fn.BlockCfg(kind=kind, num_heads=self.num_heads, alpha=alpha, ln_mode=ln_mode, training=self.training,
                          attn_dropout=self.dropout.p, out_dropout=out_dropout, need_weights=need_weights)
        wts = self._weights(rt, kind)
        params = self._attn_params()
        k_, v_, q_, o_ = self.k_layer, self.v_layer, self.q_layer, self.output_layer
        smap = {"w_out": [o_.weight], "b_out": [o_.bias]}
        if kind == "self":
            smap.update(w_in=[k_.weight, v_.weight, q_.weight], b_in=[k_.bias, v_.bias, q_.bias])
        else:
            smap.update(w_kv=[k_.weight, v_.weight], b_kv=[k_.bias, v_.bias], w_q=[q_.weight], b_q=[q_.bias])
        if ln is not None:
            wts["ln_g"], wts["ln_b"] = ln.weight.data, ln.bias.data
            params = params + [ln.weight, ln.bias]
            smap.update(ln_g=[ln.weight], ln_b=[ln.bias])
            if ln_mode == "pre" and Fn.LN_FOLD:  # LayerNorm folded into the q/k/v (cross: q) projection, see functional._LN_STATS
                first_w, first_b = ([k_.weight, v_.weight, q_.weight], [k_.bias, v_.bias, q_.bias]) if kind == "self" else ([q_.weight], [q_.bias])
                wts["fold"] = rt.ln_fold(first_w, first_b, ln)
        wts["emit_stats"] = ln_mode == "pre" and rt.store is not None and rt.compute_dtype == torch.bfloat16
        if self.rel_pos_bias is not None:
            if kind != "self":
                raise NotImplementedError("relative-position bias is defined for self-attention")
            wts["rel_bias"] = self.rel_pos_bias.data
            params = params + [self.rel_pos_bias]
            smap.update(rel_bias=[self.rel_pos_bias])
        wts["sink"], wts["notify"] = rt.sinks(smap), rt.grads_ready
        if pack is not None:  # ops.PackedRows: x holds the live rows of a ragged batch (encoders.TransformerEncoder)
            if kind != "self":
                raise NotImplementedError("packed rows: self-attention blocks only")
            wts["pack"] = pack
        if mem_pack is not None:  # ops.PackedRows: memory_kv are the projections of the PACKED encoder states (decoders.py)
            if memory_kv is None:
                raise NotImplementedError("packed encoder states: grouped K | V projections only")
            wts["mem_pack"] = mem_pack
        if memory_kv is not None:  # k_layer / v_layer ran in MemoryKVFn, which also owns their gradients
            wts["kv_off"] = int(memory_kv[1])
            wts["notify_skip"] = {id(k_.weight), id(v_.weight), id(k_.bias), id(v_.bias)}
        rng = _rng(rt, x) if cfg.any_dropout else None
        if mask is not None and not mask.is_contiguous():
            mask = mask.contiguous()
        out = Fn.ResidualBlockFn.apply(cfg, rng, x, memory, mask, wts, *params)
        return out if kind == "cross" else (out, None)

    def forward(self, k: Tensor, v: Tensor, q: Tensor, mask: Optional[Tensor] = None,
                return_weights: Optional[bool] = None):
        """Stand-alone attention (no residual, no norm) — reference :49-115.  `k` and `v` must be the same
        tensor (every call site of the reference passes them so: :282, :383, :394-396)."""
        if k is not v:
            raise NotImplementedError("MultiHeadedAttention: k and v must be the same tensor on the HIP path")
        memory = None if q is k else k
        out, att = self.run_block(q, memory, mask, ln=None, ln_mode="none", alpha=0.0, out_dropout=0.0,
                                  need_weights=bool(return_weights))
        if return_weights and memory is None:
            raise NotImplementedError("attention weights are only exported for cross-attention (decoders.py:606-615)")
        return out, att


class PositionwiseFeedForward(nn.Module):
    """LN -> Linear -> act -> Dropout -> Linear -> Dropout -> + alpha*x (reference :118-168)."""

    def __init__(self, input_size: int, ff_size: int, dropout: float = 0.1, alpha: float = 1.0,
                 layer_norm: str = "post", activation: str = "relu") -> None:
        super().__init__()
        activation_fnc = build_activation(activation=activation)
        self.layer_norm = nn.LayerNorm(input_size, eps=1e-6)
        self.pwff_layer = nn.Sequential(
            nn.Linear(input_size, ff_size),
            activation_fnc(),
            nn.Dropout(dropout),
            nn.Linear(ff_size, input_size),
            nn.Dropout(dropout),
        )
        self.alpha = alpha
        self._layer_norm_position = layer_norm
        self._activation = activation
        assert self._layer_norm_position in {"pre", "post"}

    def forward(self, x: Tensor) -> Tensor:
        rt = runtime_of(self)
        x = rt.act_in(x)
        l1, l2 = self.pwff_layer[0], self.pwff_layer[3]
        p = self.pwff_layer[2].p
        cfg = Fn.BlockCfg(kind="ffn", alpha=self.alpha, ln_mode=self._layer_norm_position, act=self._activation,
                          training=self.training, attn_dropout=p, out_dropout=p)
        wts = {"w1": rt.weight([l1.weight]), "b1": rt.bias([l1.bias]), "w2": rt.weight([l2.weight]),
               "b2": rt.bias([l2.bias]), "ln_g": self.layer_norm.weight.data, "ln_b": self.layer_norm.bias.data,
               "w1_t": rt.weight_t([l1.weight]), "w2_t": rt.weight_t([l2.weight]),
               "fold": rt.ln_fold([l1.weight], [l1.bias], self.layer_norm)
               if (self._layer_norm_position == "pre" and self._activation == "relu" and Fn.LN_FOLD) else None,
               "emit_stats": self._layer_norm_position == "pre" and rt.store is not None and rt.compute_dtype == torch.bfloat16,
               "sink": rt.sinks({"w1": [l1.weight], "b1": [l1.bias], "w2": [l2.weight], "b2": [l2.bias],
                                 "ln_g": [self.layer_norm.weight], "ln_b": [self.layer_norm.bias]}),
               "notify": rt.grads_ready}
        rng = _rng(rt, x) if cfg.any_dropout else None
        return Fn.ResidualBlockFn.apply(cfg, rng, x, None, None, wts, l1.weight, l1.bias, l2.weight, l2.bias,
                                        self.layer_norm.weight, self.layer_norm.bias)


class PositionalEncoding(nn.Module):
    """Sinusoidal position table (reference :171-213): pe[p,2i] = sin(p * exp(-2i ln(1e4)/d)), pe[p,2i+1] = cos."""

    def __init__(self, size: int = 0, max_len: int = 5000) -> None:
        if size % 2 != 0:
            raise ValueError(f"Cannot use sin/cos positional encoding with odd dim (got dim={size})")
        super().__init__()
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, size, 2, dtype=torch.float) * -(math.log(10000.0) / size))
        pe = torch.zeros(max_len, size)
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))  # (1, max_len, size) as in the reference state_dict
        self.dim = size

    def forward(self, emb: Tensor, extra: Optional[Tensor] = None, dropout: float = 0.0, training: bool = False) -> Tensor:
        """emb + pe[:, :T] (+ extra), then dropout — the add and the encoder/decoder emb_dropout are one kernel."""
        rt = runtime_of(self)
        emb = rt.act_in(emb)
        if emb.size(1) > self.pe.size(1):
            raise ValueError(f"sequence length {emb.size(1)} exceeds positional table {self.pe.size(1)}")
        p = dropout if training else 0.0
        rng = _rng(rt, emb) if p > 0 else None
        if extra is not None:
            extra = rt.act_in(extra)
        return Fn.AddPeDropoutFn.apply(emb, self.pe[0], extra, p, rng)


class TransformerEncoderLayer(nn.Module):
    """Self-attention block + feed-forward block (reference :216-289)."""

    def __init__(self, size: int = 0, ff_size: int = 0, num_heads: int = 0, dropout: float = 0.1, alpha: float = 1.0,
                 layer_norm: str = "post", activation: str = "relu") -> None:
        super().__init__()
        self.layer_norm = nn.LayerNorm(size, eps=1e-6)
        self.src_src_att = MultiHeadedAttention(num_heads, size, dropout=dropout)
        self.feed_forward = PositionwiseFeedForward(size, ff_size=ff_size, dropout=dropout, alpha=alpha,
                                                    layer_norm=layer_norm, activation=activation)
        self.dropout = nn.Dropout(dropout)
        self.size = size
        self.alpha = alpha
        self._layer_norm_position = layer_norm
        assert self._layer_norm_position in {"pre", "post"}

    def forward(self, x: Tensor, mask: Tensor, pack=None) -> Tensor:
        h, _ = self.src_src_att.run_block(x, None, mask, ln=self.layer_norm, ln_mode=self._layer_norm_position,
                                          alpha=self.alpha, out_dropout=self.dropout.p, pack=pack)
        return self.feed_forward(h)


class TransformerDecoderLayer(nn.Module):
    """Masked self-attention, encoder-decoder attention, feed-forward (reference :292-407)."""

    def __init__(self, size: int = 0, ff_size: int = 0, num_heads: int = 0, dropout: float = 0.1, alpha: float = 1.0,
                 layer_norm: str = "post", activation: str = "relu") -> None:
        super().__init__()
        self.size = size
        self.trg_trg_att = MultiHeadedAttention(num_heads, size, dropout=dropout)
        self.src_trg_att = MultiHeadedAttention(num_heads, size, dropout=dropout)
        self.feed_forward = PositionwiseFeedForward(size, ff_size=ff_size, dropout=dropout, alpha=alpha,
                                                    layer_norm=layer_norm, activation=activation)
        self.x_layer_norm = nn.LayerNorm(size, eps=1e-6)
        self.dec_layer_norm = nn.LayerNorm(size, eps=1e-6)
        self.dropout = nn.Dropout(dropout)
        self.alpha = alpha
        self._layer_norm_position = layer_norm
        assert self._layer_norm_position in {"pre", "post"}

    def forward(self, x: Tensor, memory: Tensor, src_mask: Tensor, trg_mask: Tensor, return_attention: bool = False,
                memory_kv: Optional[Tuple[Tensor, int]] = None, **kwargs):
        h1, _ = self.trg_trg_att.run_block(x, None, trg_mask, ln=self.x_layer_norm, ln_mode=self._layer_norm_position,
                                           alpha=self.alpha, out_dropout=self.dropout.p)
        h2, att = self.src_trg_att.run_block(h1, memory, src_mask, ln=self.dec_layer_norm,
                                             ln_mode=self._layer_norm_position, alpha=self.alpha,
                                             out_dropout=self.dropout.p, need_weights=return_attention, memory_kv=memory_kv,
                                             mem_pack=kwargs.get("mem_pack"))
        out = self.feed_forward(h2)
        return out, (att if return_attention else None)


class ConvolutionModule(nn.Module):
    """Conformer convolution block (reference :410-475): LayerNorm -> pointwise conv C->2C -> GLU -> depthwise conv ->
    BatchNorm1d -> Hardswish -> pointwise conv -> dropout.  Parameter names / shapes as in the reference (Conv1d weights
    [out, in, k]).  forward() takes the layer's [B, T, C] activations: the reference layer transposes dims 0 and 1 before
    calling its module (:549-552), so the depthwise convolution and the BatchNorm length axis run over the BATCH index -
    reproduced, not "fixed"."""

    def __init__(self, hidden_size: int, channels: int, depthwise_kernel_size: int, dropout: float):
        super().__init__()
        assert (depthwise_kernel_size - 1) % 2 == 0, "kernel_size should be a odd number for 'SAME' padding"
        self.layer_norm = nn.LayerNorm(hidden_size, eps=1e-6)
        self.pointwise_conv1 = nn.Conv1d(hidden_size, 2 * channels, kernel_size=1, stride=1, padding=0)
        self.glu = nn.GLU(dim=1)
        self.depthwise_conv = nn.Conv1d(channels, channels, depthwise_kernel_size, stride=1,
                                        padding=(depthwise_kernel_size - 1) // 2, groups=channels)
        self.batch_norm = nn.BatchNorm1d(channels)
        self.swish = nn.Hardswish()
        self.pointwise_conv2 = nn.Conv1d(channels, hidden_size, kernel_size=1, stride=1, padding=0)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: Tensor) -> Tensor:
        rt = runtime_of(self)
        x = rt.act_in(x)
        x = _layer_norm(rt, self.layer_norm, x)
        bn = self.batch_norm
        p = self.dropout.p
        rng = _rng(rt, x) if (self.training and p > 0) else None
        if self.training:
            bn.num_batches_tracked += 1
        lp = rt.store is not None and rt.compute_dtype != torch.float32  # the flat store's bf16 shadow: stable addresses
        return Fn.ConvModuleFn.apply(x, self.pointwise_conv1.weight, self.pointwise_conv1.bias, self.depthwise_conv.weight,
                                     self.depthwise_conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     self.pointwise_conv2.weight, self.pointwise_conv2.bias, p, rng, self.training, rt.compute_dtype,
                                     rt.weight([self.pointwise_conv1.weight]) if lp else None,
                                     rt.weight([self.pointwise_conv2.weight]) if lp else None,
                                     # the two pointwise products' parameter gradients: straight into the flat store, the weight
                                     # gradients deferred and grouped with the other layers' (runtime.WgradQueue)
                                     rt.sinks({"w1": [self.pointwise_conv1.weight], "b1": [self.pointwise_conv1.bias],
                                               "w2": [self.pointwise_conv2.weight], "b2": [self.pointwise_conv2.bias]}), rt.grads_ready)


class ConformerEncoderLayer(nn.Module):
    """Conformer block as the reference builds it (:478-565): half-step residuals around the two feed-forward modules
    (which carry their own LayerNorm and residual), self-attention, the convolution module, final LayerNorm."""

    def __init__(self, size: int = 512, ff_size: int = 2048, num_heads: int = 4, dropout: float = 0.1,
                 depthwise_conv_kernel_size: int = 31, alpha: float = 1.0, layer_norm: str = "pre",
                 rel_pos_clip: Optional[int] = None):
        super().__init__()
        self.initial_feed_forward = PositionwiseFeedForward(size, ff_size=ff_size, dropout=dropout, alpha=alpha, layer_norm=layer_norm)
        self.src_att_layer_norm = nn.LayerNorm(size, eps=1e-6)
        self.src_att_dropout = nn.Dropout(dropout)
        self.src_src_att = MultiHeadedAttention(num_heads, size, dropout=dropout, rel_pos_clip=rel_pos_clip)
        self.conv_module = ConvolutionModule(hidden_size=size, channels=size, depthwise_kernel_size=depthwise_conv_kernel_size,
                                             dropout=dropout)
        self.final_feed_forward = PositionwiseFeedForward(size, ff_size=ff_size, dropout=dropout, alpha=alpha, layer_norm=layer_norm)
        self.final_layer_norm = nn.LayerNorm(size, eps=1e-6)
        self.alpha = alpha
        self.size = size
        self._layer_norm_position = layer_norm
        assert self._layer_norm_position in {"pre", "post"}

    def forward(self, x: Tensor, mask: Tensor) -> Tensor:
        rt = runtime_of(self)
        x = rt.act_in(x)
        x = Fn.AxpbyFn.apply(self.initial_feed_forward(x), 0.5, x, 1.0)
        x, _ = self.src_src_att.run_block(x, None, mask, ln=self.src_att_layer_norm, ln_mode=self._layer_norm_position,
                                          alpha=self.alpha, out_dropout=self.src_att_dropout.p)
        x = Fn.AxpbyFn.apply(self.conv_module(x), 1.0, x, self.alpha)
        residual = x
        if self._layer_norm_position == "pre":
            x = _layer_norm(rt, self.final_layer_norm, x)
        x = Fn.AxpbyFn.apply(self.final_feed_forward(x), 0.5, residual, 1.0)
        if self._layer_norm_position == "post":
            x = _layer_norm(rt, self.final_layer_norm, x)
        return x
