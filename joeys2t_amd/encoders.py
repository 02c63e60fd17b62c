"""Speech encoder with the reference's module / parameter names (joeynmt/encoders.py): Conv1dSubsampler
(:311-373) and TransformerEncoder (:175-308), computed by HIP kernels.

Layout note: the reference transposes to [B,C,T] for nn.Conv1d and back (:363,:368); here activations stay
[B,T,C] end to end and each Conv1d(k, stride 2)+GLU is an implicit-im2col GEMM (functional.Conv1dGluFn)."""
import os
from typing import List, Tuple

import torch
from torch import Tensor, nn

from joeys2t_amd import functional as Fn
from joeys2t_amd import ops
from joeys2t_amd.helpers import freeze_params, lengths_to_padding_mask, pad
from joeys2t_amd.runtime import runtime_of
from joeys2t_amd.transformer_layers import ConformerEncoderLayer, PositionalEncoding, TransformerEncoderLayer


# the encoder stack of a ragged S2T batch on its live rows only (TransformerEncoder._packing); JS2T_PACKED_ENCODER=0: padded layout
PACK_RAGGED = os.environ.get("JS2T_PACKED_ENCODER", "1") != "0"
PACK_MIN_SAVING = 0.04  # below this share of dead rows the two gather / scatter passes are not worth it
PACK_ROUND = 64         # packed row count rounded up (whole 64-row groups for the row-wise kernels; the tail is kept zero)
_PACK_CACHE: dict = {}  # (lengths, T', device) -> ops.PackedRows of the last few ragged batches (TransformerEncoder._packing)


class Encoder(nn.Module):
    """Base encoder class."""

    @property
    def output_size(self):
        return self._output_size


class Conv1dSubsampler(nn.Module):
    """Stack of Conv1d(k, stride=2, padding=k//2) + GLU (reference :311-373).  `conv_layers.{i}.weight` keeps
    torch's [C_out, C_in, k] layout for checkpoint compatibility."""

    def __init__(self, in_channels: int, mid_channels: int, out_channels: int = None, kernel_sizes: List[int] = (3, 3)):
        super().__init__()
        self.kernel_sizes = list(kernel_sizes)
        self.n_layers = len(self.kernel_sizes)
        self.conv_layers = nn.ModuleList(
            nn.Conv1d(in_channels if i == 0 else mid_channels // 2,
                      mid_channels if i < self.n_layers - 1 else out_channels * 2, k, stride=2, padding=k // 2)
            for i, k in enumerate(self.kernel_sizes))

    def get_out_seq_lens_tensor(self, in_seq_lens_tensor: Tensor) -> Tensor:
        """floor((len + 2*(k//2) - (k-1) - 1)/2 + 1) per layer (reference :348-352), for host tensors / tests.
        The device path computes the same integers in js2t_subsample_lengths_mask."""
        out = in_seq_lens_tensor.long()
        for k in self.kernel_sizes:  # stride 2, padding k // 2: in whole numbers
            out = torch.div(out + 2 * (k // 2) - k, 2, rounding_mode="floor") + 1
        return out

    def out_len(self, t_in: int) -> int:
        for k in self.kernel_sizes:
            t_in = Fn.conv_out_len(t_in, k)
        return t_in

    def forward(self, src_tokens: Tensor, src_lengths: Tensor, crop: Tensor = None) -> Tuple[Tensor, Tensor, Tensor]:
        """-> (x [B,T',C], out_seq_lens [B], padding mask [B,1,T']).  The input is expected to be cropped to the
        longest utterance already (pad_features guarantees it, helpers_for_audio.py:159-161); the reference's
        re-crop (:356-359) costs a host sync and is only taken for host-side length tensors.
        crop (device int64 [n_layers], optional): the batch is padded to a BUCKET length (graphed.GraphedTrainStep) and
        crop[i] is the number of output positions layer i has for the longest real utterance - each layer but the last
        zeroes what lies beyond, so the next convolution sees the zero padding nn.Conv1d gives the reference's cropped tensor."""
        rt = runtime_of(self)
        if not src_lengths.is_cuda:
            max_len = int(src_lengths.max())
            assert max_len > 0, "empty batch!"
            if src_tokens.size(1) != max_len:
                src_tokens = src_tokens[:, :max_len, :]
            src_lengths = src_lengths.to(src_tokens.device)
        x = rt.act_in(src_tokens.contiguous())
        for i, conv in enumerate(self.conv_layers):
            valid = crop[i:i + 1] if (crop is not None and i + 1 < self.n_layers) else None
            x = Fn.Conv1dGluFn.apply(x, conv.weight, conv.bias, rt.compute_dtype,
                                     rt.sinks({"w": [conv.weight], "b": [conv.bias]}), rt.grads_ready, valid)
        out_lens, mask = ops.subsample_lengths_mask(src_lengths, x.size(1), self.kernel_sizes)
        # the mask's row sums, which the CTC loss asks for (model.py:125): = min(out_lens, T'), and T' is the sub-sampled
        # length of the longest utterance, so out_lens itself; saves a cast + reduction per step
        mask.js2t_row_sums = out_lens
        return x, out_lens, mask


class TransformerEncoder(Encoder):
    """Transformer encoder (reference :175-308)."""

    def __init__(self, hidden_size: int = 512, ff_size: int = 2048, num_layers: int = 8, num_heads: int = 4,
                 dropout: float = 0.1, emb_dropout: float = 0.1, freeze: bool = False, **kwargs):
        super().__init__()
        self._output_size = hidden_size
        self.layers = nn.ModuleList([
            TransformerEncoderLayer(size=hidden_size, ff_size=ff_size, num_heads=num_heads, dropout=dropout,
                                    alpha=kwargs.get("alpha", 1.0), layer_norm=kwargs.get("layer_norm", "pre"),
                                    activation=kwargs.get("activation", "relu")) for _ in range(num_layers)
        ])
        self.pe = PositionalEncoding(hidden_size)
        self.emb_dropout = nn.Dropout(p=emb_dropout)
        # final norm only for pre-LN stacks; note the differing defaults ("pre" for layers, "post" here), as in
        # the reference (:215 vs :225)
        self.layer_norm = nn.LayerNorm(hidden_size, eps=1e-6) if kwargs.get("layer_norm", "post") == "pre" else None
        if freeze:
            freeze_params(self)
        self.subsample = kwargs.get("subsample", False)
        if self.subsample:
            self.subsampler = Conv1dSubsampler(kwargs["in_channels"], kwargs["conv_channels"], hidden_size,
                                               kwargs.get("conv_kernel_sizes", [3, 3]))
            self.pad_index = kwargs.get("pad_index", 1)
            assert self.pad_index is not None

    def forward(self, src_embed: Tensor, src_length: Tensor, mask: Tensor = None, **kwargs):
        """-> (hidden states [B,T',d], None, mask [B,1,T'])."""
        if self.subsample:
            src_embed, src_length, ss_mask = self.subsampler(src_embed, src_length, kwargs.get("src_crop", None))
            if mask is None:
                mask = ss_mask
        if mask is None:
            mask = lengths_to_padding_mask(src_length, src_embed.size(1)).unsqueeze(1)
        x = self.pe(src_embed, extra=kwargs.get("src_prompt_mask", None), dropout=self.emb_dropout.p,
                    training=self.training)
        pack = self._packing(x, mask, kwargs)
        self.last_pack = pack  # for the decoder of the same forward (model.py hands it on as `memory_pack`)
        if pack is not None:  # ragged batch: the stack runs on the live positions only, [1, sum of lengths (rounded), d]
            x = Fn.PackRowsFn.apply(x, pack)
        for layer in self.layers:
            x = layer(x, mask) if pack is None else layer(x, mask, pack=pack)
        if self.layer_norm is not None:
            rt = runtime_of(self)
            sk = rt.sinks({"g": [self.layer_norm.weight], "b": [self.layer_norm.bias]})
            x = Fn.LayerNormFn.apply(x, self.layer_norm.weight, self.layer_norm.bias, None if sk is None else (sk["g"], sk["b"], sk.get("_copies")),
                                     rt.grads_ready)
        if pack is not None:  # back to [B, T', d] for the decoder / CTC layer: zeros behind every length
            x = Fn.UnpackRowsFn.apply(x, pack)
        if kwargs.get("repad", False) and "src_max_len" in kwargs and self.subsample:
            x, mask = self._repad(x, mask, kwargs["src_max_len"])
        assert src_length.size() == (x.size(0), ), (src_length.size(), x.size())
        assert mask.size() == (x.size(0), 1, x.size(1)), (mask.size(), x.size())
        return x, None, mask

    def _packing(self, x: Tensor, mask: Tensor, kwargs):
        """ops.PackedRows for this batch, or None (the padded layout).  Positions behind an utterance's sub-sampled length are dead
        in the reference - masked as keys (transformer_layers.py:86-105), beyond the CTC input lengths (loss.py:156-161), masked in
        the decoder's cross-attention - so an S2T encoder may drop them: every row-wise kernel then runs on sum(T'_i) rows and the
        fused self-attention walks each utterance's own tiles (js2t_attn_desc.seg).  Needs the fused attention kernels (bf16, head
        size 64 / 128) and lengths known on the HOST: `src_pack` (a PackedRows prepared by the caller, graphed.GraphedTrainStep)
        or `src_length_host` (frame counts, batch.Batch).  Taken when it drops at least PACK_MIN_SAVING of the rows."""
        if not (PACK_RAGGED and self.subsample and type(self) is TransformerEncoder and x.is_cuda and x.dtype == torch.bfloat16 and Fn.USE_FLASH):
            return None
        B, T, d = x.shape
        dh = d // self.layers[0].src_src_att.num_heads
        if dh not in (64, 128) or (d * 2) % 16 or mask is None or tuple(mask.shape) != (B, 1, T):
            return None
        pack = kwargs.get("src_pack", None)
        if pack is None:
            host = kwargs.get("src_length_host", None)
            if host is None or len(host) != B:
                return None
            lens = [min(self.subsampler.out_len(int(n)), T) for n in host]
            if sum(lens) > (1.0 - PACK_MIN_SAVING) * B * T:
                return None
            # the row offsets reach the device by a copy from a pinned temporary: never inside a hipGraph capture (the graph would
            # replay the copy from memory that is long gone).  A step that is captured has run eagerly on the same batch before
            # (bench.py --ragged, GraphedTrainStep hands its own `src_pack`), so the offsets of recent batches are kept
            key = (tuple(lens), T, str(x.device))
            pack = _PACK_CACHE.get(key)
            if pack is None:
                if torch.cuda.is_current_stream_capturing():
                    return None
                pack = ops.PackedRows.from_lengths(lens, T, x.device, round_to=PACK_ROUND)
                if len(_PACK_CACHE) >= 16:
                    _PACK_CACHE.pop(next(iter(_PACK_CACHE)))
                _PACK_CACHE[key] = pack
        if pack.B != B or pack.T != T:
            raise ops.Js2tError(f"packed encoder: {pack.B} x {pack.T} prepared for a [{B}, {T}, {d}] batch")
        return pack

    def _repad(self, x, mask, src_max_len):
        """Pad x / mask to the subsampled length of src_max_len (reference :290-298; mask padded with True)."""
        src_max_len = self.subsampler.out_len(int(src_max_len))
        return pad(x, src_max_len, pad_index=self.pad_index, dim=1), pad(mask, src_max_len, pad_index=self.pad_index, dim=-1)

    def __repr__(self):
        return (f"{self.__class__.__name__}(num_layers={len(self.layers)}, "
                f"num_heads={self.layers[0].src_src_att.num_heads}, alpha={self.layers[0].alpha}, "
                f'layer_norm="{self.layers[0]._layer_norm_position}", subsample={self.subsample})')


class ConformerEncoder(TransformerEncoder):
    """Conformer encoder (reference :376-445): subsampler -> positional encoding -> Linear -> emb_dropout -> Conformer
    layers; no final LayerNorm.  As in the reference it is not reachable from build_model (model.py:417-421 accepts
    `recurrent` / `transformer` only) and is constructed directly; like the reference, TransformerEncoder.__init__() runs
    first with its defaults and its layers are then replaced (:392)."""

    def __init__(self, hidden_size: int = 512, ff_size: int = 2048, num_layers: int = 8, num_heads: int = 4, dropout: float = 0.1,
                 emb_dropout: float = 0.1, freeze: bool = False, **kwargs):
        super().__init__()
        self._output_size = hidden_size
        self.layers = nn.ModuleList([
            ConformerEncoderLayer(size=hidden_size, ff_size=ff_size, num_heads=num_heads, dropout=dropout,
                                  alpha=kwargs.get("alpha", 1.0), layer_norm=kwargs.get("layer_norm", "pre"),
                                  depthwise_conv_kernel_size=kwargs.get("depthwise_conv_kernel_size", 31),
                                  rel_pos_clip=kwargs.get("rel_pos_clip")) for _ in range(num_layers)
        ])
        self.pe = PositionalEncoding(hidden_size)
        self.emb_dropout = nn.Dropout(p=emb_dropout)
        self.linear = nn.Linear(hidden_size, hidden_size)
        if freeze:
            freeze_params(self)
        self.subsampler = Conv1dSubsampler(kwargs["in_channels"], kwargs["conv_channels"], hidden_size,
                                           kwargs.get("conv_kernel_sizes", [3, 3]))
        self.pad_index = kwargs.get("pad_index", 1)
        assert self.pad_index is not None

    def forward(self, src_embed: Tensor, src_length: Tensor, mask: Tensor = None, **kwargs):
        rt = runtime_of(self)
        x, src_length, mask = self.subsampler(src_embed, src_length)  # always subsample; the mask is recomputed
        x = self.pe(x)
        x = Fn.LinearFn.apply(x, rt.weight([self.linear.weight]), self.linear.weight, self.linear.bias, None, None, None)
        p = self.emb_dropout.p if self.training else 0.0
        if p > 0:
            x = Fn.AddPeDropoutFn.apply(x, None, None, p, rt.rng)
        for layer in self.layers:
            x = layer(x, mask)
        if kwargs.get("repad", False) and "src_max_len" in kwargs:
            self.subsample = True
            x, mask = self._repad(x, mask, kwargs["src_max_len"])
        assert src_length.size() == (x.size(0), ), (src_length.size(), x.size())
        assert mask.size() == (x.size(0), 1, x.size(1)), (mask.size(), x.size())
        return x, None, mask
