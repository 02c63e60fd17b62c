"""Transformer decoder with the reference's module / parameter names (joeynmt/decoders.py:498-635),
computed by HIP kernels; includes the CTC projection of the encoder output (:560-565,622-623)."""
from typing import Optional

import torch
from torch import Tensor, nn

from joeys2t_amd import functional as Fn
from joeys2t_amd.helpers import freeze_params, subsequent_mask
from joeys2t_amd.runtime import runtime_of
from joeys2t_amd.transformer_layers import PositionalEncoding, TransformerDecoderLayer


class Decoder(nn.Module):
    """Base decoder class."""

    @property
    def output_size(self):
        return self._output_size


PACK_MEMORY = True  # the decoder reads the encoder states of a ragged batch as packed rows when the encoder packed them (tests flip this)


class TransformerDecoder(Decoder):
    def __init__(self, num_layers: int = 4, num_heads: int = 8, hidden_size: int = 512, ff_size: int = 2048,
                 dropout: float = 0.1, emb_dropout: float = 0.1, vocab_size: int = 1, freeze: bool = False, **kwargs):
        super().__init__()
        self._hidden_size = hidden_size
        self._output_size = vocab_size
        self.layers = nn.ModuleList([
            TransformerDecoderLayer(size=hidden_size, ff_size=ff_size, num_heads=num_heads, dropout=dropout,
                                    alpha=kwargs.get("alpha", 1.0), layer_norm=kwargs.get("layer_norm", "post"),
                                    activation=kwargs.get("activation", "relu")) for _ in range(num_layers)
        ])
        self.pe = PositionalEncoding(hidden_size)
        self.layer_norm = nn.LayerNorm(hidden_size, eps=1e-6) if kwargs.get("layer_norm", "post") == "pre" else None
        self.emb_dropout = nn.Dropout(p=emb_dropout)
        self.output_layer = nn.Linear(hidden_size, vocab_size, bias=False)
        if freeze:
            freeze_params(self)
        self.ctc_output_layer = None
        encoder_output_size = kwargs.get("encoder_output_size_for_ctc", None)
        if encoder_output_size is not None:
            self.ctc_output_layer = nn.Linear(encoder_output_size, vocab_size, bias=False)

    def _cross_kv_params(self):
        atts = [layer.src_trg_att for layer in self.layers]
        return ([p for a in atts for p in (a.k_layer.weight, a.v_layer.weight)],
                [p for a in atts for p in (a.k_layer.bias, a.v_layer.bias)])

    def fuse_groups(self):
        """The cross-attention k / v weights of ALL layers adjacent in the flat store ([k_0; v_0; k_1; ...], biases likewise):
        their projections of the encoder states are one product (memory_kv).  The decoder comes before its layers in
        module order, so these groups win over MultiHeadedAttention's [k; v; q] (whose q then stands alone, as the cross
        block wants it)."""
        return list(self._cross_kv_params())

    def memory_kv(self, memory: Tensor) -> Optional[Tensor]:
        """k_layer / v_layer of every layer's src_trg_att applied to the encoder states in ONE product (the reference:
        per layer, transformer_layers.py:66-68 under :383) -> [B*S, L*2d], layer i's [k | v] from column i*2d; None when
        that cannot be had - parameters not adjacent in a flat store, or gradients wanted outside TrainStep.micro_step (see
        functional.MemoryKVFn) - and the layers project for themselves."""
        rt = runtime_of(self)
        if not Fn.GROUP_MEMORY_KV or len(self.layers) < 2 or rt.store is None or memory.dim() != 3:
            return None
        ws, bs = self._cross_kv_params()
        if any(b is None for b in bs):
            return None
        w, b = rt.store.view(ws, rt.compute_dtype), rt.store.view(bs, torch.float32)
        if w is None or b is None:
            return None
        memory = rt.act_in(memory)
        need_grad = torch.is_grad_enabled() and (memory.requires_grad or any(p.requires_grad for p in ws + bs))
        if need_grad and not Fn.chain_active():
            return None
        wts = {"w_kv": w, "b_kv": b, "w_kv_t": rt.weight_t(ws), "sink": rt.sinks({"w_kv": ws, "b_kv": bs}), "notify": rt.grads_ready}
        return Fn.MemoryKVFn.apply(memory, wts, *ws, *bs)

    def project(self, layer: nn.Linear, x: Tensor, out_dtype) -> Tensor:
        rt = runtime_of(self)
        x = rt.act_in(x)
        bias = None if layer.bias is None else layer.bias
        smap = {"w": [layer.weight]}
        if bias is not None:
            smap["b"] = [bias]
        sk = rt.sinks(smap)
        if sk is not None:
            sk["_w_t"] = rt.weight_t([layer.weight])
        return Fn.LinearFn.apply(x, rt.weight([layer.weight]), layer.weight, bias, out_dtype, sk, rt.grads_ready)

    def forward(self, trg_embed: Tensor, encoder_output: Tensor, encoder_hidden: Tensor, src_mask: Tensor,
                unroll_steps: int, hidden: Tensor, trg_mask: Tensor, **kwargs):
        """-> (logits [B,L,V] fp32, hidden [B,L,d], cross-attention weights of the last layer | None, None,
        CTC logits [B,T',V] | None)."""
        assert trg_mask is not None, "trg_mask required for Transformer"
        rt = runtime_of(self)
        x = self.pe(trg_embed, extra=kwargs.get("trg_prompt_mask", None), dropout=self.emb_dropout.p,
                    training=self.training)
        L = trg_embed.size(1)
        trg_mask = (trg_mask & subsequent_mask(L, device=trg_mask.device)).contiguous()  # [B|1, L, L]
        last_layer = len(self.layers) - 1
        return_attention = kwargs.get("return_attention", False)
        att = None
        # Ragged batch whose encoder ran on packed rows (encoders.TransformerEncoder, `memory_pack` = its ops.PackedRows): the K | V
        # projections of all layers, their gradients and the cross-attention keys stay on the LIVE positions too - the padded
        # [B, T', d] states are packed again right here (a 12 MB copy; the tensor the backward pass is cut at stays the padded
        # one), the cross-attention kernels read each utterance's own row range (js2t_attn_desc.seg_keys).  Needs the grouped
        # projections and the fused kernels (no attention weights to return).
        mem_pack = kwargs.get("memory_pack")
        kv_all = None
        if (mem_pack is not None and not return_attention and Fn.USE_FLASH and rt.compute_dtype == torch.bfloat16 and PACK_MEMORY and
                encoder_output.dim() == 3 and (mem_pack.B, mem_pack.T) == tuple(encoder_output.shape[:2]) and
                (self._hidden_size // self.layers[0].src_trg_att.num_heads) in (64, 128)):
            kv_all = self.memory_kv(Fn.PackRowsFn.apply(rt.act_in(encoder_output), mem_pack))
        if kv_all is None:
            mem_pack = None
            kv_all = self.memory_kv(encoder_output)
        d2 = 2 * self._hidden_size
        for i, layer in enumerate(self.layers):
            x, att = layer(x=x, memory=encoder_output, src_mask=src_mask, trg_mask=trg_mask,
                           return_attention=(return_attention and i == last_layer),
                           memory_kv=None if kv_all is None else (kv_all, i * d2), mem_pack=mem_pack)
        if self.layer_norm is not None:
            sk = rt.sinks({"g": [self.layer_norm.weight], "b": [self.layer_norm.bias]})
            x = Fn.LayerNormFn.apply(x, self.layer_norm.weight, self.layer_norm.bias, None if sk is None else (sk["g"], sk["b"], sk.get("_copies")),
                                     rt.grads_ready)
        # decoding only scores the newest position (search.py:534 `logits[:, -1]`): project just that row
        out = self.project(self.output_layer, x[:, -1:].contiguous() if kwargs.get("last_only", False) else x, torch.float32)
        ctc_output = None
        if self.ctc_output_layer is not None and kwargs.get("compute_ctc", True):
            ctc_pack = kwargs.get("ctc_pack")  # ops.PackedRows: project the packed encoder rows -> [1, rows, V] (model._ctc_packing)
            ctc_in = encoder_output if ctc_pack is None else Fn.PackRowsFn.apply(rt.act_in(encoder_output), ctc_pack)
            ctc_output = self.project(self.ctc_output_layer, ctc_in, rt.compute_dtype)
        return out, x, att, None, ctc_output

    def __repr__(self):
        return (f"{self.__class__.__name__}(num_layers={len(self.layers)}, "
                f"num_heads={self.layers[0].trg_trg_att.num_heads}, alpha={self.layers[0].alpha}, "
                f'layer_norm="{self.layers[0]._layer_norm_position}", ctc_layer={self.ctc_output_layer is not None})')
