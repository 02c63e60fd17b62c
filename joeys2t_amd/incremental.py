"""KV-cached incremental decoding for greedy / beam search.

The reference runs the whole decoder over the growing prefix at every step and, per step and layer, re-projects the
encoder states of every hypothesis (search.py:518-534 -> decoders.py:567-625 -> transformer_layers.py:348-407): O(L^2)
decoder work and 2*k*B*S*d^2 projection FLOP per layer and step.  Results do not depend on that: position t of a causal
decoder only needs the keys / values of positions <= t.  Here

  * the encoder-side keys / values of every decoder layer are computed ONCE per utterance (not per hypothesis: a
    hypothesis carries the index of its utterance instead of a tiled copy of the encoder states);
  * self-attention keys / values are appended to a per-layer cache; beam re-ordering rewrites a small per-position
    ancestry table (js2t_attn_decode reads through it), the cache is never copied;
  * one step costs the projections / FFN of ONE position per hypothesis plus two single-query attentions per layer;
  * the ~85 kernel launches of a step are captured once per live-hypothesis count as a hipGraph and replayed: the position
    enters only through device-side counters (positional-table row, cache slot, ancestry column, key count).

Same arithmetic as the full pass in the same order per position (LayerNorm -> projection -> attention -> output projection
+ alpha * residual ...), so fp32 results agree with the reference to rounding; tests check ids bit-exact and scores 1e-4
against the reference captures."""
from typing import List

import torch
from torch import Tensor

from joeys2t_amd import ops
from joeys2t_amd.functional import linear_fwd
from joeys2t_amd.runtime import runtime_of


class IncrementalDecoder:
    def __init__(self, model, encoder_output: Tensor, src_mask: Tensor, rows_per_utt: int, max_len: int, use_graph: bool = True):
        dec = model.decoder
        self.model, self.dec = model, dec
        rt = runtime_of(dec)
        self.rt = rt
        B, S, d = encoder_output.shape
        self.B, self.S, self.d = B, S, d
        self.H = dec.layers[0].trg_trg_att.num_heads
        self.dh = d // self.H
        self.max_len = int(max_len)
        dev = encoder_output.device
        rows = B * rows_per_utt
        mem2d = rt.act_in(encoder_output).reshape(B * S, d).contiguous()
        self.kmask = src_mask.reshape(B, S).to(torch.uint8).contiguous()
        self.layers: List[dict] = []
        for layer in dec.layers:
            ws = layer.trg_trg_att._weights(rt, "self")
            wc = layer.src_trg_att._weights(rt, "cross")
            ff = layer.feed_forward
            l1, l2 = ff.pwff_layer[0], ff.pwff_layer[3]
            kvm = linear_fwd(mem2d, wc["w_kv"], wc["b_kv"])  # [B*S, 2d]: keys | values of the encoder states, once
            self.layers.append(dict(
                ln_mode=layer._layer_norm_position, alpha=layer.alpha, act=ff._activation, ws=ws, wc=wc, kvm=kvm,
                x_ln=(layer.x_layer_norm.weight.data, layer.x_layer_norm.bias.data),
                dec_ln=(layer.dec_layer_norm.weight.data, layer.dec_layer_norm.bias.data),
                ff_ln=(ff.layer_norm.weight.data, ff.layer_norm.bias.data), ff_mode=ff._layer_norm_position, ff_alpha=ff.alpha,
                w1=rt.weight([l1.weight]), b1=rt.bias([l1.bias]), w2=rt.weight([l2.weight]), b2=rt.bias([l2.bias]),
                cache=torch.empty((rows, self.max_len, 2 * d), dtype=kvm.dtype, device=dev)))
        self.final_ln = None if dec.layer_norm is None else (dec.layer_norm.weight.data, dec.layer_norm.bias.data)
        self.w_vocab = rt.weight([dec.output_layer.weight])
        self.rows0 = rows
        self.group = rows_per_utt  # the beams of an utterance stay adjacent (search.py keeps [utterance, beam] order)
        self.table = torch.zeros((rows, self.max_len), dtype=torch.int32, device=dev)  # [hypothesis, position] -> cache row
        self.mem_idx = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(rows_per_utt)
        self.row_ids = torch.arange(rows, dtype=torch.int32, device=dev)
        # position counters live on the device so that ONE captured hipGraph of a step serves every position:
        # t_idx (int64[1]) indexes the positional table / cache slot / ancestry column, len_dev (int32[1]) = t + 1 keys
        self.t_idx = torch.zeros((1, ), dtype=torch.int64, device=dev)
        self.len_dev = torch.ones((1, ), dtype=torch.int32, device=dev)
        self.ids_buf = torch.zeros((rows, ), dtype=torch.long, device=dev)
        self.t = 0
        self.use_graph = use_graph and dev.type == "cuda"
        self._graphs = {}  # live rows -> (graph, logits)

    @staticmethod
    def _ln(x2, gb):
        return ops.layernorm_fwd(x2, gb[0], gb[1], 1e-6)[0]

    def reorder(self, select_indices: Tensor):
        """Hypothesis i of the next step continues hypothesis select_indices[i] of this one (search.py:640-646,757-765).
        In place while the number of live hypotheses is unchanged (a captured graph keeps reading the same buffers)."""
        n = select_indices.shape[0]
        tab, mem = self.table[:self.rows].index_select(0, select_indices), self.mem_idx[:self.rows].index_select(0, select_indices)
        self.table[:n].copy_(tab)
        self.mem_idx[:n].copy_(mem)
        self.rows = n

    def _body(self, rows: int) -> Tensor:
        """One decoding position for the first `rows` hypotheses; every position-dependent access goes through
        t_idx / len_dev (device memory)."""
        d, H, dh = self.d, self.H, self.dh
        ids = self.ids_buf[:rows]
        emb = self.model.trg_embed(ids.reshape(rows, 1))  # lut * sqrt(d), compute dtype
        pe_row = self.dec.pe.pe[0].index_select(0, self.t_idx)  # [1, d]
        x = ops.add_pe_dropout(emb.contiguous(), pe_row, None, 0.0, None, 0).view(rows, d)
        table, mem_idx = self.table[:rows], self.mem_idx[:rows]
        table.index_copy_(1, self.t_idx, self.row_ids[:rows].unsqueeze(1))
        for L in self.layers:
            pre = L["ln_mode"] == "pre"
            alpha = L["alpha"]
            # masked self-attention over the cached prefix
            n = self._ln(x, L["x_ln"]) if pre else x
            qkv = linear_fwd(n, L["ws"]["w_in"], L["ws"]["b_in"])  # columns k | v | q
            cache = L["cache"]
            cache[:rows].index_copy_(1, self.t_idx, qkv[:, :2 * d].unsqueeze(1))
            c = ops.attn_decode(qkv[:, 2 * d:], cache, cache[0, 0, d:], 2 * d, table, self.max_len, self.max_len, 0, None, H, dh,
                                len_dev=self.len_dev)
            u = linear_fwd(c, L["ws"]["w_out"], L["ws"]["b_out"], residual=x if alpha != 0.0 else None, res_scale=alpha)
            h1 = u if pre else self._ln(u, L["x_ln"])
            # encoder-decoder attention over the utterance's cached keys / values
            n = self._ln(h1, L["dec_ln"]) if pre else h1
            q = linear_fwd(n, L["wc"]["w_q"], L["wc"]["b_q"])
            kvm = L["kvm"]
            c = ops.attn_decode(q, kvm, kvm[0, d:], 2 * d, mem_idx, 0, self.S, self.S, self.kmask, H, dh, group=self.group)
            u = linear_fwd(c, L["wc"]["w_out"], L["wc"]["b_out"], residual=h1 if alpha != 0.0 else None, res_scale=alpha)
            h2 = u if pre else self._ln(u, L["dec_ln"])
            # position-wise feed-forward (own LayerNorm and residual)
            fpre = L["ff_mode"] == "pre"
            n = self._ln(h2, L["ff_ln"]) if fpre else h2
            c = linear_fwd(n, L["w1"], L["b1"], act=L["act"])
            u = linear_fwd(c, L["w2"], L["b2"], residual=h2 if L["ff_alpha"] != 0.0 else None, res_scale=L["ff_alpha"])
            x = u if fpre else self._ln(u, L["ff_ln"])
        if self.final_ln is not None:
            x = self._ln(x, self.final_ln)
        logits = linear_fwd(x, self.w_vocab, None, out_dtype=torch.float32)
        self.t_idx.add_(1)
        self.len_dev.add_(1)
        return logits

    def step(self, last_ids: Tensor) -> Tensor:
        """Logits f32[rows, V] of the position after `last_ids` (the newest token of every live hypothesis)."""
        if self.t >= self.max_len:
            raise ops.Js2tError(f"incremental decoder: step {self.t} exceeds the cache length {self.max_len}")
        rows = last_ids.shape[0]
        self.rows = rows
        with torch.no_grad():
            self.ids_buf[:rows].copy_(last_ids)
            if not self.use_graph:
                logits = self._body(rows)
            else:
                entry = self._graphs.get(rows)
                if entry is None:
                    # first position with this many live hypotheses: run it eagerly twice is not possible (the position
                    # advances), so capture directly after a side-stream warm-up of a throw-away copy of the counters
                    t_save, l_save = self.t_idx.clone(), self.len_dev.clone()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self._body(rows)
                    torch.cuda.current_stream().wait_stream(side)
                    self.t_idx.copy_(t_save)
                    self.len_dev.copy_(l_save)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        out = self._body(rows)
                    # capture does not execute: restore nothing, replay now for this position
                    self._graphs[rows] = entry = (g, out)
                g, out = entry
                g.replay()
                logits = out
        self.t += 1
        return logits
