"""CPU ORACLE for the JoeyS2T hot path — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch / NumPy fp32 restatement of what the reference computes on its CPU path, written as pure
functions over a `state_dict` (no nn.Module).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this file; the product package `joeys2t_amd` never does (it fails loudly without
its HIP library instead).

Pinning: every function is checked against golden vectors produced by the *real* reference imported in the
build container (`oracle/make_golden.py` -> `tests/golden/*.npz`, `tests/test_oracle_golden.py`), which in turn
contain the reference's own unit-test constants.  The Kaldi fbank has no in-tree reference (torchaudio is a
third-party dependency that is not installed): it is restated from the published Kaldi/torchaudio algorithm and
pinned by the reference's own known-answer test (test/unit/test_tokenizer.py:310-329) and frame counts
(test/data/speech/test.tsv).

Each function cites the reference file:line (relative to the reference repo) it follows.
"""
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------------------
# helpers (reference helpers.py)
# --------------------------------------------------------------------------------------------------
def subsequent_mask(size: int) -> Tensor:
    """helpers.py:81-90"""
    return torch.tril(torch.ones(size, size, dtype=torch.bool)).unsqueeze(0)


def lengths_to_padding_mask(lengths: Tensor) -> Tensor:
    """helpers.py:459-469"""
    max_len = int(lengths.max().item())
    return torch.arange(max_len).unsqueeze(0) < lengths.view(-1, 1)


def subsample_lengths(lengths: Tensor, kernel_sizes: List[int]) -> Tensor:
    """encoders.py:348-352 (float arithmetic then floor, as the reference does)"""
    out = lengths.clone()
    for k in kernel_sizes:
        out = ((out.float() + 2 * (k // 2) - (k - 1) - 1) / 2 + 1).floor().long()
    return out


def positional_table(max_len: int, size: int) -> Tensor:
    """transformer_layers.py:192-199"""
    pe = torch.zeros(max_len, size)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, size, 2, dtype=torch.float) * -(math.log(10000.0) / size))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


def activation(name: str):
    """builders.py:24-41"""
    return {"relu": F.relu, "gelu": F.gelu, "tanh": torch.tanh, "swish": F.silu}[name]


# --------------------------------------------------------------------------------------------------
# bf16 emulation (a YARDSTICK for the tests' bf16 bounds, not part of the reference's arithmetic)
# --------------------------------------------------------------------------------------------------
# With `bf16_emulation(True)` the same fp32 functions round where a bf16 compute path must round whatever its kernels look
# like: every activation that is stored between two operators (and its gradient on the way back) and both operands of every
# matrix product take a bf16 round trip; sums stay in fp32.  How far THIS run's gradients move from the plain fp32 run is what
# bf16 storage costs at a given model width and depth - an estimate that owes nothing to the HIP kernels, against which their
# bf16 gradients are then judged (tests/test_hip_config_width.py).
_EMULATE_BF16 = False


class bf16_emulation:
    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        global _EMULATE_BF16
        self.prev, _EMULATE_BF16 = _EMULATE_BF16, self.on
        return self

    def __exit__(self, *exc):
        global _EMULATE_BF16
        _EMULATE_BF16 = self.prev


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


class _RoundValue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _stored(x: Tensor) -> Tensor:
    """an activation kept in bf16 between two operators: value and gradient both rounded"""
    return _RoundBoth.apply(x) if _EMULATE_BF16 and x.is_floating_point() else x


def _operand(w: Tensor) -> Tensor:
    """a parameter entering a product as bf16 (its gradient is accumulated in fp32)"""
    return _RoundValue.apply(w) if _EMULATE_BF16 and w.is_floating_point() else w


# --------------------------------------------------------------------------------------------------
# layers (reference transformer_layers.py / encoders.py / decoders.py)
# --------------------------------------------------------------------------------------------------
def linear(sd: SD, prefix: str, x: Tensor) -> Tensor:
    return _stored(F.linear(_stored(x), _operand(sd[prefix + ".weight"]), sd.get(prefix + ".bias")))


def layer_norm(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """nn.LayerNorm(eps=1e-6): transformer_layers.py:146,248,339-340"""
    return _stored(F.layer_norm(_stored(x), (x.size(-1), ), sd[prefix + ".weight"], sd[prefix + ".bias"], eps=1e-6))


def rel_pos_scores(rel_bias: Tensor, Tq: int, Tk: int) -> Tensor:
    """EXTENSION (BASELINE config 5 "rel-pos attn"; the reference's attention has no relative term): the additive score
    term [H, Tq, Tk] of a learned bias per head and clipped distance, rel_bias[h, clamp(j - i, -R, R) + R], R = (cols-1)/2."""
    R = (rel_bias.size(1) - 1) // 2
    idx = (torch.arange(Tk)[None, :] - torch.arange(Tq)[:, None]).clamp(-R, R) + R
    return rel_bias[:, idx]


def mha(sd: SD, prefix: str, k: Tensor, v: Tensor, q: Tensor, mask: Optional[Tensor], num_heads: int,
        return_weights: bool = False):
    """MultiHeadedAttention.forward, transformer_layers.py:49-115 (eval mode: dropout is the identity).  A
    `<prefix>.rel_pos_bias` entry in sd adds rel_pos_scores() to the scaled scores (extension, see there)."""
    B, d = k.size(0), q.size(-1)
    dh = d // num_heads
    k = linear(sd, prefix + ".k_layer", k).view(B, -1, num_heads, dh).transpose(1, 2)
    v = linear(sd, prefix + ".v_layer", v).view(B, -1, num_heads, dh).transpose(1, 2)
    q = linear(sd, prefix + ".q_layer", q).view(B, -1, num_heads, dh).transpose(1, 2)
    q = q / math.sqrt(dh)  # scaled BEFORE the product (:86)
    scores = torch.matmul(q, k.transpose(2, 3))
    if prefix + ".rel_pos_bias" in sd:
        scores = scores + rel_pos_scores(sd[prefix + ".rel_pos_bias"], scores.size(2), scores.size(3)).unsqueeze(0)
    if mask is not None:
        scores = scores.masked_fill(~mask.unsqueeze(1), float("-inf"))
    weights = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(_stored(weights), v).transpose(1, 2).contiguous().view(B, -1, d)
    out = linear(sd, prefix + ".output_layer", ctx)
    if return_weights:
        return out, weights.sum(dim=1) / num_heads
    return out, None


def feed_forward(sd: SD, prefix: str, x: Tensor, alpha: float, ln_pos: str, act: str) -> Tensor:
    """PositionwiseFeedForward.forward, transformer_layers.py:159-168"""
    residual = x
    if ln_pos == "pre":
        x = layer_norm(sd, prefix + ".layer_norm", x)
    x = linear(sd, prefix + ".pwff_layer.3", activation(act)(linear(sd, prefix + ".pwff_layer.0", x))) + alpha * residual
    if ln_pos == "post":
        x = layer_norm(sd, prefix + ".layer_norm", x)
    return x


def encoder_layer(sd: SD, prefix: str, x: Tensor, mask: Tensor, cfg: dict) -> Tensor:
    """TransformerEncoderLayer.forward, transformer_layers.py:267-289"""
    alpha, ln_pos = cfg["alpha"], cfg["layer_norm"]
    residual = x
    if ln_pos == "pre":
        x = layer_norm(sd, prefix + ".layer_norm", x)
    x, _ = mha(sd, prefix + ".src_src_att", x, x, x, mask, cfg["num_heads"])
    x = x + alpha * residual
    if ln_pos == "post":
        x = layer_norm(sd, prefix + ".layer_norm", x)
    return feed_forward(sd, prefix + ".feed_forward", x, alpha, ln_pos, cfg["activation"])


def decoder_layer(sd: SD, prefix: str, x: Tensor, memory: Tensor, src_mask: Tensor, trg_mask: Tensor, cfg: dict,
                  return_attention: bool = False):
    """TransformerDecoderLayer.forward, transformer_layers.py:348-407"""
    alpha, ln_pos, H = cfg["alpha"], cfg["layer_norm"], cfg["num_heads"]
    residual = x
    if ln_pos == "pre":
        x = layer_norm(sd, prefix + ".x_layer_norm", x)
    h1, _ = mha(sd, prefix + ".trg_trg_att", x, x, x, trg_mask, H)
    h1 = h1 + alpha * residual
    if ln_pos == "post":
        h1 = layer_norm(sd, prefix + ".x_layer_norm", h1)
    h1_residual = h1
    if ln_pos == "pre":
        h1 = layer_norm(sd, prefix + ".dec_layer_norm", h1)
    h2, att = mha(sd, prefix + ".src_trg_att", memory, memory, h1, src_mask, H, return_weights=return_attention)
    h2 = h2 + alpha * h1_residual
    if ln_pos == "post":
        h2 = layer_norm(sd, prefix + ".dec_layer_norm", h2)
    return feed_forward(sd, prefix + ".feed_forward", h2, alpha, ln_pos, cfg["activation"]), att


def conv_subsample(sd: SD, prefix: str, x: Tensor, lengths: Tensor, kernel_sizes: List[int]) -> Tuple[Tensor, Tensor]:
    """Conv1dSubsampler.forward, encoders.py:354-373 (padded frames are convolved like any other)."""
    max_len = int(lengths.max().item())
    if x.size(1) != max_len:
        x = x[:, :max_len, :]
    x = x.transpose(1, 2).contiguous()
    for i, k in enumerate(kernel_sizes):
        x = _stored(F.conv1d(_stored(x), _operand(sd[f"{prefix}.conv_layers.{i}.weight"]), sd[f"{prefix}.conv_layers.{i}.bias"], stride=2,
                             padding=k // 2))
        x = _stored(F.glu(x, dim=1))
    return x.transpose(1, 2).contiguous(), subsample_lengths(lengths, kernel_sizes)


def encoder_forward(sd: SD, cfg: dict, src: Tensor, src_length: Tensor, prefix: str = "encoder"):
    """TransformerEncoder.forward, encoders.py:241-288 (eval mode) -> (x, mask [B,1,T'], lengths)."""
    e = cfg["encoder"]
    x, lengths = src, src_length
    if e.get("subsample", False):
        x, lengths = conv_subsample(sd, prefix + ".subsampler", x, lengths, e["conv_kernel_sizes"])
    mask = lengths_to_padding_mask(lengths).unsqueeze(1)
    x = x + positional_table(5000, x.size(-1))[: x.size(1)].unsqueeze(0)
    for i in range(e["num_layers"]):
        x = encoder_layer(sd, f"{prefix}.layers.{i}", x, mask, e)
    if e["layer_norm"] == "pre":
        x = layer_norm(sd, prefix + ".layer_norm", x)
    return x, mask, lengths


def encoder_forward_text(sd: SD, cfg: dict, src_ids: Tensor, pad_index: int, prefix: str = "encoder"):
    """Text source (task "MT", configs/transformer_small.yaml): Model._encode embeds the ids with `src_embed`
    (model.py:225-238, embeddings.py:55-64), Batch supplies src_mask = (src != pad) (batch.py:99-100), and the encoder runs
    without its sub-sampler (encoders.py:241-288) -> (x, mask [B,1,S])."""
    e = cfg["encoder"]
    w = sd["src_embed.lut.weight"]
    x = F.embedding(src_ids, w)
    if e["embeddings"].get("scale", False):
        x = x * math.sqrt(w.size(1))
    mask = (src_ids != pad_index).unsqueeze(1)
    x = x + positional_table(5000, x.size(-1))[: x.size(1)].unsqueeze(0)
    for i in range(e["num_layers"]):
        x = encoder_layer(sd, f"{prefix}.layers.{i}", x, mask, e)
    if e["layer_norm"] == "pre":
        x = layer_norm(sd, prefix + ".layer_norm", x)
    return x, mask


# ------------------------------------------------------------------------------------------------ Conformer (a30)
def conv_module(sd: SD, prefix: str, x: Tensor, train: bool, new_stats: Optional[dict] = None) -> Tensor:
    """ConvolutionModule.forward, transformer_layers.py:458-475, on the tensor it is GIVEN: the layer hands it
    x.transpose(0, 1) of a [B, T, C] tensor (:549-552), so "batch" = T and the depthwise convolution and the BatchNorm
    length axis run over B.  This function takes the layer's [B, T, C] view and does that transposition itself.
    BatchNorm1d: batch statistics (biased variance) in train mode, running statistics in eval mode; `new_stats` receives
    the running statistics after the momentum-0.1 update (unbiased variance) when training."""
    x = layer_norm(sd, prefix + ".layer_norm", x)
    xt = x.transpose(0, 1).transpose(1, 2)  # [T, C, B]
    xt = F.conv1d(xt, sd[prefix + ".pointwise_conv1.weight"], sd[prefix + ".pointwise_conv1.bias"])
    xt = F.glu(xt, dim=1)
    wd = sd[prefix + ".depthwise_conv.weight"]
    xt = F.conv1d(xt, wd, sd[prefix + ".depthwise_conv.bias"], padding=(wd.size(-1) - 1) // 2, groups=wd.size(0))
    rm, rv = sd[prefix + ".batch_norm.running_mean"], sd[prefix + ".batch_norm.running_var"]
    if train:
        mean = xt.mean(dim=(0, 2))
        var = xt.var(dim=(0, 2), unbiased=False)
        n = xt.size(0) * xt.size(2)
        if new_stats is not None:
            new_stats[prefix + ".batch_norm.running_mean"] = (0.9 * rm + 0.1 * mean).detach()
            new_stats[prefix + ".batch_norm.running_var"] = (0.9 * rv + 0.1 * var * n / max(n - 1, 1)).detach()
    else:
        mean, var = rm, rv
    xt = (xt - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5)
    xt = xt * sd[prefix + ".batch_norm.weight"][None, :, None] + sd[prefix + ".batch_norm.bias"][None, :, None]
    xt = F.hardswish(xt)  # named "swish" in the reference (:449)
    xt = F.conv1d(xt, sd[prefix + ".pointwise_conv2.weight"], sd[prefix + ".pointwise_conv2.bias"])
    return xt.transpose(1, 2).transpose(0, 1)


def conformer_layer(sd: SD, prefix: str, x: Tensor, mask: Tensor, cfg: dict, train: bool, new_stats: Optional[dict] = None) -> Tensor:
    """ConformerEncoderLayer.forward, transformer_layers.py:526-565 (dropout 0): half-step residuals on top of the
    feed-forward modules' own residual, self-attention, the convolution module, final LayerNorm for post-LN."""
    alpha, ln_pos = cfg["alpha"], cfg["layer_norm"]
    residual = x
    x = 0.5 * feed_forward(sd, prefix + ".initial_feed_forward", x, alpha, ln_pos, "relu") + residual
    residual = x
    if ln_pos == "pre":
        x = layer_norm(sd, prefix + ".src_att_layer_norm", x)
    x, _ = mha(sd, prefix + ".src_src_att", x, x, x, mask, cfg["num_heads"])
    x = x + alpha * residual
    if ln_pos == "post":
        x = layer_norm(sd, prefix + ".src_att_layer_norm", x)
    x = conv_module(sd, prefix + ".conv_module", x, train, new_stats) + alpha * x
    residual = x
    if ln_pos == "pre":
        x = layer_norm(sd, prefix + ".final_layer_norm", x)
    x = 0.5 * feed_forward(sd, prefix + ".final_feed_forward", x, alpha, ln_pos, "relu") + residual
    if ln_pos == "post":
        x = layer_norm(sd, prefix + ".final_layer_norm", x)
    return x


def conformer_encoder_forward(sd: SD, cfg: dict, src: Tensor, src_length: Tensor, train: bool = False, prefix: str = "encoder",
                              new_stats: Optional[dict] = None):
    """ConformerEncoder.forward, encoders.py:425-445: subsample -> mask -> pe -> Linear -> layers (no final LayerNorm)."""
    e = cfg["encoder"]
    x, lengths = conv_subsample(sd, prefix + ".subsampler", src, src_length, e["conv_kernel_sizes"])
    mask = lengths_to_padding_mask(lengths).unsqueeze(1)
    x = x + positional_table(5000, x.size(-1))[: x.size(1)].unsqueeze(0)
    x = linear(sd, prefix + ".linear", x)
    for i in range(e["num_layers"]):
        x = conformer_layer(sd, f"{prefix}.layers.{i}", x, mask, e, train, new_stats)
    return x, mask, lengths


def embed(sd: SD, cfg: dict, ids: Tensor) -> Tensor:
    """Embeddings.forward, embeddings.py:55-64"""
    w = sd["trg_embed.lut.weight"]
    x = F.embedding(ids, w)
    return x * math.sqrt(w.size(1)) if cfg["decoder"]["embeddings"].get("scale", False) else x


def decoder_forward(sd: SD, cfg: dict, trg_input: Tensor, memory: Tensor, src_mask: Tensor, trg_mask: Tensor,
                    return_attention: bool = False, prefix: str = "decoder", trg_prompt_mask: Optional[Tensor] = None):
    """TransformerDecoder.forward, decoders.py:567-625 (eval mode) -> (logits, hidden, att, ctc_logits|None).
    trg_prompt_mask (0/1 ids [B, L]): embedded with the target table and added after the positional encoding
    (model.py:271-282, decoders.py:600-602)."""
    x = embed(sd, cfg, trg_input)
    return decoder_forward_embedded(sd, cfg, x, memory, src_mask, trg_mask, return_attention, prefix,
                                    None if trg_prompt_mask is None else embed(sd, cfg, trg_prompt_mask))


def decoder_forward_embedded(sd: SD, cfg: dict, trg_embed: Tensor, memory: Tensor, src_mask: Tensor, trg_mask: Tensor,
                             return_attention: bool = False, prefix: str = "decoder", prompt_embed: Optional[Tensor] = None):
    """TransformerDecoder.forward proper (decoders.py:567-625): takes the embedded targets, as the reference's own
    known-answer test does (test/unit/test_transformer_decoder.py:45-172)."""
    dcfg = cfg["decoder"]
    x = trg_embed + positional_table(5000, trg_embed.size(-1))[: trg_embed.size(1)].unsqueeze(0)
    if prompt_embed is not None:
        x = x + prompt_embed
    tmask = trg_mask & subsequent_mask(trg_embed.size(1))
    att = None
    n = dcfg["num_layers"]
    for i in range(n):
        x, att = decoder_layer(sd, f"{prefix}.layers.{i}", x, memory, src_mask, tmask, dcfg,
                               return_attention=(return_attention and i == n - 1))
    if dcfg["layer_norm"] == "pre":
        x = layer_norm(sd, prefix + ".layer_norm", x)
    out = F.linear(_stored(x), _operand(sd[prefix + ".output_layer.weight"]))  # the vocabulary logits stay fp32
    ctc = None
    if prefix + ".ctc_output_layer.weight" in sd:
        ctc = _stored(F.linear(_stored(memory), _operand(sd[prefix + ".ctc_output_layer.weight"])))
    return out, x, att, ctc


# --------------------------------------------------------------------------------------------------
# losses (reference loss.py, model.py:113-148)
# --------------------------------------------------------------------------------------------------
def smooth_targets(targets: Tensor, vocab_size: int, pad_index: int, smoothing: float) -> Tensor:
    """XentLoss._smooth_targets, loss.py:35-58"""
    dist = torch.full((targets.size(0), vocab_size), smoothing / (vocab_size - 2))
    dist.scatter_(1, targets.unsqueeze(1), 1.0 - smoothing)
    dist[:, pad_index] = 0
    dist[targets == pad_index] = 0.0
    return dist


def xent_loss(log_probs: Tensor, trg: Tensor, pad_index: int, smoothing: float) -> Tensor:
    """XentLoss.forward, loss.py:85-101"""
    V = log_probs.size(-1)
    lp = log_probs.contiguous().view(-1, V)
    t = trg.contiguous().view(-1)
    if smoothing > 0:
        return F.kl_div(lp, smooth_targets(t, V, pad_index, smoothing), reduction="sum")
    return F.nll_loss(lp, t, ignore_index=pad_index, reduction="sum")


def ctc_loss(ctc_log_probs: Tensor, trg: Tensor, input_lengths: Tensor, target_lengths: Tensor, blank: int) -> Tensor:
    """XentCTCLoss.forward, loss.py:156-161: nn.CTCLoss(blank=bos, reduction='sum', zero_infinity=True)"""
    return F.ctc_loss(ctc_log_probs.transpose(0, 1).contiguous(), trg, input_lengths, target_lengths, blank=blank,
                      reduction="sum", zero_infinity=True)


def model_loss(sd: SD, cfg: dict, batch: dict, specials: dict, smoothing: float, ctc_weight: Optional[float], train: bool = False,
               new_stats: Optional[dict] = None):
    """Model.forward(return_type="loss"), model.py:113-148 -> (total, xent, ctc|None, n_correct, logits, ctc_logits).
    `train` / `new_stats` only matter for a Conformer encoder (BatchNorm takes batch statistics in training mode)."""
    if "src_embed.lut.weight" in sd:  # text source (MT): no sub-sampler, mask from the pad positions
        enc, src_mask = encoder_forward_text(sd, cfg, batch["src"], specials["pad"])
    elif cfg["encoder"].get("type", "transformer") == "conformer":
        # EXTENSION (BASELINE.json configs[4]): the reference's Model is never built over its ConformerEncoder (model.py:417-421);
        # the composition is Model._encode_decode's (model.py:170-239) with that encoder class in the encoder's place
        enc, src_mask, _ = conformer_encoder_forward(sd, cfg, batch["src"], batch["src_length"], train=train, new_stats=new_stats)
    else:
        enc, src_mask, _ = encoder_forward(sd, cfg, batch["src"], batch["src_length"])
    out, _, _, ctc_out = decoder_forward(sd, cfg, batch["trg_input"], enc, src_mask, batch["trg_mask"])
    log_probs = F.log_softmax(out, dim=-1)
    xent = xent_loss(log_probs, batch["trg"], specials["pad"], smoothing)
    ctc = None
    total = xent
    if ctc_weight is not None and ctc_out is not None:
        ctc = ctc_loss(F.log_softmax(ctc_out, dim=-1), batch["trg"], src_mask.squeeze(1).sum(dim=1), batch["trg_length"],
                       specials["bos"])
        total = (1.0 - ctc_weight) * xent + ctc_weight * ctc
    tmask = batch["trg_mask"].squeeze(1)
    n_correct = torch.sum(log_probs.argmax(-1).masked_select(tmask).eq(batch["trg"].masked_select(tmask)))
    return total, xent, ctc, n_correct, out, ctc_out


# --------------------------------------------------------------------------------------------------
# search (reference search.py)
# --------------------------------------------------------------------------------------------------
def _forbid(log_probs: Tensor, ids: List[Optional[int]]):
    for i in ids:
        if i is not None and i < log_probs.size(1):
            log_probs[:, i] = float("-inf")


def adjust_mask_size(mask: Optional[Tensor], batch_size: int, hyp_len: int) -> Optional[Tensor]:
    """helpers.py:307-326"""
    if mask is None:
        return None
    if mask.size(1) < hyp_len:
        out = mask.new_zeros((batch_size, hyp_len))
        out[:, :mask.size(1)] = mask
        return out
    return mask[:, :hyp_len]


def block_repeat_ngrams(tokens: Tensor, scores: Tensor, n: int, step: int, src_tokens: Optional[Tensor] = None,
                        exclude_tokens: Optional[List[int]] = None) -> Tensor:
    """search.py:915-969: ban every token that would complete an n-gram already present in the hypothesis (or the source)."""
    trg = tokens.tolist()
    src = None if src_tokens is None else src_tokens.tolist()
    check_end_pos, offset = step + 2 - n, n - 1
    for h in range(tokens.size(0)):
        banned = set()
        if len(trg[h]) > n:
            ngram = trg[h][-offset:]
            for i in range(1, check_end_pos):
                if ngram == trg[h][i:i + offset]:
                    banned.add(trg[h][i + offset])
            if src is not None:
                for i in range(len(src[h]) + 1 - n):
                    if ngram == src[h][i:i + offset]:
                        banned.add(src[h][i + offset])
        banned -= set(exclude_tokens or [])
        scores[h, list(banned)] = float("-inf")
    return scores


def penalize_repetition(tokens: Tensor, scores: Tensor, penalty: float) -> Tensor:
    """search.py:972-1001: gather, x*penalty if x<0 else x/penalty, scatter.  The reference's `exclude_tokens` restore copies
    from an alias of `scores` itself (:986), i.e. restores nothing - so there is no such parameter here."""
    score = torch.gather(scores, 1, tokens)
    score = torch.where(score < 0, score * penalty, score / penalty)
    scores.scatter_(1, tokens, score)
    return scores


def greedy(sd: SD, cfg: dict, specials: dict, enc: Tensor, src_mask: Tensor, max_output_length: int,
           min_output_length: int = 1, generate_unk: bool = True, return_prob: bool = False, repetition_penalty: float = -1,
           no_repeat_ngram_size: int = -1, encoder_input: Optional[Tensor] = None, decoder_prompt: Optional[Tensor] = None,
           trg_prompt_mask: Optional[Tensor] = None, return_attention: bool = False):
    """transformer_greedy, search.py:162-342 -> (ids [B,L], scores|None) or, with return_attention, (ids, scores, att).
    specials may carry "sep", "lang_tags" (list) and "all" (model.specials)."""
    B, _, src_len = src_mask.size()
    tags = list(specials.get("lang_tags", []))
    excl = list(specials.get("all", [])) + tags
    ys = torch.full((B, 1), specials["bos"], dtype=torch.long)
    yv = torch.zeros((B, 1)) if return_prob else None
    yt = torch.zeros((B, 1, src_len)) if return_attention else None
    trg_mask = torch.ones(1, 1, 1, dtype=torch.bool)
    finished = torch.zeros(B, 1, dtype=torch.uint8)
    compute_softmax = return_prob or repetition_penalty > 0 or no_repeat_ngram_size > 0 or encoder_input is not None
    for step in range(max_output_length):
        forced_word = decoder_prompt[:, step + 1].unsqueeze(1) if decoder_prompt is not None and decoder_prompt.size(1) > step + 1 \
            else torch.full((B, 1), specials["pad"], dtype=torch.long)
        forced_mask = trg_prompt_mask[:, step + 1].unsqueeze(1).bool() if trg_prompt_mask is not None and trg_prompt_mask.size(1) > step + 1 \
            else torch.zeros((B, 1), dtype=torch.bool)
        if torch.any(~forced_mask).item():
            logits, _, att, _ = decoder_forward(sd, cfg, ys, enc, src_mask, trg_mask, return_attention=return_attention,
                                                trg_prompt_mask=adjust_mask_size(trg_prompt_mask, B, ys.size(1)))
            lp = logits[:, -1]
            if compute_softmax:
                lp = F.log_softmax(lp, dim=-1)
                if no_repeat_ngram_size > 1:
                    lp = block_repeat_ngrams(ys, lp, no_repeat_ngram_size, step, encoder_input, excl)
                if repetition_penalty > 1.0:
                    lp = penalize_repetition(ys, lp, repetition_penalty)
                    if encoder_input is not None:
                        lp = penalize_repetition(encoder_input, lp, repetition_penalty)
            _forbid(lp, [specials["bos"], specials.get("sep")] + tags)
            if not generate_unk:
                lp[:, specials["unk"]] = float("-inf")
            if step < min_output_length:
                lp[:, specials["eos"]] = float("-inf")
            prob, nxt = torch.max(lp, dim=1)
            nxt = torch.where(forced_mask, forced_word, nxt.unsqueeze(-1))
            prob = torch.where(forced_mask, torch.zeros(B, 1), prob.unsqueeze(-1))
            if return_attention:
                att = torch.where(forced_mask.expand(-1, src_len).unsqueeze(1), torch.zeros(B, 1, src_len), att[:, -1, :].unsqueeze(1))
        else:
            nxt, prob = forced_word, torch.zeros(B, 1)
            att = torch.zeros(B, 1, src_len) if return_attention else None
        ys = torch.cat([ys, nxt], dim=1)
        if return_prob:
            yv = torch.cat([yv, prob], dim=1)
        if return_attention:
            yt = torch.cat([yt, att], dim=1)
        finished += nxt.eq(specials["eos"]).to(torch.uint8)
        if (finished >= 1).sum() == B:
            break
    if return_attention:
        return ys[:, 1:], (yv[:, 1:] if return_prob else None), yt[:, 1:]
    return ys[:, 1:], (yv[:, 1:] if return_prob else None)


def beam_search(sd: SD, cfg: dict, specials: dict, enc: Tensor, src_mask: Tensor, beam_size: int, max_output_length: int,
                alpha: float, n_best: int = 1, min_output_length: int = 1, generate_unk: bool = True,
                repetition_penalty: float = -1, no_repeat_ngram_size: int = -1, encoder_input: Optional[Tensor] = None,
                decoder_prompt: Optional[Tensor] = None, trg_prompt_mask: Optional[Tensor] = None):
    """beam_search, search.py:345-825 (Transformer branch) -> (ids [B*n_best,L], scores [B*n_best,1])"""
    bos, eos, pad, unk = specials["bos"], specials["eos"], specials["pad"], specials["unk"]
    tags = list(specials.get("lang_tags", []))
    excl = list(specials.get("all", [])) + tags
    B = src_mask.size(0)
    V = sd["decoder.output_layer.weight"].size(0)
    enc = enc.repeat_interleave(beam_size, dim=0)
    src_mask = src_mask.repeat_interleave(beam_size, dim=0)
    if encoder_input is not None:
        encoder_input = encoder_input.repeat_interleave(beam_size, dim=0)
    if decoder_prompt is not None:
        decoder_prompt = decoder_prompt.repeat_interleave(beam_size, dim=0)
    if trg_prompt_mask is not None:
        trg_prompt_mask = trg_prompt_mask.repeat_interleave(beam_size, dim=0)
    trg_mask = torch.ones(1, 1, 1, dtype=torch.bool)
    batch_offset = torch.arange(B)
    beam_offset = torch.arange(0, B * beam_size, step=beam_size)
    alive_seq = torch.full((B * beam_size, 1), bos, dtype=torch.long)
    topk_log_probs = torch.zeros(B, beam_size)
    topk_log_probs[:, 1:] = float("-inf")
    hypotheses = [[] for _ in range(B)]
    results = {"predictions": [[] for _ in range(B)], "scores": [[] for _ in range(B)]}
    is_finished = torch.zeros(B, beam_size, dtype=torch.bool)
    for step in range(max_output_length):
        rows, alive_len = alive_seq.size()
        forced_token_ids = decoder_prompt[:, step + 1] if decoder_prompt is not None and decoder_prompt.size(1) > step + 1 \
            else torch.full((rows, ), pad, dtype=torch.long)
        padding_mask = trg_prompt_mask[:, step + 1].bool() if trg_prompt_mask is not None and trg_prompt_mask.size(1) > step + 1 \
            else torch.zeros((rows, ), dtype=torch.bool)
        if torch.any(~padding_mask).item():
            logits, _, _, _ = decoder_forward(sd, cfg, alive_seq, enc, src_mask, trg_mask,
                                              trg_prompt_mask=adjust_mask_size(trg_prompt_mask, rows, alive_len))
            log_probs = F.log_softmax(logits[:, -1], dim=-1)
            if no_repeat_ngram_size > 0:
                log_probs = block_repeat_ngrams(alive_seq, log_probs, no_repeat_ngram_size, step, encoder_input, excl)
            if repetition_penalty > 1.0:
                log_probs = penalize_repetition(alive_seq, log_probs, repetition_penalty)
                if encoder_input is not None:
                    log_probs = penalize_repetition(encoder_input, log_probs, repetition_penalty)
            _forbid(log_probs, [bos, pad, specials.get("sep")] + tags)
            if not generate_unk:
                log_probs[:, unk] = float("-inf")
            if step < min_output_length:
                log_probs[:, eos] = float("-inf")
        else:
            log_probs = torch.full((rows, V), float("-inf"))
        forced_rows = padding_mask.nonzero(as_tuple=False).view(-1)
        if len(forced_rows):
            log_probs = log_probs.index_put([forced_rows, forced_token_ids[forced_rows]], torch.zeros(len(forced_rows)))
        log_probs += topk_log_probs.view(-1).unsqueeze(1)
        curr_scores = log_probs.clone()
        if alpha > 0:
            length_penalty = ((5.0 + (step + 1)) / 6.0)**alpha
            curr_scores /= length_penalty
        curr_scores = curr_scores.reshape(-1, beam_size * V)
        topk_scores, topk_ids = curr_scores.topk(beam_size, dim=-1)
        topk_log_probs = topk_scores * length_penalty if alpha > 0 else topk_scores.clone()
        topk_beam_index = topk_ids.div(V, rounding_mode="floor")
        topk_ids = topk_ids.fmod(V)
        if len(forced_rows):  # the picks themselves are overwritten too (:648-655)
            topk_ids = topk_ids.view(-1).index_put((forced_rows, ), forced_token_ids[forced_rows]).view(-1, beam_size)
            topk_scores = topk_scores.view(-1).index_put((forced_rows, ), torch.zeros(len(forced_rows))).view(-1, beam_size)
        batch_index = topk_beam_index + beam_offset[:topk_ids.size(0)].unsqueeze(1)
        select_indices = batch_index.view(-1)
        alive_seq = torch.cat([alive_seq.index_select(0, select_indices), topk_ids.view(-1, 1)], -1)
        is_finished = topk_ids.eq(eos) | is_finished | topk_scores.eq(-np.inf)
        if step + 1 == max_output_length:
            is_finished.fill_(True)
        end_condition = is_finished.all(-1)
        if is_finished.any():
            predictions = alive_seq.view(-1, beam_size, alive_seq.size(-1))
            for i in range(is_finished.size(0)):
                b = batch_offset[i].item()
                if end_condition[i]:
                    is_finished[i].fill_(True)
                for j in is_finished[i].nonzero(as_tuple=False).view(-1):
                    n_eos = (predictions[i, j, 1:] == eos).count_nonzero().item()
                    if n_eos > 1:
                        continue
                    if (n_eos == 0 and step + 1 == max_output_length) or (n_eos == 1 and predictions[i, j, -1] == eos):
                        hypotheses[b].append((topk_scores[i, j], predictions[i, j, 1:]))
                if end_condition[i]:
                    for n, (score, pred) in enumerate(sorted(hypotheses[b], key=lambda x: x[0], reverse=True)):
                        if n >= n_best:
                            break
                        results["scores"][b].append(score)
                        results["predictions"][b].append(pred)
            unfinished = end_condition.eq(False).nonzero(as_tuple=False).view(-1)
            if len(unfinished) == 0:
                break
            batch_index = batch_index.index_select(0, unfinished)
            topk_log_probs = topk_log_probs.index_select(0, unfinished)
            is_finished = is_finished.index_select(0, unfinished)
            batch_offset = batch_offset.index_select(0, unfinished)
            alive_seq = predictions.index_select(0, unfinished).view(-1, alive_seq.size(-1))
            if encoder_input is not None:
                encoder_input = encoder_input.view(-1, beam_size, encoder_input.size(1)).index_select(0, unfinished).view(-1, encoder_input.size(1))
            if decoder_prompt is not None:
                decoder_prompt = decoder_prompt.view(-1, beam_size, decoder_prompt.size(1)).index_select(0, unfinished).view(-1, decoder_prompt.size(1))
            if trg_prompt_mask is not None:
                trg_prompt_mask = trg_prompt_mask.view(-1, beam_size, trg_prompt_mask.size(1)).index_select(0, unfinished).view(-1, trg_prompt_mask.size(1))
        select_indices = batch_index.view(-1)
        enc = enc.index_select(0, select_indices)
        src_mask = src_mask.index_select(0, select_indices)
    for b in range(B):
        for _ in range(n_best - len(results["predictions"][b])):
            results["predictions"][b].append(torch.tensor([unk]).long())
            results["scores"][b].append(torch.tensor([-1]).float())
    preds = [u for r in results["predictions"] for u in r]
    max_len = max(p.shape[0] for p in preds)
    out = torch.full((len(preds), max_len), pad, dtype=torch.int64)
    for j, p in enumerate(preds):
        out[j, :p.shape[0]] = p
    scores = torch.tensor([[float(u)] for r in results["scores"] for u in r])
    return out, scores



def search_text(sd: SD, cfg: dict, specials: dict, batch: dict, max_output_length: int, beam_size: int, beam_alpha: float,
                n_best: int = 1, **kwargs):
    """search, search.py:828-912, for a text-source batch {src, src_length[, trg_input, trg_prompt_mask]}: encode once; the
    options it derives from the batch (:866-873) - source tokens for the source-side penalty / n-gram block, the forced prefix
    of a prompted batch - then greedy (beam_size < 2) or beam search -> (ids, scores)."""
    enc, src_mask = encoder_forward_text(sd, cfg, batch["src"], specials["pad"])
    if max_output_length < 0:
        max_output_length = int(max(batch["src_length"].numpy()) * 1.5)
    if kwargs.get("no_repeat_ngram_size", -1) > 1 or kwargs.get("repetition_penalty", -1) > 1:
        kwargs["encoder_input"] = batch["src"]
    if batch.get("trg_prompt_mask") is not None:
        kwargs["decoder_prompt"], kwargs["trg_prompt_mask"] = batch["trg_input"], batch["trg_prompt_mask"]
    if beam_size < 2:
        return greedy(sd, cfg, specials, enc, src_mask, max_output_length, return_prob=True, **kwargs)[:2]
    return beam_search(sd, cfg, specials, enc, src_mask, beam_size, max_output_length, beam_alpha, n_best=n_best, **kwargs)

# --------------------------------------------------------------------------------------------------
# audio front-end (reference helpers_for_audio.py, data_augmentation.py; Kaldi fbank from the published spec)
# --------------------------------------------------------------------------------------------------
def get_n_frames(wave_length: int, sample_rate: int) -> int:
    """helpers_for_audio.py:93-96"""
    duration_ms = int(wave_length / sample_rate * 1000)
    return int(1 + (duration_ms - 25) / 10)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_banks(num_bins: int = 80, n_fft: int = 512, sample_rate: float = 16000.0, low: float = 20.0, high: float = 0.0):
    """Kaldi triangular mel filters (torchaudio.compliance.kaldi.get_mel_banks): returns [num_bins, n_fft/2]."""
    nyquist = 0.5 * sample_rate
    if high <= 0.0:
        high += nyquist
    mel_low, mel_high = mel_scale(low), mel_scale(high)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
    mel = mel_scale((sample_rate / n_fft) * np.arange(n_fft // 2, dtype=np.float64))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    return np.maximum(0.0, np.minimum(up, down)).astype(np.float32)


def povey_window(n: int = 400) -> np.ndarray:
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / (n - 1)))**0.85


def fbank(waveform: np.ndarray, sample_rate: int = 16000, n_bins: int = 80) -> np.ndarray:
    """ta_kaldi.fbank(waveform * 2**15, num_mel_bins=80, sample_frequency=sr) with Kaldi defaults, as called at
    helpers_for_audio.py:30-37,54: 25 ms / 10 ms frames (snip_edges), no dither, DC removal, pre-emphasis 0.97,
    Povey window, 512-point power spectrum, 80 mel bins from 20 Hz to Nyquist, log with floor eps(float32).
    waveform: float array [N] in [-1, 1] (first channel).  Returns float32 [T, n_bins]."""
    x = np.asarray(waveform, dtype=np.float32) * np.float32(2**15)
    win_len, shift = int(sample_rate * 0.025), int(sample_rate * 0.010)
    n_fft = 1 << (win_len - 1).bit_length()
    if x.shape[0] < win_len:
        return np.zeros((0, n_bins), dtype=np.float32)
    T = 1 + (x.shape[0] - win_len) // shift
    idx = np.arange(win_len)[None, :] + shift * np.arange(T)[:, None]
    frames = x[idx].astype(np.float32)
    frames = frames - frames.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)
    frames = frames - np.float32(0.97) * prev
    frames = frames * povey_window(win_len).astype(np.float32)[None, :]
    spec = np.fft.rfft(frames.astype(np.float32), n=n_fft, axis=1)
    power = (spec.real.astype(np.float32)**2 + spec.imag.astype(np.float32)**2).astype(np.float32)
    banks = mel_banks(n_bins, n_fft, float(sample_rate))  # [n_bins, n_fft/2]; Nyquist bin has zero weight
    mel = power[:, : n_fft // 2] @ banks.T
    return np.log(np.maximum(mel, np.finfo(np.float32).eps)).astype(np.float32)


def cmvn(x: np.ndarray, norm_means: bool = True, norm_vars: bool = True) -> np.ndarray:
    """CMVN.__call__, data_augmentation.py:96-109"""
    mean = x.mean(axis=0)
    square_sums = (x**2).sum(axis=0)
    if norm_means:
        x = np.subtract(x, mean)
    if norm_vars:
        var = square_sums / x.shape[0] - mean**2
        x = np.divide(x, np.sqrt(np.maximum(var, 1e-10)))
    return x


def specaugment_params(num_frames: int, num_freqs: int, rng: np.random.RandomState, freq_mask_n=2, freq_mask_f=27,
                       time_mask_n=2, time_mask_t=40, time_mask_p=1.0):
    """RNG draw order of SpecAugment.__call__, data_augmentation.py:54-68 -> (freq masks [(f0,f)], time masks [(t0,t)])
    or None when the reference returns its input unchanged (:48-52)."""
    if num_frames == 0 or num_freqs < freq_mask_f:
        return None
    fm, tm = [], []
    for _ in range(freq_mask_n):
        f = rng.randint(0, freq_mask_f)
        f0 = rng.randint(0, num_freqs - f)
        fm.append((f0, f))
    max_t = min(time_mask_t, math.floor(num_frames * time_mask_p))
    if max_t >= 1:
        for _ in range(time_mask_n):
            t = rng.randint(0, max_t)
            t0 = rng.randint(0, num_frames - t)
            tm.append((t0, t))
    return fm, tm


def specaugment_apply(x: np.ndarray, params) -> np.ndarray:
    """Mask fill = mean of the INPUT spectrogram (data_augmentation.py:45-46,57-68)."""
    if params is None:
        return x
    out = x.copy()
    fill = x.mean()
    for f0, f in params[0]:
        if f != 0:
            out[:, f0:f0 + f] = fill
    for t0, t in params[1]:
        if t != 0:
            out[t0:t0 + t, :] = fill
    return out


def pad_features(feat_list: List[np.ndarray], embed_size: int = 80, pad_index: int = 1):
    """helpers_for_audio.py:130-170 (pads with float(pad_index) = 1.0)"""
    max_len = max(int(f.shape[0]) for f in feat_list)
    out = np.full((len(feat_list), max_len, embed_size), float(pad_index), dtype=np.float32)
    lengths = []
    for i, f in enumerate(feat_list):
        n = min(int(f.shape[0]), max_len)
        out[i, :n, :] = f[:n, :]
        lengths.append(n)
    return out, lengths, None


# --------------------------------------------------------------------------------------------------
# batch bookkeeping (reference batch.py:79-96)
# --------------------------------------------------------------------------------------------------
def make_batch(src: Tensor, src_length: Tensor, trg_full: Tensor, trg_length_full: Tensor, pad: int, eos: int) -> dict:
    trg_input = torch.where(trg_full == eos, torch.full_like(trg_full, pad), trg_full)[:, :-1]
    trg = trg_full[:, 1:]
    return {"src": src, "src_length": src_length, "trg_input": trg_input, "trg": trg, "trg_length": trg_length_full - 1,
            "trg_mask": (trg != pad).unsqueeze(1)}


def ctc_best_path(logits, in_len, blank, pad):
    """CTC best-path decoding (the standard collapse rule of Graves et al. 2006, which `nn.CTCLoss` of
    loss.py:132-168 is the training-side counterpart of): frame-wise arg-max (first maximum), merge repeats, drop blanks,
    only inside the first in_len[b] frames.  logits [B, T, V] -> (ids [B, T] pad-filled, lengths [B])."""
    logits = np.asarray(logits)
    B, T, _ = logits.shape
    ids = np.full((B, T), pad, dtype=np.int64)
    lens = np.zeros((B,), dtype=np.int64)
    for b in range(B):
        prev, n = -1, 0
        for t in range(min(int(in_len[b]), T)):
            cur = int(np.argmax(logits[b, t]))
            if cur != blank and cur != prev:
                ids[b, n] = cur
                n += 1
            prev = cur
        lens[b] = n
    return ids, lens



# --------------------------------------------------------------------------------------------------
# joint CTC / attention decoding (EXTENSION: the reference returns `ctc_out` for return_type="decode_ctc",
# model.py:162-166, and has no consumer).  The algorithm is the published one: Watanabe, Hori, Kim, Hershey, Hayashi, "Hybrid
# CTC/Attention Architecture for End-to-End Speech Recognition", IEEE JSTSP 2017, section IV-B / Algorithm 2 (CTC prefix score),
# with the forward variables r^n_t(g), r^b_t(g) of Graves' prefix search.  Pinned by enumeration: ctc_prefix_brute() below sums
# the probability of every alignment of a tiny input by brute force (tests/test_oracle_golden.py).
# --------------------------------------------------------------------------------------------------
def _collapse(path, blank):
    out, prev = [], None
    for s in path:
        if s != blank and s != prev:
            out.append(s)
        prev = s
    return out


def ctc_prefix_brute(logp: np.ndarray, prefix: List[int], blank: int, whole: bool = False) -> float:
    """log of the total probability of all length-T alignments whose collapsed labelling STARTS WITH `prefix` (whole=False: the
    CTC prefix probability psi of the paper, eq. 44-51) or IS `prefix` (whole=True: what a hypothesis ending in EOS scores).
    Exponential in T: for T <= 6, V <= 4."""
    import itertools
    T, V = logp.shape
    tot = -np.inf
    for path in itertools.product(range(V), repeat=T):
        lab = _collapse(path, blank)
        ok = lab == list(prefix) if whole else lab[:len(prefix)] == list(prefix)
        if ok:
            tot = np.logaddexp(tot, sum(logp[t, s] for t, s in enumerate(path)))
    return float(tot)


def ctc_prefix_init(logp: np.ndarray, in_len: int, blank: int) -> np.ndarray:
    """forward variables of the EMPTY prefix: r[t] = (r^n_t, r^b_t) = (log 0, sum of the blank log-probabilities up to t)"""
    T = logp.shape[0]
    r = np.full((T, 2), -np.inf)
    r[:in_len, 1] = np.cumsum(logp[:in_len, blank])
    return r


def ctc_prefix_score(logp: np.ndarray, in_len: int, y: List[int], cands: List[int], r_prev: np.ndarray, blank: int, eos: int):
    """One extension step of the CTC prefix score (Algorithm 2 of the paper): y = the hypothesis so far INCLUDING its leading BOS,
    r_prev [T, 2] = its forward variables -> (log psi [C] of y + c for every candidate c, r_new [T, 2, C])."""
    T = logp.shape[0]
    C = len(cands)
    n_out = len(y) - 1
    xs = logp[:, cands]
    r = np.full((T, 2, C), -np.inf)
    if n_out == 0 and in_len > 0:
        r[0, 0] = xs[0]
    r_sum = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
    log_phi = np.repeat(r_sum[:, None], C, axis=1)
    if n_out > 0:
        for i, c in enumerate(cands):
            if c == y[-1]:
                log_phi[:, i] = r_prev[:, 1]  # a repeated label needs a blank in between
    start = max(n_out, 1)
    # (a hypothesis with more labels than the input has frames cannot be emitted at all: every extension scores log 0)
    log_psi = r[start - 1, 0].copy() if start - 1 < in_len else np.full((C, ), -np.inf)
    for t in range(start, in_len):
        r[t, 0] = np.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
        r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + logp[t, blank]
        log_psi = np.logaddexp(log_psi, log_phi[t - 1] + xs[t])
    for i, c in enumerate(cands):
        if c == eos:
            log_psi[i] = r_sum[in_len - 1]  # the hypothesis ends here: probability of the labelling itself
        if c == blank:
            log_psi[i] = -np.inf
    return log_psi, r


def joint_ctc_beam_search(sd: SD, cfg: dict, specials: dict, enc: Tensor, src_mask: Tensor, beam_size: int, max_output_length: int,
                          alpha: float, ctc_weight: float, n_cand: int = 8, n_best: int = 1, min_output_length: int = 1,
                          generate_unk: bool = True):
    """beam_search() above with the CTC prefix score mixed into every step (EXTENSION, see the section header):
      per live hypothesis: the n_cand best next tokens by attention log-probability; for each, local score
      (1 - w) * log p_att + w * (log psi_ctc(y + c) - log psi_ctc(y)); per utterance the beam_size best of beam_size * n_cand
      by (accumulated score + local) / length penalty.  Bookkeeping of finished hypotheses as in the reference's beam search.
    -> (ids [B*n_best, L], scores [B*n_best, 1])"""
    bos, eos, pad, unk = specials["bos"], specials["eos"], specials["pad"], specials["unk"]
    B = src_mask.size(0)
    V = sd["decoder.output_layer.weight"].size(0)
    ctc_lp = F.log_softmax(linear(sd, "decoder.ctc_output_layer", enc), dim=-1).double().numpy()  # [B, T', V], blank = BOS (loss.py:156-161)
    in_len = src_mask.squeeze(1).sum(-1).tolist()
    k = beam_size
    enc = enc.repeat_interleave(k, dim=0)
    src_mask = src_mask.repeat_interleave(k, dim=0)
    trg_mask = torch.ones(1, 1, 1, dtype=torch.bool)
    alive_seq = torch.full((B * k, 1), bos, dtype=torch.long)
    topk_log_probs = torch.zeros(B, k, dtype=torch.float64)
    topk_log_probs[:, 1:] = float("-inf")
    r_state = [ctc_prefix_init(ctc_lp[b], int(in_len[b]), bos) for b in range(B) for _ in range(k)]
    ctc_prev = [0.0] * (B * k)
    hypotheses = [[] for _ in range(B)]
    results = {"predictions": [[] for _ in range(B)], "scores": [[] for _ in range(B)]}
    is_finished = torch.zeros(B, k, dtype=torch.bool)
    live = list(range(B))  # every utterance stays in the batch until the search ends (utterances do not interact)
    done = [False] * B
    for step in range(max_output_length):
        logits, _, _, _ = decoder_forward(sd, cfg, alive_seq, enc, src_mask, trg_mask)
        att = F.log_softmax(logits[:, -1], dim=-1).double()
        _forbid(att, [bos, pad, specials.get("sep")])
        if not generate_unk:
            att[:, unk] = float("-inf")
        if step < min_output_length:
            att[:, eos] = float("-inf")
        length_penalty = ((5.0 + (step + 1)) / 6.0)**alpha if alpha > 0 else 1.0
        cand_lp, cand_id = att.topk(n_cand, dim=-1)
        joint = torch.full((B * k, n_cand), float("-inf"), dtype=torch.float64)
        new_r, new_psi = [None] * (B * k), [None] * (B * k)
        for row in range(B * k):
            b = row // k
            psi, r_new = ctc_prefix_score(ctc_lp[b], int(in_len[b]), alive_seq[row].tolist(), cand_id[row].tolist(), r_state[row], bos, eos)
            new_r[row], new_psi[row] = r_new, psi
            loc = (1.0 - ctc_weight) * cand_lp[row].numpy()
            if ctc_weight > 0:  # log 0 stays log 0: an extension (or a hypothesis) CTC rules out is out, whatever the weight
                with np.errstate(invalid="ignore"):
                    inc = np.where(np.isfinite(psi) & np.isfinite(ctc_prev[row]), psi - ctc_prev[row], -np.inf)
                loc = loc + ctc_weight * inc
            loc[~np.isfinite(cand_lp[row].numpy())] = -np.inf
            joint[row] = torch.from_numpy(topk_log_probs.view(-1)[row].item() + loc)
        curr = (joint / length_penalty).reshape(B, k * n_cand)
        topk_scores, flat = curr.topk(k, dim=-1)
        topk_log_probs = topk_scores * length_penalty if alpha > 0 else topk_scores.clone()
        beam_idx = flat.div(n_cand, rounding_mode="floor")
        slot = flat.fmod(n_cand)
        rows_from = (beam_idx + torch.arange(B).unsqueeze(1) * k).view(-1)
        topk_ids = cand_id[rows_from, slot.view(-1)].view(B, k)
        r_state = [new_r[int(rf)][:, :, int(s)] for rf, s in zip(rows_from, slot.view(-1))]
        ctc_prev = [float(new_psi[int(rf)][int(s)]) for rf, s in zip(rows_from, slot.view(-1))]
        alive_seq = torch.cat([alive_seq.index_select(0, rows_from), topk_ids.view(-1, 1)], -1)
        is_finished = topk_ids.eq(eos) | is_finished | topk_scores.eq(-np.inf)
        if step + 1 == max_output_length:
            is_finished.fill_(True)
        end_condition = is_finished.all(-1)
        predictions = alive_seq.view(B, k, -1)
        for i in range(B):
            if done[i] or not is_finished[i].any():
                continue
            if end_condition[i]:
                is_finished[i].fill_(True)
            for j in is_finished[i].nonzero(as_tuple=False).view(-1):
                n_eos = (predictions[i, j, 1:] == eos).count_nonzero().item()
                if n_eos > 1:
                    continue
                if (n_eos == 0 and step + 1 == max_output_length) or (n_eos == 1 and predictions[i, j, -1] == eos):
                    hypotheses[i].append((topk_scores[i, j].item(), predictions[i, j, 1:].clone()))
            if end_condition[i]:
                for n, (score, pred) in enumerate(sorted(hypotheses[i], key=lambda x: x[0], reverse=True)):
                    if n >= n_best:
                        break
                    results["scores"][i].append(score)
                    results["predictions"][i].append(pred)
                done[i] = True
        if all(done):
            break
        enc = enc.index_select(0, rows_from)
        src_mask = src_mask.index_select(0, rows_from)
    for b in range(B):
        for _ in range(n_best - len(results["predictions"][b])):
            results["predictions"][b].append(torch.tensor([unk]).long())
            results["scores"][b].append(-1.0)
    preds = [u for r in results["predictions"] for u in r]
    max_len = max(p.shape[0] for p in preds)
    out = torch.full((len(preds), max_len), pad, dtype=torch.int64)
    for j, p in enumerate(preds):
        out[j, :p.shape[0]] = p
    scores = torch.tensor([[float(u)] for r in results["scores"] for u in r])
    return out, scores
