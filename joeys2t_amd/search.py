"""Greedy and beam search over the HIP model with the reference's call surface (joeynmt/search.py):
`search(model, batch, max_output_length, beam_size, beam_alpha, n_best, **kw)` (:828-912),
`beam_search(model, beam_size, encoder_output, encoder_hidden, src_mask, max_output_length, alpha, n_best, **kw)`
(:345-825) and `transformer_greedy(src_mask, max_output_length, model, encoder_output, encoder_hidden, **kw)`
(:162-342).  Outputs are CPU tensors / NumPy arrays exactly like the reference's.

Per step the scoring chain (log-softmax, forbidden-token masks, beam score add, length penalty, top-k over beam*V) is
ONE kernel (js2t_beam_step); the hypothesis bookkeeping (:671-755) is integer logic kept in the reference's order so
that finished / n-best handling is identical.  Unsupported reference options (forced decoding prompts, repetition
penalty, n-gram blocking: all off in the S2T configs, config.py:422,445) raise."""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from joeys2t_amd import ops
from joeys2t_amd.helpers import tile
from joeys2t_amd.incremental import IncrementalDecoder
from joeys2t_amd.helpers_for_ddp import ddp_merge


def _check_unsupported(kwargs):
    if kwargs.get("repetition_penalty", -1) > 1.0 or kwargs.get("no_repeat_ngram_size", -1) > 0:
        raise NotImplementedError("repetition_penalty / no_repeat_ngram_size are not on the HIP decoding path")
    if kwargs.get("decoder_prompt") is not None or kwargs.get("trg_prompt_mask") is not None:
        raise NotImplementedError("forced (prompt) decoding is not on the HIP decoding path")


def _forbidden(model, include_pad: bool, generate_unk: bool, step: int, min_output_length: int, vocab: int) -> List[int]:
    ids = [model.bos_index, model.sep_index] + list(model.lang_tags)
    if include_pad:
        ids.insert(1, model.pad_index)  # beam search also bans PAD (search.py:591); greedy does not (:288)
    if not generate_unk:
        ids.append(model.unk_index)
    if step < min_output_length:
        ids.append(model.eos_index)
    return sorted({int(i) for i in ids if i is not None and i < vocab})


def _decode_last(model, ys: Tensor, encoder_output: Tensor, src_mask: Tensor, trg_mask: Tensor) -> Tensor:
    """Logits of the newest position for every hypothesis: the decoder runs over the whole prefix (as the reference
    does, search.py:518-534) but projects only the last row onto the vocabulary."""
    with torch.no_grad():
        logits, _, _, _ = model(return_type="decode", trg_input=ys, encoder_output=encoder_output, encoder_hidden=None,
                                src_mask=src_mask, unroll_steps=None, decoder_hidden=None, trg_mask=trg_mask, last_only=True)
    return logits[:, -1].contiguous()


def transformer_greedy(src_mask: Tensor, max_output_length: int, model, encoder_output: Tensor, encoder_hidden: Tensor,
                       **kwargs) -> Tuple[Tensor, Optional[Tensor], None]:
    _check_unsupported(kwargs)
    generate_unk = kwargs.get("generate_unk", True)
    return_prob = kwargs.get("return_prob", "none") == "hyp"
    min_output_length = kwargs.get("min_output_length", 1)
    if kwargs.get("return_attention", False):
        raise NotImplementedError("attention scores are not exported by the HIP greedy search")
    B = src_mask.size(0)
    dev = encoder_output.device
    V = model.decoder.output_size
    ys = torch.full((B, 1), model.bos_index, dtype=torch.long, device=dev)
    yv = torch.zeros((B, 1), dtype=torch.float32, device=dev) if return_prob else None
    trg_mask = torch.ones((1, 1, 1), dtype=torch.bool, device=dev)
    finished = torch.zeros((B, 1), dtype=torch.bool, device=dev)
    zero_lp = torch.zeros((B, ), dtype=torch.float32, device=dev)
    inc = IncrementalDecoder(model, encoder_output, src_mask, 1, max_output_length) if kwargs.get("incremental", True) else None
    for step in range(max_output_length):
        logits = inc.step(ys[:, -1]) if inc is not None else _decode_last(model, ys, encoder_output, src_mask, trg_mask)
        forbid = _forbidden(model, False, generate_unk, step, min_output_length, V)
        # arg-max of the masked row == top-1 of a beam of one; scores are log-probs (the reference only normalises
        # when probabilities are requested, search.py:257-258, which does not change the arg-max)
        scores, ids, _ = ops.beam_step(logits, zero_lp, B, 1, forbid, 0.0)
        ys = torch.cat([ys, ids], dim=1)
        if return_prob:
            yv = torch.cat([yv, scores], dim=1)
        finished |= ids.eq(model.eos_index)
        if bool(finished.all()):
            break
    ys = ddp_merge(ys, model.pad_index)
    yv = ddp_merge(yv, 0.0) if return_prob else None
    output = ys[:, 1:].detach().cpu().long()
    scores = yv[:, 1:].detach().cpu().float() if return_prob else None
    return output, scores, None


def beam_search(model, beam_size: int, encoder_output: Tensor, encoder_hidden: Tensor, src_mask: Tensor,
                max_output_length: int, alpha: float, n_best: int = 1, **kwargs) -> Tuple[Tensor, Optional[Tensor], None]:
    assert beam_size > 0, "Beam size must be >0."
    assert n_best <= beam_size, f"Can only return {beam_size} best hypotheses."
    _check_unsupported(kwargs)
    bos, eos, pad, unk = model.bos_index, model.eos_index, model.pad_index, model.unk_index
    generate_unk = kwargs.get("generate_unk", True)
    return_prob = kwargs.get("return_prob", "none") == "hyp"
    min_output_length = kwargs.get("min_output_length", 1)
    B = src_mask.size(0)
    V = model.decoder.output_size
    dev = encoder_output.device

    # KV-cached decoding (default): hypotheses index their utterance's encoder keys / values instead of carrying a tiled
    # copy of the encoder states; incremental=False keeps the reference's full-prefix pass (used to cross-check)
    inc = IncrementalDecoder(model, encoder_output, src_mask, beam_size, max_output_length) if kwargs.get("incremental", True) else None
    if inc is None:
        encoder_output = tile(encoder_output.contiguous(), beam_size, dim=0)  # [B*k, S, d]
        src_mask = tile(src_mask, beam_size, dim=0)
    trg_mask = torch.ones((1, 1, 1), dtype=torch.bool, device=dev)
    batch_offset = torch.arange(B, dtype=torch.long)  # host: live example -> original position
    beam_offset = torch.arange(0, B * beam_size, step=beam_size, dtype=torch.long, device=dev)
    alive_seq = torch.full((B * beam_size, 1), bos, dtype=torch.long, device=dev)
    topk_log_probs = torch.zeros((B, beam_size), device=dev)
    topk_log_probs[:, 1:] = float("-inf")  # only the first beam is live at step 0 (search.py:477-479)
    hypotheses = [[] for _ in range(B)]
    results = {"predictions": [[] for _ in range(B)], "scores": [[] for _ in range(B)]}
    is_finished = torch.zeros((B, beam_size), dtype=torch.bool, device=dev)

    for step in range(max_output_length):
        nb = alive_seq.size(0) // beam_size
        logits = inc.step(alive_seq[:, -1]) if inc is not None else _decode_last(model, alive_seq, encoder_output, src_mask, trg_mask)
        forbid = _forbidden(model, True, generate_unk, step, min_output_length, V)
        length_penalty = ((5.0 + (step + 1)) / 6.0)**alpha if alpha > 0 else 0.0
        topk_scores, topk_flat, _ = ops.beam_step(logits, topk_log_probs.reshape(-1), nb, beam_size, forbid, length_penalty)
        topk_log_probs = topk_scores * length_penalty if alpha > 0 else topk_scores.clone()
        topk_beam_index = topk_flat.div(V, rounding_mode="floor")
        topk_ids = topk_flat.fmod(V)
        batch_index = topk_beam_index + beam_offset[:nb].unsqueeze(1)
        select_indices = batch_index.view(-1)
        alive_seq = torch.cat([alive_seq.index_select(0, select_indices), topk_ids.view(-1, 1)], -1)
        is_finished = topk_ids.eq(eos) | is_finished | topk_scores.eq(float("-inf"))
        if step + 1 == max_output_length:
            is_finished.fill_(True)
        end_condition = is_finished.all(-1)

        if bool(is_finished.any()):
            # one device->host transfer per step for the bookkeeping below (the reference syncs per hypothesis, :683-717)
            fin_h, end_h = is_finished.cpu(), end_condition.cpu()
            pred_h = alive_seq.view(-1, beam_size, alive_seq.size(-1)).cpu()
            score_h = topk_scores.cpu()
            for i in range(fin_h.size(0)):
                b = int(batch_offset[i])
                if end_h[i]:
                    fin_h[i].fill_(True)
                for j in fin_h[i].nonzero(as_tuple=False).view(-1).tolist():
                    n_eos = int((pred_h[i, j, 1:] == eos).count_nonzero())
                    if n_eos > 1:
                        continue  # already collected at an earlier step
                    if (n_eos == 0 and step + 1 == max_output_length) or (n_eos == 1 and pred_h[i, j, -1] == eos):
                        hypotheses[b].append((score_h[i, j], pred_h[i, j, 1:]))
                if end_h[i]:
                    for n, (score, pred) in enumerate(sorted(hypotheses[b], key=lambda x: x[0], reverse=True)):
                        if n >= n_best:
                            break
                        results["scores"][b].append(score)
                        results["predictions"][b].append(pred)
            unfinished_h = end_h.eq(False).nonzero(as_tuple=False).view(-1)
            if len(unfinished_h) == 0:
                break
            unfinished = unfinished_h.to(dev)
            is_finished = fin_h.to(dev).index_select(0, unfinished)
            batch_index = batch_index.index_select(0, unfinished)
            topk_log_probs = topk_log_probs.index_select(0, unfinished)
            batch_offset = batch_offset.index_select(0, unfinished_h)
            alive_seq = alive_seq.view(-1, beam_size, alive_seq.size(-1)).index_select(0, unfinished).view(-1, alive_seq.size(-1))

        select_indices = batch_index.view(-1)
        if inc is not None:
            inc.reorder(select_indices)
        else:
            encoder_output = encoder_output.index_select(0, select_indices)
            src_mask = src_mask.index_select(0, select_indices)

    for b in range(B):
        for _ in range(n_best - len(results["predictions"][b])):
            results["predictions"][b].append(torch.tensor([unk]).long())
            results["scores"][b].append(torch.tensor([-1]).float())
    preds = [u for r in results["predictions"] for u in r]
    max_len = max(p.shape[0] for p in preds)
    final_outputs = torch.full((len(preds), max_len), pad, dtype=torch.int64)
    for j, p in enumerate(preds):
        final_outputs[j, :p.shape[0]] = p
    scores = torch.tensor([[float(u)] for r in results["scores"] for u in r]) if return_prob else None
    assert final_outputs.shape[0] == B * n_best
    return final_outputs, scores, None


def search(model, batch, max_output_length: int, beam_size: int, beam_alpha: float, n_best: int = 1, **kwargs):
    """Encode once, then greedy (beam_size < 2) or beam search; returns NumPy arrays (reference :828-912)."""
    kwargs.pop("autocast", None)
    with torch.no_grad():
        encoder_output, encoder_hidden, src_mask, _ = model(return_type="encode", **vars(batch))
    src_mask = src_mask if batch.src_mask is None else batch.src_mask
    if max_output_length < 0:
        # un-subsampled frame count * 1.5, as the reference computes it (:863-864)
        max_output_length = int(max(batch.src_length.cpu().numpy()) * 1.5)
    if beam_size < 2:
        out, scores, att = transformer_greedy(src_mask=src_mask, max_output_length=max_output_length, model=model,
                                              encoder_output=encoder_output, encoder_hidden=encoder_hidden, **kwargs)
    else:
        out, scores, att = beam_search(model=model, beam_size=beam_size, encoder_output=encoder_output,
                                       encoder_hidden=encoder_hidden, src_mask=src_mask,
                                       max_output_length=max_output_length, alpha=beam_alpha, n_best=n_best, **kwargs)

    def _np(t):
        return t.detach().cpu().numpy() if torch.is_tensor(t) else t

    return _np(out), _np(scores), _np(att)


def ctc_greedy(model, batch):
    """CTC best-path decoding of the encoder-side output layer (the reference's `decode_ctc` return type hands
    out ctc_out, model.py:162-166, without a consumer): encode once, project with `decoder.ctc_output_layer`, take the
    frame-wise arg-max inside each utterance's sub-sampled length (model.py:125), merge repeats, drop blanks
    (blank = BOS, model.py:84).  Returns NumPy (ids i64[B, T'] pad-filled, lengths i64[B])."""
    from joeys2t_amd import ops
    layer = getattr(model.decoder, "ctc_output_layer", None)
    if layer is None:
        raise ValueError("ctc_greedy: the model has no CTC output layer (loss: crossentropy-ctc)")
    with torch.no_grad():
        encoder_output, _, src_mask, _ = model(return_type="encode", **vars(batch))
        ctc_out = model.decoder.project(layer, encoder_output, model.runtime.compute_dtype)  # [B, T', V]
        B, T, V = ctc_out.shape
        _, best = ops.row_lse(ctc_out.reshape(B * T, V), want_argmax=True)
        in_len = src_mask.squeeze(1).sum(dim=1)
        ids, n = ops.ctc_collapse(best.view(B, T), in_len, model.bos_index, model.pad_index)
    return ids.cpu().numpy(), n.cpu().numpy()

