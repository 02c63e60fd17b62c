"""Greedy and beam search over the HIP model with the reference's call surface (joeynmt/search.py):
`search(model, batch, max_output_length, beam_size, beam_alpha, n_best, **kw)` (:828-912),
`beam_search(model, beam_size, encoder_output, encoder_hidden, src_mask, max_output_length, alpha, n_best, **kw)`
(:345-825) and `transformer_greedy(src_mask, max_output_length, model, encoder_output, encoder_hidden, **kw)`
(:162-342).  Outputs are CPU tensors / NumPy arrays exactly like the reference's.

Per step the scoring chain (log-softmax, forbidden-token masks, beam score add, length penalty, top-k over beam*V) is
ONE kernel (js2t_beam_step); the hypothesis bookkeeping (:671-755) is integer logic kept in the reference's order so
that finished / n-best handling is identical.

The decoding options of the reference (all off in the S2T configs, config.py:422,445) edit the log-softmax output
between those stages, so with any of them on the step runs as js2t_log_softmax -> edits -> js2t_beam_step_logp:
  * `no_repeat_ngram_size`: banned continuations are found on the host exactly like block_repeat_ngrams (:915-969;
    the reference walks Python lists too) and written as -inf by js2t_logp_set;
  * `repetition_penalty` (+ `encoder_input` source tokens): js2t_rep_penalty (:972-1001);
  * `decoder_prompt` / `trg_prompt_mask` (forced decoding, :489-499,603-618,648-655): the forced token's log-prob is set
    to 0 AFTER the forbidden ids were masked; the prompt mask is embedded and added to the decoder input (model.py:271-282),
    which needs the full-prefix decoder pass (no KV cache on this path);
  * `return_attention` (greedy only, :318-325): cross-attention weights of the last layer, full-prefix pass."""
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from joeys2t_amd import ops
from joeys2t_amd.helpers import adjust_mask_size, tile
from joeys2t_amd.incremental import IncrementalDecoder

SYNC_EVERY = 8  # plain greedy / beam decoding looks at the device (has every hypothesis ended?) once per this many steps
from joeys2t_amd.helpers_for_ddp import ddp_merge


class _Options:
    """Generation options of search.py:195-207,389-397 that are not part of the fused step."""

    def __init__(self, kwargs, greedy: bool):
        self.repetition_penalty = float(kwargs.get("repetition_penalty", -1))
        self.ngram = int(kwargs.get("no_repeat_ngram_size", -1))
        self.encoder_input = kwargs.get("encoder_input", None)
        self.decoder_prompt = kwargs.get("decoder_prompt", None)
        self.trg_prompt_mask = kwargs.get("trg_prompt_mask", None)
        self.return_attention = bool(kwargs.get("return_attention", False)) and greedy
        # greedy blocks for n > 1 (:248), beam search for n > 0 (:565); the penalty needs > 1.0 in both (:259,:576)
        self.block = self.ngram > (1 if greedy else 0)
        self.penalize = self.repetition_penalty > 1.0
        self.prompted = self.decoder_prompt is not None or self.trg_prompt_mask is not None
        self.edits = self.block or self.penalize or self.prompted
        self.full_prefix = self.prompted or self.return_attention


def banned_ngram_tokens(trg_tokens: List[List[int]], n: int, step: int, src_tokens: Optional[List[List[int]]],
                        exclude: List[int]):
    """block_repeat_ngrams (search.py:915-969), the search for banned continuations: -> (rows, tokens) to set to -inf."""
    rows, toks = [], []
    check_end_pos, offset = step + 2 - n, n - 1
    for h, seq in enumerate(trg_tokens):
        banned = set()
        if len(seq) > n:
            # (n-1)-token suffix; for n == 1 the reference's seq[-0:] is the WHOLE sequence, which never equals an empty
            # slice: nothing is banned - kept as is
            ngram = seq[-offset:]
            for i in range(1, check_end_pos):  # position 0 is BOS
                if ngram == seq[i:i + offset]:
                    banned.add(seq[i + offset])
            if src_tokens is not None:
                src = src_tokens[h]
                for i in range(len(src) + 1 - n):
                    if ngram == src[i:i + offset]:
                        banned.add(src[i + offset])
        for t in sorted(banned - set(exclude)):
            rows.append(h)
            toks.append(t)
    return rows, toks


def _edit_log_probs(model, opt: _Options, log_probs: Tensor, seqs: Tensor, step: int, encoder_input: Optional[Tensor]) -> Tensor:
    """n-gram blocking, then the repetition penalty on the hypothesis tokens, then on the source tokens (:564-588)."""
    if opt.block:
        src = None if encoder_input is None else encoder_input.cpu().tolist()
        rows, toks = banned_ngram_tokens(seqs.cpu().tolist(), opt.ngram, step, src, model.specials + model.lang_tags)
        if rows:
            ops.logp_set(log_probs, rows, toks, float("-inf"))
    if opt.penalize:
        ops.rep_penalty(log_probs, seqs, opt.repetition_penalty)
        if encoder_input is not None:
            ops.rep_penalty(log_probs, encoder_input, opt.repetition_penalty)
    return log_probs


def _forbidden(model, include_pad: bool, generate_unk: bool, step: int, min_output_length: int, vocab: int) -> List[int]:
    ids = [model.bos_index, model.sep_index] + list(model.lang_tags)
    if include_pad:
        ids.insert(1, model.pad_index)  # beam search also bans PAD (search.py:591); greedy does not (:288)
    if not generate_unk:
        ids.append(model.unk_index)
    if step < min_output_length:
        ids.append(model.eos_index)
    return sorted({int(i) for i in ids if i is not None and i < vocab})


def _decode_last(model, ys: Tensor, encoder_output: Tensor, src_mask: Tensor, trg_mask: Tensor,
                 trg_prompt_mask: Optional[Tensor] = None, return_attention: bool = False):
    """Logits of the newest position for every hypothesis: the decoder runs over the whole prefix (as the reference
    does, search.py:518-534) but projects only the last row onto the vocabulary.  With return_attention also the last
    layer's cross-attention weights of that position [B, src_len]."""
    with torch.no_grad():
        logits, _, att, _ = model(return_type="decode", trg_input=ys, encoder_output=encoder_output, encoder_hidden=None,
                                  src_mask=src_mask, unroll_steps=None, decoder_hidden=None, trg_mask=trg_mask, last_only=True,
                                  return_attention=return_attention,
                                  trg_prompt_mask=adjust_mask_size(trg_prompt_mask, ys.size(0), ys.size(1)))
    if return_attention:
        return logits[:, -1].contiguous(), att[:, -1, :].float()
    return logits[:, -1].contiguous()


def transformer_greedy(src_mask: Tensor, max_output_length: int, model, encoder_output: Tensor, encoder_hidden: Tensor,
                       **kwargs) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    opt = _Options(kwargs, greedy=True)
    generate_unk = kwargs.get("generate_unk", True)
    return_prob = kwargs.get("return_prob", "none") == "hyp"
    min_output_length = kwargs.get("min_output_length", 1)
    B, _, src_len = src_mask.size()
    dev = encoder_output.device
    V = model.decoder.output_size
    pad = model.pad_index
    ys = torch.full((B, 1), model.bos_index, dtype=torch.long, device=dev)
    yv = torch.zeros((B, 1), dtype=torch.float32, device=dev) if return_prob else None
    yt = torch.zeros((B, 1, src_len), dtype=torch.float32, device=dev) if opt.return_attention else None
    trg_mask = torch.ones((1, 1, 1), dtype=torch.bool, device=dev)
    finished = torch.zeros((B, 1), dtype=torch.bool, device=dev)
    zero_lp = torch.zeros((B, ), dtype=torch.float32, device=dev)
    enc_in = None if opt.encoder_input is None else opt.encoder_input.to(dev).long()
    prompt = None if opt.decoder_prompt is None else opt.decoder_prompt.to(dev).long()
    pmask = None if opt.trg_prompt_mask is None else opt.trg_prompt_mask.to(dev).long()
    pmask_host = None if pmask is None else pmask.bool().cpu()
    # the reference normalises only when something downstream reads probabilities (:204-207)
    compute_softmax = return_prob or opt.repetition_penalty > 0 or opt.ngram > 0 or enc_in is not None
    inc = None
    if kwargs.get("incremental", True) and not opt.full_prefix:
        inc = IncrementalDecoder(model, encoder_output, src_mask, 1, max_output_length)
    done_flags, n_steps = [], None
    for step in range(max_output_length):
        has_col = prompt is not None and prompt.size(1) > step + 1
        forced_word = prompt[:, step + 1:step + 2] if has_col else None
        has_mask = pmask is not None and pmask.size(1) > step + 1
        forced_mask = pmask[:, step + 1:step + 2].bool() if has_mask else None
        all_forced = has_mask and bool(pmask_host[:, step + 1].all())
        att = None
        if not all_forced:
            if inc is not None:
                logits = inc.step(ys[:, -1])
            elif opt.return_attention:
                logits, att = _decode_last(model, ys, encoder_output, src_mask, trg_mask, pmask, True)
            else:
                logits = _decode_last(model, ys, encoder_output, src_mask, trg_mask, pmask)
            forbid = _forbidden(model, False, generate_unk, step, min_output_length, V)
            if opt.edits and compute_softmax:
                log_probs = _edit_log_probs(model, opt, ops.log_softmax(logits), ys, step, enc_in)
                scores, ids, _ = ops.beam_step(log_probs, zero_lp, B, 1, forbid, 0.0, normalized=True)
            else:
                # arg-max of the masked row == top-1 of a beam of one; scores are log-probs (without compute_softmax the
                # reference takes the arg-max of the raw logits, :257-258 - the same token, and no score is returned)
                scores, ids, _ = ops.beam_step(logits, zero_lp, B, 1, forbid, 0.0)
            if forced_mask is not None:
                fw = forced_word if forced_word is not None else torch.full_like(ids, pad)
                ids = torch.where(forced_mask, fw, ids)
                scores = torch.where(forced_mask, torch.zeros_like(scores), scores)
                if att is not None:
                    att = torch.where(forced_mask.expand(-1, src_len), torch.zeros_like(att), att)
        else:
            ids = forced_word if forced_word is not None else torch.full((B, 1), pad, dtype=torch.long, device=dev)
            scores = torch.zeros((B, 1), dtype=torch.float32, device=dev)
            att = torch.zeros((B, src_len), dtype=torch.float32, device=dev) if opt.return_attention else None
        ys = torch.cat([ys, ids], dim=1)
        if return_prob:
            yv = torch.cat([yv, scores], dim=1)
        if opt.return_attention:
            yt = torch.cat([yt, att.unsqueeze(1)], dim=1)
        finished |= ids.eq(model.eos_index)
        # the reference stops at the first step at which every hypothesis has emitted EOS (:332-334) - a device -> host round
        # trip per step.  Here the flag of every step is kept on the device and looked at every SYNC_EVERY steps; the steps
        # that ran past the stopping point are cut off below, so the result is the reference's, column for column.
        done_flags.append(finished.all().view(1))
        if (step + 1) % SYNC_EVERY == 0 or step + 1 == max_output_length:
            flags = torch.cat(done_flags).cpu()
            if bool(flags.any()):
                n_steps = int(torch.nonzero(flags)[0]) + 1
                break
    if n_steps is not None:  # drop what was generated after the stopping point
        ys = ys[:, :n_steps + 1]
        yv = yv[:, :n_steps + 1] if return_prob else None
        yt = yt[:, :n_steps + 1] if opt.return_attention else None
    ys = ddp_merge(ys, model.pad_index)
    yv = ddp_merge(yv, 0.0) if return_prob else None
    yt = ddp_merge(yt, 0.0) if opt.return_attention else None
    output = ys[:, 1:].detach().cpu().long()
    scores = yv[:, 1:].detach().cpu().float() if return_prob else None
    attention = yt[:, 1:, :].detach().cpu().float() if opt.return_attention else None
    return output, scores, attention


def beam_search(model, beam_size: int, encoder_output: Tensor, encoder_hidden: Tensor, src_mask: Tensor,
                max_output_length: int, alpha: float, n_best: int = 1, **kwargs) -> Tuple[Tensor, Optional[Tensor], None]:
    assert beam_size > 0, "Beam size must be >0."
    assert n_best <= beam_size, f"Can only return {beam_size} best hypotheses."
    opt = _Options(kwargs, greedy=False)
    bos, eos, pad, unk = model.bos_index, model.eos_index, model.pad_index, model.unk_index
    generate_unk = kwargs.get("generate_unk", True)
    return_prob = kwargs.get("return_prob", "none") == "hyp"
    min_output_length = kwargs.get("min_output_length", 1)
    B = src_mask.size(0)
    V = model.decoder.output_size
    dev = encoder_output.device

    # KV-cached decoding (default): hypotheses index their utterance's encoder keys / values instead of carrying a tiled
    # copy of the encoder states; incremental=False keeps the reference's full-prefix pass (used to cross-check)
    inc = None
    if kwargs.get("incremental", True) and not opt.full_prefix:
        inc = IncrementalDecoder(model, encoder_output, src_mask, beam_size, max_output_length)
    if inc is None:
        encoder_output = tile(encoder_output.contiguous(), beam_size, dim=0)  # [B*k, S, d]
        src_mask = tile(src_mask, beam_size, dim=0)
    # per-hypothesis copies of the option tensors, filtered with the live examples below (:441-458,764-781)
    enc_in = None if opt.encoder_input is None else tile(opt.encoder_input.to(dev).long().contiguous(), beam_size, dim=0).view(B * beam_size, -1)
    prompt = None if opt.decoder_prompt is None else tile(opt.decoder_prompt.to(dev).long().contiguous(), beam_size, dim=0).view(B * beam_size, -1)
    pmask = None if opt.trg_prompt_mask is None else tile(opt.trg_prompt_mask.to(dev).long().contiguous(), beam_size, dim=0).view(B * beam_size, -1)
    if pmask is not None:
        assert prompt is not None and prompt.size(1) == pmask.size(1)
    trg_mask = torch.ones((1, 1, 1), dtype=torch.bool, device=dev)
    batch_offset = torch.arange(B, dtype=torch.long)  # host: live example -> original position
    beam_offset = torch.arange(0, B * beam_size, step=beam_size, dtype=torch.long, device=dev)
    alive_seq = torch.full((B * beam_size, 1), bos, dtype=torch.long, device=dev)
    topk_log_probs = torch.zeros((B, beam_size), device=dev)
    topk_log_probs[:, 1:] = float("-inf")  # only the first beam is live at step 0 (search.py:477-479)
    hypotheses = [[] for _ in range(B)]
    results = {"predictions": [[] for _ in range(B)], "scores": [[] for _ in range(B)]}
    is_finished = torch.zeros((B, beam_size), dtype=torch.bool, device=dev)
    # plain decoding (no option that edits the scores on the host) with the key/value cache: the search runs without a
    # device -> host round trip per step, see `if fast:` below
    fast = inc is not None and not opt.edits and kwargs.get("sync_free", True)
    # EXTENSION (the reference has no consumer for its CTC head at decoding time, model.py:162-166): joint CTC /
    # attention decoding after Watanabe et al. 2017.  ctc_weight w > 0: per live hypothesis the `ctc_candidates` best next tokens by
    # attention log-probability, each scored (1 - w) * log p_att + w * (CTC prefix score of the extension - of the hypothesis);
    # the beam keeps the best beam_size of beam_size * ctc_candidates per utterance.  w = 0: the reference's beam search, untouched.
    ctc_w = float(kwargs.get("ctc_weight", 0.0) or 0.0)
    n_cand = 0
    if ctc_w > 0.0:
        if not fast:
            raise ops.Js2tError("beam_search: ctc_weight needs the key/value-cached search without score-editing options")
        if getattr(model.decoder, "ctc_output_layer", None) is None:
            raise ops.Js2tError("beam_search: ctc_weight > 0 but the model has no CTC output layer")
        n_cand = int(kwargs.get("ctc_candidates", 8))
        if not (beam_size <= n_cand <= min(8, V)):
            raise ops.Js2tError("beam_search: ctc_candidates must lie in beam_size .. min(8, vocabulary)")
        with torch.no_grad():
            ctc_lp = ops.log_softmax(model.decoder.project(model.decoder.ctc_output_layer, encoder_output, model.runtime.compute_dtype).float())
        ctc_in_len = src_mask.view(B, -1).sum(-1).long()
        ctc_r = ops.ctc_prefix_init(ctc_lp, ctc_in_len, beam_size, bos)  # blank = BOS (loss.py:156-161)
        ctc_psi = torch.zeros((B * beam_size, ), dtype=torch.float32, device=dev)
    if fast:
        # what a step leaves behind is its picks (token + the beam it extends), its flags and its scores: O(L * B * k); the
        # hypotheses themselves are rebuilt on the host from the back-pointers, only for those that get filed
        tok_all = torch.zeros((max_output_length, B, beam_size), dtype=torch.long, device=dev)
        parent_all = torch.zeros((max_output_length, B, beam_size), dtype=torch.long, device=dev)
        fin_all = torch.zeros((max_output_length, B, beam_size), dtype=torch.bool, device=dev)
        score_all = torch.zeros((max_output_length, B, beam_size), dtype=torch.float32, device=dev)
        ended = torch.zeros((B, ), dtype=torch.bool, device=dev)
        n_run = 0

    for step in range(max_output_length):
        nb = alive_seq.size(0) // beam_size
        rows = nb * beam_size
        forbid = _forbidden(model, True, generate_unk, step, min_output_length, V)
        length_penalty = ((5.0 + (step + 1)) / 6.0)**alpha if alpha > 0 else 0.0
        forced_rows = None
        if not opt.edits:
            logits = inc.step(alive_seq[:, -1]) if inc is not None else _decode_last(model, alive_seq, encoder_output, src_mask, trg_mask)
            if ctc_w > 0.0:
                cand_lp, cand_id, _ = ops.beam_pick(logits, n_cand, forbid)
                local, psi_new, r_new = ops.ctc_prefix_step(ctc_lp, ctc_in_len, ctc_r, alive_seq[:, -1].contiguous(), cand_id, cand_lp, ctc_psi,
                                                           step, beam_size, bos, eos, ctc_w)
                topk_scores, topk_flat, _ = ops.beam_step(local, topk_log_probs.reshape(-1), nb, beam_size, [], length_penalty, normalized=True)
            else:
                topk_scores, topk_flat, _ = ops.beam_step(logits, topk_log_probs.reshape(-1), nb, beam_size, forbid, length_penalty)
        else:
            # forced tokens of this step (:489-499): hypotheses whose prompt mask is set at position step + 1
            has_mask = pmask is not None and pmask.size(1) > step + 1
            padding_mask = pmask[:, step + 1].bool() if has_mask else torch.zeros((rows, ), dtype=torch.bool, device=dev)
            forced_tok = prompt[:, step + 1] if (prompt is not None and prompt.size(1) > step + 1) else torch.full((rows, ), pad, dtype=torch.long, device=dev)
            pm_host = padding_mask.cpu()
            if not bool(pm_host.all()):
                if inc is not None:
                    logits = inc.step(alive_seq[:, -1])
                else:
                    logits = _decode_last(model, alive_seq, encoder_output, src_mask, trg_mask, pmask)
                log_probs = _edit_log_probs(model, opt, ops.log_softmax(logits), alive_seq, step, enc_in)
                if forbid:  # masked BEFORE the forced overwrite: a forced SEP / tag must survive (:590-618)
                    ops.logp_set(log_probs, [r for r in range(rows) for _ in forbid], forbid * rows, float("-inf"))
            else:
                log_probs = torch.full((rows, V), float("-inf"), dtype=torch.float32, device=dev)  # dummy (:607-611)
            if bool(pm_host.any()):
                forced_rows = pm_host.nonzero(as_tuple=False).view(-1).to(dev)
                ops.logp_set(log_probs, forced_rows, forced_tok.index_select(0, forced_rows), 0.0)
            topk_scores, topk_flat, _ = ops.beam_step(log_probs, topk_log_probs.reshape(-1), nb, beam_size, [], length_penalty,
                                                      normalized=True)
        topk_log_probs = topk_scores * length_penalty if alpha > 0 else topk_scores.clone()
        if ctc_w > 0.0:  # flat index over beam * candidates: which hypothesis, which of its candidates
            topk_beam_index = topk_flat.div(n_cand, rounding_mode="floor")
            pair = ((topk_beam_index + beam_offset[:nb].unsqueeze(1)) * n_cand + topk_flat.fmod(n_cand)).view(-1)
            topk_ids = cand_id.view(-1).index_select(0, pair).view(nb, beam_size)
            ctc_r = r_new.view(rows * n_cand, -1).index_select(0, pair).view(rows, -1, 2)  # the winners' forward variables
            ctc_psi = psi_new.view(-1).index_select(0, pair)
        else:
            topk_beam_index = topk_flat.div(V, rounding_mode="floor")
            topk_ids = topk_flat.fmod(V)
        if forced_rows is not None:
            # forced decoding overwrites the picks themselves as well (:648-655): flat position r of [nb, k] <- hypothesis r
            topk_ids = topk_ids.view(-1).index_put((forced_rows, ), forced_tok.index_select(0, forced_rows)).view(-1, beam_size)
            topk_scores = topk_scores.view(-1).index_put((forced_rows, ), torch.zeros_like(forced_rows, dtype=topk_scores.dtype)).view(-1, beam_size)
        batch_index = topk_beam_index + beam_offset[:nb].unsqueeze(1)
        select_indices = batch_index.view(-1)
        if fast:  # only the newest token is read on the device (the decoder keeps its own key / value history)
            alive_seq = topk_ids.view(-1, 1)
        else:
            alive_seq = torch.cat([alive_seq.index_select(0, select_indices), topk_ids.view(-1, 1)], -1)
        is_finished = topk_ids.eq(eos) | is_finished | topk_scores.eq(float("-inf"))
        if step + 1 == max_output_length:
            is_finished.fill_(True)
        end_condition = is_finished.all(-1)

        if fast:
            # No look at the device in this step.  The reference removes an utterance from the batch once all its beams have
            # finished (:757-781) and files finished hypotheses as they appear (:683-717); both only READ the step's tensors, and
            # utterances do not interact - so every utterance stays in the batch (one captured step graph for the whole search),
            # the step's tensors are kept, and the filing happens once, after the loop, exactly as it would have step by step.
            # Whether everything has ended is asked every SYNC_EVERY steps.
            tok_all[step], parent_all[step] = topk_ids, topk_beam_index
            fin_all[step], score_all[step] = is_finished, topk_scores
            ended |= end_condition
            n_run = step + 1
            if (step + 1) % SYNC_EVERY == 0 and bool(ended.all()):
                break
        elif bool(is_finished.any()):
            # one device->host transfer per step for the bookkeeping below (the reference syncs per hypothesis, :683-717)
            # and the tests of :683-717 for all (utterance, beam) pairs at once: with a trained model some hypothesis ends at
            # most steps, and a Python loop of small tensor operations per pair (~1 ms per step at 32 x 5) costs more than the
            # replayed decoder step itself
            fin_h, end_h = is_finished.cpu(), end_condition.cpu()
            pred_h = alive_seq.view(-1, beam_size, alive_seq.size(-1)).cpu()
            score_h = topk_scores.cpu()
            fin_h[end_h] = True
            pred_np = pred_h.numpy()
            n_eos = (pred_np[:, :, 1:] == eos).sum(-1)  # 0: still open, 1: ends here or ended earlier, > 1: collected earlier
            take = fin_h.numpy() & (((n_eos == 0) & (step + 1 == max_output_length)) | ((n_eos == 1) & (pred_np[:, :, -1] == eos)))
            offs = batch_offset.tolist()
            for i, j in zip(*np.nonzero(take)):  # row-major: utterances ascending, beams ascending, as the reference's loops
                hypotheses[offs[i]].append((score_h[i, j], pred_h[i, j, 1:]))
            for i in np.nonzero(end_h.numpy())[0]:
                b = offs[i]
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
            if enc_in is not None:
                enc_in = enc_in.view(-1, beam_size, enc_in.size(1)).index_select(0, unfinished).view(-1, enc_in.size(1))
            if prompt is not None:
                prompt = prompt.view(-1, beam_size, prompt.size(1)).index_select(0, unfinished).view(-1, prompt.size(1))
            if pmask is not None:
                pmask = pmask.view(-1, beam_size, pmask.size(1)).index_select(0, unfinished).view(-1, pmask.size(1))

        select_indices = batch_index.view(-1)
        if inc is not None:
            inc.reorder(select_indices)
        else:
            encoder_output = encoder_output.index_select(0, select_indices)
            src_mask = src_mask.index_select(0, select_indices)

    if fast:
        # the filing the reference does inside the loop (:683-717,757-781), step by step over what the steps left behind
        tok_h, parent_h = tok_all[:n_run].cpu().numpy(), parent_all[:n_run].cpu().numpy()
        fin_h_all, score_h_all = fin_all[:n_run].cpu().numpy(), score_all[:n_run].cpu()

        def backtrace(step, i, j):  # hypothesis (i, j) of `step` without BOS, through the beams it extended
            out = np.empty((step + 1, ), dtype=np.int64)
            for s in range(step, -1, -1):
                out[s] = tok_h[s, i, j]
                j = parent_h[s, i, j]
            return torch.from_numpy(out)

        live = np.ones((B, ), dtype=bool)
        n_eos = np.zeros((B, beam_size), dtype=np.int64)  # EOS tokens in each live hypothesis so far
        for step in range(n_run):
            tok = tok_h[step]
            n_eos = np.take_along_axis(n_eos, parent_h[step], 1) + (tok == eos)
            fin_np = fin_h_all[step]
            if not (fin_np & live[:, None]).any():
                continue
            end_np = fin_np.all(-1)
            # n_eos 0: still open, 1: ends here or ended earlier, > 1: collected earlier
            take = fin_np & live[:, None] & (((n_eos == 0) & (step + 1 == max_output_length)) | ((n_eos == 1) & (tok == eos)))
            for i, j in zip(*np.nonzero(take)):  # row-major: utterances ascending, beams ascending, as the reference's loops
                hypotheses[i].append((score_h_all[step, i, j], backtrace(step, i, j)))
            for i in np.nonzero(end_np & live)[0]:
                for n, (score, pred) in enumerate(sorted(hypotheses[i], key=lambda x: x[0], reverse=True)):
                    if n >= n_best:
                        break
                    results["scores"][i].append(score)
                    results["predictions"][i].append(pred)
            live &= ~end_np
            if not live.any():
                break
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
    # options the reference derives from the batch (:866-873): source tokens for the source-side penalty / n-gram block,
    # the forced prefix of a prompted batch
    if kwargs.get("no_repeat_ngram_size", -1) > 1 or kwargs.get("repetition_penalty", -1) > 1:
        if batch.src.is_floating_point():
            # the reference hands the float features to a gather as indices and fails inside torch; say why instead
            raise ValueError("repetition_penalty / no_repeat_ngram_size read source TOKENS (search.py:866-870); "
                             "this batch carries speech features")
        kwargs["encoder_input"] = batch.src
    if batch.has_trg and batch.trg_prompt_mask is not None:
        kwargs["decoder_prompt"] = batch.trg_input
        kwargs["trg_prompt_mask"] = batch.trg_prompt_mask
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

