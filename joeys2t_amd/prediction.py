"""Decode / validation driver: the numerically relevant part of the reference's `predict` batch loop
(joeynmt/prediction.py:154-245): sort by source length -> [validation loss | reference scoring] -> search -> un-sort (n-best
expanded) -> ids to tokens (cut at EOS).  Dataset plumbing, logging, BLEU/chrF and checkpoint handling are out of scope.

The validation leg (prediction.py:165-200): the model runs "as during training" under no_grad, the batch loss, the number of
correct tokens and the token count are summed over ranks (`ddp_reduce`) and over batches; with `return_prob="ref"` no search runs -
the log-probabilities of the reference tokens are looked up instead (`ddp_merge` of log-probs and targets, `Batch.score`).
Reference quirks, stated rather than copied:
  * its loop asks the model for return_type="loss" also when it goes on to score the references - that 4-tuple carries the loss
    components in slots 1-2, not log-probabilities (model.py:133-150), so `Batch.score` cannot have worked there; this driver asks
    for "loss_probs" when it needs log-probabilities.  tests/golden/predict_loss.npz holds both calls' outputs of the reference model.
  * it computes the normaliser of the validation loss and then drops it: its valid_scores keep loss / acc / ppl at NaN.  `totals`
    below are what it accumulates; `valid_scores` are upstream JoeyNMT's formulas over them (flagged: not this reference's output)."""
import math
from typing import Dict, Iterable

import numpy as np
import torch

from joeys2t_amd.batch import Batch
from joeys2t_amd.helpers import expand_reverse_index
from joeys2t_amd.helpers_for_ddp import ddp_merge, ddp_reduce, use_ddp
from joeys2t_amd.search import search


def predict(model, batches: Iterable[Batch], *, beam_size: int = 1, beam_alpha: float = -1.0, n_best: int = 1,
            max_output_length: int = -1, min_output_length: int = 1, generate_unk: bool = True,
            return_prob: str = "none", repetition_penalty: float = -1, no_repeat_ngram_size: int = -1,
            return_attention: bool = False, compute_loss: bool = False, normalization: str = "batch", n_gpu: int = 1):
    """Returns (id arrays in the ORIGINAL batch order, decoded token lists, scores or None); with `return_attention` a fourth
    element: the attention arrays of greedy search (prediction.py:205-218 hands the same options to `search`, which derives
    `encoder_input` / the forced prefix from the batch); with `compute_loss` a last element, the validation record
    {"totals": {loss, n_correct, ntokens, nseqs}, "normalizer", "valid_scores": {loss, acc, ppl}}.
    return_prob="ref": ids are the reference tokens and scores their log-probabilities (one array per sentence), no search.
    Under a process group every rank passes ITS batches; totals are sums over ranks, and rank 0's outputs hold all ranks'
    sentences in dataset order (`batch.indices`), as in the reference (prediction.py:222-231, 250-257)."""
    model.eval()
    all_ids, all_scores, all_att, by_index = [], [], [], {}
    totals: Dict[str, float] = {"loss": 0.0, "n_correct": 0, "ntokens": 0, "nseqs": 0}
    ddp = use_ddp()
    for batch in batches:
        device = batch.src.device
        batch_nseqs = int(ddp_reduce(batch.nseqs, device, torch.long).item())  # = all ranks' sentences under DDP
        sort_reverse_index = expand_reverse_index(batch.sort_by_src_length(), n_best)
        ids = scores = att = None
        if compute_loss and batch.has_trg:
            assert model.loss_function is not None
            with torch.no_grad():  # run as during training to get the validation loss; log-probabilities only when they are used
                batch_loss, slot1, _, n_correct = model(return_type="loss_probs" if return_prob == "ref" else "loss",
                                                        return_prob=return_prob, return_attention=return_attention, **vars(batch))
            batch_loss, n_correct = ddp_reduce(batch_loss.detach().float()), ddp_reduce(n_correct)
            batch_ntokens = ddp_reduce(int(batch.ntokens), device, torch.long)
            batch_loss = batch.normalize(batch_loss, "sum", n_gpu=n_gpu)  # sum over multiple GPUs (DataParallel's vector of losses)
            n_correct = batch.normalize(n_correct, "sum", n_gpu=n_gpu)
            if return_prob == "ref":
                log_probs = ddp_merge(slot1, 0.0)
                batch_trg = ddp_merge(batch.trg, model.pad_index)
                scores = Batch.score(log_probs, batch_trg, model.pad_index)
                ids = batch_trg.detach().cpu().numpy()
            totals["loss"] += float(batch_loss.sum().item())
            totals["n_correct"] += int(n_correct.sum().item())
            totals["ntokens"] += int(batch_ntokens.sum().item())
        if return_prob != "ref":
            ids, scores, att = search(model=model, batch=batch, beam_size=beam_size, beam_alpha=beam_alpha, n_best=n_best,
                                      max_output_length=max_output_length, min_output_length=min_output_length,
                                      generate_unk=generate_unk, return_prob=return_prob,
                                      repetition_penalty=repetition_penalty, no_repeat_ngram_size=no_repeat_ngram_size,
                                      return_attention=return_attention)
        if ddp:
            # Hypotheses are merged ONCE: greedy search returns all ranks' rows already (its tail merges ids, scores and attention as
            # the reference's does, search.py:333-335); beam search returns this rank's rows, so they are merged here (extension: the
            # reference's beam search does not merge, and its assert below then fails for world_size > 1).  The order of merged
            # outputs is unknown: they are put back by `indices` after the loop (prediction.py:222-231).
            if return_prob != "ref" and beam_size >= 2:
                ids = ddp_merge(torch.as_tensor(np.asarray(ids), device=device), model.pad_index).cpu().numpy()
                if scores is not None:
                    scores = ddp_merge(torch.as_tensor(np.asarray(scores), device=device), 0.0).cpu().numpy()
            batch_indices = ddp_merge(batch.indices.to(device).unsqueeze(1), -1).squeeze(1)
            assert bool(torch.all(batch_indices >= 0)) and len(batch_indices) * n_best == len(ids) == batch_nseqs * n_best, \
                (len(batch_indices), len(ids), batch_nseqs)
            have_scores = scores is not None and len(scores) == len(ids)
            for k, index in enumerate(batch_indices.cpu().tolist()):
                # keyed by dataset index: a sentence the sampler handed out twice (its padding to a multiple of the world size)
                # is kept once, the later copy overwriting the earlier one as in the reference (`_all_outputs[i] = row`, :250-257)
                rows = slice(k * n_best, (k + 1) * n_best)
                by_index[int(index)] = (list(ids[rows]), list(scores[rows]) if have_scores else None,
                                        att[k] if att is not None else None)
        else:
            all_ids.extend(ids[sort_reverse_index])  # either hypotheses or references
            if att is not None:
                all_att.extend(att[sort_reverse_index])
            if scores is not None and len(scores) == len(sort_reverse_index):
                all_scores.extend(scores[sort_reverse_index])
        totals["nseqs"] += batch_nseqs
    if ddp:
        for index in sorted(by_index):  # dataset order; the n-best rows of a sentence stay together
            rows, row_scores, row_att = by_index[index]
            all_ids.extend(rows)
            if row_scores is not None:
                all_scores.extend(row_scores)
            if row_att is not None:
                all_att.append(row_att)
    sentences = model.trg_vocab.arrays_to_sentences(all_ids, cut_at_eos=True)
    out = [all_ids, sentences, (all_scores if len(all_scores) else None)]
    if return_attention:
        out.append(all_att if all_att else None)
    if compute_loss:
        normalizer = {"batch": totals["nseqs"], "tokens": totals["ntokens"], "none": 1}[normalization]
        assert normalizer > 0 and totals["ntokens"] > 0, (normalizer, totals)
        valid_scores = {"loss": totals["loss"] / normalizer, "acc": totals["n_correct"] / totals["ntokens"],
                        "ppl": math.exp(totals["loss"] / totals["ntokens"])}  # upstream JoeyNMT; this reference leaves them NaN
        out.append({"totals": totals, "normalizer": normalizer, "valid_scores": valid_scores})
    return tuple(out)
