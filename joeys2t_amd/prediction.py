"""Minimal decode driver: the numerically relevant part of the reference's `predict` batch loop
(joeynmt/prediction.py:154-245): sort by source length -> search -> un-sort (n-best expanded) -> ids to tokens
(cut at EOS).  Dataset plumbing, logging, BLEU/chrF and checkpoint handling of the reference are out of scope."""
from typing import Iterable, List, Optional, Tuple

import numpy as np

from joeys2t_amd.batch import Batch
from joeys2t_amd.helpers import expand_reverse_index
from joeys2t_amd.search import search


def predict(model, batches: Iterable[Batch], *, beam_size: int = 1, beam_alpha: float = -1.0, n_best: int = 1,
            max_output_length: int = -1, min_output_length: int = 1, generate_unk: bool = True,
            return_prob: str = "none", repetition_penalty: float = -1, no_repeat_ngram_size: int = -1,
            return_attention: bool = False):
    """Returns (hypothesis id arrays in the ORIGINAL batch order, decoded token lists, scores or None); with
    `return_attention` a fourth element: the attention arrays of greedy search (prediction.py:205-218 hands the same
    options to `search`, which derives `encoder_input` / the forced prefix from the batch)."""
    model.eval()
    all_ids, all_scores, all_att = [], [], []
    for batch in batches:
        sort_reverse_index = expand_reverse_index(batch.sort_by_src_length(), n_best)
        ids, scores, att = search(model=model, batch=batch, beam_size=beam_size, beam_alpha=beam_alpha, n_best=n_best,
                                max_output_length=max_output_length, min_output_length=min_output_length,
                                generate_unk=generate_unk, return_prob=return_prob,
                                repetition_penalty=repetition_penalty, no_repeat_ngram_size=no_repeat_ngram_size,
                                return_attention=return_attention)
        all_ids.extend(ids[sort_reverse_index])
        if att is not None:
            all_att.extend(att[sort_reverse_index])
        if scores is not None:
            all_scores.extend(scores[sort_reverse_index])
    sentences = model.trg_vocab.arrays_to_sentences(all_ids, cut_at_eos=True)
    if return_attention:
        return all_ids, sentences, (all_scores if all_scores else None), (all_att if all_att else None)
    return all_ids, sentences, (all_scores if all_scores else None)
