"""SpeechProcessor — the source-side "tokenizer" of the S2T path (reference tokenizers.py:433-508), batched on the GPU -
and the evaluation tail of the target side: `post_process` of the reference's text tokenizers (:133-165,230-260,334-366),
i.e. hypothesis token lists -> the strings WER is computed on.

Reference per-utterance order (:458-494): length filter (drop if min_length > T > 0; if T > max_length drop in
training, truncate in evaluation) -> CMVN(before) -> SpecAugment (training only) -> CMVN(after).  Training-side text
tokenisation (BPE learning / application, Moses) is CPU string processing outside the hot path: `__call__` is provided
for word / char level and, when the sentencepiece package and a model file are present, for SentencePiece."""
import re
from typing import List, Optional, Sequence, Tuple, Union

import torch

from joeys2t_amd.data_augmentation import CMVN, SpecAugment, finalize_features
from joeys2t_amd.helpers_for_audio import get_extractor


class SpeechProcessor:
    def __init__(self, level: str = "frame", num_freq: int = 80, normalize: bool = False, max_length: int = -1,
                 min_length: int = -1, sample_rate: int = 16000, **kwargs):
        self.level = level
        self.num_freq = num_freq
        self.normalize = normalize
        self.max_length = max_length
        self.min_length = min_length
        self.sample_rate = sample_rate
        self.specaugment: Optional[SpecAugment] = SpecAugment(**kwargs["specaugment"]) if "specaugment" in kwargs else None
        self.cmvn: Optional[CMVN] = CMVN(**kwargs["cmvn"]) if "cmvn" in kwargs else None

    def keep_mask(self, frames: Sequence[int], is_train: bool) -> List[bool]:
        """Which utterances survive the length filter (reference :461-478)."""
        keep = []
        for t in frames:
            ok = not (self.min_length > t > 0)
            if ok and 0 < self.max_length < t and is_train:
                ok = False
            keep.append(ok)
        return keep

    def draw_masks(self, frames: Sequence[int]):
        """SpecAugment parameters for a batch (host, np.random order of the reference) as an int32 [U,8] array."""
        import numpy as np
        eff = [min(int(t), self.max_length) if self.max_length > 0 else int(t) for t in frames]
        return np.stack([self.specaugment.draw(t, self.num_freq) for t in eff])

    def batch_from_waveforms(self, wave: torch.Tensor, n_samples: Sequence[int], is_train: bool = False,
                             out_dtype=torch.float32, sample_off: Optional[Sequence[int]] = None,
                             masks_dev: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, List[int]]:
        """Raw waveforms resident in HBM -> (features [B,Tmax,F] padded with 1.0, frame counts).  Utterances are NOT
        filtered here (call keep_mask first); over-long ones are truncated to max_length as in evaluation."""
        ex = get_extractor(wave.device, self.sample_rate, self.num_freq)
        feat, frame_off, frames = ex.batch(wave, n_samples, sample_off)
        sa = self.specaugment if is_train else None
        return finalize_features(feat, frame_off, frames, cmvn=self.cmvn, specaugment=sa, out_dtype=out_dtype,
                                 max_length=self.max_length if self.max_length > 0 else None, masks_dev=masks_dev)

    def batch_from_tables(self, wave: torch.Tensor, d_soff: torch.Tensor, d_foff: torch.Tensor, n_utts: int, t_pad: int,
                          crop_t: torch.Tensor, is_train: bool = True, out_dtype=torch.float32,
                          masks_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """batch_from_waveforms for hipGraph replay over VARYING batches: utterance offsets / frame counts come from device
        tables, the output is [n_utts, t_pad, F] (t_pad: a bucket length >= the longest utterance) with 1.0 behind each
        utterance and 0 from *crop_t (the longest utterance's frames) on; SpecAugment parameters from masks_dev."""
        ex = get_extractor(wave.device, self.sample_rate, self.num_freq)
        feat = ex.batch_tables(wave, d_soff, d_foff, n_utts, n_utts * t_pad)
        sa = self.specaugment if is_train else None
        if sa is not None and masks_dev is None:
            raise ValueError("batch_from_tables: SpecAugment parameters must be supplied in masks_dev")
        out, _ = finalize_features(feat, d_foff, [t_pad] * n_utts, cmvn=self.cmvn, specaugment=sa, out_dtype=out_dtype,
                                   max_length=None, masks_dev=masks_dev, t_pad=t_pad, crop_t=crop_t)
        return out

    def __repr__(self):
        return (f"{self.__class__.__name__}(level={self.level}, normalize={self.normalize}, "
                f"filter_by_length=({self.min_length}, {self.max_length}), specaugment={self.specaugment}, cmvn={self.cmvn})")


# ------------------------------------------------------------------------------------------------ text side: evaluation tail
_SPACE_RUNS = re.compile("[ \u3000]+")


def remove_extra_spaces(s: str) -> str:
    """Zero-width spaces out, runs of (ideographic) spaces to one, no space in front of ? ! , . : (helpers.py:409-431 of the reference)."""
    s = _SPACE_RUNS.sub(" ", s.replace("\u200b", ""))
    for mark in "?!,.:":
        s = s.replace(" " + mark, mark)
    return s.strip()


class BasicTokenizer:
    """Word / character level (reference tokenizers.py:24-188): `post_process` undoes it."""
    SPACE = chr(32)
    SPACE_ESCAPE = chr(9601)  # the SentencePiece whitespace marker

    def __init__(self, level: str = "word", lowercase: bool = False, normalize: bool = False, max_length: int = -1,
                 min_length: int = -1, **kwargs):
        self.level, self.lowercase, self.normalize = level, lowercase, normalize
        self.max_length, self.min_length = max_length, min_length
        self.pretokenizer = kwargs.get("pretokenizer", "none").lower()
        if self.pretokenizer == "moses":
            raise NotImplementedError("the Moses (de)tokenizer (sacremoses) is not part of this package")
        self.unk_token = self.eos_token = self.sep_token = None
        self.specials: List[str] = []
        self.lang_tags: List[str] = []

    def __call__(self, raw_input: str, is_train: bool = False) -> Optional[List[str]]:
        if raw_input is None:
            return None
        if self.level == "word":
            sequence = raw_input.split(self.SPACE)
        elif self.level == "char":
            sequence = list(raw_input.replace(self.SPACE, self.SPACE_ESCAPE))
        else:
            raise NotImplementedError(self.level)
        if is_train and self._filter_by_length(len(sequence)):
            return None
        return sequence

    def _filter_by_length(self, length: int) -> bool:
        return length > self.max_length > 0 or self.min_length > length > 0

    def _remove_special(self, sequence: List[str], generate_unk: bool = False) -> List[str]:
        specials = self.specials if generate_unk else self.specials + [self.unk_token]
        valid = [token for token in sequence if token not in specials]
        return valid if valid else [self.unk_token]  # never empty

    def _cut_prompt(self, sequence: List[str], keep_sep: bool) -> List[str]:
        """drop everything up to the prompt marker; the subword tokenizers keep the marker itself (it is a special token and
        goes in _remove_special): reference :244 / :349 `sequence[sep_pos:]` against :146 `sequence[sep_pos + 1:]`"""
        try:
            pos = sequence.index(self.sep_token)
        except ValueError:
            return sequence
        return sequence[pos:] if keep_sep else sequence[pos + 1:]

    def post_process(self, sequence: Union[List[str], str], generate_unk: bool = True, cut_at_sep: bool = True) -> str:
        if isinstance(sequence, list):
            if cut_at_sep:
                sequence = self._cut_prompt(sequence, keep_sep=False)
            sequence = self._remove_special(sequence, generate_unk=generate_unk)
            if self.level == "word":
                sequence = self.SPACE.join(sequence)
            elif self.level == "char":
                sequence = "".join(sequence).replace(self.SPACE_ESCAPE, self.SPACE)
        if self.normalize:
            sequence = remove_extra_spaces(sequence)
        assert sequence is not None and len(sequence) > 0, sequence
        return sequence

    def set_vocab(self, vocab) -> None:
        """vocab: joeys2t_amd.vocabulary.Vocabulary (reference :167-177)"""
        self.unk_token = vocab.specials[vocab.unk_index]
        self.eos_token = vocab.specials[vocab.eos_index]
        self.sep_token = vocab.specials[vocab.sep_index] if vocab.sep_index else None
        specials = list(vocab.specials) + list(vocab.lang_tags)
        self.specials = [token for token in specials if token != self.unk_token]
        self.lang_tags = list(vocab.lang_tags)

    def __repr__(self):
        return (f"{self.__class__.__name__}(level={self.level}, lowercase={self.lowercase}, normalize={self.normalize}, "
                f"filter_by_length=({self.min_length}, {self.max_length}), pretokenizer={self.pretokenizer})")


class SentencePieceTokenizer(BasicTokenizer):
    """reference tokenizers.py:191-288.  Decoding pieces back to text needs no model: SentencePiece's DecodePieces
    concatenates the pieces, turns the whitespace marker into a space and drops the leading one (byte-fallback and
    user-defined pieces, which the S2T vocabularies do not contain, would need the model: pass `model_file` then)."""

    def __init__(self, level: str = "bpe", lowercase: bool = False, normalize: bool = False, max_length: int = -1,
                 min_length: int = -1, **kwargs):
        super().__init__(level, lowercase, normalize, max_length, min_length, **kwargs)
        assert self.level == "bpe"
        self.model_file = kwargs.get("model_file")
        self.nbest_size, self.alpha = kwargs.get("nbest_size", 5), kwargs.get("alpha", 0.0)
        self.spm = None
        if self.model_file is not None:
            import sentencepiece as sp
            self.spm = sp.SentencePieceProcessor()
            self.spm.load(str(self.model_file))

    def __call__(self, raw_input: str, is_train: bool = False) -> Optional[List[str]]:
        if raw_input is None:
            return None
        if self.spm is None:
            raise RuntimeError("SentencePieceTokenizer needs `model_file` (and the sentencepiece package) to tokenize")
        if is_train and self.alpha > 0:
            tokenized = self.spm.sample_encode_as_pieces(raw_input, nbest_size=self.nbest_size, alpha=self.alpha)
        else:
            tokenized = self.spm.encode(raw_input, out_type=str)
        if is_train and self._filter_by_length(len(tokenized)):
            return None
        return tokenized

    def _decode(self, pieces: List[str]) -> str:
        if self.spm is not None:
            return self.spm.decode(pieces)
        # the unknown piece is rendered as SentencePiece's default unk surface " \u2047 " (DecodePieces)
        text = "".join(" \u2047 " if p == self.unk_token else p for p in pieces).replace(self.SPACE_ESCAPE, self.SPACE)
        return text[1:] if text.startswith(self.SPACE) else text

    def post_process(self, sequence: Union[List[str], str], generate_unk: bool = True, cut_at_sep: bool = True) -> str:
        if isinstance(sequence, list):
            if cut_at_sep:
                sequence = self._cut_prompt(sequence, keep_sep=True)
            sequence = self._remove_special(sequence, generate_unk=generate_unk)
            sequence = self._decode(sequence).replace(self.SPACE_ESCAPE, self.SPACE).strip()
        if self.normalize:
            sequence = remove_extra_spaces(sequence)
        assert sequence is not None and len(sequence) > 0, sequence
        return sequence


class SubwordNMTTokenizer(BasicTokenizer):
    """reference tokenizers.py:291-389: merge markers (`@@ `) are glued back; applying BPE codes (training side) needs the
    subword-nmt package and is not provided."""

    def __init__(self, level: str = "bpe", lowercase: bool = False, normalize: bool = False, max_length: int = -1,
                 min_length: int = -1, **kwargs):
        super().__init__(level, lowercase, normalize, max_length, min_length, **kwargs)
        assert self.level == "bpe"
        self.separator: str = kwargs.get("separator", "@@")
        self.codes = kwargs.get("codes")

    def __call__(self, raw_input: str, is_train: bool = False):
        raise NotImplementedError("applying BPE merges needs subword-nmt; hypotheses are post-processed without it")

    def post_process(self, sequence: Union[List[str], str], generate_unk: bool = True, cut_at_sep: bool = True) -> str:
        if isinstance(sequence, list):
            if cut_at_sep:
                sequence = self._cut_prompt(sequence, keep_sep=True)
            sequence = self._remove_special(sequence, generate_unk=generate_unk)
            sequence = self.SPACE.join(sequence).replace(self.separator + self.SPACE, "")
            if sequence.endswith(self.separator):
                sequence = sequence[:-len(self.separator)]
        if self.normalize:
            sequence = remove_extra_spaces(sequence)
        assert sequence is not None and len(sequence) > 0, sequence
        return sequence
