"""Minimal target vocabulary: token <-> id with the special-symbol ids of the S2T configs
(reference vocabulary.py:20-154, special ids from configs/librispeech_100h.yaml:45-53)."""
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np

DEFAULT_SPECIALS = SimpleNamespace(unk_token="<unk>", unk_id=0, pad_token="<pad>", pad_id=1, bos_token="<s>", bos_id=2,
                                   eos_token="</s>", eos_id=3, sep_token=None, sep_id=None, lang_tags=[])


class Vocabulary:
    def __init__(self, tokens: List[str], cfg: SimpleNamespace = DEFAULT_SPECIALS) -> None:
        self.specials = [cfg.unk_token, cfg.pad_token, cfg.bos_token, cfg.eos_token]
        self.lang_tags = list(cfg.lang_tags)
        if cfg.sep_token:
            self.specials.append(cfg.sep_token)
        self._stoi: Dict[str, int] = {}
        self._itos: List[str] = []
        self.add_tokens(self.specials + self.lang_tags + list(tokens))
        self.pad_index, self.bos_index, self.eos_index, self.unk_index = cfg.pad_id, cfg.bos_id, cfg.eos_id, cfg.unk_id
        self.sep_index: Optional[int] = cfg.sep_id if cfg.sep_token else None
        assert self.pad_index == self.lookup(cfg.pad_token) and self.bos_index == self.lookup(cfg.bos_token)
        assert self.eos_index == self.lookup(cfg.eos_token) and self._itos[cfg.unk_id] == cfg.unk_token

    @classmethod
    def synthetic(cls, size: int) -> "Vocabulary":
        """A vocabulary of `size` entries (4 specials + placeholder tokens) for synthetic-data runs."""
        return cls([f"tok{i}" for i in range(size - 4)])

    def add_tokens(self, tokens: List[str]) -> None:
        for t in tokens:
            if t not in self._stoi:
                self._stoi[t] = len(self._itos)
                self._itos.append(t)

    def lookup(self, token: str) -> int:
        return self._stoi.get(token, getattr(self, "unk_index", 0))

    def is_unk(self, token: str) -> bool:
        return self.lookup(token) == self.unk_index

    def __len__(self) -> int:
        return len(self._itos)

    def __eq__(self, other) -> bool:
        return isinstance(other, Vocabulary) and self._itos == other._itos

    def array_to_sentence(self, array: np.ndarray, cut_at_eos: bool = True, skip_pad: bool = True) -> List[str]:
        out = []
        for i in array:
            if skip_pad and i == self.pad_index:
                continue
            out.append(self._itos[i])
            if cut_at_eos and i == self.eos_index:
                break
        return out

    def arrays_to_sentences(self, arrays: np.ndarray, cut_at_eos: bool = True, skip_pad: bool = True) -> List[List[str]]:
        return [self.array_to_sentence(a, cut_at_eos, skip_pad) for a in arrays]
