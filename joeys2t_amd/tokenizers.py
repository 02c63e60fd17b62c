"""SpeechProcessor — the source-side "tokenizer" of the S2T path (reference tokenizers.py:433-508), batched on the GPU.

Reference per-utterance order (:458-494): length filter (drop if min_length > T > 0; if T > max_length drop in
training, truncate in evaluation) -> CMVN(before) -> SpecAugment (training only) -> CMVN(after).  Text-side tokenizers
of the reference are CPU string processing outside the hot path and are not part of this package."""
from typing import List, Optional, Sequence, Tuple

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

    def __repr__(self):
        return (f"{self.__class__.__name__}(level={self.level}, normalize={self.normalize}, "
                f"filter_by_length=({self.min_length}, {self.max_length}), specaugment={self.specaugment}, cmvn={self.cmvn})")
