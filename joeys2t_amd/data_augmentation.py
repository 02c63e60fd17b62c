"""SpecAugment and CMVN with the reference's class names (joeynmt/data_augmentation.py).

The reference transforms one NumPy spectrogram at a time on the host.  Here the mask *parameters* are still drawn
on the host from np.random in the reference's order (:54-68) — that is what keeps seeded runs reproducible —
while the statistics, normalisation, masking and batch padding run on the GPU for a whole batch at once
(js2t_cmvn_stats + js2t_feature_finalize)."""
import ctypes as C
import math
from typing import Optional, Sequence

import numpy as np
import torch

from joeys2t_amd import ops
from joeys2t_amd._lib import check, lib

_p, _stream = ops._p, ops._stream


class SpecAugment:
    """SpecAugment (https://arxiv.org/abs/1904.08779), reference :14-80."""

    def __init__(self, freq_mask_n: int = 2, freq_mask_f: int = 27, time_mask_n: int = 2, time_mask_t: int = 40,
                 time_mask_p: float = 1.0, mask_value: Optional[float] = None):
        self.freq_mask_n, self.freq_mask_f = freq_mask_n, freq_mask_f
        self.time_mask_n, self.time_mask_t, self.time_mask_p = time_mask_n, time_mask_t, time_mask_p
        self.mask_value = mask_value

    @property
    def slots(self):
        """(frequency, time) mask slots of draw()'s layout: two of each - the fused kernel's - unless more are configured."""
        return max(2, self.freq_mask_n), max(2, self.time_mask_n)

    @property
    def fused(self) -> bool:
        return self.freq_mask_n <= 2 and self.time_mask_n <= 2

    def draw(self, num_frames: int, num_freqs: int) -> np.ndarray:
        """The mask integers (start, width) of one utterance - frequency slots first, then time slots; (f0,f, f0,f, t0,t, t0,t)
        for the usual two of each - consuming np.random exactly like SpecAugment.__call__ (:54-68).  All-zero widths when the
        reference returns its input unchanged (:48-52)."""
        nf, nt = self.slots
        m = np.zeros(2 * (nf + nt), dtype=np.int32)
        if num_frames == 0 or num_freqs < self.freq_mask_f:
            return m
        for i in range(self.freq_mask_n):
            f = np.random.randint(0, self.freq_mask_f)
            f0 = np.random.randint(0, num_freqs - f)
            m[2 * i], m[2 * i + 1] = f0, f
        max_t = min(self.time_mask_t, math.floor(num_frames * self.time_mask_p))
        if max_t < 1:
            return m
        for i in range(self.time_mask_n):
            t = np.random.randint(0, max_t)
            t0 = np.random.randint(0, num_frames - t)
            m[2 * nf + 2 * i], m[2 * nf + 2 * i + 1] = t0, t
        return m

    def __repr__(self):
        return (f"{self.__class__.__name__}(freq_mask_n={self.freq_mask_n}, freq_mask_f={self.freq_mask_f}, "
                f"time_mask_n={self.time_mask_n}, time_mask_t={self.time_mask_t}, time_mask_p={self.time_mask_p})")


class CMVN:
    """Utterance-level mean / variance normalisation, reference :83-115."""

    def __init__(self, norm_means: bool = True, norm_vars: bool = True, before: bool = True):
        self.norm_means, self.norm_vars, self.before = norm_means, norm_vars, before

    def __repr__(self):
        return f"{self.__class__.__name__}(norm_means={self.norm_means}, norm_vars={self.norm_vars}, before={self.before})"


def finalize_features(feat: torch.Tensor, frame_off: torch.Tensor, frames: Sequence[int], *, cmvn: Optional[CMVN],
                      specaugment: Optional[SpecAugment], out_dtype=torch.float32, pad_value: float = 1.0,
                      max_length: Optional[int] = None, masks_dev: Optional[torch.Tensor] = None, t_pad: Optional[int] = None,
                      crop_t: Optional[torch.Tensor] = None):
    """feat f32 [sum T, F] (ragged, frame_off int64[U+1]) -> padded batch [U, Tmax, F] on the device:
    CMVN (before) -> SpecAugment -> pad with 1.0, i.e. SpeechProcessor.__call__ (tokenizers.py:480-492) followed by
    pad_features (helpers_for_audio.py:130-170), in two launches.  `max_length` truncates (evaluation-time rule,
    tokenizers.py:474-478) BEFORE the CMVN statistics are taken, as the reference does.
    t_pad / crop_t (graph replay of varying batches): the output is [U, t_pad, F] whatever `frames` says (the real counts are
    in frame_off on the device) and positions >= *crop_t (device int64 scalar: the longest utterance) are 0, not 1.0."""
    ops._dev(feat, frame_off)
    U, F = len(frames), feat.shape[1]
    dev = feat.device
    eff = [min(int(t), max_length) if max_length else int(t) for t in frames]
    Tmax = max(eff) if t_pad is None else int(t_pad)
    mean = istd = fill = masks = None
    fused = ((cmvn is None or cmvn.before) and (specaugment is None or specaugment.fused) and
             not (specaugment is not None and cmvn is None and specaugment.mask_value is None))
    if not fused:
        if masks_dev is not None or crop_t is not None:
            raise NotImplementedError("graph-replayed batches (masks_dev / crop_t) take the fused front-end: CMVN before SpecAugment, "
                                      "at most two masks of each kind")
        return _finalize_general(feat, frame_off, eff, cmvn, specaugment, out_dtype, pad_value, max_length, Tmax)
    if cmvn is not None:
        mean = torch.empty((U, F), dtype=torch.float32, device=dev)
        istd = torch.empty((U, F), dtype=torch.float32, device=dev)
        fill = torch.empty((U, ), dtype=torch.float32, device=dev)
        ws = torch.empty((int(lib().js2t_cmvn_stats_workspace(C.c_int32(U), C.c_int32(F))), ), dtype=torch.float64, device=dev)
        check(lib().js2t_cmvn_stats_ws(_p(feat), _p(frame_off), C.c_int32(U), C.c_int32(F), _p(mean), _p(istd), _p(fill),
                                       C.c_int32(int(cmvn.norm_means)), C.c_int32(int(cmvn.norm_vars)),
                                       C.c_int64(int(max_length) if max_length else 0), _p(ws), _stream()),
              "js2t_cmvn_stats_ws")
    if specaugment is not None:
        if masks_dev is not None:  # caller-managed static int32[U,8] buffer (hipGraph replay): already drawn
            masks = masks_dev
        else:
            m = np.stack([specaugment.draw(t, F) for t in eff])
            masks = torch.from_numpy(m).to(dev)
        if specaugment.mask_value is not None:
            fill = torch.full((U, ), float(specaugment.mask_value), dtype=torch.float32, device=dev)
    out = torch.empty((U, Tmax, F), dtype=out_dtype, device=dev)
    check(lib().js2t_feature_finalize_crop(_p(feat), _p(frame_off), _p(mean), _p(istd), _p(fill), _p(masks), _p(out),
                                           ops.dt_code(out), C.c_int64(U), C.c_int64(Tmax), C.c_int32(F), C.c_float(pad_value),
                                           _p(crop_t), _stream()), "js2t_feature_finalize")
    return out, eff


def _stats(feat, frame_off, U, F, norm_means, norm_vars, max_length):
    dev = feat.device
    mean = torch.empty((U, F), dtype=torch.float32, device=dev)
    istd = torch.empty((U, F), dtype=torch.float32, device=dev)
    fill = torch.empty((U, ), dtype=torch.float32, device=dev)
    check(lib().js2t_cmvn_stats(_p(feat), _p(frame_off), C.c_int32(U), C.c_int32(F), _p(mean), _p(istd), _p(fill), C.c_int32(int(norm_means)),
                                C.c_int32(int(norm_vars)), C.c_int64(int(max_length) if max_length else 0), _stream()), "js2t_cmvn_stats")
    return mean, istd, fill


def _finalize_general(feat, frame_off, eff, cmvn, specaugment, out_dtype, pad_value, max_length, Tmax):
    """SpeechProcessor.__call__ (tokenizers.py:480-492) for the settings the fused pair does not cover - CMVN after SpecAugment,
    more than two masks of a kind, SpecAugment's local-mean fill without CMVN: the same steps in the reference's order, each a
    kernel over the ragged features (js2t_cmvn_stats, js2t_feature_transform in place on a copy), then the padding launch."""
    U, F = len(eff), feat.shape[1]
    dev = feat.device
    x = feat.clone()
    # (evaluation-time truncation, tokenizers.py:474-478, comes first in the reference: js2t_cmvn_stats(max_frames) and the padding
    # launch (Tmax) both stop at max_length, masks are drawn for the kept frames)
    mean = istd = None
    if cmvn is not None and cmvn.before:
        mean, istd, fill_n = _stats(x, frame_off, U, F, cmvn.norm_means, cmvn.norm_vars, max_length)
    masks = fill = None
    if specaugment is not None:
        nf, nt = specaugment.slots
        masks = torch.from_numpy(np.stack([specaugment.draw(t, F) for t in eff])).to(dev)
        if specaugment.mask_value is not None:
            fill = torch.full((U, ), float(specaugment.mask_value), dtype=torch.float32, device=dev)
        elif mean is not None:
            fill = fill_n  # the mean of the normalised utterance, as the statistics kernel derives it
        else:  # the mean of the utterance as it is when SpecAugment sees it: un-normalised
            fill = _stats(x, frame_off, U, F, False, False, max_length)[2]
    if mean is not None or masks is not None:
        nf, nt = specaugment.slots if specaugment is not None else (0, 0)
        check(lib().js2t_feature_transform(_p(x), _p(frame_off), C.c_int32(U), C.c_int32(F), _p(mean), _p(istd), _p(fill), _p(masks),
                                           C.c_int32(nf), C.c_int32(nt), _stream()), "js2t_feature_transform")
    mean = istd = None
    if cmvn is not None and not cmvn.before:
        mean, istd, _ = _stats(x, frame_off, U, F, cmvn.norm_means, cmvn.norm_vars, max_length)
    out = torch.empty((U, Tmax, F), dtype=out_dtype, device=dev)
    check(lib().js2t_feature_finalize_crop(_p(x), _p(frame_off), _p(mean), _p(istd), None, None, _p(out), ops.dt_code(out), C.c_int64(U),
                                           C.c_int64(Tmax), C.c_int32(F), C.c_float(pad_value), None, _stream()), "js2t_feature_finalize")
    return out, eff
