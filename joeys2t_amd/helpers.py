"""Host-side helpers of the hot path (integer / bool logic; reference helpers.py).

subsequent_mask (:81-90), tile (:264-293), adjust_mask_size (:307-326), expand_reverse_index (:384-406),
lengths_to_padding_mask (:459-469), pad (:472-497), set_seed (:93-104), freeze_params (:296-304).
These build masks / index maps; they carry no floating-point arithmetic of the path."""
import random
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor, nn


_SUBSEQUENT_MASKS = {}


def subsequent_mask(size: int, device=None) -> Tensor:
    """Lower-triangular bool mask of shape (1, size, size).  A constant per (size, device): built once (two launches per
    decoder pass otherwise) and handed out read-only - callers combine it with `&`, none writes into it."""
    key = (int(size), str(device))
    m = _SUBSEQUENT_MASKS.get(key)
    if m is None:
        m = torch.ones(size, size, dtype=torch.bool, device=device).tril_().unsqueeze(0)
        # Built while a hipGraph is captured, the mask is filled only when THAT graph replays: cached, a graph captured later (another
        # bucket of graphed.GraphedTrainStep.precapture, same target length) would read it before anything has written it.  So a mask
        # made under capture belongs to its graph alone, and only eagerly built ones are shared.
        if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            if len(_SUBSEQUENT_MASKS) > 256:
                _SUBSEQUENT_MASKS.clear()
            _SUBSEQUENT_MASKS[key] = m
    return m


def set_seed(seed: int) -> None:
    """One seed for every generator a run draws from: torch (CPU and all GPUs), numpy (SpecAugment), random."""
    for seeder in (torch.manual_seed, np.random.seed, random.seed):
        seeder(seed)
    if torch.cuda.device_count() > 0 and torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def tile(x: Tensor, count: int, dim: int = 0) -> Tensor:
    """Repeat every slice along `dim` `count` times, keeping copies adjacent: [a,b] -> [a,a,b,b]."""
    if isinstance(x, tuple):
        return tuple(tile(t, count, dim) for t in x)
    return x.repeat_interleave(count, dim=dim)


def freeze_params(module: nn.Module) -> None:
    for p in module.parameters():
        p.requires_grad = False


def adjust_mask_size(mask: Optional[Tensor], batch_size: int, hyp_len: int) -> Optional[Tensor]:
    """Pad with zeros / crop a (batch, len) prompt mask to hyp_len columns."""
    if mask is None:
        return None
    cur = mask.size(1)
    if cur < hyp_len:
        out = mask.new_zeros((batch_size, hyp_len))
        out[:, :cur] = mask
        return out
    return mask[:, :hyp_len] if cur > hyp_len else mask


def expand_reverse_index(reverse_index: List[int], n_best: int = 1) -> List[int]:
    if n_best == 1:
        return reverse_index
    return [ix * n_best + n for ix in reverse_index for n in range(n_best)]


def lengths_to_padding_mask(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """mask[b, t] = t < lengths[b]  (True on valid positions).  Pass max_len to avoid the host sync of
    `lengths.max().item()` that the reference pays (helpers.py:466)."""
    if max_len is None:
        max_len = int(lengths.max().item())
    steps = torch.arange(max_len, device=lengths.device).unsqueeze(0)
    return steps < lengths.view(-1, 1)


def pad(x: Tensor, max_len: int, pad_index: int = 1, dim: int = 1) -> Tensor:
    """Right-pad dim 1 (of a 3-D tensor) or the last dim with pad_index up to max_len; bool masks are padded
    with True because pad_index == 1 (reference quirk, encoders.py:297)."""
    if pad_index is None:
        pad_index = 1
    cur = x.size(dim)
    if cur >= max_len:
        assert cur == max_len, (x.size(), max_len)
        return x
    shape = list(x.shape)
    shape[dim] = max_len - cur
    filler = torch.full(shape, pad_index, dtype=x.dtype, device=x.device)
    return torch.cat([x, filler], dim=dim if dim >= 0 else x.dim() + dim)


# ------------------------------------------------------------------------------------------------ checkpoint I/O
def load_checkpoint(path, map_location="cpu") -> dict:
    """Read a JoeyNMT / JoeyS2T checkpoint (reference helpers.py:232-242): a dict with `model_state` (the state_dict this
    package's Model loads as is - identical parameter names), `optimizer_state` (torch.optim.AdamW layout),
    `scheduler_state`, `scaler_state`, `train_iter_state`, `stats_state`."""
    from pathlib import Path
    path = Path(path)
    assert path.is_file(), f"Checkpoint {path} not found."
    return torch.load(path, map_location=map_location, weights_only=False)


def init_layers(model: nn.Module, path, layer: str, map_location="cpu") -> None:
    """Initialise the `encoder` / `decoder` sub-tree from a checkpoint (reference training.py:294-309)."""
    ckpt = load_checkpoint(path, map_location)
    model.load_state_dict({k: v for k, v in ckpt["model_state"].items() if k.startswith(layer)}, strict=False)


def average_checkpoints(inputs: List[str]) -> dict:
    """Average the `model_state` of several checkpoints (reference scripts/average_checkpoints.py:17-73): floating tensors
    are averaged, integer ones floor-divided; everything else is taken from the first file; key lists must agree."""
    import collections
    params, keys, new_state = collections.OrderedDict(), None, None
    for f in inputs:
        state = load_checkpoint(f, "cpu")
        if new_state is None:
            new_state = state
        ks = list(state["model_state"].keys())
        if keys is None:
            keys = ks
        elif keys != ks:
            raise KeyError(f"For checkpoint {f}, expected list of params: {keys}, but found: {ks}")
        for k in keys:
            p = state["model_state"][k]
            p = p.float() if p.dtype == torch.float16 else p
            if k not in params:
                params[k] = p.clone()
            else:
                params[k] += p
    for k, v in params.items():
        if v.is_floating_point():
            v.div_(len(inputs))
        else:
            v //= len(inputs)
    new_state["model_state"] = params
    return new_state
