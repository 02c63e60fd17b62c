"""Target-side embedding table with the reference's module layout (embeddings.py:17-64): `lut.weight` is an
nn.Embedding parameter (checkpoint key `trg_embed.lut.weight`); the gather * sqrt(d) runs in js2t_embed_fwd."""
import math

import torch
from torch import Tensor, nn

from joeys2t_amd import functional as Fn
from joeys2t_amd.helpers import freeze_params
from joeys2t_amd.runtime import runtime_of


class Embeddings(nn.Module):
    def __init__(self, embedding_dim: int = 64, scale: bool = False, vocab_size: int = 0, padding_idx: int = 1,
                 freeze: bool = False, **kwargs):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.scale = scale
        self.vocab_size = vocab_size
        self.lut = nn.Embedding(vocab_size, self.embedding_dim, padding_idx=padding_idx)
        if freeze:
            freeze_params(self)

    def forward(self, x: Tensor) -> Tensor:
        rt = runtime_of(self)
        factor = math.sqrt(self.embedding_dim) if self.scale else 1.0
        return Fn.EmbedFn.apply(x, self.lut.weight, factor, self.lut.padding_idx, rt.compute_dtype,
                                rt.grad_sink([self.lut.weight]) if torch.is_grad_enabled() else None, rt.grads_ready)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(embedding_dim={self.embedding_dim}, vocab_size={self.vocab_size})"
