"""Activation selection (reference builders.py:24-41).  The returned nn.Module classes are placeholders inside
`pwff_layer` (so that repr()/indices match the reference: pwff_layer.0 / .3 are the Linears); the activation
itself is evaluated in the GEMM epilogue of libjoeys2t_hip.so."""
from typing import Callable

from torch import nn


class ConfigurationError(Exception):
    """Custom exception for misspecifications of configuration (reference config.py:22)."""


class _Tanh(nn.Module):
    def forward(self, x):  # pragma: no cover - placeholder, never called on the HIP path
        raise RuntimeError("placeholder activation module: the HIP GEMM epilogue applies tanh")


def build_activation(activation: str = "relu") -> Callable:
    if activation == "relu":
        return nn.ReLU
    if activation == "gelu":
        return nn.GELU
    if activation == "tanh":
        return _Tanh
    if activation == "swish":
        return nn.SiLU
    raise ConfigurationError("Invalid activation function. Valid options: 'relu', 'gelu', 'tanh', 'swish'.")
