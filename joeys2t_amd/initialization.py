"""Parameter initialisation following the reference's config keys (initialization.py:78-236): xavier/uniform/
normal/zeros per group ("embed" in name -> embed initializer, "bias" -> bias initializer, >1-D -> main), DeepNet
alpha/beta when initializer == xavier_normal (:63-76,:138-152,:193-203), zeroed padding row (:213-215).
torch.nn.init is used as the RNG / filler; this is one-off setup, not part of the hot path."""
from typing import Dict

import torch
from torch import nn

from joeys2t_amd.builders import ConfigurationError
from joeys2t_amd.embeddings import Embeddings


def compute_alpha_beta(num_enc_layers: int, num_dec_layers: int) -> Dict[str, Dict]:
    """DeepNet (arXiv:2203.00555) residual / init scales."""
    n, m = num_enc_layers, num_dec_layers
    return {
        "alpha": {"encoder": 0.81 * (n**4 * m)**(1 / 16), "decoder": (3 * m)**(1 / 4)},
        "beta": {"encoder": 0.87 * (n**4 * m)**(-1 / 16), "decoder": (12 * m)**(-1 / 4)},
    }


def _initializer(kind: str, scale: float, gain: float):
    scale = float(scale)
    assert scale > 0.0, "incorrect init_weight"
    kind = kind.lower()
    if kind == "xavier":
        kind = "xavier_uniform"
    table = {
        "xavier_uniform": lambda p: nn.init.xavier_uniform_(p, gain=gain),
        "xavier_normal": lambda p: nn.init.xavier_normal_(p, gain=gain),
        "uniform": lambda p: nn.init.uniform_(p, a=-scale, b=scale),
        "normal": lambda p: nn.init.normal_(p, mean=0.0, std=scale),
        "zeros": lambda p: nn.init.zeros_(p),
    }
    if kind not in table:
        raise ConfigurationError("Unknown initializer.")
    return table[kind]


def initialize_model(model: nn.Module, cfg: dict, src_padding_idx: int, trg_padding_idx: int) -> None:
    gain = float(cfg.get("init_gain", 1.0))
    init = cfg.get("initializer", "xavier_uniform")
    init = "xavier_uniform" if init == "xavier" else init
    init_fn = _initializer(init, cfg.get("init_weight", 0.01), gain)
    embed_fn = _initializer(cfg.get("embed_initializer", "xavier_uniform"), cfg.get("embed_init_weight", 0.01),
                            float(cfg.get("embed_init_gain", 1.0)))
    bias_fn = _initializer(cfg.get("bias_initializer", "zeros"), cfg.get("bias_init_weight", 0.01), gain)

    deepnet = None
    if init == "xavier_normal" and cfg["encoder"].get("type", "transformer") == cfg["decoder"].get(
            "type", "transformer") == "transformer":
        deepnet = compute_alpha_beta(cfg["encoder"]["num_layers"], cfg["decoder"]["num_layers"])
        for side, stack in (("encoder", model.encoder.layers), ("decoder", model.decoder.layers)):
            for layer in stack:
                layer.alpha = deepnet["alpha"][side]
                layer.feed_forward.alpha = deepnet["alpha"][side]

    with torch.no_grad():
        for name, p in model.named_parameters():
            if "embed" in name:
                embed_fn(p)
            elif "bias" in name:
                bias_fn(p)
            elif p.dim() > 1:
                if deepnet is not None:
                    beta = 1.0
                    if "pwff_layer" in name or "v_layer" in name or "output_layer" in name:
                        if "encoder" in name:
                            beta = deepnet["beta"]["encoder"]
                        elif "decoder" in name:
                            beta = deepnet["beta"]["decoder"]
                    nn.init.xavier_normal_(p, gain=beta)
                else:
                    init_fn(p)
            # 1-D non-bias parameters (LayerNorm weights) keep their constructor values (reference :176-210)
        if isinstance(model.src_embed, Embeddings):
            model.src_embed.lut.weight.data[src_padding_idx].zero_()
        model.trg_embed.lut.weight.data[trg_padding_idx].zero_()
