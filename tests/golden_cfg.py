"""Model configurations of the golden fixtures (must match oracle/make_golden.py:tiny_cfg)."""
SPECIALS = dict(unk=0, pad=1, bos=2, eos=3, sep=None)


def tiny_cfg(layer_norm="pre", initializer="xavier_uniform", act="relu", heads=2):
    return {
        "initializer": initializer, "bias_initializer": "zeros", "embed_initializer": "xavier_uniform",
        "tied_embeddings": False, "tied_softmax": False,
        "encoder": {"type": "transformer", "num_layers": 2, "num_heads": heads, "embeddings": {"embedding_dim": 8},
                    "hidden_size": 16, "ff_size": 32, "dropout": 0.0, "freeze": False, "subsample": True,
                    "conv_kernel_sizes": [5, 5], "conv_channels": 24, "in_channels": 8, "layer_norm": layer_norm,
                    "activation": act},
        "decoder": {"type": "transformer", "num_layers": 2, "num_heads": heads,
                    "embeddings": {"embedding_dim": 16, "scale": True, "dropout": 0.0}, "hidden_size": 16, "ff_size": 32,
                    "dropout": 0.0, "freeze": False, "layer_norm": layer_norm, "activation": act},
    }


FIXTURES = {
    "model_pre": dict(cfg=tiny_cfg("pre"), ctc_weight=0.3),
    "model_post": dict(cfg=tiny_cfg("post", act="gelu"), ctc_weight=0.3),
    "model_deepnet": dict(cfg=tiny_cfg("pre", initializer="xavier_normal", heads=4), ctc_weight=0.1),
}


def oracle_cfg(cfg):
    """Add the residual scale the reference derives at init time (DeepNet alpha for xavier_normal)."""
    import copy
    c = copy.deepcopy(cfg)
    a_enc = a_dec = 1.0
    if c["initializer"] == "xavier_normal":
        n, m = c["encoder"]["num_layers"], c["decoder"]["num_layers"]
        a_enc, a_dec = 0.81 * (n**4 * m)**(1 / 16), (3 * m)**(1 / 4)
    c["encoder"]["alpha"], c["decoder"]["alpha"] = a_enc, a_dec
    return c


def mt_cfg():
    """Model section of the reference's configs/transformer_small.yaml, as oracle/make_golden.py read it from the file."""
    import json
    from pathlib import Path
    return json.loads((Path(__file__).resolve().parent / "golden" / "model_mt_cfg.json").read_text())
