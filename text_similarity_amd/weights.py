"""Loading real checkpoints from a LOCAL HuggingFace directory (no network): config.json plus
model.safetensors or pytorch_model.bin (torch.load(weights_only=True)).  Mirrors what
``transformers.AutoModel.from_pretrained(path)`` feeds the reference at
/root/reference/src/models/sentence_encoder.py:194-195."""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np

from .presets import EncoderConfig


def config_from_hf(d: dict) -> EncoderConfig:
    mt = d.get("model_type", "bert")
    if mt not in ("bert", "mpnet"):
        raise ValueError(f"model_type {mt!r} is not supported (bert, mpnet)")
    act = d.get("hidden_act", "gelu")
    if act != "gelu":
        raise ValueError(f"hidden_act {act!r} is not supported (gelu)")
    if mt == "bert" and d.get("position_embedding_type", "absolute") != "absolute":
        raise ValueError("only absolute position embeddings are supported for BERT")
    return EncoderConfig(arch=mt, num_layers=d["num_hidden_layers"], hidden=d["hidden_size"],
                         heads=d["num_attention_heads"], ffn=d["intermediate_size"], vocab=d["vocab_size"],
                         max_pos=d["max_position_embeddings"], ln_eps=d.get("layer_norm_eps", 1e-12),
                         type_vocab=d.get("type_vocab_size", 2) if mt == "bert" else 0,
                         pad_id=d.get("pad_token_id", 0 if mt == "bert" else 1),
                         rel_buckets=d.get("relative_attention_num_buckets", 32))


def load_hf_dir(path: str) -> Tuple[EncoderConfig, Dict[str, np.ndarray]]:
    with open(os.path.join(path, "config.json")) as f:
        cfg = config_from_hf(json.load(f))
    st = os.path.join(path, "model.safetensors")
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        raw = load_file(st)
    elif os.path.exists(pt):
        import torch
        raw = {k: v.float().numpy() for k, v in torch.load(pt, map_location="cpu", weights_only=True).items()}
    else:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
    out = {}
    for k, v in raw.items():
        for prefix in ("bert.", "mpnet.", "model."):
            if k.startswith(prefix):
                k = k[len(prefix):]
        out[k] = np.asarray(v, dtype=np.float32)
    return cfg, out


def save_hf_dir(path: str, cfg: EncoderConfig, weights: Dict[str, np.ndarray]) -> None:
    """Write config.json + model.safetensors (used by save_pretrained and by tests)."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    d = {"model_type": cfg.arch, "num_hidden_layers": cfg.num_layers, "hidden_size": cfg.hidden,
         "num_attention_heads": cfg.heads, "intermediate_size": cfg.ffn, "vocab_size": cfg.vocab,
         "max_position_embeddings": cfg.max_pos, "layer_norm_eps": cfg.ln_eps, "hidden_act": "gelu",
         "pad_token_id": cfg.pad_id}
    if cfg.arch == "bert":
        d["type_vocab_size"] = cfg.type_vocab
    else:
        d["relative_attention_num_buckets"] = cfg.rel_buckets
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(d, f, indent=1)
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()},
              os.path.join(path, "model.safetensors"))
