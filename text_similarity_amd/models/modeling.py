"""``BaseEncoderModel`` (/root/reference/src/models/modeling.py:11-87): holds ``params`` and ``context_embedder``.
Here ``context_embedder`` is a :class:`text_similarity_amd.native_encoder.NativeEncoder`."""
from __future__ import annotations

import os

import torch
from torch import nn

from ..native_encoder import NativeEncoder


class BaseEncoderModel(nn.Module):
    def __init__(self, params, context_embedder, input_dict: bool = False, normalize: bool = False):
        super().__init__()
        self.params = params
        self.normalize = normalize      # stored, never read — as in the reference (modeling.py:20,24)
        self.input_dict = input_dict
        # not registered as a submodule: the native handle has no torch parameters
        object.__setattr__(self, "context_embedder", context_embedder)

    @classmethod
    def from_pretrained(cls, path, params=None):
        if params is None:
            raise ValueError("params are required: model_config.bin is a pickle and is not loaded")
        return cls(params=params, context_embedder=_native_from_dir(path, params))

    def save_pretrained(self, path):
        """modeling.py:52-59: encoder weights, tokenizer files and the model parameters under ``path``.  The reference
        pickles ``self.params`` into model_config.bin; here the same file holds a plain dict of the primitive fields
        (``torch.load(..., weights_only=True)`` reads it) — device handles and the tokenizer object are not serialised."""
        assert path is not None
        if not os.path.exists(path):
            os.makedirs(path)
        self.context_embedder.save_pretrained(path)
        tok = getattr(self.params, "tokenizer", None)
        if tok is not None and hasattr(tok, "save_pretrained"):
            tok.save_pretrained(path)
        torch.save(_plain_params(self.params), os.path.join(path, "model_config.bin"))

    @property
    def model_name(self):
        return self.params.model_parameters.model_name

    @property
    def config(self):
        return self.context_embedder.config

    @property
    def embedding_size(self):
        embed_size = self.config.dim if "distilbert" in self.params.model else self.config.hidden_size
        mp = self.params.model_parameters
        if mp is not None and mp.hidden_size is not None and mp.hidden_size < embed_size:
            return mp.hidden_size
        return embed_size

    @property
    def params_num(self):
        return 0

    def forward(self):
        raise NotImplementedError()

    def encode(self):
        raise NotImplementedError()


def _plain_params(params) -> dict:
    out = {}
    for k, v in vars(params).items():
        if isinstance(v, (int, float, str, bool, type(None))):
            out[k] = v
        elif k == "device":
            out[k] = str(v)
        elif k == "model_parameters" and v is not None:
            out[k] = {kk: vv for kk, vv in vars(v).items() if isinstance(vv, (int, float, str, bool, type(None)))}
    return out


def _native_from_dir(path, params) -> NativeEncoder:
    return NativeEncoder.from_pretrained(path, max_tokens=getattr(params, "max_tokens_per_batch", 65536),
                                         max_seqs=getattr(params, "max_seqs_per_batch", 8192),
                                         device=params.device if torch.device(params.device).type == "cuda" else None)
