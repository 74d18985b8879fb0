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
        assert path is not None
        os.makedirs(path, exist_ok=True)
        raise NotImplementedError("weights live in the native handle as bf16; keep the source checkpoint directory")

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


def _native_from_dir(path, params) -> NativeEncoder:
    return NativeEncoder.from_pretrained(path, max_tokens=getattr(params, "max_tokens_per_batch", 65536),
                                         max_seqs=getattr(params, "max_seqs_per_batch", 8192),
                                         device=params.device if torch.device(params.device).type == "cuda" else None)
