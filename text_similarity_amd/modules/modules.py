"""Pooling modules on the hot path: ``AvgPoolingStrategy`` (/root/reference/src/modules/modules.py:154-171).
The masked mean runs in the HIP kernel ``mean_pool_kernel`` through ``tsim_mean_pool``."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..dataset.dataset import EmbeddingsFeatures


class PoolingStrategy(nn.Module):
    """Base class (modules.py:44-55).  ``params`` is optional here: the reference's own ``from_pretrained`` calls
    ``AvgPoolingStrategy()`` without it (sentence_encoder.py:201), which its constructor rejects."""

    def __init__(self, params=None, *args, **kwargs):
        super().__init__()
        self.params = params

    def forward(self, embeddings: torch.Tensor, features: EmbeddingsFeatures = None):
        raise NotImplementedError()


class AvgPoolingStrategy(PoolingStrategy):
    def forward(self, embeddings: torch.Tensor, features: EmbeddingsFeatures):
        assert len(embeddings.shape) == 3  # batch, seq_len, embed_size
        mask = features.to_dict()["attention_mask"]
        return ops.mean_pool(embeddings, mask)
