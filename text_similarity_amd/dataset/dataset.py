"""Feature container of the hot path: ``EmbeddingsFeatures``
(/root/reference/src/dataset/dataset.py:213-251).  ``to_dict`` drops ``token_type_ids`` when it is None, which is
how the reference ends up calling ``context_embedder(input_ids=..., attention_mask=...)``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class EmbeddingsFeatures:
    input_ids: torch.Tensor
    attention_mask: torch.Tensor
    token_type_ids: Optional[torch.Tensor] = None

    @classmethod
    def from_dict(cls, dictionary, *args, **kwargs):
        return cls(dictionary["input_ids"], dictionary["attention_mask"], dictionary.get("token_type_ids"),
                   *args, **kwargs)

    def to_dict(self):
        d = {"input_ids": self.input_ids, "attention_mask": self.attention_mask}
        if self.token_type_ids is not None:
            d["token_type_ids"] = self.token_type_ids
        return d

    def generate_labels(self, model):
        with torch.no_grad():
            return model.encode(self)

    def to(self, device):
        self.input_ids = self.input_ids.to(device)
        self.attention_mask = self.attention_mask.to(device)
        if self.token_type_ids is not None:
            self.token_type_ids = self.token_type_ids.to(device)
        return self
