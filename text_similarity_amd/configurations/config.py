"""Configuration containers with the reference's field names and positional order
(/root/reference/src/configurations/config.py:7-44) so call sites such as
``Configuration(model_parameters=..., model=..., save_path=..., tokenizer=tok, batch_size=16)`` keep working.

Differences, all deliberate:
* ``SearchConfiguration.ef / ef_construction / M`` are plain ints.  In the reference the trailing commas at
  config.py:41-43 turn the defaults into 1-tuples, which hnswlib would reject.
* ``max_tokens_per_batch`` / ``max_seqs_per_batch`` size the native encoder's activation workspace (packed tokens per
  launch).  ``batch_size`` is kept and still honoured as the tokenizer batch size.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional, Tuple, Union

import torch


@dataclass
class ModelParameters:
    model_name: str
    hidden_size: Optional[int] = 768
    num_classes: int = 2
    use_pretrained_embeddings: bool = False
    freeze_weights: bool = True
    context_layers: Tuple[int, ...] = (-1,)
    output_attention = False


@dataclass
class Configuration:
    model_parameters: Union[ModelParameters, None]
    model: str
    save_path: str
    tokenizer: Any = None                 # a HuggingFace tokenizer (callable with the kwargs of sentence_encoder.py:144-153)
    sequence_max_len: int = 256
    dropout_prob: float = 0.1
    lr: float = 2e-5
    batch_size: int = 16
    epochs: int = 1
    device: torch.device = torch.device("cuda")
    warmup_steps: int = 0
    fp16: bool = True
    model_path: Optional[str] = None
    # native-engine sizing (not in the reference)
    max_tokens_per_batch: int = 65536
    max_seqs_per_batch: int = 8192


@dataclass
class SearchConfiguration(Configuration):
    ef: int = 50
    ef_construction: int = 400
    M: int = 64
