"""SBERT-style bi-encoder wrappers with the reference's names and signatures
(/root/reference/src/models/sentence_encoder.py): ``OnnxSentenceTransformerWrapper`` (:17-39) and
``SentenceTransformerWrapper`` (:71-217), running on the native MI355X encoder.

What is reproduced is the *intended* behaviour: ``encode_text`` in the reference calls
``self.encode(features, parallel_mode=False)`` (:161) although ``encode`` is ``encode(documents, output_np)``
(:133) — a TypeError as shipped; its evident meaning, encoder forward + pooler on the batch, is what runs here.
"""
from __future__ import annotations

import os
from typing import List, Union

import numpy as np
import torch
from torch import nn

from ..dataset.dataset import EmbeddingsFeatures
from ..modules.modules import AvgPoolingStrategy, PoolingStrategy
from ..native_encoder import NativeEncoder
from .modeling import BaseEncoderModel, _native_from_dir


def _native_wordpiece(tokenizer):
    """The tokenizer's NativeWordPiece (built once, kept on the tokenizer object), or None when its pipeline is not BERT's,
    the library is not built, or TSIM_NATIVE_TOKENIZER=0."""
    wp = getattr(tokenizer, "_tsim_native_wordpiece", None)
    if wp is None:
        wp = False
        if os.environ.get("TSIM_NATIVE_TOKENIZER", "1") != "0":
            from ..wordpiece import NativeWordPiece
            wp = NativeWordPiece.from_tokenizer(tokenizer) or False
        try:
            tokenizer._tsim_native_wordpiece = wp
        except Exception:
            pass
    return wp or None


def _tokenize_packed(tokenizer, docs: List[str], max_len: int, batch_size: int):
    """Tokenise without padding -> (flat ids int32, lengths): the native WordPiece path for the sentences it handles (pure
    ASCII; csrc/wordpiece.cpp — same ids, tests/test_wordpiece_cpu.py), the library for the rest."""
    wp = _native_wordpiece(tokenizer)
    if wp is not None:
        return wp.tokenize_packed(docs, int(max_len), lambda rest: _tokenize_library(tokenizer, rest, max_len, batch_size))
    return _tokenize_library(tokenizer, docs, max_len, batch_size)


def _tokenize_library(tokenizer, docs: List[str], max_len: int, batch_size: int):
    """Tokenise without padding -> (flat ids int32, lengths).  Same tokenizer kwargs as
    sentence_encoder.py:144-153 except padding: the packed layout has no pad tokens, which is equivalent
    because padded positions are masked out of attention and pooling.

    Fast (Rust-backed) tokenizers are driven through their backend ``encode_batch`` on the whole chunk: the same code the
    ``tokenizer(text=...)`` call ends in — identical ids (tests/test_host_cpu.py) — without the per-call Python wrapping
    (BatchEncoding construction, per-batch option handling), which is single-threaded and was about half of the tokenizer
    time; the backend itself spreads a batch over the host cores (rayon), so one call per chunk uses all of them."""
    bt = getattr(tokenizer, "backend_tokenizer", None) if getattr(tokenizer, "is_fast", False) else None
    if bt is not None and hasattr(bt, "encode_batch"):
        prev_trunc, prev_pad = bt.truncation, bt.padding
        try:
            bt.enable_truncation(max_length=int(max_len), stride=0, strategy="longest_first",
                                 direction=getattr(tokenizer, "truncation_side", "right"))
            bt.no_padding()
            encs = bt.encode_batch(docs, add_special_tokens=True)
        finally:
            if prev_trunc is None:
                bt.no_truncation()
            else:
                bt.enable_truncation(**prev_trunc)
            if prev_pad is not None:
                bt.enable_padding(**prev_pad)
        lens = np.fromiter((len(e) for e in encs), dtype=np.int64, count=len(encs))
        flat = np.empty(int(lens.sum()), dtype=np.int32)
        o = 0
        for e, n in zip(encs, lens):
            flat[o:o + n] = e.ids
            o += n
        return flat, lens
    flat, lens = [], []
    for s in range(0, len(docs), batch_size):
        enc = tokenizer(text=docs[s:s + batch_size], add_special_tokens=True, padding=False, truncation=True,
                        max_length=max_len, return_attention_mask=False, return_token_type_ids=False)
        for ids in enc["input_ids"]:
            flat.extend(ids)
            lens.append(len(ids))
    return np.asarray(flat, dtype=np.int32), np.asarray(lens, dtype=np.int64)


class _EncodeMixin:
    """encode / encode_text shared by both wrappers."""

    def encode(self, documents, output_np: bool = False):
        # evaluators call model.encode(EmbeddingsFeatures) (src/evaluation/evaluators.py:75): dispatch on type
        if isinstance(documents, EmbeddingsFeatures):
            return self._encode_features(documents)
        return self.encode_text(documents, output_np)

    def _encode_features(self, features: EmbeddingsFeatures) -> torch.Tensor:
        d = features.to_dict()
        hidden = self.context_embedder(input_ids=d["input_ids"], attention_mask=d["attention_mask"])[0]
        pooler = getattr(self, "pooler", None) or AvgPoolingStrategy(self.params)
        return self.projection(pooler(hidden, features))

    def encode_packed(self, flat_ids: torch.Tensor, cu: torch.Tensor, unit: bool = False, cu_host: np.ndarray = None):
        """Device-resident pre-tokenised input (the benchmark path): pooled f32 [B,H] (+ unit float16 rows).
        ``cu_host``: the same offsets on the host, when the caller has them (saves a device->host copy and its sync)."""
        enc: NativeEncoder = self.context_embedder
        B = cu.numel() - 1
        outs, units = [], []
        cu_h = (cu.cpu().numpy() if cu_host is None else np.asarray(cu_host)).astype(np.int64)
        s = 0
        while s < B:
            e = s + 1
            while e < B and e - s < enc.max_seqs and cu_h[e + 1] - cu_h[s] <= enc.max_tokens:
                e += 1
            if cu_h[e] - cu_h[s] > enc.max_tokens:
                raise ValueError("a single sequence exceeds the encoder token capacity")
            r = enc.forward_packed(flat_ids[cu_h[s]:cu_h[e]], cu[s:e + 1] - cu[s], pooled=True, unit=unit,
                                   max_len=int(np.diff(cu_h[s:e + 1]).max()))
            outs.append(r["pooled"])
            if unit:
                units.append(r["unit"])
            s = e
        pooled = torch.cat(outs) if outs else torch.empty((0, enc.cfg.hidden), device=cu.device)
        return (pooled, torch.cat(units)) if unit else pooled

    def encode_text(self, documents: List[str], output_np: bool = False) -> Union[torch.Tensor, np.ndarray]:
        """sentence_encoder.py:136-173: sort by character length, encode in batches, un-sort, stack.
        Returns float32 [N, H] on params.device (or numpy when ``output_np``), un-normalised.
        The host tokenizer is the end-to-end bottleneck of this path, so it runs one chunk AHEAD on a
        host thread: chunk i+1 is tokenised while the GPU encodes chunk i (kernel launches are asynchronous; fast
        tokenizers release the GIL).  ``self.last_encode_stats`` holds the split of the wall time."""
        import time
        from concurrent.futures import ThreadPoolExecutor
        enc: NativeEncoder = self.context_embedder
        dev = enc.device
        n = len(documents)
        if n == 0:
            out = torch.empty((0, enc.cfg.hidden), dtype=torch.float32, device=dev)
            return out.cpu().numpy() if output_np else out
        t_start = time.perf_counter()
        order = np.argsort([len(s) for s in documents], kind="stable")
        docs = [documents[i] for i in order]
        tok_batch = max(int(self.params.batch_size), 1) * 64
        chunk = max(tok_batch, int(getattr(self.params, "encode_chunk_sentences", 8192)))
        t_tok = [0.0]

        def tokenize(lo):
            t0 = time.perf_counter()
            r = _tokenize_packed(self.params.tokenizer, docs[lo:lo + chunk], self.params.sequence_max_len, tok_batch)
            t_tok[0] += time.perf_counter() - t0
            return r

        parts = []
        with ThreadPoolExecutor(max_workers=1) as pool, torch.no_grad():
            fut = pool.submit(tokenize, 0)
            for lo in range(0, n, chunk):
                flat, lens = fut.result()
                if lo + chunk < n:
                    fut = pool.submit(tokenize, lo + chunk)
                cu = np.zeros(len(lens) + 1, dtype=np.int64)
                np.cumsum(lens, out=cu[1:])
                flat_d = torch.from_numpy(flat).to(dev, non_blocking=True)
                cu_d = torch.from_numpy(cu.astype(np.int32)).to(dev, non_blocking=True)
                parts.append(self.projection(self.encode_packed(flat_d, cu_d, cu_host=cu)))
        pooled = torch.cat(parts) if len(parts) > 1 else parts[0]
        out = torch.empty_like(pooled)
        out[torch.from_numpy(order).to(dev)] = pooled      # un-sort (sentence_encoder.py:168)
        enc.check()      # out-of-range ids / positions: HF would have raised IndexError (synchronises, as the return would)
        self.last_encode_stats = {"sentences": n, "wall_s": time.perf_counter() - t_start, "tokenizer_s": t_tok[0]}
        return out.cpu().numpy() if output_np else out

    def get_sentence_embedding_dimension(self):
        return self.context_embedder.config.hidden_size


class OnnxSentenceTransformerWrapper(_EncodeMixin, BaseEncoderModel):
    """Inference wrapper with fixed average pooling (sentence_encoder.py:17-39)."""

    def __init__(self, *args, projection: nn.Module = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.projection = projection if projection is not None else nn.Identity()

    def forward(self, input_ids, attention_mask, **kwargs):
        token_embeddings = self.context_embedder(input_ids=input_ids, attention_mask=attention_mask, **kwargs)[0]
        token_embeddings = self.projection(token_embeddings)
        from .. import ops
        return ops.mean_pool(token_embeddings, attention_mask)

    @classmethod
    def from_pretrained(cls, path, projection: nn.Module = None, params=None):
        assert params is not None, "Parameters not found, need to pass model parameters for the model to work"
        return cls(params=params, context_embedder=_native_from_dir(path, params), projection=projection)


class SentenceTransformerWrapper(_EncodeMixin, BaseEncoderModel):
    """sentence_encoder.py:71-217.  ``merge_strategy`` and ``loss`` are training-time modules: accepted and kept,
    unused on the embed-and-search path."""

    def __init__(self, pooler: PoolingStrategy = None, merge_strategy=None, loss=None, *args,
                 parallel_mode: bool = True, projection: nn.Module = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.pooler = pooler if pooler is not None else AvgPoolingStrategy(self.params)
        self.merge_strategy = merge_strategy
        self.loss = loss
        self.parallel_mode = parallel_mode
        self.projection = projection if projection is not None else nn.Identity()

    def forward(self, features, return_output=False, head_mask=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is a pruning-time feature outside the embed-and-search path")
        if self.parallel_mode:
            e1 = self._encode_features(features.sentence_1_features)
            e2 = self._encode_features(features.sentence_2_features)
            merged = torch.cat((e1, e2, torch.abs(e1 - e2)), dim=-1)
        else:
            merged = self._encode_features(features)
        if self.loss is None:
            return merged
        return self.loss(merged, features)

    @classmethod
    def from_pretrained(cls, path, pooler=None, merge_strategy=None, loss=None, params=None, parallel_mode=True):
        assert params is not None, "Parameters not found, need to pass model parameters for the model to work"
        return cls(pooler=pooler, merge_strategy=merge_strategy, loss=loss, params=params,
                   context_embedder=_native_from_dir(path, params), parallel_mode=parallel_mode)

    @classmethod
    def from_preset(cls, preset: str, params, **kw):
        """Architecture preset with synthetic weights (offline stand-in for a hub checkpoint name)."""
        enc = NativeEncoder.from_preset(preset, max_tokens=params.max_tokens_per_batch,
                                        max_seqs=params.max_seqs_per_batch,
                                        device=params.device if torch.device(params.device).type == "cuda" else None)
        return cls(params=params, context_embedder=enc, **kw)
