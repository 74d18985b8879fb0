"""Host tokenizer of encode_text, native for ASCII sentences (csrc/wordpiece.cpp, include/tsim.h "tokenizer").

The reference calls the HuggingFace tokenizer it is configured with (sentence_encoder.py:144-153).  ``NativeWordPiece`` is
built FROM that tokenizer — vocabulary, special ids, lower-casing and truncation side are read out of its `tokenizers`
backend — and produces the same ids for the sentences it handles (pure ASCII, no added-token text inside); every other
sentence goes through the library itself, so the result is the library's for any input.  A tokenizer whose backend is not
the BERT pipeline (BertNormalizer / BertPreTokenizer / WordPiece / CLS..SEP template) is not supported: ``from_tokenizer``
returns None and the caller keeps using the library for everything."""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _lib


def _blob(keys: List[str]) -> Tuple[bytes, np.ndarray]:
    enc = [k.encode("utf-8") for k in keys]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    return b"".join(enc), off


class NativeWordPiece:
    def __init__(self, handle, n_special: int, owner):
        self._h = handle
        self.n_special = n_special
        self._owner = owner          # the HF tokenizer this was built from (fallback for unhandled sentences)
        self.threads = int(os.environ.get("TSIM_TOKENIZER_THREADS", "0"))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().tsim_wordpiece_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------------------------------------------------------
    @classmethod
    def from_tokenizer(cls, tokenizer) -> Optional["NativeWordPiece"]:
        """None when the tokenizer's pipeline is not the one csrc/wordpiece.cpp restates."""
        bt = getattr(tokenizer, "backend_tokenizer", None) if getattr(tokenizer, "is_fast", False) else None
        if bt is None:
            return None
        try:
            cfg = json.loads(bt.to_str())
        except Exception:
            return None
        norm, pre, model, post = cfg.get("normalizer"), cfg.get("pre_tokenizer"), cfg.get("model"), cfg.get("post_processor")
        if not (norm and norm.get("type") == "BertNormalizer" and norm.get("clean_text", True)):
            return None
        if not (pre and pre.get("type") == "BertPreTokenizer"):
            return None
        if not (model and model.get("type") == "WordPiece" and isinstance(model.get("vocab"), dict)):
            return None
        if getattr(tokenizer, "truncation_side", "right") != "right":
            return None
        vocab = model["vocab"]
        size = max(vocab.values()) + 1
        keys = [None] * size
        for k, i in vocab.items():
            keys[i] = k
        if any(k is None for k in keys):          # holes in the id range: keep the library
            return None
        unk = vocab.get(model.get("unk_token", "[UNK]"))
        if unk is None:
            return None
        prefix_ids, suffix_ids = [], []
        if post is None:
            pass
        elif post.get("type") == "TemplateProcessing":
            seen_seq = False
            for item in post.get("single", []):
                if "Sequence" in item:
                    if seen_seq:
                        return None
                    seen_seq = True
                elif "SpecialToken" in item:
                    ids = post["special_tokens"][item["SpecialToken"]["id"]]["ids"]
                    (suffix_ids if seen_seq else prefix_ids).extend(ids)
                else:
                    return None
            if not seen_seq:
                return None
        elif post.get("type") in ("BertProcessing", "RobertaProcessing"):
            prefix_ids, suffix_ids = [int(post["cls"][1])], [int(post["sep"][1])]
        else:
            return None
        added = [a["content"] for a in cfg.get("added_tokens", [])]
        vtext, voff = _blob(keys)
        atext, aoff = _blob(added)
        pre_a = np.asarray(prefix_ids, dtype=np.int32)
        suf_a = np.asarray(suffix_ids, dtype=np.int32)
        h = C.c_void_p()
        _lib.check(_lib.lib().tsim_wordpiece_create(
            vtext, voff.ctypes.data, size, model.get("continuing_subword_prefix", "##").encode(), int(unk),
            pre_a.ctypes.data if len(pre_a) else None, len(pre_a), suf_a.ctypes.data if len(suf_a) else None, len(suf_a),
            1 if norm.get("lowercase", True) else 0, int(model.get("max_input_chars_per_word", 100)),
            atext, aoff.ctypes.data, len(added), C.byref(h)), "tsim_wordpiece_create")
        return cls(h, len(prefix_ids) + len(suffix_ids), tokenizer)

    # ------------------------------------------------------------------------------------------------------------------
    def encode_ascii(self, docs: List[str], max_len: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """docs must all be ``str.isascii()``.  -> (ids of the handled sentences back to back, lens [n], handled [n] bool)."""
        n = len(docs)
        text = "".join(docs).encode("ascii")
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, docs), dtype=np.int64, count=n), out=off[1:])
        cap = int(off[-1]) + n * max(self.n_special, 1)
        ids = np.empty(cap, dtype=np.int32)
        lens = np.empty(n, dtype=np.int32)
        handled = np.empty(n, dtype=np.uint8)
        _lib.check(_lib.lib().tsim_wordpiece_encode(self._h, text, off.ctypes.data, n, int(max_len), self.threads, ids.ctypes.data,
                                                   cap, lens.ctypes.data, handled.ctypes.data), "tsim_wordpiece_encode")
        return ids[:int(lens.sum(dtype=np.int64))], lens.astype(np.int64), handled.astype(bool)

    def tokenize_packed(self, docs: List[str], max_len: int, fallback) -> Tuple[np.ndarray, np.ndarray]:
        """(flat ids int32, lens int64) of ``docs`` in order.  ``fallback(list of str) -> (flat, lens)`` is the library path,
        called once with the sentences the native code does not handle (non-ASCII, added-token text)."""
        n = len(docs)
        asc = np.fromiter((s.isascii() for s in docs), dtype=bool, count=n)
        if asc.all():
            a_docs, a_idx = docs, None
        else:
            a_idx = np.flatnonzero(asc)
            a_docs = [docs[i] for i in a_idx]
        ids, lens_a, handled = self.encode_ascii(a_docs, max_len) if a_docs else (np.empty(0, np.int32), np.empty(0, np.int64),
                                                                                 np.empty(0, bool))
        if a_idx is None and handled.all():
            return ids, lens_a
        # splice: sentences of the native path and of the library path back into input order
        ok = np.zeros(n, dtype=bool)
        ok[(np.arange(n) if a_idx is None else a_idx)[handled]] = True
        rest = np.flatnonzero(~ok)
        r_flat, r_lens = fallback([docs[i] for i in rest])
        lens = np.empty(n, dtype=np.int64)
        lens[ok] = lens_a[handled]
        lens[rest] = np.asarray(r_lens, dtype=np.int64)
        cu = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=cu[1:])
        flat = np.empty(int(cu[-1]), dtype=np.int32)

        def scatter(src, which):
            if len(which) == 0:
                return
            ln = lens[which]
            starts = np.repeat(cu[which] - (np.cumsum(ln) - ln), ln)     # destination start minus source start, per id
            flat[np.arange(int(ln.sum())) + starts] = src

        scatter(ids, np.flatnonzero(ok))
        scatter(np.asarray(r_flat, dtype=np.int32), rest)
        return flat, lens
