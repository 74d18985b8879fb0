"""ORACLE — test infrastructure only.  Never imported by the product package.

CPU (torch, float32) restatement of the encoder forward the reference delegates to HuggingFace
``AutoModel`` (`/root/reference/src/models/sentence_encoder.py:33` ``context_embedder(...)[0]``),
and of the masked mean-pool the reference applies to it.  No ``transformers`` import: the layer
arithmetic is restated from the only in-tree statement of it,

* embeddings   /root/reference/src/models/bert_of_theseus.py:185-211
* attention    /root/reference/src/models/bert_of_theseus.py:244-336  (scores :292,310-313, softmax :316, PV :326)
* projections  /root/reference/src/models/bert_of_theseus.py:346-350, 411-414, 424-428
* model        /root/reference/src/models/bert_of_theseus.py:902-1024 (extended mask (1-m)*-10000 :972)
* mean-pool    /root/reference/src/modules/modules.py:158-171 == src/models/sentence_encoder.py:35-38

plus the MPNet deltas from the pinned third-party dependency ``transformers`` (requirements.txt:5 pins
4.2.0; the installed 5.15.0 mpnet/modeling_mpnet.py:71-95,135-174,312-348,873-881 was read for the
published algorithm): position ids = cumsum(ids != pad) * (ids != pad) + pad, no token-type table,
one T5-style relative-position bias table shared by all layers computed from arange(S).

Pinned (tests/test_oracle_golden.py) against golden vectors produced in the build container by
importing the reference's own ``OnnxSentenceTransformerWrapper.forward`` / ``AvgPoolingStrategy`` around
HF ``BertModel`` / ``MPNetModel`` (tools/make_golden.py -> tests/golden/*.npz).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def _t(w: Dict[str, np.ndarray], name: str) -> torch.Tensor:
    v = w[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def _layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _gelu_erf(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mpnet_relative_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """rel = memory_position - context_position (key index - query index)."""
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def encoder_forward(cfg, w: Dict[str, np.ndarray], input_ids, attention_mask, linear=F.linear) -> torch.Tensor:
    """last_hidden_state [B,S,H] float32 for padded ``input_ids``/``attention_mask`` [B,S].  ``linear`` replaces
    F.linear in the six projections of a layer (oracle.fp8_ref.mx_linear for the fp8 variant)."""
    ids = torch.as_tensor(np.asarray(input_ids)).long()
    mask = torch.as_tensor(np.asarray(attention_mask)).long()
    B, S = ids.shape
    H, nh, dh = cfg.hidden, cfg.heads, cfg.head_dim
    word = _t(w, "embeddings.word_embeddings.weight")
    pos = _t(w, "embeddings.position_embeddings.weight")
    if cfg.arch == "bert":
        pos_ids = torch.arange(S).unsqueeze(0).expand(B, S)
        x = word[ids] + _t(w, "embeddings.token_type_embeddings.weight")[0] + pos[pos_ids]
    else:
        ne = (ids != cfg.pad_id).long()
        pos_ids = torch.cumsum(ne, 1) * ne + cfg.pad_id
        x = word[ids] + pos[pos_ids]
    x = _layer_norm(x, _t(w, "embeddings.LayerNorm.weight"), _t(w, "embeddings.LayerNorm.bias"), cfg.ln_eps)
    ext = (1.0 - mask[:, None, None, :].float()) * -10000.0
    bias = None
    if cfg.arch == "mpnet":
        ar = torch.arange(S)
        bucket = mpnet_relative_bucket(ar[None, :] - ar[:, None], cfg.rel_buckets)
        bias = _t(w, "encoder.relative_attention_bias.weight")[bucket].permute(2, 0, 1).unsqueeze(0)
    for l in range(cfg.num_layers):
        p = f"encoder.layer.{l}."
        if cfg.arch == "bert":
            nq, nk, nv, no = (p + "attention.self.query", p + "attention.self.key",
                              p + "attention.self.value", p + "attention.output.dense")
            ln1 = p + "attention.output.LayerNorm"
        else:
            nq, nk, nv, no = (p + "attention.attn.q", p + "attention.attn.k",
                              p + "attention.attn.v", p + "attention.attn.o")
            ln1 = p + "attention.LayerNorm"
        lin = lambda t, n: linear(t, _t(w, n + ".weight"), _t(w, n + ".bias"))
        q = lin(x, nq).view(B, S, nh, dh).transpose(1, 2)
        k = lin(x, nk).view(B, S, nh, dh).transpose(1, 2)
        v = lin(x, nv).view(B, S, nh, dh).transpose(1, 2)
        sc = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        if bias is not None:
            sc = sc + bias
        sc = sc + ext
        pr = torch.softmax(sc, dim=-1)
        ctx = torch.matmul(pr, v).transpose(1, 2).reshape(B, S, H)
        x = _layer_norm(lin(ctx, no) + x, _t(w, ln1 + ".weight"), _t(w, ln1 + ".bias"), cfg.ln_eps)
        h = _gelu_erf(lin(x, p + "intermediate.dense"))
        x = _layer_norm(lin(h, p + "output.dense") + x, _t(w, p + "output.LayerNorm.weight"),
                        _t(w, p + "output.LayerNorm.bias"), cfg.ln_eps)
    return x


def mean_pool(hidden, attention_mask) -> torch.Tensor:
    """/root/reference/src/modules/modules.py:158-171: sum_s(h*m) / clamp(sum_s m, 1e-9)."""
    hidden = torch.as_tensor(hidden).float()
    assert hidden.dim() == 3
    m = torch.as_tensor(np.asarray(attention_mask)).unsqueeze(-1).expand(hidden.size()).float()
    return (hidden * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)


def encode(cfg, w, input_ids, attention_mask, linear=F.linear) -> torch.Tensor:
    """OnnxSentenceTransformerWrapper.forward (sentence_encoder.py:32-39) with Identity projection."""
    with torch.no_grad():
        return mean_pool(encoder_forward(cfg, w, input_ids, attention_mask, linear=linear), attention_mask)


def pad_batch(flat_ids: np.ndarray, cu: np.ndarray, rows, pad_id: int = 0):
    """Packed tokens -> right-padded [B,S] ids + mask (what tokenizer(padding='longest') yields,
    sentence_encoder.py:144-153)."""
    lens = [int(cu[r + 1] - cu[r]) for r in rows]
    S = max(max(lens), 1)
    ids = np.full((len(rows), S), pad_id, dtype=np.int64)
    mask = np.zeros((len(rows), S), dtype=np.int64)
    for i, r in enumerate(rows):
        ids[i, :lens[i]] = flat_ids[cu[r]:cu[r + 1]]
        mask[i, :lens[i]] = 1
    return ids, mask


def encode_packed(cfg, w, flat_ids, cu, batch_size: int = 16, linear=F.linear) -> np.ndarray:
    """The reference's encode_text loop (sentence_encoder.py:136-173) on pre-tokenised input:
    sort by length, batches of ``batch_size`` padded to the longest, un-sort.  float32 [n,H]."""
    n = len(cu) - 1
    lens = np.diff(cu)
    order = np.argsort(lens, kind="stable")
    out = np.zeros((n, cfg.hidden), dtype=np.float32)
    for s in range(0, n, batch_size):
        rows = order[s:s + batch_size]
        ids, mask = pad_batch(flat_ids, cu, rows, cfg.pad_id)
        out[rows] = encode(cfg, w, ids, mask, linear=linear).numpy()
    return out
