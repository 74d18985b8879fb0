"""Architecture presets and regenerable synthetic weights / sentences.

There is no network on the build or GPU boxes, so the three checkpoints BASELINE.json names
(all-MiniLM-L6-v2, all-mpnet-base-v2, bert-base-uncased) are honoured as *architecture presets*:
same layer count / widths / vocab as the public model cards, weights drawn from a counter-based
generator so that every process (oracle, golden tool, GPU box) regenerates bit-identical tensors
instead of shipping 400 MB of random numbers.  Real checkpoints load through
``weights.load_hf_dir`` with the same tensor names.

Tensor names are the HuggingFace ``state_dict`` names of ``BertModel`` / ``MPNetModel`` (the objects
the reference holds as ``context_embedder``: /root/reference/src/models/sentence_encoder.py:49,195).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


@dataclass(frozen=True)
class EncoderConfig:
    """Shape of one encoder.  ``arch`` is "bert" (absolute positions, token-type 0 row added,
    /root/reference/src/models/bert_of_theseus.py:185-211) or "mpnet" (RoBERTa-style position ids,
    one relative-position bias table shared by all layers; transformers mpnet/modeling_mpnet.py)."""
    arch: str
    num_layers: int
    hidden: int
    heads: int
    ffn: int
    vocab: int
    max_pos: int
    ln_eps: float = 1e-12
    type_vocab: int = 2
    pad_id: int = 0
    rel_buckets: int = 32

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def to_dict(self):
        return asdict(self)


PRESETS: Dict[str, EncoderConfig] = {
    # SURVEY.md §8 preset table
    "all-MiniLM-L6-v2": EncoderConfig("bert", 6, 384, 12, 1536, 30522, 512, 1e-12),
    "all-mpnet-base-v2": EncoderConfig("mpnet", 12, 768, 12, 3072, 30527, 514, 1e-5, type_vocab=0, pad_id=1),
    "bert-base-uncased": EncoderConfig("bert", 12, 768, 12, 3072, 30522, 512, 1e-12),
    # small shapes for golden fixtures / unit tests
    "tiny-bert": EncoderConfig("bert", 2, 64, 4, 128, 1000, 64, 1e-12),
    "tiny-mpnet": EncoderConfig("mpnet", 2, 64, 4, 128, 1000, 66, 1e-5, type_vocab=0, pad_id=1),
}


# --------------------------------------------------------------------------- counter-based RNG
def _fnv1a64(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(stream: str, n: int, offset: int = 0) -> np.ndarray:
    """n uniform [0,1) float32 values with 24 random bits each; value i depends only on
    (stream, offset+i)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64) + _fnv1a64(stream) * np.uint64(0x2545F4914F6CDD1D)
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float32)) * np.float32(1.0 / (1 << 24))


def randint(stream: str, n: int, lo: int, hi: int, offset: int = 0) -> np.ndarray:
    """n integers uniform in [lo, hi)."""
    u = uniform01(stream, n, offset).astype(np.float64)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64)


def normal(stream: str, n: int, offset: int = 0) -> np.ndarray:
    """n standard-normal float32 (Box-Muller on two uniform streams)."""
    u1 = uniform01(stream + "/bm1", n, offset).astype(np.float64)
    u2 = uniform01(stream + "/bm2", n, offset).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest bfloat16 (ties to even), returned as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    out = r.astype(np.uint32).view(np.float32)
    return np.where(np.isnan(x), x, out).astype(np.float32)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> uint16 bf16 bit patterns (RNE)."""
    return (bf16_round(x).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


# --------------------------------------------------------------------------- weights
def weight_names(cfg: EncoderConfig) -> List[tuple]:
    """(name, shape, kind) for every tensor of the encoder, HF naming."""
    H, F = cfg.hidden, cfg.ffn
    out = [("embeddings.word_embeddings.weight", (cfg.vocab, H), "w"),
           ("embeddings.position_embeddings.weight", (cfg.max_pos, H), "w")]
    if cfg.arch == "bert":
        out.append(("embeddings.token_type_embeddings.weight", (cfg.type_vocab, H), "w"))
    out += [("embeddings.LayerNorm.weight", (H,), "g"), ("embeddings.LayerNorm.bias", (H,), "b")]
    for l in range(cfg.num_layers):
        p = f"encoder.layer.{l}."
        if cfg.arch == "bert":
            q, k, v, o = (p + "attention.self.query", p + "attention.self.key",
                          p + "attention.self.value", p + "attention.output.dense")
            ln1 = p + "attention.output.LayerNorm"
        else:
            q, k, v, o = (p + "attention.attn.q", p + "attention.attn.k",
                          p + "attention.attn.v", p + "attention.attn.o")
            ln1 = p + "attention.LayerNorm"
        for lin, shp in ((q, (H, H)), (k, (H, H)), (v, (H, H)), (o, (H, H))):
            out += [(lin + ".weight", shp, "w"), (lin + ".bias", (shp[0],), "b")]
        out += [(ln1 + ".weight", (H,), "g"), (ln1 + ".bias", (H,), "b")]
        out += [(p + "intermediate.dense.weight", (F, H), "w"), (p + "intermediate.dense.bias", (F,), "b"),
                (p + "output.dense.weight", (H, F), "w"), (p + "output.dense.bias", (H,), "b"),
                (p + "output.LayerNorm.weight", (H,), "g"), (p + "output.LayerNorm.bias", (H,), "b")]
    if cfg.arch == "mpnet":
        out.append(("encoder.relative_attention_bias.weight", (cfg.rel_buckets, cfg.heads), "r"))
    return out


def synthetic_weights(preset: str, cfg: EncoderConfig | None = None, bf16_exact: bool = True) -> Dict[str, np.ndarray]:
    """Deterministic float32 weights for a preset.  Matrices/embeddings ~ U(-a, a) with std 0.02
    scaled up x2 for linear layers so that activations are not degenerate; biases std 0.02;
    LayerNorm gamma = 1 + U(-.1,.1), beta = U(-.05,.05); relative bias U(-.5,.5).
    With ``bf16_exact`` every value is rounded to a bf16-representable float32, so the fp32 CPU
    reference and the bf16 GPU path consume *identical* weights."""
    cfg = cfg or PRESETS[preset]
    out = {}
    a = 0.02 * np.sqrt(3.0)
    for name, shape, kind in weight_names(cfg):
        n = int(np.prod(shape))
        u = uniform01(f"{preset}/{name}", n) * np.float32(2.0) - np.float32(1.0)
        if kind == "w":
            scale = a * (2.0 if name.startswith("encoder.") else 1.0)
            x = u * np.float32(scale)
        elif kind == "b":
            x = u * np.float32(a if "LayerNorm" not in name else 0.05)
        elif kind == "g":
            x = np.float32(1.0) + u * np.float32(0.1)
        else:
            x = u * np.float32(0.5)
        x = x.reshape(shape).astype(np.float32)
        out[name] = bf16_round(x) if bf16_exact else x
    return out


# --------------------------------------------------------------------------- synthetic text
def synthetic_vocab(size: int = 30522) -> Dict[str, int]:
    """WordPiece-style vocab: BERT special tokens at their usual ids, then ``w00000``-style
    whole-word entries (no '##' pieces: one token per word, so token counts are predictable)."""
    vocab = {"[PAD]": 0}
    for i in range(1, 100):
        vocab[f"[unused{i}]"] = i
    vocab["[UNK]"], vocab["[CLS]"], vocab["[SEP]"], vocab["[MASK]"] = 100, 101, 102, 103
    i = 104
    while i < size:
        vocab[f"w{i:05d}"] = i
        i += 1
    return vocab


def synthetic_lengths(n: int, seed: str = "sent1234", median: int = 12, sigma: float = 0.6,
                      max_words: int = 254) -> np.ndarray:
    """Clipped log-normal word counts, median 12."""
    z = normal(seed + "/len", n)
    return np.clip(np.rint(np.exp(np.log(median) + sigma * z.astype(np.float64))), 1, max_words).astype(np.int64)


def synthetic_sentences(n: int, seed: str = "sent1234", vocab_size: int = 30522, max_words: int = 254) -> List[str]:
    lens = synthetic_lengths(n, seed, max_words=max_words)
    ids = randint(seed + "/tok", int(lens.sum()), 104, vocab_size)
    out, p = [], 0
    for L in lens:
        out.append(" ".join(f"w{t:05d}" for t in ids[p:p + L]))
        p += L
    return out


def synthetic_token_batch(n: int, seed: str = "sent1234", vocab_size: int = 30522, max_len: int = 256,
                          cls_id: int = 101, sep_id: int = 102):
    """Tokenised form of ``synthetic_sentences`` without going through a tokenizer:
    returns (flat token ids int32 [T], cu_seqlens int32 [n+1]); each row is CLS w.. SEP."""
    lens = synthetic_lengths(n, seed, max_words=max_len - 2)
    words = randint(seed + "/tok", int(lens.sum()), 104, vocab_size)
    tl = lens + 2
    cu = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(tl, out=cu[1:])
    flat = np.empty(int(cu[-1]), dtype=np.int32)
    starts = cu[:-1]
    flat[starts] = cls_id
    flat[cu[1:] - 1] = sep_id
    mask = np.ones(int(cu[-1]), dtype=bool)
    mask[starts] = False
    mask[cu[1:] - 1] = False
    flat[mask] = words.astype(np.int32)
    return flat, cu.astype(np.int32)


def synthetic_embeddings(n: int, d: int, seed: str) -> np.ndarray:
    """n x d standard-normal rows, L2-normalised in float64, rounded to bf16-exact float32
    (the search-only benchmark input of SURVEY.md §8(d))."""
    x = normal(seed, n * d).reshape(n, d).astype(np.float64)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-8)
    return bf16_round(x.astype(np.float32))
