"""NativeEncoder — the object that stands where the reference keeps a HuggingFace ``AutoModel``
(``context_embedder``, /root/reference/src/models/modeling.py:25,
/root/reference/src/models/sentence_encoder.py:33,107-108,118).

It owns a ``tsim_encoder`` handle of libtsim.so (weights in HBM as bf16, activation workspace) and runs the
encoder forward on *packed* tokens.  ``__call__(input_ids=..., attention_mask=...)`` keeps the HF contract the
wrappers rely on: element ``[0]`` of the result is ``last_hidden_state`` of shape ``[B, S, H]``.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from .presets import EncoderConfig, PRESETS, synthetic_weights


def _f32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().float().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def layer_names(cfg: EncoderConfig, l: int) -> Dict[str, str]:
    p = f"encoder.layer.{l}."
    if cfg.arch == "bert":
        a = {"q": p + "attention.self.query", "k": p + "attention.self.key", "v": p + "attention.self.value",
             "o": p + "attention.output.dense", "ln1": p + "attention.output.LayerNorm"}
    else:
        a = {"q": p + "attention.attn.q", "k": p + "attention.attn.k", "v": p + "attention.attn.v",
             "o": p + "attention.attn.o", "ln1": p + "attention.LayerNorm"}
    a.update({"f1": p + "intermediate.dense", "f2": p + "output.dense", "ln2": p + "output.LayerNorm"})
    return a


class NativeEncoder:
    def __init__(self, cfg: EncoderConfig, weights: Dict[str, np.ndarray], max_tokens: int = 65536,
                 max_seqs: int = 8192, device: Optional[torch.device] = None, weight_dtype: str = "bf16"):
        """``weight_dtype``: "bf16" (default) or "mxfp8" — projections on OCP MXFP8 operands (e4m3 + one power-of-two
        scale per 32 elements) through the block-scaled fp8 MFMA; base-size models only (hidden, ffn % 256 == 0)."""
        if weight_dtype not in ("bf16", "mxfp8"):
            raise ValueError(f"weight_dtype must be 'bf16' or 'mxfp8', got {weight_dtype!r}")
        self.weight_dtype = weight_dtype
        if not torch.cuda.is_available():
            raise _lib.TsimError("NativeEncoder needs an MI355X: torch.cuda.is_available() is False and "
                                 "there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.max_tokens, self.max_seqs = int(max_tokens), int(max_seqs)
        # what BaseEncoderModel.config / get_sentence_embedding_dimension read (modeling.py:65-77)
        self.config = SimpleNamespace(hidden_size=cfg.hidden, dim=cfg.hidden, num_hidden_layers=cfg.num_layers,
                                      num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                                      vocab_size=cfg.vocab, max_position_embeddings=cfg.max_pos,
                                      model_type=cfg.arch)
        w = {k: _f32(v) for k, v in weights.items() if not k.endswith("position_ids")}
        keep = []
        layers = (_lib.LayerWeightsC * cfg.num_layers)()
        for l in range(cfg.num_layers):
            n = layer_names(cfg, l)
            lw = layers[l]
            for field, key in (("wq", n["q"] + ".weight"), ("bq", n["q"] + ".bias"), ("wk", n["k"] + ".weight"),
                               ("bk", n["k"] + ".bias"), ("wv", n["v"] + ".weight"), ("bv", n["v"] + ".bias"),
                               ("wo", n["o"] + ".weight"), ("bo", n["o"] + ".bias"),
                               ("ln1_g", n["ln1"] + ".weight"), ("ln1_b", n["ln1"] + ".bias"),
                               ("w1", n["f1"] + ".weight"), ("b1", n["f1"] + ".bias"),
                               ("w2", n["f2"] + ".weight"), ("b2", n["f2"] + ".bias"),
                               ("ln2_g", n["ln2"] + ".weight"), ("ln2_b", n["ln2"] + ".bias")):
                if key not in w:
                    raise KeyError(f"missing weight {key}")
                keep.append(w[key])
                setattr(lw, field, _ptr(w[key]))
        ew = _lib.EncoderWeightsC()
        ew.word_emb = _ptr(w["embeddings.word_embeddings.weight"])
        ew.pos_emb = _ptr(w["embeddings.position_embeddings.weight"])
        if cfg.arch == "bert":
            ew.type_emb = _ptr(w["embeddings.token_type_embeddings.weight"])
        ew.emb_ln_g = _ptr(w["embeddings.LayerNorm.weight"])
        ew.emb_ln_b = _ptr(w["embeddings.LayerNorm.bias"])
        if cfg.arch == "mpnet":
            ew.rel_bias = _ptr(w["encoder.relative_attention_bias.weight"])
        ew.layers = layers
        cc = _lib.EncoderConfigC(arch=_lib.ARCH_BERT if cfg.arch == "bert" else _lib.ARCH_MPNET,
                                 num_layers=cfg.num_layers, hidden=cfg.hidden, heads=cfg.heads, ffn=cfg.ffn,
                                 vocab=cfg.vocab, max_pos=cfg.max_pos, pad_id=cfg.pad_id,
                                 rel_buckets=cfg.rel_buckets, ln_eps=cfg.ln_eps, max_tokens=self.max_tokens,
                                 max_seqs=self.max_seqs,
                                 weight_dtype=_lib.W_MXFP8 if weight_dtype == "mxfp8" else _lib.W_BF16)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tsim_encoder_create(C.byref(cc), C.byref(ew), C.byref(handle)), "encoder_create")
        self._h = handle
        self._weights_host = w  # float32 source weights: what save_pretrained writes (the handle holds bf16 / fp8 copies)

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_preset(cls, preset: str, **kw) -> "NativeEncoder":
        """Architecture preset with regenerable synthetic weights (no checkpoints exist offline)."""
        return cls(PRESETS[preset], synthetic_weights(preset), **kw)

    @classmethod
    def from_pretrained(cls, path: str, **kw) -> "NativeEncoder":
        """Local HF directory: config.json + model.safetensors (or pytorch_model.bin, loaded weights_only)."""
        from .weights import load_hf_dir
        cfg, w = load_hf_dir(path)
        return cls(cfg, w, **kw)

    def save_pretrained(self, path: str) -> None:
        """config.json + model.safetensors of the float32 source weights (what ``from_pretrained(path)`` reads back) — the
        ``context_embedder.save_pretrained(path)`` of /root/reference/src/models/modeling.py:56."""
        from .weights import save_hf_dir
        save_hf_dir(path, self.cfg, self._weights_host)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().tsim_encoder_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ input validation
    ERR_BITS = {1: "a token id outside [0, vocab_size)", 2: "a position id outside the position table",
                4: "a sequence longer than the max_len passed to forward_packed"}

    @staticmethod
    def check_lengths(cfg: EncoderConfig, max_len: int) -> None:
        """HF raises IndexError when a sequence needs a position row the table does not have: BERT rows 0..len-1, MPNet
        rows pad_id+1..pad_id+len (max_pos 514 holds 512 tokens).  Raised here, before any launch."""
        need = int(max_len) + (cfg.pad_id + 1 if cfg.arch == "mpnet" else 0)
        if need > cfg.max_pos:
            raise ValueError(f"sequences of {max_len} tokens need position rows up to {need - 1}; "
                             f"{cfg.arch} table has {cfg.max_pos} (max {cfg.max_pos - need + int(max_len)} tokens)")

    def check(self) -> None:
        """Raise if any forward since the last check saw an out-of-range token id / position id or a sequence longer than
        its promised max_len (the kernels clamp and go on; HF would have raised IndexError).  Synchronises the stream."""
        flags = C.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tsim_encoder_error_flags(self._h, C.byref(flags),
                                                           torch.cuda.current_stream(self.device).cuda_stream), "encoder_error_flags")
        if flags.value:
            what = "; ".join(msg for bit, msg in self.ERR_BITS.items() if flags.value & bit)
            raise IndexError(f"encoder input out of range: {what}")

    # torch.nn.Module-ish no-ops used by the reference wrappers (`self.to(device)`, `self.eval()`)
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    # ------------------------------------------------------------------ packed forward
    def positions(self, flat_ids: torch.Tensor, cu: torch.Tensor, cols: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(position-embedding rows, padded-batch columns) for packed tokens.
        BERT: row = column (bert_of_theseus.py:199-200).  MPNet: cumsum(ids != pad) * (ids != pad) + pad over the
        tokens present (mpnet create_position_ids_from_input_ids)."""
        T = flat_ids.numel()
        seq_of = torch.repeat_interleave(torch.arange(cu.numel() - 1, device=cu.device), (cu[1:] - cu[:-1]).long(),
                                         output_size=T)
        if cols is None:
            cols = torch.arange(T, device=cu.device, dtype=torch.int32) - cu[seq_of.long()].to(torch.int32)
        if self.cfg.arch == "bert":
            return cols.to(torch.int32), cols.to(torch.int32)
        ne = (flat_ids != self.cfg.pad_id).to(torch.int32)
        csum = torch.cumsum(ne, 0, dtype=torch.int32)
        start = torch.zeros(cu.numel() - 1, dtype=torch.int32, device=cu.device)
        if T:
            excl = csum - ne
            start = excl[cu[:-1].clamp(max=max(T - 1, 0)).long()]
        pos = (csum - start[seq_of.long()]) * ne + self.cfg.pad_id
        return pos.to(torch.int32), cols.to(torch.int32)

    def forward_packed(self, flat_ids: torch.Tensor, cu: torch.Tensor, pos: Optional[torch.Tensor] = None,
                       cols: Optional[torch.Tensor] = None, max_len: Optional[int] = None, pooled: bool = True,
                       unit: bool = False, hidden: bool = False, rho: Optional[torch.Tensor] = None):
        """flat_ids int32 [T], cu int32 [B+1] on the GPU.  Returns dict with 'pooled' f32 [B,H],
        'unit' float16 [B,pad_dim(H)] (L2-normalised rows for the search kernel), 'hidden' bf16 [T,H] as requested.
        ``rho``: a device float32 word raised to the largest rounding residual of the unit rows (ops.l2norm_rows)."""
        ops._need_gpu(flat_ids, cu)
        flat_ids = flat_ids.to(torch.int32).contiguous()
        cu = cu.to(torch.int32).contiguous()
        T, B = flat_ids.numel(), cu.numel() - 1
        if pos is None:
            pos, cols2 = self.positions(flat_ids, cu, cols)
            cols = cols2 if cols is None else cols
        pos = pos.to(torch.int32).contiguous()
        cols = None if cols is None else cols.to(torch.int32).contiguous()
        if max_len is None:   # longest sequence in the batch: sizes the attention grid (one host sync; pass it to avoid)
            max_len = int((cu[1:] - cu[:-1]).max().item()) if B else 0
        self.check_lengths(self.cfg, max_len)
        H = self.cfg.hidden
        out = {}
        dev = flat_ids.device
        p = torch.empty((B, H), dtype=torch.float32, device=dev) if pooled else None
        u = torch.empty((B, ops.pad_dim(H)), dtype=ops.UNIT_DTYPE, device=dev) if unit else None
        hd = torch.empty((T, H), dtype=torch.bfloat16, device=dev) if hidden else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().tsim_encoder_forward(
                self._h, flat_ids.data_ptr(), pos.data_ptr(), cols.data_ptr() if cols is not None else None,
                cu.data_ptr(), T, B, int(max_len), p.data_ptr() if p is not None else None,
                u.data_ptr() if u is not None else None, u.shape[1] if u is not None else 0,
                rho.data_ptr() if (rho is not None and u is not None) else None,
                hd.data_ptr() if hd is not None else None, torch.cuda.current_stream(dev).cuda_stream),
                "encoder_forward")
        if pooled:
            out["pooled"] = p
        if unit:
            out["unit"] = u
        if hidden:
            out["hidden"] = hd
        return out

    # ------------------------------------------------------------------ padded (HF-style) call
    @staticmethod
    def pack(input_ids: torch.Tensor, attention_mask: torch.Tensor):
        m = attention_mask.bool()
        lens = m.sum(1)
        cu = torch.zeros(m.shape[0] + 1, dtype=torch.int32, device=m.device)
        cu[1:] = torch.cumsum(lens, 0)
        nz = m.nonzero(as_tuple=False)          # row-major: exactly the packed order
        flat = input_ids[m].to(torch.int32)
        cols = nz[:, 1].to(torch.int32)
        return flat, cu, cols, nz

    def __call__(self, input_ids=None, attention_mask=None, **kwargs):
        """HF AutoModel contract used by the wrappers: returns (last_hidden_state [B,S,H] float32,).
        Positions whose mask is 0 come back as zeros (the reference never reads them: the pooler multiplies
        by the mask, modules.py:165)."""
        ops._need_gpu(input_ids)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        B, S = input_ids.shape
        flat, cu, cols, nz = self.pack(input_ids, attention_mask)
        if self.cfg.arch == "mpnet":
            # position ids come from input_ids over the WHOLE padded row (masked non-pad tokens count too)
            ne = (input_ids != self.cfg.pad_id).to(torch.int32)
            pos_full = torch.cumsum(ne, 1, dtype=torch.int32) * ne + self.cfg.pad_id
            pos = pos_full[attention_mask.bool()].to(torch.int32)
        else:
            pos = cols
        out = torch.zeros((B, S, self.cfg.hidden), dtype=torch.float32, device=input_ids.device)
        T = flat.numel()
        for s in range(0, max(B, 1), self.max_seqs):  # capacity-sized slices of the batch
            e = min(B, s + self.max_seqs)
            t0, t1 = int(cu[s]), int(cu[e])
            if t1 - t0 > self.max_tokens:
                raise ValueError(f"batch slice has {t1 - t0} tokens > encoder capacity {self.max_tokens}")
            if t1 > t0:
                r = self.forward_packed(flat[t0:t1], (cu[s:e + 1] - cu[s]), pos[t0:t1], cols[t0:t1], S,
                                        pooled=False, hidden=True)
                out[nz[t0:t1, 0], nz[t0:t1, 1]] = r["hidden"].float()
        return (out,)
