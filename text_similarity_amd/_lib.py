"""ctypes binding of libtsim.so (include/tsim.h).  The library is the product: nothing here falls back
to torch or to the CPU oracle — if the shared object is missing or a call fails, the caller gets an
exception."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSIM_LIB") or os.path.join(_HERE, "libtsim.so")   # TSIM_LIB: a variant build (build.py TSIM_BUILD_TAG)

TSIM_F32, TSIM_BF16 = 0, 1
ARCH_BERT, ARCH_MPNET = 0, 1
W_BF16, W_MXFP8 = 0, 1


class TsimError(RuntimeError):
    pass


class EncoderConfigC(C.Structure):
    _fields_ = [("arch", C.c_int32), ("num_layers", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32),
                ("ffn", C.c_int32), ("vocab", C.c_int32), ("max_pos", C.c_int32), ("pad_id", C.c_int32),
                ("rel_buckets", C.c_int32), ("ln_eps", C.c_float), ("max_tokens", C.c_int32),
                ("max_seqs", C.c_int32), ("weight_dtype", C.c_int32)]


_FP = C.POINTER(C.c_float)


class LayerWeightsC(C.Structure):
    _fields_ = [(n, _FP) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b",
                                   "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")]


class EncoderWeightsC(C.Structure):
    _fields_ = [("word_emb", _FP), ("pos_emb", _FP), ("type_emb", _FP), ("emb_ln_g", _FP), ("emb_ln_b", _FP),
                ("rel_bias", _FP), ("layers", C.POINTER(LayerWeightsC))]


_lib = None

_SIGS = {
    "tsim_version": (C.c_int, []),
    "tsim_last_error": (C.c_char_p, []),
    "tsim_pad_dim": (C.c_int, [C.c_int]),
    "tsim_l2norm_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_int,
                                   C.c_float, C.c_void_p, C.c_void_p]),
    "tsim_cosine_topk_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int]),
    "tsim_cosine_topk_plan": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "tsim_cosine_topk_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "tsim_cosine_topk": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tsim_time_next_topk": (None, [C.c_void_p, C.c_void_p]),
    "tsim_topk_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "tsim_topk_merge_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsim_wordpiece_create": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "tsim_wordpiece_destroy": (None, [C.c_void_p]),
    "tsim_wordpiece_encode": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p]),
    "tsim_cos_sim": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "tsim_mean_pool": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "tsim_quantize_mxfp8": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsim_gemm_mxfp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
    "tsim_encoder_create": (C.c_int, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC),
                                      C.POINTER(C.c_void_p)]),
    "tsim_encoder_destroy": (None, [C.c_void_p]),
    "tsim_encoder_error_flags": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "tsim_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
}

DECLARED_SYMBOLS = tuple(_SIGS)


def lib() -> C.CDLL:
    """Load libtsim.so once.  Raises TsimError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TsimError(f"{LIB_PATH} not found: build it with `python -m text_similarity_amd.build` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                def fn(*a, _n=name, **kw):
                    raise TsimError(f"libtsim.so does not export {_n}: rebuild with python -m text_similarity_amd.build")
                setattr(L, name, fn)
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().tsim_last_error().decode(errors="replace")
        exc = ValueError if rc in (1, 4) else TsimError
        raise exc(f"{what or 'tsim'} failed (code {rc}): {msg}")
