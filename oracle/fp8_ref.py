"""ORACLE — test infrastructure only.  Never imported by the product package.

CPU restatement (numpy) of the MXFP8 operand format the fp8 encoder variant (BASELINE.json configs[4]:
"bert-base-uncased fp8 weights (CDNA4 fp8 MFMA)") feeds to v_mfma_scale_f32_32x32x64_f8f6f4, and of the encoder
forward with every projection computed on so-quantised operands.

The reference has no fp8 path (its encoder is HF fp32, /root/reference/src/models/sentence_encoder.py:33), so this
variant has no reference golden of its own: it is pinned (a) element format — against the OCP 8-bit floating point
specification's e4m3 value table, restated in `e4m3_values` and checked against torch.float8_e4m3fn here
(tests/test_oracle_golden.py), and (b) end to end — against the fp32 goldens under a stated fp8 tolerance
(tests/test_fp8_gpu.py).  "parity unpinned" beyond that tolerance.

Format (OCP Microscaling Formats v1.0, MXFP8 E4M3): blocks of 32 consecutive elements along the contraction axis share
one power-of-two scale X = 2^(floor(log2(amax)) - 8) stored as E8M0 (byte = exponent + 127; amax = 0 -> byte 127);
elements are x / X clamped to +-448 and rounded to e4m3 (1-4-3, bias 7, no infinities, S.1111.111 = NaN) nearest-even.
"""
from __future__ import annotations

import numpy as np

BLOCK = 32
E4M3_MAX = 448.0


def e4m3_values() -> np.ndarray:
    """float32 value of every e4m3fn byte (NaN for 0x7f / 0xff)."""
    b = np.arange(256)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    v = np.where(s == 1, -v, v)
    v[(b & 0x7F) == 0x7F] = np.nan
    return v.astype(np.float32)


def f32_to_e4m3(v: np.ndarray) -> np.ndarray:
    """Round float values with |v| <= 448 to e4m3 bytes, nearest, ties to even (subnormals kept)."""
    v = np.asarray(v, dtype=np.float32)
    a = np.abs(v).astype(np.float64)
    sign = (np.signbit(v)).astype(np.uint8) << 7
    mant, exp = np.frexp(a)                     # a = mant * 2^exp, mant in [0.5, 1)
    e = exp - 1                                 # floor(log2 a) for a > 0
    normal = a >= 2.0 ** -6
    # normal: q = RNE((a / 2^e - 1) * 8) in 0..8 ; subnormal: q = RNE(a * 2^9) in 0..8 (8 = the smallest normal)
    qn = np.rint((mant * 2.0 - 1.0) * 8.0)
    byte_n = ((e + 7).astype(np.int64) << 3) + qn.astype(np.int64)   # q == 8 carries into the exponent field
    byte_s = np.rint(a * 2.0 ** 9).astype(np.int64)
    out = np.where(normal, byte_n, byte_s)
    out = np.minimum(out, 0x7E)                 # 448 is the largest finite value
    return (out.astype(np.uint8) | sign).astype(np.uint8)


def mx_quantize(x: np.ndarray):
    """x [..., K] (K % 32 == 0) -> (e4m3 bytes [..., K] uint8, E8M0 scale bytes [..., K/32] uint8)."""
    x = np.asarray(x, dtype=np.float32)
    K = x.shape[-1]
    assert K % BLOCK == 0
    xb = x.reshape(x.shape[:-1] + (K // BLOCK, BLOCK))
    amax = np.abs(xb).max(-1)
    _, exp = np.frexp(amax.astype(np.float64))
    sexp = np.where(amax > 0, exp - 1 - 8, 0)                       # floor(log2 amax) - emax(e4m3)
    sexp = np.clip(sexp, -127, 127)
    scale = np.ldexp(1.0, sexp).astype(np.float64)
    y = np.clip(xb.astype(np.float64) / scale[..., None], -E4M3_MAX, E4M3_MAX).astype(np.float32)
    q = f32_to_e4m3(y).reshape(x.shape)
    return q, (sexp + 127).astype(np.uint8)


def mx_dequantize(q: np.ndarray, s: np.ndarray) -> np.ndarray:
    vals = e4m3_values()[q]
    K = q.shape[-1]
    sc = np.ldexp(1.0, s.astype(np.int64) - 127).astype(np.float32)
    return (vals.reshape(q.shape[:-1] + (K // BLOCK, BLOCK)) * sc[..., None]).reshape(q.shape).astype(np.float32)


def fake_quant(x: np.ndarray) -> np.ndarray:
    return mx_dequantize(*mx_quantize(x))


_wq_cache = {}


def mx_linear(t, weight, bias):
    """Drop-in for F.linear in oracle.encoder_ref.encoder_forward: both operands through MXFP8 along the contraction
    axis (the activation is first rounded to bf16, as it is stored on the device), product and sum in float32."""
    import torch
    from .search_ref import bf16_round
    tq = fake_quant(bf16_round(t.detach().numpy().astype(np.float32)))
    key = (weight.data_ptr(), tuple(weight.shape))          # weights are views of the caller's numpy arrays
    wq = _wq_cache.get(key)
    if wq is None:
        wq = _wq_cache[key] = torch.from_numpy(fake_quant(weight.detach().numpy().astype(np.float32)))
    return torch.from_numpy(tq) @ wq.T + bias
