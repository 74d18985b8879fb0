"""GPU parity of the MXFP8 encoder variant (BASELINE.json configs[4]: bert-base-uncased, fp8 weights, CDNA4 fp8 MFMA).

* the quantiser (bytes and block scales) is integer work: bit-exact against oracle/fp8_ref.mx_quantize;
* the encoder computes every projection on MXFP8 operands (3 significand bits per element): compared with
  (0) one projection on given MXFP8 operands against the oracle's dequantise-and-multiply in float64: products of e4m3
      values and power-of-two scales are exact in fp32, only the summation order differs: |err| <= 1e-4 * (1 + |ref|);
  (a) the oracle running the same algorithm (oracle.fp8_ref.mx_linear inside oracle.encoder_ref) — 12 layers amplify
      every flipped e4m3 rounding (bf16 storage of activations, fp32 summation order), so two correct fp8 runs agree only
      to fp8 accuracy: row cosine >= 0.98 (measured 0.987);
  (b) the fp32 reference golden — the cost of fp8 itself on these synthetic (random, unstructured) weights, stated:
      row cosine >= 0.95 (the fp8 oracle itself measures 0.967 against the fp32 golden).
The reference has no fp8 path, so (b) is the only reference-anchored statement: "parity unpinned" beyond it."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import encoder_ref, fp8_ref
from text_similarity_amd import ops, presets
from text_similarity_amd.native_encoder import NativeEncoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cos_rows(a, b):
    num = (a * b).sum(1)
    return num / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-30)


@pytest.mark.parametrize("rows,K", [(1, 32), (37, 768), (300, 3072)])
def test_quantizer_bit_exact(rows, K):
    rng = np.random.default_rng(rows * 1000 + K)
    x = (rng.standard_normal((rows, K)) * np.exp(rng.uniform(-20, 12, (rows, 1)))).astype(np.float32)
    x[0, :32] = 0.0                                   # an all-zero block: scale byte 127, zero bytes
    if rows > 2:
        x[1, 5] = 3.0e38                              # huge: exponent clamp of the shared scale
        x[2, :64] = 1e-40                             # fp32-denormal inputs flush to bf16 zero or denormal
    xb = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    q, s = ops.quantize_mxfp8(xb)
    torch.cuda.synchronize()
    rq, rs = fp8_ref.mx_quantize(xb.float().cpu().numpy())
    np.testing.assert_array_equal(s.cpu().numpy(), rs)
    np.testing.assert_array_equal(q.cpu().numpy(), rq)


@pytest.mark.parametrize("M,N,K", [(1, 256, 256), (300, 768, 768), (517, 256, 3072)])
def test_mxfp8_gemm_vs_oracle(M, N, K):
    rng = np.random.default_rng(M + N + K)
    x = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-3, 3, (M, 1)))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    xq, xs = fp8_ref.mx_quantize(x)
    wq, ws = fp8_ref.mx_quantize(w)
    out = ops.gemm_mxfp8(*(torch.from_numpy(a).to(DEV) for a in (xq, xs, wq, ws, bias)))
    torch.cuda.synchronize()
    ref = fp8_ref.mx_dequantize(xq, xs).astype(np.float64) @ fp8_ref.mx_dequantize(wq, ws).astype(np.float64).T + bias
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * (1 + np.abs(ref).max())
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_fp8_encoder_bert_base_vs_oracle_and_fp32_golden():
    preset = "bert-base-uncased"
    g = golden(f"encoder_{preset}.npz")
    cfg, w = presets.PRESETS[preset], presets.synthetic_weights(preset)
    enc = NativeEncoder.from_preset(preset, max_tokens=4096, max_seqs=64, weight_dtype="mxfp8")
    flat, cu = g["flat_ids"], g["cu_seqlens"].astype(np.int64)
    r = enc.forward_packed(torch.from_numpy(flat).to(DEV), torch.from_numpy(cu.astype(np.int32)).to(DEV), pooled=True)
    torch.cuda.synchronize()
    p = r["pooled"].cpu().numpy()
    assert np.isfinite(p).all()
    ref8 = encoder_ref.encode_packed(cfg, w, flat, cu, batch_size=8, linear=fp8_ref.mx_linear)
    err8, cos8 = np.abs(p - ref8).max(), _cos_rows(p, ref8).min()
    cos32 = _cos_rows(p, g["pooled"]).min()
    print(f"fp8 {preset}: vs fp8 oracle max|err|={err8:.4f} min cos={cos8:.6f}; vs fp32 golden min cos={cos32:.6f}")
    assert cos8 >= 0.98
    assert cos32 >= 0.95


def test_fp8_rejects_small_models():
    with pytest.raises(Exception):
        NativeEncoder.from_preset("all-MiniLM-L6-v2", max_tokens=256, max_seqs=8, weight_dtype="mxfp8")
