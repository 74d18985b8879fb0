"""Pins oracle/ (the CPU restatement) to golden vectors produced by running the reference
(tools/make_golden.py).  CPU only."""
import zlib

import numpy as np
import pytest

from conftest import golden
from oracle import encoder_ref, search_ref
from text_similarity_amd import presets


def _crc(w):
    c = 0
    for k in sorted(w):
        c = zlib.crc32(np.ascontiguousarray(w[k]).tobytes(), c)
    return np.uint32(c)


@pytest.mark.parametrize("preset", ["tiny-bert", "tiny-mpnet"])
def test_tiny_encoder_matches_reference(preset):
    g = golden(f"encoder_{preset}.npz")
    cfg = presets.PRESETS[preset]
    w = presets.synthetic_weights(preset)
    assert _crc(w) == g["weights_crc"], "synthetic weight generator drifted from the golden fixtures"
    import torch
    with torch.no_grad():
        h = encoder_ref.encoder_forward(cfg, w, g["input_ids"], g["attention_mask"]).numpy()
    # Rows with at least one valid token: every position (masked ones too) must match HF.
    # A row whose mask is all zero is a don't-care: the reference's pinned transformers 4.2 adds
    # (1-m)*-10000 (bert_of_theseus.py:972), which shifts all keys equally, while the installed 5.x
    # masks with finfo.min; the pooled output of such a row is exactly 0 either way (checked below).
    live = g["attention_mask"].sum(1) > 0
    np.testing.assert_allclose(h[live], g["last_hidden_state"][live], rtol=0, atol=2e-5)
    p = encoder_ref.encode(cfg, w, g["input_ids"], g["attention_mask"]).numpy()
    np.testing.assert_allclose(p, g["pooled"], rtol=0, atol=1e-5)
    assert np.all(p[3] == 0.0)  # all-zero mask row


@pytest.mark.parametrize("preset", ["all-MiniLM-L6-v2", "all-mpnet-base-v2", "bert-base-uncased"])
def test_preset_encoder_matches_reference(preset):
    g = golden(f"encoder_{preset}.npz")
    cfg = presets.PRESETS[preset]
    w = presets.synthetic_weights(preset)
    assert _crc(w) == g["weights_crc"]
    out = encoder_ref.encode_packed(cfg, w, g["flat_ids"], g["cu_seqlens"], batch_size=16)
    np.testing.assert_allclose(out, g["pooled"], rtol=0, atol=3e-5)


def test_pool_edge_cases():
    g = golden("pool_edge.npz")
    p = encoder_ref.mean_pool(g["hidden"], g["attention_mask"]).numpy()
    np.testing.assert_array_equal(p, g["pooled"])


def test_cos_sim_and_cosine_similarity():
    g = golden("search_cos.npz")
    np.testing.assert_allclose(search_ref.cos_sim(g["a"], g["b"]), g["cos_sim"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(search_ref.cos_sim(g["a"][0], g["b"]), g["cos_sim_1d"], rtol=0, atol=5e-7)
    z = search_ref.cos_sim(g["a"], g["b_zero"])
    assert np.array_equal(np.isnan(z), np.isnan(g["cos_sim_zero"]))
    assert np.isnan(z[:, 5]).all()
    np.testing.assert_allclose(np.nan_to_num(z), np.nan_to_num(g["cos_sim_zero"]), rtol=0, atol=5e-7)
    for q in range(4):
        qe = np.broadcast_to(g["a"][q], g["b_zero"].shape)
        r = search_ref.cosine_similarity_rows(qe, g["b_zero"])
        np.testing.assert_allclose(r, g["cosine_similarity_rows"][q], rtol=0, atol=5e-7)
        assert r[5] == 0.0
    # A7 == A8 away from zero rows
    np.testing.assert_allclose(g["cosine_similarity_rows"][0][:5], g["cos_sim_zero"][0][:5], atol=5e-7)


def _tie_aware_equal(values_ref, idx_ref, values, idx, scores, atol):
    """torch.topk's tie order is implementation-defined: require equal value lists and that every
    index we return carries the value torch reports at that rank."""
    np.testing.assert_allclose(values, values_ref, rtol=0, atol=atol)
    got = np.take_along_axis(scores, idx, 1)
    np.testing.assert_allclose(got, values_ref, rtol=0, atol=atol)
    for r in range(idx.shape[0]):
        assert len(set(idx[r].tolist())) == idx.shape[1]


@pytest.mark.parametrize("k", [1, 3, 10, 300])
def test_topk_vs_torch(k):
    g = golden("search_topk.npz")
    sc = g["scores_mm"]
    v, i = search_ref.topk_rows(sc, k)
    _tie_aware_equal(g[f"topk{k}_values"], g[f"topk{k}_indices"], v, i, sc, 0.0)
    # defined tie rule: query 0 is corpus row 7, duplicated at rows 40, 41, 200
    if k >= 3:
        assert i[0, :min(k, 4)].tolist() == [7, 40, 41, 200][:min(k, 4)]
    if k == 300:  # k == N: a permutation
        assert sorted(i[0].tolist()) == list(range(300))


def test_canonical_scores_close_to_reference_loop():
    """Unit-row-only callers get the inner product of L2-normalised rows *as stored* (half precision; north_star: "MFMA
    syrk-style matmul over L2-normalised rows").  A half-rounded unit row typically has norm 1 +- ~2e-5; this fixture is
    the bad case — its rows are bf16-exact with norms 1 - 2e-4, and x / 0.9998 rounds straight back to x in half precision
    for EVERY element, so the stored row keeps that norm and the score is 4e-4 low.  Still inside the 1e-3 fp32 tolerance
    north_star states, asserted here.  Callers that hold the float32 embeddings get the reference's value itself
    (cosine_topk_f32, exact)."""
    g = golden("search_topk.npz")
    q, c = g["queries"], g["corpus"]
    assert np.array_equal(search_ref.unit_rows(q)[0], q[0])      # renormalising + half rounding gives row 0 back as it was
    v, i = search_ref.cosine_topk(q, c, 10)
    np.testing.assert_allclose(v, g["loop_top10_values"], rtol=0, atol=1e-3)
    can = search_ref.canonical_scores(q, c)
    # dividing by the true norms of the stored rows recovers the reference's cosine to fp32 rounding
    nq = np.sqrt((q.astype(np.float64) ** 2).sum(1))[:, None]
    nc = np.sqrt((c.astype(np.float64) ** 2).sum(1))[None, :]
    cosv = (can.astype(np.float64) / (nq * nc)).astype(np.float32)
    np.testing.assert_allclose(np.take_along_axis(cosv, g["loop_top10_indices"], 1), g["loop_top10_values"],
                               rtol=0, atol=3e-7)
    assert i[0, :4].tolist() == [7, 40, 41, 200]


def test_merge_topk_equals_unsharded():
    g = golden("search_topk.npz")
    q, c = g["queries"], g["corpus"]
    full_v, full_i = search_ref.cosine_topk(q, c, 10)
    parts = [search_ref.cosine_topk(q, c[s:s + 77], 10, idx_offset=s) for s in range(0, 300, 77)]
    mv, mi = search_ref.merge_topk([p[0] for p in parts], [p[1] for p in parts], 10)
    np.testing.assert_array_equal(mi, full_i)
    np.testing.assert_array_equal(mv, full_v)


def test_e2e_config1_encode_and_search():
    g = golden("e2e_config1.npz")
    preset = "all-MiniLM-L6-v2"
    cfg = presets.PRESETS[preset]
    w = presets.synthetic_weights(preset)
    assert _crc(w) == g["weights_crc"]
    # token ids: the synthetic generator reproduces what the reference's tokenizer call produced
    flat, cu = presets.synthetic_token_batch(1000, seed="sent1234", vocab_size=cfg.vocab, max_len=256)
    np.testing.assert_array_equal(cu, g["cu_seqlens"])
    np.testing.assert_array_equal(flat, g["flat_ids"])
    sub = np.arange(0, 1000, 25)
    cu_sub = np.zeros(len(sub) + 1, dtype=np.int64)
    parts = [flat[cu[r]:cu[r + 1]] for r in sub]
    np.cumsum([len(p) for p in parts], out=cu_sub[1:])
    emb = encoder_ref.encode_packed(cfg, w, np.concatenate(parts), cu_sub, batch_size=16)
    np.testing.assert_allclose(emb, g["embeddings"][sub], rtol=0, atol=3e-5)
    # search on the reference's own embeddings (fp32): same values as the reference loop, tie-aware
    E = g["embeddings"]
    q = E[:50]
    sc = np.stack([search_ref.cosine_similarity_rows(np.broadcast_to(x, E.shape), E) for x in q])
    v, i = search_ref.topk_rows(sc, 10)
    np.testing.assert_allclose(v, g["top10_values"][:50], rtol=0, atol=3e-7)
    agree = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(i.tolist(), g["top10_indices"][:50].tolist())])
    assert agree > 0.99


def test_f64_to_bf16_is_a_single_correct_rounding():
    """Values a hair below / above a bf16 midpoint must NOT be dragged onto the midpoint by an intermediate float32
    rounding (they would then follow the ties-to-even rule instead of their true side)."""
    lo, hi = np.float32(1.0), np.float32(1.0078125)           # neighbouring bf16 values; midpoint 1.00390625
    mid = np.float64(1.00390625)
    eps = 2.0 ** -40
    got = search_ref.f64_to_bf16(np.array([mid - eps, mid + eps, mid, 1.01171875, 1.01171875 + eps, -mid + eps]))
    assert got.tolist() == [lo, hi, lo, np.float32(1.015625), np.float32(1.015625), -lo]
    x = np.random.default_rng(0).standard_normal(10000)
    ref = search_ref.bf16_round(x.astype(np.float32))          # agrees away from the double-rounding cases
    assert (search_ref.f64_to_bf16(x) != ref).mean() < 1e-3
    u = search_ref.unit_rows(np.random.default_rng(1).standard_normal((7, 100)).astype(np.float32))
    assert np.array_equal(u, u.astype(np.float16).astype(np.float32)) and abs(float((u[0].astype(np.float64) ** 2).sum()) - 1) < 3e-4


# ---------------------------------------------------------------- MXFP8 operand format (oracle/fp8_ref.py)
def test_fp8_oracle_e4m3_table_and_rounding_match_torch_float8():
    """The e4m3 restatement is pinned against torch.float8_e4m3fn (an independent implementation of the OCP format):
    the value of every byte, and round-to-nearest-even of random values, of every midpoint and of the subnormal range."""
    import torch
    from oracle import fp8_ref
    vals = fp8_ref.e4m3_values()
    t = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(np.isnan(vals), np.isnan(t))
    np.testing.assert_array_equal(vals[~np.isnan(vals)], t[~np.isnan(t)])
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(100000) * np.exp(rng.uniform(-12, 6, 100000))).astype(np.float32)
    fin = vals[:0x7F].astype(np.float64)
    mid = ((fin[:-1] + fin[1:]) / 2).astype(np.float32)
    x = np.clip(np.concatenate([x, mid, -mid, np.linspace(0, 2.0 ** -5, 4097, dtype=np.float32)]), -448, 448)
    ref = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    np.testing.assert_array_equal(fp8_ref.f32_to_e4m3(x), ref)


def test_fp8_oracle_mx_blocks():
    from oracle import fp8_ref
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((5, 96)) * np.exp(rng.uniform(-8, 8, (5, 1)))).astype(np.float32)
    x[0, :32] = 0
    q, s = fp8_ref.mx_quantize(x)
    assert q.shape == (5, 96) and s.shape == (5, 3) and s[0, 0] == 127 and (q[0, :32] == 0).all()
    d = fp8_ref.mx_dequantize(q, s)
    blk = np.abs(x.reshape(5, 3, 32)).max(-1)
    # the block maximum lands in [256, 512) before clamping to 448: every element is within 2^-4 relative of the block max
    assert (np.abs(d - x).reshape(5, 3, 32).max(-1) <= blk * 2.0 ** -3 + 1e-30).all()
    # power-of-two scaling commutes with the format
    q2, s2 = fp8_ref.mx_quantize(x * 4.0)
    np.testing.assert_array_equal(q2, q)
    np.testing.assert_array_equal(s2[x.reshape(5, 3, 32).any(-1)], s[x.reshape(5, 3, 32).any(-1)] + 2)


# ---------------------------------------------------------------- adjacent callers (oracle/adjacent_ref.py, §8(f) N3 / N4)
@pytest.mark.parametrize("case", ["separated", "overlap", "norms"])
def test_kmeans_oracle_matches_sklearn_fixture(case):
    from oracle import adjacent_ref
    g = golden("kmeans.npz")
    lab, c, inertia = adjacent_ref.kmeans_lloyd(g[f"{case}_x"], g[f"{case}_init"])
    assert (lab == g[f"{case}_labels"]).mean() >= 0.999            # float32 (sklearn) vs float64 ties at cluster borders
    np.testing.assert_allclose(c, g[f"{case}_centers"], rtol=0, atol=2e-3)
    assert abs(inertia - float(g[f"{case}_inertia"])) <= 1e-4 * inertia
    if case == "norms":   # the fixture where cosine assignment would NOT reproduce the reference
        u = g["norms_x"] / np.linalg.norm(g["norms_x"], axis=1, keepdims=True)
        cu = g["norms_centers"] / np.linalg.norm(g["norms_centers"], axis=1, keepdims=True)
        assert ((u @ cu.T).argmax(1) != g["norms_labels"]).sum() >= 3


def test_meter_oracles_match_reference_fixture():
    from oracle import adjacent_ref
    g = golden("meters.npz")
    s2t, t2s, avg, wrong = adjacent_ref.retrieval_accuracy(g["src"], g["tgt"])
    assert (s2t, t2s, avg) == (float(g["src2tgt"]), float(g["tgt2src"]), float(g["avg"]))
    np.testing.assert_array_equal(wrong, g["wrong_pairs"])
    corr, val = adjacent_ref.sts_correlations(g["sts_a"], g["sts_b"], g["sts_gold"])
    np.testing.assert_allclose(corr, g["sts_corr"], rtol=0, atol=2e-6)
    assert abs(val - float(g["sts_val"])) <= 2e-6 and float(g["sts_avg"]) == float(g["sts_val"])
