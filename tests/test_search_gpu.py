"""GPU parity of the search half of the hot path against the oracle (bit-exact: scores and indices).
Every call goes through the C ABI of libtsim.so."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import search_ref
from text_similarity_amd import ops, presets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _unit_dev(x_f32: np.ndarray, d: int):
    """upload rows as the padded float16 matrix the kernels consume (callers pass half-exact unit rows; anything else is
    rounded to half here and by _h below, so GPU and oracle still see identical operands)"""
    ld = ops.pad_dim(d)
    t = torch.zeros((x_f32.shape[0], ld), dtype=torch.float16)
    t[:, :d] = torch.from_numpy(np.ascontiguousarray(x_f32, dtype=np.float32)).to(torch.float16)
    return t.to(DEV)


def _h(x: np.ndarray) -> np.ndarray:
    """the values _unit_dev stores: float32 -> half (nearest even) -> float32"""
    return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def _check_exact(q, c, k, idx_offset=0, oracle_queries=None):
    """GPU top-k vs the oracle, bit for bit.  For big shapes the O(Q*N*d) oracle runs on a sample of the queries
    (``oracle_queries``); every query still gets the checks that need no full oracle: returned scores are the canonical
    scores of the returned (query, row) pairs, lists are ordered by (score desc, index asc), indices are distinct."""
    d = q.shape[1]
    q, c = _h(q), _h(c)
    s, i = ops.cosine_topk(_unit_dev(q, d), _unit_dev(c, d), d, k, idx_offset)
    torch.cuda.synchronize()
    s, i = s.cpu().numpy(), i.cpu().numpy()
    kk = min(k, c.shape[0])
    sel = np.arange(q.shape[0]) if oracle_queries is None else np.asarray(oracle_queries)
    rs, ri = search_ref.cosine_topk(q[sel], c, k, idx_offset)
    np.testing.assert_array_equal(i[sel][:, :kk], ri)
    np.testing.assert_array_equal(s[sel][:, :kk], rs)
    if kk < k:  # corpus smaller than k: padded with -1 / -inf
        assert (i[:, kk:] == -1).all() and np.isneginf(s[:, kk:]).all()
    qi = np.repeat(np.arange(q.shape[0]), kk)
    pair = search_ref.canonical_scores_pairs(q, c, qi, (i[:, :kk] - idx_offset).reshape(-1)).reshape(-1, kk)
    np.testing.assert_array_equal(s[:, :kk], pair)
    assert ((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (i[:, :-1] < i[:, 1:])))[:, :kk - 1].all()


def test_golden_topk_fixture_with_duplicates():
    g = golden("search_topk.npz")
    for k in (1, 3, 10):
        _check_exact(g["queries"], g["corpus"], k)
    s, i = ops.cosine_topk(_unit_dev(g["queries"], 384), _unit_dev(g["corpus"], 384), 384, 10)
    assert i[0, :4].tolist() == [7, 40, 41, 200]           # tie rule: equal scores -> lower index first
    np.testing.assert_allclose(s.cpu().numpy(), g["loop_top10_values"], rtol=0, atol=1e-3)  # north_star tolerance


@pytest.mark.parametrize("Q,N,d,k", [
    (70, 5000, 384, 10),      # ragged Q (not a multiple of 32), N not a multiple of 32
    (300, 20011, 384, 10),    # two query blocks
    (33, 777, 64, 5),         # d padded 64 -> 128
    (17, 4097, 768, 10),      # 4-wave variant
    (5, 40, 384, 12),         # tiny corpus
    (9, 2000, 256, 20),       # KL = 32 lists
    (64, 33000, 384, 1),
    (1, 1000, 384, 10),       # a single query
    (40, 3001, 512, 10),      # d = 512 (4-wave kernel)
    (12, 5000, 384, 28),      # largest supported k (KL = 32)
    (300, 262144 + 77, 384, 10),   # just above the pre-pass threshold, ragged tail, 2 query blocks
    (2600, 300000, 384, 10),  # 11 query blocks -> fewer than 8 corpus chunks (XCD-sharing block map)
])
def test_random_exact(Q, N, d, k):
    big = Q * N > 4_000_000          # keep the CPU side (generator + oracle) to a few seconds per case
    if big:                          # data drawn on the GPU, unit rows downloaded for the sampled oracle
        g = torch.Generator(device=DEV).manual_seed(Q * 7919 + N)
        cu = ops.l2norm_rows(torch.randn((N, d), generator=g, device=DEV))
        qu = ops.l2norm_rows(torch.randn((Q, d), generator=g, device=DEV))
        c, q = cu[:, :d].float().cpu().numpy(), qu[:, :d].float().cpu().numpy()
        sample = sorted(set([0, Q // 3, Q // 2, Q - 1]))
    else:
        q = presets.synthetic_embeddings(Q, d, f"tq/{Q}/{N}/{d}")
        c = presets.synthetic_embeddings(N, d, f"tc/{Q}/{N}/{d}")
        sample = None
    _check_exact(q, c, k, idx_offset=123456789012, oracle_queries=sample)


def test_prepass_threshold_is_safe_when_best_rows_sit_in_the_sample():
    """The pre-pass seeds thresholds from the first rows.  Put every query's true neighbours INSIDE that sample (so the
    seeded bound is as tight as it can be) and duplicates of them at the far end of the corpus: ties must still resolve
    to the lower index and nothing may be dropped."""
    N, d = 400_000, 384
    g = torch.Generator(device=DEV).manual_seed(7)
    c = torch.randn((N, d), generator=g, device=DEV)
    q = c[:64].clone() + 0.01 * torch.randn((64, d), generator=g, device=DEV)   # neighbours = rows 0..63 (in the sample)
    c[N - 64:] = c[:64]                                                          # exact duplicates at the end
    cu, qu = ops.l2norm_rows(c), ops.l2norm_rows(q)
    s, i = ops.cosine_topk(qu, cu, d, 10)
    torch.cuda.synchronize()
    i = i.cpu().numpy()
    assert (i[:, 0] == np.arange(64)).all() and (i[:, 1] == N - 64 + np.arange(64)).all()
    assert torch.equal(s[:, 0], s[:, 1])
    rs, ri = search_ref.cosine_topk(qu[:8, :d].float().cpu().numpy(), cu[:, :d].float().cpu().numpy(), 10)
    np.testing.assert_array_equal(i[:8], ri)
    np.testing.assert_array_equal(s[:8].cpu().numpy(), rs)


def test_corpus_smaller_than_k():
    q = presets.synthetic_embeddings(3, 384, "small/q")
    c = presets.synthetic_embeddings(7, 384, "small/c")
    _check_exact(q, c, 10)


def test_many_duplicates_and_zero_rows():
    c = presets.synthetic_embeddings(3000, 384, "dup/c")
    c[100:160] = c[5]            # 61 identical rows: more than the candidate list holds
    c[2000] = 0.0                # zero row -> score exactly 0 (A7 semantics)
    q = presets.synthetic_embeddings(40, 384, "dup/q")
    q[0] = c[5]
    q[1] = 0.0                   # zero query: every score 0 -> the k lowest indices
    _check_exact(q, c, 10)
    s, i = ops.cosine_topk(_unit_dev(q, 384), _unit_dev(c, 384), 384, 10)
    assert i[0].tolist() == [5] + list(range(100, 109))
    assert i[1].tolist() == list(range(10)) and (s[1] == 0).all()


@pytest.mark.parametrize("k", [10, 19])
def test_zero_query_and_zero_rows_with_threshold_prepass(k):
    """N >= 262144 switches the threshold pre-pass on.  A zero query scores exactly 0 against every row, so the bound it
    derives is 0.0 and every row ties: the k lowest indices must come back (found by tools/fuzz_search.py: the float
    'below' +0.0 in key order is -0.0, which compares equal, and the filter then rejected everything).  A query whose
    best scores are a mix of exact zeros (zero corpus rows) and negatives is checked as well."""
    N, d = 300_000, 128
    rng = np.random.default_rng(11)
    c = rng.standard_normal((N, d)).astype(np.float32)
    c[rng.integers(0, N, N // 10)] = 0.0
    q = rng.standard_normal((33, d)).astype(np.float32)
    q[0] = 0.0
    q[1] = -np.abs(q[1])
    c[::7] = -np.abs(c[::7]) * (c[::7] != 0)          # many rows with all-negative entries: negative cosines with q[1]
    s, i = ops.cosine_topk(ops.l2norm_rows(torch.from_numpy(q).to(DEV)), ops.l2norm_rows(torch.from_numpy(c).to(DEV)), d, k)
    torch.cuda.synchronize()
    assert i[0].tolist() == list(range(k)) and (s[0] == 0).all()
    rs, ri = search_ref.cosine_topk(search_ref.unit_rows(q[:6]), search_ref.unit_rows(c), k)
    np.testing.assert_array_equal(i[:6].cpu().numpy(), ri)
    np.testing.assert_array_equal(s[:6].cpu().numpy(), rs)


def test_anisotropic_scores_near_ties():
    # rows = common direction + small noise: all cosines ~0.99, gaps ~1e-4 (random-weight encoders look like this)
    base = presets.normal("aniso/base", 384)
    c = base[None, :] + 0.05 * presets.normal("aniso/c", 6000 * 384).reshape(6000, 384)
    q = base[None, :] + 0.05 * presets.normal("aniso/q", 50 * 384).reshape(50, 384)
    cu = search_ref.unit_rows(c)
    qu = search_ref.unit_rows(q)
    _check_exact(qu, cu, 10)


def test_self_search_full_size_properties():
    """N = 1M x 384 (BASELINE.json size): properties that need no O(N^2) oracle —
    a row's best match is itself with score ||row||^2, lists are sorted, sampled queries match the oracle."""
    N, d, Q = 1_000_000, 384, 512
    g = torch.Generator(device=DEV).manual_seed(4321)
    x = torch.randn((N, d), generator=g, device=DEV, dtype=torch.float32)
    unit = ops.l2norm_rows(x)
    rows = torch.arange(0, N, N // Q, device=DEV)[:Q]
    s, i = ops.cosine_topk(unit[rows].contiguous(), unit, d, 10)
    torch.cuda.synchronize()
    assert (i[:, 0] == rows).all()
    self_dot = (unit[rows].double() ** 2).sum(1).float()
    assert torch.equal(s[:, 0], self_dot)
    assert (s[:, :-1] >= s[:, 1:]).all()
    # oracle on 4 sampled queries against the whole corpus
    un = unit[:, :d].float().cpu().numpy()
    sel = [0, 101, 333, 511]
    rs, ri = search_ref.cosine_topk(un[rows.cpu().numpy()[sel]], un, 10)
    np.testing.assert_array_equal(i.cpu().numpy()[sel], ri)
    np.testing.assert_array_equal(s.cpu().numpy()[sel], rs)


def test_l2norm_rows_matches_oracle_bit_exact():
    x = presets.normal("l2/x", 1000 * 384).reshape(1000, 384) * 3.0
    x[7] = 0.0
    x[8] = 1e-12        # norm below eps: divided by eps, not by its norm
    u = ops.l2norm_rows(torch.from_numpy(x).to(DEV))
    torch.cuda.synchronize()
    got = u[:, :384].float().cpu().numpy()
    ref = search_ref.unit_rows(x)
    np.testing.assert_array_equal(got, ref)
    assert (got[7] == 0).all() and (u[:, 384:] == 0).all()
    xb = torch.from_numpy(x).to(torch.bfloat16).to(DEV)      # bf16 INPUT rows (encoder hidden states) are accepted too
    ub = ops.l2norm_rows(xb)
    refb = search_ref.unit_rows(xb.float().cpu().numpy())
    np.testing.assert_array_equal(ub[:, :384].float().cpu().numpy(), refb)
    # narrow rows (d = 100, not a multiple of 64) and the padded tail
    y = presets.normal("l2/y", 33 * 100).reshape(33, 100)
    uy = ops.l2norm_rows(torch.from_numpy(y).to(DEV))
    np.testing.assert_array_equal(uy[:, :100].float().cpu().numpy(), search_ref.unit_rows(y))
    assert uy.shape[1] == 128 and (uy[:, 100:] == 0).all()


def test_topk_merge_matches_unsharded():
    q = presets.synthetic_embeddings(50, 384, "mg/q")
    c = presets.synthetic_embeddings(9000, 384, "mg/c")
    c[8000] = c[10]
    qd, cd = _unit_dev(q, 384), _unit_dev(c, 384)
    full_s, full_i = ops.cosine_topk(qd, cd, 384, 10)
    ss, ii = [], []
    for lo in range(0, 9000, 2300):
        s, i = ops.cosine_topk(qd, cd[lo:lo + 2300].contiguous(), 384, 10, idx_offset=lo)
        ss.append(s)
        ii.append(i)
    ms, mi = ops.topk_merge(ss, ii, 10)
    assert torch.equal(mi, full_i) and torch.equal(ms, full_s)
    rs, ri = search_ref.merge_topk([s.cpu().numpy() for s in ss], [i.cpu().numpy() for i in ii], 10)
    np.testing.assert_array_equal(mi.cpu().numpy(), ri)


def test_cos_sim_dense_and_mean_pool_goldens():
    g = golden("search_cos.npz")
    out = ops.cos_sim_dense(torch.from_numpy(g["a"]).to(DEV), torch.from_numpy(g["b"]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(out, g["cos_sim"], rtol=0, atol=1e-6)
    z = ops.cos_sim_dense(torch.from_numpy(g["a"]).to(DEV), torch.from_numpy(g["b_zero"]).to(DEV)).cpu().numpy()
    assert np.isnan(z[:, 5]).all() and np.array_equal(np.isnan(z), np.isnan(g["cos_sim_zero"]))
    p = golden("pool_edge.npz")
    got = ops.mean_pool(torch.from_numpy(p["hidden"]).to(DEV), torch.from_numpy(p["attention_mask"]).to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), p["pooled"], rtol=0, atol=1e-6)
    assert (got[2] == 0).all()   # all-zero mask row
