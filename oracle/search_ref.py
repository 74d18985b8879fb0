"""ORACLE — test infrastructure only.  Never imported by the product package.

CPU restatement (numpy / torch float32+float64) of the similarity-search half of the hot path:

* row cosine           torch ``F.cosine_similarity`` as called at
                       /root/reference/src/pipeline/search_pipeline.py:77   (eps semantics: SURVEY.md §8 A7)
* dense cosine matrix  ``cos_sim``  /root/reference/src/utils/metrics.py:81-101 (no eps: zero rows -> NaN)
* per-query top-k      ``torch.topk(scores, k, largest=True)`` search_pipeline.py:78, with the tie rule the
                       reference leaves undefined fixed as: score descending, then index ascending
* the search loop      search_pipeline.py:60-89 (intended semantics)

torch (the pinned third-party dependency, requirements.txt:4 torch==1.6.0; installed 2.10.0) holds the
arithmetic; its published semantics are restated here in numpy.  Pinned against golden vectors produced in
the build container by the reference's own ``cos_sim`` and by torch ``cosine_similarity``/``topk``
(tools/make_golden.py -> tests/golden/search_*.npz); see tests/test_oracle_golden.py.

Exact scores.  The GPU path scores half-precision unit rows with MFMA (fp32 accumulate, hardware summation order) only to
*select candidates*; every returned score and the final order come from an exact re-score in float64 in one fixed
("lane") order, rounded once to float32, so results are reproducible bit for bit:
``exact_cosine`` — the reference's F.cosine_similarity of the float32 embeddings (what ``cosine_topk_f32`` ranks), and
``canonical_scores`` — the inner product of the stored unit rows, for callers that hold unit rows only.
"""
from __future__ import annotations

import numpy as np


def bf16_round(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return np.where(np.isnan(x), x, r.astype(np.uint32).view(np.float32)).astype(np.float32)


def f64_to_bf16(v: np.ndarray) -> np.ndarray:
    """float64 -> nearest bfloat16 (ties to even) with a single rounding, returned as bf16-exact float32.
    Same algorithm as text_similarity_amd/csrc/common.h f64_to_bf16: round to float32, and where that landed exactly on
    a bf16 midpoint step one float ulp back towards the true value before the final ties-to-even rounding."""
    v = np.asarray(v, dtype=np.float64)
    f = v.astype(np.float32)
    u = f.view(np.uint32).copy()
    mid = (u & np.uint32(0xFFFF)) == np.uint32(0x8000)
    fd, av = np.abs(f.astype(np.float64)), np.abs(v)
    u = np.where(mid & (av < fd), u - np.uint32(1), u)
    u = np.where(mid & (av > fd), u + np.uint32(1), u).astype(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.astype(np.uint32).view(np.float32).reshape(v.shape)


def l2_normalize_f64(x: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """x * (1 / max(||x||, eps)) in float64 (rows).  This is torch>=1.12 cosine_similarity's per-operand clamp (a zero
    row stays zero, so its cosine with anything is 0: SURVEY.md A7).  The sum of squares follows the kernel's order
    exactly (text_similarity_amd/csrc/common.h canonical_inv_norm): 64 lane partials over j = lane, lane+64, ... in
    element order, then an xor butterfly 32, 16, .., 1 — so GPU and oracle agree bit for bit, not merely to rounding."""
    x = np.asarray(x, dtype=np.float32)
    rows, d = x.shape
    pad = (-d) % 64
    xd = np.concatenate([x.astype(np.float64), np.zeros((rows, pad))], axis=1).reshape(rows, -1, 64)
    part = np.zeros((rows, 64), dtype=np.float64)
    for i in range(xd.shape[1]):
        part = part + xd[:, i, :] * xd[:, i, :]          # products of float32 are exact in float64: fma == mul+add
    lanes = np.arange(64)
    for o in (32, 16, 8, 4, 2, 1):
        part = part + part[:, lanes ^ o]
    inv = 1.0 / np.maximum(np.sqrt(part[:, :1]), np.float64(np.float32(eps)))
    return x.astype(np.float64) * inv


def l2_normalize(x: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """float32 view of l2_normalize_f64 (for comparisons with float32 references)."""
    return l2_normalize_f64(x, eps).astype(np.float32)


def unit_rows(x: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """The canonical search operand: L2-normalised rows rounded ONCE to IEEE half (returned as half-exact float32) — what
    tsim_l2norm_rows and the encoder's fused pooling epilogue store (text_similarity_amd/csrc/common.h f64_to_f16).
    numpy's float64 -> float16 cast is a single correct rounding (nearest even), subnormal halves included."""
    with np.errstate(over="ignore"):
        return l2_normalize_f64(x, eps).astype(np.float16).astype(np.float32)


def cosine_similarity_rows(x: np.ndarray, y: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """F.cosine_similarity(x, y, dim=-1) for x,y [N,H] float32 (search_pipeline.py:77)."""
    x = np.asarray(x, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    dot = (x.astype(np.float64) * y.astype(np.float64)).sum(-1)
    nx = np.maximum(np.sqrt((x.astype(np.float64) ** 2).sum(-1)), eps)
    ny = np.maximum(np.sqrt((y.astype(np.float64) ** 2).sum(-1)), eps)
    return (dot / (nx * ny)).astype(np.float32)


def cos_sim(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """metrics.py:81-101: a/||a|| @ (b/||b||)^T, 1-D inputs promoted to one row, no eps."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    if b.ndim == 1:
        b = b[None, :]
    with np.errstate(invalid="ignore", divide="ignore"):
        an = a / np.linalg.norm(a, axis=-1)[:, None]
        bn = b / np.linalg.norm(b, axis=-1)[:, None]
        return (an.astype(np.float64) @ bn.astype(np.float64).T).astype(np.float32)


_LANES = np.arange(64)


def _lane_sum(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Canonical float64 inner product along the last axis of two broadcastable float arrays, in the order the GPU's
    wave-cooperative re-score uses (text_similarity_amd/csrc/search.hip wave_dot_f64): lane l accumulates the products of
    elements j = l, l + 64, l + 128, ... in that order, then an xor butterfly 32, 16, .., 1 adds the 64 lane partials.
    Products of two float32 (or narrower) values are exact in float64, so fma == multiply + add and numpy reproduces
    the kernel bit for bit."""
    d = x.shape[-1]
    m = -(-d // 64)
    pad = m * 64 - d

    def lanes(a):
        a = np.asarray(a, dtype=np.float64)
        if pad:
            a = np.concatenate([a, np.zeros(a.shape[:-1] + (pad,))], axis=-1)
        return a.reshape(a.shape[:-1] + (m, 64))

    xl, yl = lanes(x), lanes(y)
    part = xl[..., 0, :] * yl[..., 0, :]
    for i in range(1, m):
        part = part + xl[..., i, :] * yl[..., i, :]
    for o in (32, 16, 8, 4, 2, 1):
        part = part + part[..., _LANES ^ o]
    return part[..., 0]


def exact_cosine(q: np.ndarray, c: np.ndarray, eps: float = 1e-8, qblock: int = 16, nblock: int = 8192) -> np.ndarray:
    """[Q,N] float32: the reference's score, F.cosine_similarity(q_row.expand_as(c), c, dim=-1)
    (/root/reference/src/pipeline/search_pipeline.py:76-77) = x.y / (max(||x||, eps) * max(||y||, eps)) on the float32
    rows, evaluated in float64 (canonical order: _lane_sum) and rounded ONCE to float32.  torch evaluates the same formula
    in float32, so its values differ from this one by float32 rounding (<= 4e-7 on the goldens); this is the definition
    the GPU's fp32 re-score (tsim_cosine_topk_ex) reproduces bit for bit."""
    q = np.asarray(q, dtype=np.float32)
    c = np.asarray(c, dtype=np.float32)
    e = np.float64(np.float32(eps))
    nq = np.maximum(np.sqrt(_lane_sum(q, q)), e)
    nc = np.maximum(np.sqrt(_lane_sum(c, c)), e)
    out = np.empty((q.shape[0], c.shape[0]), dtype=np.float32)
    for a in range(0, q.shape[0], qblock):
        for b in range(0, c.shape[0], nblock):
            dot = _lane_sum(q[a:a + qblock, None, :], c[None, b:b + nblock, :])
            out[a:a + qblock, b:b + nblock] = (dot / (nq[a:a + qblock, None] * nc[None, b:b + nblock])).astype(np.float32)
    return out


def exact_cosine_pairs(q: np.ndarray, c: np.ndarray, qi: np.ndarray, ci: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """exact_cosine of selected (query, corpus) index pairs."""
    x = np.asarray(q, dtype=np.float32)[qi]
    y = np.asarray(c, dtype=np.float32)[ci]
    e = np.float64(np.float32(eps))
    return (_lane_sum(x, y) / (np.maximum(np.sqrt(_lane_sum(x, x)), e) * np.maximum(np.sqrt(_lane_sum(y, y)), e))
            ).astype(np.float32)


def cosine_topk_f32(q: np.ndarray, c: np.ndarray, k: int, idx_offset: int = 0, block: int = 16):
    """The reference's search on float32 embeddings (search_pipeline.py:73-78: expand_as + F.cosine_similarity +
    torch.topk per query) with the tie rule fixed as (score desc, index asc): exact top-k of exact_cosine."""
    vals, idxs = [], []
    for s in range(0, q.shape[0], block):
        v, i = topk_rows(exact_cosine(q[s:s + block], c), k)
        vals.append(v)
        idxs.append(i + idx_offset)
    return np.concatenate(vals), np.concatenate(idxs)


def canonical_scores(eq: np.ndarray, ec: np.ndarray, qblock: int = 16, nblock: int = 8192) -> np.ndarray:
    """[Q,N] float32: inner product of the stored unit rows, float64 accumulation in the canonical lane order
    (_lane_sum), one final rounding to float32.  Inputs are the (half-exact) normalised rows."""
    eq = np.asarray(eq, dtype=np.float32)
    ec = np.asarray(ec, dtype=np.float32)
    out = np.empty((eq.shape[0], ec.shape[0]), dtype=np.float32)
    for a in range(0, eq.shape[0], qblock):
        for b in range(0, ec.shape[0], nblock):
            out[a:a + qblock, b:b + nblock] = _lane_sum(eq[a:a + qblock, None, :], ec[None, b:b + nblock, :]).astype(np.float32)
    return out


def canonical_scores_pairs(eq: np.ndarray, ec: np.ndarray, qi: np.ndarray, ci: np.ndarray) -> np.ndarray:
    """canonical score of selected (query, corpus) index pairs."""
    return _lane_sum(np.asarray(eq, dtype=np.float32)[qi], np.asarray(ec, dtype=np.float32)[ci]).astype(np.float32)


def topk_rows(scores: np.ndarray, k: int):
    """k largest per row, sorted by (score desc, index asc).  Returns (values [Q,k] f32, idx [Q,k] i64)."""
    scores = np.asarray(scores, dtype=np.float32)
    Q, N = scores.shape
    k = min(k, N)
    idx = np.empty((Q, k), dtype=np.int64)
    for q in range(Q):
        s = scores[q]
        if N > 4 * k + 64:
            kth = np.partition(s, N - k)[N - k]
            cand = np.nonzero(s >= kth)[0]
        else:
            cand = np.arange(N)
        order = np.lexsort((cand, -s[cand].astype(np.float64)))
        idx[q] = cand[order[:k]]
    return np.take_along_axis(scores, idx, 1), idx


def cosine_topk(eq: np.ndarray, ec: np.ndarray, k: int, idx_offset: int = 0, block: int = 64):
    """Exact top-k of canonical scores of every query row against every corpus row."""
    vals, idxs = [], []
    for s in range(0, eq.shape[0], block):
        v, i = topk_rows(canonical_scores(eq[s:s + block], ec), k)
        vals.append(v)
        idxs.append(i + idx_offset)
    return np.concatenate(vals), np.concatenate(idxs)


def merge_topk(values, indices, k: int):
    """Merge per-shard (values [Q,k_i], global indices [Q,k_i]) lists: (score desc, index asc)."""
    v = np.concatenate(values, axis=1)
    i = np.concatenate(indices, axis=1)
    out_v = np.empty((v.shape[0], min(k, v.shape[1])), dtype=np.float32)
    out_i = np.empty(out_v.shape, dtype=np.int64)
    for q in range(v.shape[0]):
        order = np.lexsort((i[q], -v[q].astype(np.float64)))[:out_v.shape[1]]
        out_v[q], out_i[q] = v[q][order], i[q][order]
    return out_v, out_i


def mining_search(query_emb: np.ndarray, corpus_emb: np.ndarray, k: int, chunk: int):
    """search_pipeline.py:60-89 as intended: for each corpus chunk, each query row is scored against
    every chunk row with F.cosine_similarity on the float32 embeddings and the k best are kept; chunks are merged."""
    vs, ix = [], []
    for s in range(0, corpus_emb.shape[0], chunk):
        v, i = cosine_topk_f32(query_emb, corpus_emb[s:s + chunk], k, idx_offset=s)
        vs.append(v)
        ix.append(i)
    return merge_topk(vs, ix, k)


# ---------------------------------------------------------------------------------------------------------------------
# The exactness guard of the GPU search (text_similarity_amd/csrc/common.h guard_eps, search.hip cos_topk_finalize):
# restated here so that tests can replay its decisions on the CPU and check the BOUND it rests on.
#
# A stored half row is u^ = u + delta, u = x / max(|x|, eps) the exact unit row, rho = |delta|_2.  For two rows
# |u^q.u^c - u_q.u_c| <= rho_q + rho_c + rho_q rho_c (Cauchy-Schwarz, |u| <= 1), u_q.u_c is the reference's cosine
# (search_pipeline.py:77), and a float32 accumulation of the ld exact products in any order adds at most
# ld 2^-23 (1 + rho_q)(1 + rho_c).
# ---------------------------------------------------------------------------------------------------------------------
def rho_rows(x: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """[rows] float64: || half(u_r) - u_r ||_2, the rounding residual of each row's stored unit image."""
    u = l2_normalize_f64(x, eps)
    with np.errstate(over="ignore"):
        h = u.astype(np.float16).astype(np.float64)
    return np.sqrt(((h - u) ** 2).sum(-1))


def rho_apriori(ld: int) -> float:
    """Bound on rho for ANY correctly rounded unit row: relative 2^-11 per normal element, 2^-25 per subnormal one."""
    return float(np.float32(4.8828125e-4) * np.float32(1.000001) + np.sqrt(np.float32(ld)) * np.float32(2.98023224e-8) * np.float32(1.000001))


def guard_eps(rho_q, rho_c, ld: int):
    """The guard's bound on |MFMA score - exact float32 cosine| for a query with residual rho_q against any row with residual
    <= rho_c (float64 evaluation of common.h guard_eps; the kernel rounds the same expression up to float32)."""
    rho_q = np.asarray(rho_q, dtype=np.float64)
    acc = ld * 2.0 ** -23 * (1.0 + rho_q) * (1.0 + rho_c)
    return (rho_q + rho_c + rho_q * rho_c + acc + 2.0 ** -22) * (1.0 + 1e-6)


def mfma_model_scores(q: np.ndarray, c: np.ndarray, order: str = "f64") -> np.ndarray:
    """[Q,N] float32 model of the selection scores: inner products of the stored half unit rows.
    order = "f64": exact dot rounded once (the centre of every possible float32 accumulation);
            "f32seq": float32 accumulation element by element (a worst-ish case for accumulation error)."""
    uq, uc = unit_rows(q).astype(np.float64), unit_rows(c).astype(np.float64)
    if order == "f64":
        return (uq @ uc.T).astype(np.float32)
    acc = np.zeros((uq.shape[0], uc.shape[0]), dtype=np.float32)
    for j in range(uq.shape[1]):
        acc = (acc + (uq[:, j:j + 1] * uc[None, :, j]).astype(np.float32)).astype(np.float32)
    return acc


def guard_replay(q: np.ndarray, c: np.ndarray, k: int, KL: int, mode: str = "bound", rho_c=None, c1: float = 4.0):
    """First pass of tsim_cosine_topk_ex on the CPU for ONE query row q [d]: the KL best rows by model MFMA score are
    re-scored exactly; returns (first-pass top-k indices, safe?, eps, cut, k-th exact score).
    mode = "bound": eps = guard_eps(rho_q, rho_c) (the shipped guard);  "sampled": eps = max(c1 x largest |MFMA - exact| seen on
    the KL candidates, d 2^-24) — the round-2 heuristic, kept to show what the adversarial fixtures defeat."""
    q = np.asarray(q, dtype=np.float32)[None, :]
    m = mfma_model_scores(q, c)[0]
    order = np.lexsort((np.arange(m.size), -m.astype(np.float64)))[:KL]
    ex = exact_cosine(q, c[order])[0]
    cut = float(m[order[-1]])
    rank = np.lexsort((order, -ex.astype(np.float64)))
    top = order[rank[:k]]
    sk = float(ex[rank[k - 1]])
    ld = c.shape[1]
    if mode == "bound":
        rc = float(rho_rows(c).max()) if rho_c is None else float(rho_c)
        eps = float(guard_eps(rho_rows(q)[0], rc, ld))
    else:
        eps = max(c1 * float(np.abs(m[order] - ex).max()), ld * 2.0 ** -24)
    return top, (cut + eps < sk), eps, cut, sk
