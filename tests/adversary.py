"""Adversarial corpora for the search's exactness guard (test data builders; numpy only).

``aligned_rounding_row``: a float32 row whose unit image sits ``frac`` of a half-precision ulp past a grid point in EVERY
element, on the side that makes the rounding LOWER q.c — the worst case of the half-row selection error (~3e-4 at d = 384,
7-10x the typical error).  ``neighbours``: rows at prescribed exact cosines.  VERDICT r2 "What's weak 1"."""
import numpy as np


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.sqrt((v * v).sum())


def row_at_cosine(q, cos_t, rng):
    """float64 unit vector with cosine cos_t to q."""
    uq = _unit(q)
    r = rng.standard_normal(uq.size)
    r -= (r @ uq) * uq
    return cos_t * uq + np.sqrt(1.0 - cos_t * cos_t) * _unit(r)


def neighbours(q, cosines, rng, scale=1.0):
    return np.stack([row_at_cosine(q, float(ct), rng) * scale for ct in cosines]).astype(np.float32)


def aligned_rounding_row(q, cos_target, rng, frac=0.49, scale=3.7):
    """Returns (row float32 [d], exact cosine with q, selection score = dot of the half images).  The row's exact cosine is
    within ~1e-5 of cos_target; its half image scores ~0.49 sum|q_i| ulp_i lower."""
    uq = _unit(q)
    sgn = np.where(uq >= 0, 1.0, -1.0)
    uqh = uq.astype(np.float16).astype(np.float64)
    cos_t = cos_target - 3e-4
    best = None
    for _ in range(8):
        v = row_at_cosine(q, cos_t, rng)
        beta = 1.0
        for _ in range(60):
            g16 = (beta * v).astype(np.float16)
            g = g16.astype(np.float64)
            ulp = np.spacing(np.abs(g16)).astype(np.float64)            # spacing towards larger magnitude
            toward_zero = (sgn * np.sign(g)) < 0
            pow2 = np.frexp(np.abs(g))[0] == 0.5
            step = np.where(toward_zero & pow2, 0.5 * ulp, ulp)         # below a power of two the grid is twice as fine
            w = g + frac * step * sgn
            n = np.sqrt((w * w).sum())
            if abs(n - 1.0) < 2e-6:
                break
            beta /= n
        j = int(np.argmax(np.abs(w)))                                    # absorb the remaining norm defect in one element
        rest = (w * w).sum() - w[j] * w[j]
        w[j] = np.sign(w[j]) * np.sqrt(max(1.0 - rest, 0.0))
        assert np.array_equal(w.astype(np.float16)[np.arange(w.size) != j], g16[np.arange(w.size) != j])
        exact = float(w @ uq)
        sel = float(w.astype(np.float16).astype(np.float64) @ uqh)
        best = ((w * scale).astype(np.float32), exact, sel)
        if abs(exact - cos_target) < 1e-5:
            break
        cos_t += cos_target - exact
    return best


def adversarial_case(seed=0, N=20000, d=384, spacing=4.2e-5):
    """30 neighbours c[100+j] spaced `spacing` apart in exact cosine and one aligned-rounding row (5000)
    whose exact cosine sits between the 9th and the 10th of them (rank 10) while its selection score falls below the 16th.
    Returns (q [d], c [N, d]) float32."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal(d).astype(np.float32)
    c = rng.standard_normal((N, d)).astype(np.float32)
    row, s_star, _ = aligned_rounding_row(q, 0.9496, rng)
    c[5000] = row
    c[100:130] = neighbours(q, s_star + (8.5 - np.arange(30)) * spacing, rng, scale=2.0)
    return q, c
