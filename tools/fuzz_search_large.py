#!/usr/bin/env python3
"""Randomised parity sweep of the fused cosine top-k in the LARGE-corpus regime (threshold pre-pass active, many query
blocks), oracle on sampled queries.  Usage: python tools/fuzz_search_large.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import search_ref
from text_similarity_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for c in range(cases):
    d = int(rng.choice([128, 384, 384, 768]))
    Q = int(rng.choice([1, 33, 256, 300, 1024, 2600, 4096]))
    N = int(rng.choice([262144, 300000, 524288, 700001, 1000000]))
    k = int(rng.choice([1, 5, 10, 12, 13, 20, 28]))
    kind = rng.choice(["normal", "dups", "aniso", "zeros", "planted"])
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(N, d, device="cuda", generator=g)
    q = torch.randn(Q, d, device="cuda", generator=g)
    if kind == "dups":
        src = torch.randint(0, N, (N // 50,), device="cuda", generator=g)
        dst = torch.randint(0, N, (N // 50,), device="cuda", generator=g)
        x[dst] = x[src]
        x[N - 40:] = x[7]                                  # 41 identical rows at the very end of the corpus
        q[: min(Q, 8)] = x[[7, 8, 9, 10, 11, 12, 13, 14]][: min(Q, 8)]
    elif kind == "aniso":
        x[:, 0] += 5.0
        q[:, 0] += 5.0
    elif kind == "zeros":
        x[torch.randint(0, N, (N // 10,), device="cuda", generator=g)] = 0.0
        q[0] = 0.0
        if Q > 2:
            q[2] = -q[2].abs()
    elif kind == "planted":                                # the best matches sit in the last rows / last chunk
        m = min(Q, 16)
        x[N - m:] = q[:m] * 3.0
    eq, ec = ops.l2norm_rows(q), ops.l2norm_rows(x)
    s, i = ops.cosine_topk(eq, ec, d, k)
    torch.cuda.synchronize()
    nq = max(1, min(Q, 4_000_000 // N))
    qs = np.unique(np.concatenate([np.arange(min(Q, 3)), rng.choice(Q, nq, replace=False)]))
    xu = search_ref.unit_rows(x.cpu().numpy())
    rs, ri = search_ref.cosine_topk(search_ref.unit_rows(q.cpu().numpy()[qs]), xu, k)
    ok = np.array_equal(i.cpu().numpy()[qs], ri) and np.array_equal(s.cpu().numpy()[qs], rs)
    # sortedness / validity of every list
    sv, iv = s.cpu().numpy(), i.cpu().numpy()
    ok = ok and bool((np.diff(sv, axis=1) <= 0).all()) and bool(((iv >= 0) & (iv < N)).all())
    # the reference's definition (float32 rows, exact re-score + guard)
    s2, i2 = ops.cosine_topk(eq, ec, d, k, eq_f32=q, ec_f32=x)
    torch.cuda.synchronize()
    fs, fi = search_ref.cosine_topk_f32(q.cpu().numpy()[qs], x.cpu().numpy(), k)
    ok = ok and np.array_equal(i2.cpu().numpy()[qs], fi) and np.array_equal(s2.cpu().numpy()[qs], fs)
    bad += not ok
    print(f"case {c:3d} d={d:3d} Q={Q:4d} N={N:7d} k={k:2d} {kind:7s} {'ok' if ok else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
    del x, q, eq, ec
print(f"fuzz_search_large: {cases - bad}/{cases} cases exact")
sys.exit(1 if bad else 0)
