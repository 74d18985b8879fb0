#!/usr/bin/env python3
"""Randomised parity sweep of the fused cosine top-k against the oracle (bit-exact indices and scores), in both modes: unit rows
only (canonical scores) and the reference's cosine of float32 rows (tsim_cosine_topk_ex); every fourth case takes k up to 64.
Usage: python tools/fuzz_search.py [cases] [seed].  Not part of the pytest suite (minutes of CPU oracle time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import search_ref
from text_similarity_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for c in range(cases):
    d = int(rng.choice([64, 100, 128, 256, 300, 384, 512, 768]))
    Q = int(rng.choice([1, 2, 31, 32, 33, 100, 255, 256, 257, 600]))
    N = int(rng.choice([1, 5, 31, 32, 63, 64, 65, 1000, 4097, 20000, 70000, 300000]))
    k = int(rng.integers(1, 65)) if c % 4 == 3 else int(rng.integers(1, 29))
    kind = rng.choice(["normal", "dups", "aniso", "zeros"])
    x = rng.standard_normal((N, d)).astype(np.float32)
    q = rng.standard_normal((Q, d)).astype(np.float32)
    if kind == "dups" and N > 4:
        x[rng.integers(0, N, N // 3)] = x[rng.integers(0, N, N // 3)]          # many exact ties
        q[: min(Q, N)] = x[: min(Q, N)]
    elif kind == "aniso":
        x[:, 0] += 6.0                                                          # scores crowd near 1: near-ties everywhere
        q[:, 0] += 6.0
    elif kind == "zeros" and N > 2:
        x[rng.integers(0, N, max(1, N // 10))] = 0.0
        q[0] = 0.0
    eq = ops.l2norm_rows(torch.from_numpy(q).cuda())
    ec = ops.l2norm_rows(torch.from_numpy(x).cuda())
    kk = min(k, N)
    s, i = ops.cosine_topk(eq, ec, d, kk)
    torch.cuda.synchronize()
    # oracle on a sample of queries when the case is large
    qs = np.arange(Q) if Q * N <= 3_000_000 else np.sort(rng.choice(Q, max(1, 3_000_000 // N), replace=False))
    rs, ri = search_ref.cosine_topk(search_ref.unit_rows(q[qs]), search_ref.unit_rows(x), kk)
    ok = np.array_equal(i.cpu().numpy()[qs], ri) and np.array_equal(s.cpu().numpy()[qs], rs)
    # the reference's definition: cosine of the float32 rows (exact re-score + guard), status must be 0/1/2
    qt, xt = torch.from_numpy(q).cuda(), torch.from_numpy(x).cuda()
    s2, i2, st = ops.cosine_topk(eq, ec, d, kk, eq_f32=qt, ec_f32=xt, return_status=True)
    torch.cuda.synchronize()
    fs, fi = search_ref.cosine_topk_f32(q[qs], x, kk)
    ok2 = np.array_equal(i2.cpu().numpy()[qs], fi) and np.array_equal(s2.cpu().numpy()[qs], fs) and int(st.max()) <= 2
    ok = ok and ok2
    bad += not ok
    print(f"case {c:3d} d={d:3d} Q={Q:4d} N={N:6d} k={kk:2d} {kind:6s} {'ok' if ok else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
print(f"fuzz_search: {cases - bad}/{cases} cases bit-exact")
sys.exit(1 if bad else 0)
