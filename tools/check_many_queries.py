#!/usr/bin/env python3
"""Many query blocks (Q = 8 192 / 16 384 / 40 000 against 1 M x 384): the main pass then has fewer corpus chunks than XCDs and
several XCDs share a chunk (k1_topk.h block mapping).  Sampled oracle + timing.  Usage: python tools/check_many_queries.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import search_ref
from text_similarity_amd import ops
N, d, k = 1_000_000, 384, 10
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn(N, d, device="cuda", generator=g)
ec = ops.l2norm_rows(x)
for Q in (8192, 16384, 40000):
    q = torch.randn(Q, d, device="cuda", generator=g)
    eq = ops.l2norm_rows(q)
    s, i = ops.cosine_topk(eq, ec, d, k, eq_f32=q, ec_f32=x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        s, i = ops.cosine_topk(eq, ec, d, k, eq_f32=q, ec_f32=x)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 2 * 1e3
    qs = np.array([0, 1, Q // 2, Q - 1])
    fs, fi = search_ref.cosine_topk_f32(q.cpu().numpy()[qs], x.cpu().numpy(), k)
    ok = np.array_equal(i.cpu().numpy()[qs], fi) and np.array_equal(s.cpu().numpy()[qs], fs)
    sv = s.cpu().numpy()
    ok = ok and bool((np.diff(sv, axis=1) <= 0).all())
    print(f"Q={Q} ms={ms:.2f} TFLOPs={2*Q*N*d/ms/1e9:.0f} {'ok' if ok else 'MISMATCH'}", flush=True)
