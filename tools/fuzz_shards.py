#!/usr/bin/env python3
"""Sharded search == unsharded search, bit for bit: random shard boundaries (including empty and 1-row shards), per-shard
tsim_cosine_topk with idx_offset, tsim_topk_merge.  Usage: python tools/fuzz_shards.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from text_similarity_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    d = int(rng.choice([64, 384, 768])); Q = int(rng.choice([1, 40, 300])); N = int(rng.choice([50, 3000, 40000, 300000]))
    k = int(rng.choice([1, 10, 20])); ns = int(rng.choice([2, 3, 8]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(N, d, device="cuda", generator=g); q = torch.randn(Q, d, device="cuda", generator=g)
    x[torch.randint(0, N, (max(1, N // 20),), device="cuda", generator=g)] = x[0].clone()      # ties across shards
    ec, eq = ops.l2norm_rows(x), ops.l2norm_rows(q)
    s0, i0 = ops.cosine_topk(eq, ec, d, min(k, N))
    cuts = np.sort(rng.integers(0, N + 1, ns - 1)); bounds = [0, *cuts.tolist(), N]
    ss, ii = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        kk = min(k, N)
        if b - a == 0:
            s = torch.full((Q, kk), float("-inf"), device="cuda"); i = torch.full((Q, kk), -1, dtype=torch.int64, device="cuda")
        else:
            s, i = ops.cosine_topk(eq, ec[a:b].contiguous(), d, min(kk, b - a), idx_offset=a)
            if s.shape[1] < kk:
                pad = kk - s.shape[1]
                s = torch.cat([s, torch.full((Q, pad), float("-inf"), device="cuda")], 1)
                i = torch.cat([i, torch.full((Q, pad), -1, dtype=torch.int64, device="cuda")], 1)
        ss.append(s); ii.append(i)
    sm, im = ops.topk_merge(ss, ii, min(k, N))
    ok = torch.equal(sm, s0) and torch.equal(im, i0)
    bad += not ok
    print(f"case {c:2d} d={d} Q={Q} N={N} k={k} shards={bounds} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"fuzz_shards: {cases - bad}/{cases} identical to the unsharded search")
sys.exit(1 if bad else 0)
