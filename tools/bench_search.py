#!/usr/bin/env python3
"""Quick timing of tsim_cosine_topk on the GPU box: python tools/bench_search.py [N] [d] [Q ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_similarity_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
Qs = [int(a) for a in sys.argv[3:]] or [256, 1024, 4096, 16384]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(4321)
corpus = ops.l2norm_rows(torch.randn((N, d), generator=g, device=dev))
for Q in Qs:
    q = ops.l2norm_rows(torch.randn((Q, d), generator=g, device=dev))
    for _ in range(2):
        ops.cosine_topk(q, corpus, d, 10)
    torch.cuda.synchronize()
    iters = 5 if Q <= 4096 else 2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.cosine_topk(q, corpus, d, 10)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    pairs = Q * N
    print(json.dumps({"Q": Q, "N": N, "d": d, "ms": round(ms, 4), "Gpairs_s": round(pairs / ms / 1e6, 1),
                      "TFLOPs": round(2 * pairs * d / ms / 1e9, 1),
                      "stream_GBs": round(-(-Q // 256) * N * d * 2 / ms / 1e6, 1)}), flush=True)
