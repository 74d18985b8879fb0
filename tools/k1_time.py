#!/usr/bin/env python3
"""Duration of the search main pass alone (HIP events around it: tsim_time_next_topk), for A/B runs of variant libraries
(TSIM_LIB=<path>).  Unit-rows-only search, so diagnostic builds with wrong scores take no fallback.
Usage: python tools/k1_time.py [N] [d] [Q ...]"""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_similarity_amd import ops, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
Qs = [int(a) for a in sys.argv[3:]] or [4096]
g = torch.Generator(device="cuda").manual_seed(4321)
ec = ops.l2norm_rows(torch.randn((N, d), generator=g, device="cuda"))
L = C.CDLL(_lib.lib()._name)
L.tsim_time_next_topk.argtypes = [C.c_void_p, C.c_void_p]
for Q in Qs:
    eq = ops.l2norm_rows(torch.randn((Q, d), generator=g, device="cuda"))
    for _ in range(3):
        ops.cosine_topk(eq, ec, d, 10)
    ts = []
    for _ in range(8):
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(); k1.record()
        L.tsim_time_next_topk(k0.cuda_event, k1.cuda_event)
        ops.cosine_topk(eq, ec, d, 10)
        torch.cuda.synchronize()
        ts.append(k0.elapsed_time(k1))
    ts.sort()
    print(json.dumps({"lib": os.path.basename(_lib.lib()._name), "Q": Q, "N": N, "main_pass_ms_median": round(ts[len(ts) // 2], 4),
                      "min": round(ts[0], 4), "TFLOPs": round(2 * Q * N * d / ts[len(ts) // 2] / 1e9, 1)}), flush=True)
