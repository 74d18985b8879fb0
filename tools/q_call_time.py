#!/usr/bin/env python3
"""Whole-call and main-pass duration of tsim_cosine_topk_ex (float32 rows given: the product path) for small query batches —
BASELINE config 4's regime (Q = 256 per batch).  A/B knobs are environment variables read by the library (TSIM_K1_PREPASS,
TSIM_K1_PHASES) or variant libraries (TSIM_LIB).  Usage: python tools/q_call_time.py [N] [d] [Q ...]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_similarity_amd import ops, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
Qs = [int(a) for a in sys.argv[3:]] or [256]
g = torch.Generator(device="cuda").manual_seed(4321)
cf = torch.randn((N, d), generator=g, device="cuda")
ec, rho = ops.l2norm_rows(cf, return_rho=True)
L = C.CDLL(_lib.lib()._name)
L.tsim_time_next_topk.argtypes = [C.c_void_p, C.c_void_p]
ev = lambda: torch.cuda.Event(enable_timing=True)
for Q in Qs:
    qf = torch.randn((Q, d), generator=g, device="cuda")
    eq = ops.l2norm_rows(qf)
    for _ in range(3):
        ops.cosine_topk(eq, ec, d, 10, eq_f32=qf, ec_f32=cf, rho_c=rho)
    tm, tc = [], []
    for _ in range(12):
        k0, k1, c0, c1 = ev(), ev(), ev(), ev()
        for e in (k0, k1, c0, c1):
            e.record()
        L.tsim_time_next_topk(k0.cuda_event, k1.cuda_event)
        c0.record()
        _, _, st = ops.cosine_topk(eq, ec, d, 10, eq_f32=qf, ec_f32=cf, rho_c=rho, return_status=True)
        c1.record()
        torch.cuda.synchronize()
        tm.append(k0.elapsed_time(k1))
        tc.append(c0.elapsed_time(c1))
    tm.sort(); tc.sort()
    print(json.dumps({"lib": os.path.basename(_lib.lib()._name), "env": {k: v for k, v in os.environ.items() if k.startswith("TSIM_")},
                      "Q": Q, "N": N, "d": d, "main_pass_ms": round(tm[len(tm) // 2], 4), "call_ms": round(tc[len(tc) // 2], 4),
                      "call_min_ms": round(tc[0], 4), "hbm_frac_main": round(N * d * 2 / tm[len(tm) // 2] / 1e6 / 8000, 4),
                      "hbm_frac_call": round(N * d * 2 / tc[len(tc) // 2] / 1e6 / 8000, 4), "flagged": int((st > 0).sum())}), flush=True)
