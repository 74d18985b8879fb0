#!/usr/bin/env python3
"""Randomised checks of the small ops: tsim_gemm_mxfp8 (+ quantiser) on odd shapes and extreme scales, tsim_mean_pool,
tsim_cos_sim, tsim_l2norm_rows.  Usage: python tools/fuzz_ops.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import fp8_ref, search_ref, encoder_ref
from text_similarity_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    what = rng.choice(["mx", "pool", "cos", "l2"])
    ok, desc = True, ""
    if what == "mx":
        M = int(rng.choice([1, 31, 255, 256, 257, 700])); N = int(rng.choice([256, 512, 768])); K = int(rng.choice([256, 384 + 128, 768, 1024, 3072]))
        x = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-15, 10, (M, 1)))).astype(np.float32)
        w = (rng.standard_normal((N, K)) * np.exp(rng.uniform(-8, 2, (N, 1)))).astype(np.float32)
        x[rng.random((M, K)) < 0.05] = 0.0
        if M > 2: x[1, :64] = 0.0
        bias = rng.standard_normal(N).astype(np.float32)
        xb = torch.from_numpy(x).cuda().to(torch.bfloat16)
        xq, xs = ops.quantize_mxfp8(xb)
        rq, rs = fp8_ref.mx_quantize(xb.float().cpu().numpy())
        ok = np.array_equal(xq.cpu().numpy(), rq) and np.array_equal(xs.cpu().numpy(), rs)
        wq, ws = fp8_ref.mx_quantize(w)
        out = ops.gemm_mxfp8(xq, xs, torch.from_numpy(wq).cuda(), torch.from_numpy(ws).cuda(), torch.from_numpy(bias).cuda()).cpu().numpy()
        ref = fp8_ref.mx_dequantize(rq, rs).astype(np.float64) @ fp8_ref.mx_dequantize(wq, ws).astype(np.float64).T + bias
        scale = np.abs(fp8_ref.mx_dequantize(rq, rs)).astype(np.float64) @ np.abs(fp8_ref.mx_dequantize(wq, ws)).astype(np.float64).T + np.abs(bias)
        ok = ok and np.isfinite(out).all() and (np.abs(out - ref) <= 4e-5 * scale + 1e-30).all()   # the scaled MFMA aligns the 64 products of a step with limited width: measured <= 1.6e-5 * sum|a*b|
        desc = f"M={M} N={N} K={K}"
    elif what == "pool":
        B = int(rng.integers(1, 40)); S = int(rng.choice([1, 2, 17, 64, 300])); H = int(rng.choice([64, 384, 768, 100]))
        h = rng.standard_normal((B, S, H)).astype(np.float32); m = (rng.random((B, S)) < 0.7).astype(np.int64)
        if B > 1: m[0] = 0
        got = ops.mean_pool(torch.from_numpy(h).cuda(), torch.from_numpy(m).cuda()).cpu().numpy()
        ref = encoder_ref.mean_pool(h, m).numpy()
        ok = np.allclose(got, ref, rtol=2e-6, atol=2e-6); desc = f"B={B} S={S} H={H}"
    elif what == "cos":
        A = int(rng.choice([1, 37, 300])); Bn = int(rng.choice([1, 101, 1000])); d = int(rng.choice([3, 64, 384, 500]))
        a = rng.standard_normal((A, d)).astype(np.float32); b = rng.standard_normal((Bn, d)).astype(np.float32)
        got = ops.cos_sim_dense(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
        ok = np.allclose(got, search_ref.cos_sim(a, b), rtol=0, atol=3e-6); desc = f"A={A} B={Bn} d={d}"
    else:
        n = int(rng.choice([1, 63, 64, 1000])); d = int(rng.choice([1, 64, 100, 384, 768]))
        x = (rng.standard_normal((n, d)) * np.exp(rng.uniform(-30, 30, (n, 1)))).astype(np.float32)
        if n > 1: x[0] = 0
        got = ops.l2norm_rows(torch.from_numpy(x).cuda())[:, :d].float().cpu().numpy()
        ok = np.array_equal(got, search_ref.unit_rows(x)); desc = f"n={n} d={d}"
    bad += not ok
    print(f"case {c:2d} {what:4s} {desc:24s} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"fuzz_ops: {cases - bad}/{cases} ok")
sys.exit(1 if bad else 0)
