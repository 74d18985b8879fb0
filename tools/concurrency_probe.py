#!/usr/bin/env python3
"""Do a small forward (the ~1.7 k remainder tokens of a 67 k-token batch) and a large one overlap when issued on two streams?
Two encoder handles (separate activation buffers), same weights.  Usage: python tools/concurrency_probe.py [big] [small]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from text_similarity_amd import presets
from text_similarity_amd.native_encoder import NativeEncoder
preset = "all-MiniLM-L6-v2"
cfg = presets.PRESETS[preset]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 100
def mk(n, seed):
    flat, cu = presets.synthetic_token_batch(n, seed=seed, vocab_size=cfg.vocab, max_len=256)
    enc = NativeEncoder.from_preset(preset, max_tokens=int(cu[-1]), max_seqs=n)
    fd, cd = torch.from_numpy(flat).cuda(), torch.from_numpy(cu).cuda()
    pos, cols = enc.positions(fd, cd)
    return enc, fd, cd, pos, cols, int(np.diff(cu).max()), int(cu[-1])
A, B = mk(nb, "sent1234"), mk(ns, "tail")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(which, iters=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        if "b" in which:
            with torch.cuda.stream(sb):
                sb.wait_event(e0) if _ == 0 else None
                B[0].forward_packed(B[1], B[2], B[3], B[4], B[5], pooled=True)
        if "a" in which:
            with torch.cuda.stream(sa):
                sa.wait_event(e0) if _ == 0 else None
                A[0].forward_packed(A[1], A[2], A[3], A[4], A[5], pooled=True)
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for w in ("a", "b", "ab", "a", "b", "ab"):
    run(w, 3)
    print(json.dumps({"which": w, "tokens": {"a": A[6], "b": B[6]}, "ms_per_iter": round(run(w), 4)}), flush=True)
