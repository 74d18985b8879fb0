#!/usr/bin/env python3
"""Diagnostic: the large-batch MiniLM forward of tests/test_encoder_gpu.py repeated (a fresh encoder every 10 forwards); lists the
forwards with non-finite pooled / hidden rows.  Usage: python tools/nan_probe.py [sentences] [forwards]; A/B with TSIM_* switches
or TSIM_LIB=<variant library>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from text_similarity_amd import presets
from text_similarity_amd.native_encoder import NativeEncoder
preset = "all-MiniLM-L6-v2"
cfg = presets.PRESETS[preset]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
flat, cu = presets.synthetic_token_batch(n, seed="big/" + preset, vocab_size=cfg.vocab, max_len=64)
bad = []
fd, cd = torch.from_numpy(flat).cuda(), torch.from_numpy(cu.astype(np.int32)).cuda()
enc = None
for rep in range(reps):
    if rep % 10 == 0:
        enc = NativeEncoder.from_preset(preset, max_tokens=int(cu[-1]), max_seqs=n)
    out = enc.forward_packed(fd, cd, hidden=True)
    torch.cuda.synchronize()
    p = out["pooled"]
    rows = (~torch.isfinite(p).all(dim=1)).nonzero().flatten().tolist()
    lh = out.get("hidden")
    trows = (~torch.isfinite(lh.float()).all(dim=1)).nonzero().flatten().tolist() if lh is not None else []
    if rows or trows:
        bad.append((rep, rows[:6], len(trows), trows[:3], trows[-3:]))
print({k: v for k, v in os.environ.items() if k.startswith("TSIM_")}, "T=", int(cu[-1]), bad)
