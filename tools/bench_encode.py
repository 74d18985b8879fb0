#!/usr/bin/env python3
"""Quick encoder timing: python tools/bench_encode.py [preset] [n_sentences] [iters] [bf16|mxfp8]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from text_similarity_amd import presets
from text_similarity_amd.native_encoder import NativeEncoder

preset = sys.argv[1] if len(sys.argv) > 1 else "all-MiniLM-L6-v2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
wdtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
cfg = presets.PRESETS[preset]
flat, cu = presets.synthetic_token_batch(n, seed="sent1234", vocab_size=cfg.vocab, max_len=256)
T = int(cu[-1])
enc = NativeEncoder.from_preset(preset, max_tokens=T, max_seqs=n, weight_dtype=wdtype)
fd, cd = torch.from_numpy(flat).cuda(), torch.from_numpy(cu).cuda()
pos, cols = enc.positions(fd, cd)
for _ in range(2):
    enc.forward_packed(fd, cd, pos, cols, int(np.diff(cu).max()), pooled=True, unit=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    enc.forward_packed(fd, cd, pos, cols, int(np.diff(cu).max()), pooled=True, unit=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
H, F, L = cfg.hidden, cfg.ffn, cfg.num_layers
sbar = float((np.diff(cu).astype(np.float64) ** 2).sum() / T)
flops = T * L * (2 * (4 * H * H + 2 * H * F) + 4 * sbar * H)
print(json.dumps({"preset": preset, "weight_dtype": wdtype, "sentences": n, "tokens": T, "ms": round(ms, 3),
                  "sentences_per_s": round(n / ms * 1e3), "tokens_per_s": round(T / ms * 1e3),
                  "TFLOPs": round(flops / ms / 1e9, 1)}))
