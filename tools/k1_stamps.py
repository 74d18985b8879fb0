#!/usr/bin/env python3
"""Where do the search main pass's cycles go?  Diagnostic build only: python -m text_similarity_amd.build --stamps
(rebuild with --force afterwards).  Usage: python tools/k1_stamps.py [N] [d] [Q]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_similarity_amd import ops, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
ec = ops.l2norm_rows(torch.randn(N, d, device="cuda", generator=g))
eq = ops.l2norm_rows(torch.randn(Q, d, device="cuda", generator=g))
L = C.CDLL(_lib.lib()._name)
buf = (C.c_ulonglong * 24)()
for _ in range(2):
    ops.cosine_topk(eq, ec, d, 10)
torch.cuda.synchronize()
L.tsim_debug_k1_stamps(buf, 24, 1)
ops.cosine_topk(eq, ec, d, 10)
torch.cuda.synchronize()
L.tsim_debug_k1_stamps(buf, 24, 0)
if buf[8]:      # ping-pong schedule (TSIM_K1_PP=1)
    for name, o in (("group 0 (wave 0)", 8), ("group 1 (wave 4)", 16)):
        tiles, l0, l1, b1, m, b2, tot = [buf[o + i] for i in range(7)]
        print(f"ping-pong {name}: cycles per tile: M (reads + MFMAs) {m / tiles:.0f}, barrier after M {b1 / tiles:.0f}, "
              f"L (filter + DMA issue) {l0 / tiles:.0f} of which filter {l1 / tiles:.0f}, barrier after L {b2 / tiles:.0f}, "
              f"whole loop {tot / tiles:.0f}; candidate events per tile (wave 0) {buf[15] / max(buf[8], 1):.3f}")
    sys.exit(0)
pairs, dma, bar, iss, cmp_, pro, epi, wgs = [buf[i] for i in range(8)]
print(f"main pass Q={Q} N={N} d={d}: {wgs} workgroups, {pairs / wgs:.1f} tile pairs each; cycles of wave 0 per pair: wait for DMA "
      f"{dma / pairs:.0f}, barrier {bar / pairs:.0f}, DMA issue {iss / pairs:.0f}, reads + MFMAs + selection {cmp_ / pairs:.0f}; "
      f"per workgroup: prologue {pro / wgs:.0f}, epilogue {epi / wgs:.0f}")
