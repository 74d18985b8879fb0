#!/usr/bin/env python3
"""Where do a ping-pong GEMM tile's cycles go?  Needs the diagnostic build: python -m text_similarity_amd.build --stamps
(rebuild with --force afterwards).  Usage: python tools/pp_stamps.py [preset] [n_sentences] [bf16|mxfp8]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from text_similarity_amd import presets, _lib
from text_similarity_amd.native_encoder import NativeEncoder

preset = sys.argv[1] if len(sys.argv) > 1 else "bert-base-uncased"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wd = sys.argv[3] if len(sys.argv) > 3 else "bf16"
cfg = presets.PRESETS[preset]
flat, cu = presets.synthetic_token_batch(n, seed="sent1234", vocab_size=cfg.vocab, max_len=256)
enc = NativeEncoder.from_preset(preset, max_tokens=int(cu[-1]), max_seqs=n, weight_dtype=wd)
fd, cd = torch.from_numpy(flat).cuda(), torch.from_numpy(cu).cuda()
pos, cols = enc.positions(fd, cd)
L = C.CDLL(_lib.lib()._name)
buf = (C.c_ulonglong * 8)()
for _ in range(2):
    enc.forward_packed(fd, cd, pos, cols, int(np.diff(cu).max()), pooled=True, unit=True)
torch.cuda.synchronize()
L.tsim_debug_pp_stamps(buf, 1)
L.tsim_debug_xr_stamps(buf, 1)
enc.forward_packed(fd, cd, pos, cols, int(np.diff(cu).max()), pooled=True, unit=True)
torch.cuda.synchronize()
L.tsim_debug_pp_stamps(buf, 0)
tiles, top, loop, epi, mf, ls, bw, dw = [buf[i] for i in range(8)]
if tiles:
    print(f"{preset} {wd}: {tiles} tiles (all ping-pong launches of one forward); per tile (cycles of wave 0): "
          f"top wait {top / tiles:.0f}, k loop {loop / tiles:.0f} (MFMA sections {mf / tiles:.0f}, load sections {ls / tiles:.0f}, barriers {bw / tiles:.0f}, "
          f"wait for the next k-tile's DMA {dw / tiles:.0f}), epilogue {epi / tiles:.0f}")
    
L.tsim_debug_xr_stamps(buf, 0)
steps, wait, iss, comp, epi, xl, items = [buf[i] for i in range(7)]
if steps:
    print(f"resident-X kernel: {steps} tile-steps, {items} items; cycles of wave 0 per tile-step: wait {wait / steps:.0f}, DMA issue "
          f"{iss / steps:.0f}, reads + MFMAs {comp / steps:.0f}, activation reload {xl / steps:.0f}; per item: epilogue {epi / max(items, 1):.0f}")
