#!/usr/bin/env python3
"""Where do gemm_xres2's cycles go (QKV + FFN1 of one MiniLM forward)?  Diagnostic build only:
TSIM_BUILD_TAG=stamps python -m text_similarity_amd.build --stamps [-DTSIM_XR_STAMP_TID=256]; TSIM_LIB=.../libtsim_stamps.so."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from text_similarity_amd import _lib, presets
from text_similarity_amd.native_encoder import NativeEncoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = presets.PRESETS["all-MiniLM-L6-v2"]
flat, cu = presets.synthetic_token_batch(n, seed="sent1234", vocab_size=cfg.vocab, max_len=256)
enc = NativeEncoder.from_preset("all-MiniLM-L6-v2", max_tokens=int(cu[-1]), max_seqs=n)
fd, cd = torch.from_numpy(flat).cuda(), torch.from_numpy(cu).cuda()
pos, cols = enc.positions(fd, cd)
L = C.CDLL(_lib.lib()._name)
buf = (C.c_ulonglong * 8)()
for _ in range(3):
    enc.forward_packed(fd, cd, pos, cols, int(np.diff(cu).max()))
torch.cuda.synchronize()
L.tsim_debug_xr_stamps(buf, 1)
enc.forward_packed(fd, cd, pos, cols, int(np.diff(cu).max()))
torch.cuda.synchronize()
L.tsim_debug_xr_stamps(buf, 0)
steps, wait, _, comp, store, pro, items = [buf[i] for i in range(7)]
print(f"gemm_xres2 (QKV + FFN1, 6 layers): {steps} steps, {items} items of the stamped wave over all workgroups; cycles per step: "
      f"wait (vmcnt + barrier) {wait / steps:.0f}, reads + MFMAs + shadow epilogue {comp / steps:.0f}, swap + stores {store / steps:.0f}; "
      f"per item: fragment reload / bias init {pro / max(items, 1):.0f}")
