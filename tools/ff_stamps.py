#!/usr/bin/env python3
"""Where do the fused-FFN kernel's cycles go?  Diagnostic build only (TSIM_BUILD_TAG=stamps python -m text_similarity_amd.build
--stamps --only=encoder.hip; TSIM_LIB=.../libtsim_stamps.so).  Usage: python tools/ff_stamps.py [sentences]"""
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
L.tsim_debug_ff_stamps(buf, 1)
enc.forward_packed(fd, cd, pos, cols, int(np.diff(cu).max()))
torch.cuda.synchronize()
L.tsim_debug_ff_stamps(buf, 0)
for name, o in (("producer wave 0", 0), ("consumer wave 4", 4)):
    ph = max(buf[o], 1)
    print(f"{name}: {buf[o]} phases; cycles per phase: wait (vmcnt + barrier) {buf[o+1]/ph:.0f}, DMA issue {buf[o+2]/ph:.0f}, work {buf[o+3]/ph:.0f}")
