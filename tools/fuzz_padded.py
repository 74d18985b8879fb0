#!/usr/bin/env python3
"""The HF-contract entry (padded [B,S] ids + attention mask, masks with HOLES, pad tokens inside MPNet rows) against the
fp32 oracle forward on the same padded batch: hidden states of valid positions and mean-pooled rows.
Usage: python tools/fuzz_padded.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import encoder_ref
from text_similarity_amd import ops, presets
from text_similarity_amd.native_encoder import NativeEncoder

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
encs, bad = {}, 0
for c in range(cases):
    preset = str(rng.choice(["tiny-bert", "tiny-mpnet", "all-MiniLM-L6-v2", "all-mpnet-base-v2"]))
    cfg = presets.PRESETS[preset]
    if preset not in encs:
        encs[preset] = (NativeEncoder.from_preset(preset, max_tokens=4096, max_seqs=32), presets.synthetic_weights(preset))
    enc, w = encs[preset]
    B = int(rng.integers(1, 9)); S = int(rng.choice([1, 2, 7, 16, 33, 60]))
    S = min(S, cfg.max_pos - 2)
    ids = rng.integers(5, cfg.vocab, (B, S)).astype(np.int64)
    mask = np.ones((B, S), dtype=np.int64)
    for b in range(B):
        kind = rng.choice(["full", "prefix", "holes", "single", "empty"], p=[0.3, 0.3, 0.25, 0.1, 0.05])
        if kind == "prefix": mask[b, int(rng.integers(1, S + 1)):] = 0
        elif kind == "holes": mask[b] = rng.random(S) < 0.6
        elif kind == "single": mask[b] = 0; mask[b, int(rng.integers(S))] = 1
        elif kind == "empty": mask[b] = 0
    if cfg.arch == "mpnet":
        ids[mask == 0] = cfg.pad_id                         # what a tokenizer pads with; also sprinkle pad ids inside rows
        ids[rng.random((B, S)) < 0.03] = cfg.pad_id
    hid = enc(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda())[0]
    pooled = ops.mean_pool(hid, torch.from_numpy(mask).cuda()).cpu().numpy()
    hid = hid.cpu().numpy()
    with torch.no_grad():
        rh = encoder_ref.encoder_forward(cfg, w, ids, mask)
        rp = encoder_ref.mean_pool(rh, mask).numpy()
    rh = rh.numpy()
    rows_ok = mask.sum(1) > 0                                # all-masked rows: documented don't-care for hidden states
    m3 = (mask.astype(bool) & rows_ok[:, None])
    eh = float(np.abs(hid - rh)[m3].max()) if m3.any() else 0.0
    ep = float(np.abs(pooled - rp)[rows_ok].max()) if rows_ok.any() else 0.0
    # a pooled row is an average of hidden rows, so its error is bounded by theirs (a one-token row IS its hidden row: the old
    # 5e-2 pooled bound was tighter than the 8e-2 hidden bound and tripped on such a row of the 12-layer model); the pooling itself
    # is checked tightly against the mean of the kernel's own hidden states
    den = np.maximum(mask.sum(1, keepdims=True), 1e-9)
    self_pool = (hid.astype(np.float64) * mask[:, :, None]).sum(1) / den
    es = float(np.abs(pooled - self_pool).max())
    ok = (np.isfinite(hid).all() and eh <= 8e-2 and ep <= 8e-2 and es <= 1e-5 and (hid[mask == 0] == 0).all()
          and (pooled[~rows_ok] == 0).all())
    bad += not ok
    print(f"case {c:2d} {preset:18s} B={B} S={S:2d} hidden err={eh:.4f} pooled err={ep:.4f} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"fuzz_padded: {cases - bad}/{cases} ok")
sys.exit(1 if bad else 0)
