#!/usr/bin/env python3
"""Randomised checks of the reference-named API: encode_text on odd strings (empty, whitespace, unknown words, far too long)
against the oracle on the same tokenisation, order invariance, and SentenceMiningPipeline with random chunk sizes against a
one-chunk search.  Usage: python tools/fuzz_api.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transformers import BertTokenizer
from oracle import encoder_ref
from text_similarity_amd import presets
from text_similarity_amd.configurations.config import Configuration, ModelParameters
from text_similarity_amd.models.sentence_encoder import SentenceTransformerWrapper
from text_similarity_amd.pipeline.search_pipeline import SentenceMiningPipeline

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
preset = "all-MiniLM-L6-v2"
cfg, w = presets.PRESETS[preset], presets.synthetic_weights(preset)
tok = BertTokenizer(vocab=presets.synthetic_vocab(30522), do_lower_case=True)
params = Configuration(model_parameters=ModelParameters(preset, hidden_size=384), model=preset, save_path="", tokenizer=tok,
                       device=torch.device("cuda:0"), batch_size=16, max_tokens_per_batch=2048, max_seqs_per_batch=64,
                       sequence_max_len=64)
model = SentenceTransformerWrapper.from_preset(preset, params, parallel_mode=False)
words = [f"w{i:05d}" for i in range(200, 5000)]
bad = 0
for c in range(cases):
    n = int(rng.integers(1, 300))
    docs = []
    for _ in range(n):
        kind = rng.choice(["normal", "empty", "space", "unk", "long"], p=[0.7, 0.05, 0.05, 0.1, 0.1])
        if kind == "empty": docs.append("")
        elif kind == "space": docs.append("   ")
        elif kind == "unk": docs.append("zzzqqq " + " ".join(rng.choice(words, 3)))
        elif kind == "long": docs.append(" ".join(rng.choice(words, 400)))
        else: docs.append(" ".join(rng.choice(words, int(rng.integers(1, 30)))))
    emb = model.encode_text(docs)
    e = emb.cpu().numpy()
    enc = tok(text=docs, add_special_tokens=True, padding=False, truncation=True, max_length=64,
              return_attention_mask=False, return_token_type_ids=False)["input_ids"]
    lens = np.array([len(x) for x in enc]); cu = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=cu[1:])
    flat = np.array([t for x in enc for t in x], dtype=np.int32)
    ref = encoder_ref.encode_packed(cfg, w, flat, cu, batch_size=8)
    err = float(np.abs(e - ref).max())
    perm = rng.permutation(n)
    e2 = model.encode_text([docs[i] for i in perm]).cpu().numpy()
    ok = emb.shape == (n, 384) and np.isfinite(e).all() and err <= 5e-2 and np.array_equal(e2, e[perm]) and lens.max() <= 64
    # search through the pipeline with a random chunk size == one-chunk search
    if n >= 3:
        k = int(rng.integers(1, min(n, 12) + 1)); chunk = int(rng.integers(1, n + 5))
        p1 = SentenceMiningPipeline(chunk, params, model, corpus=emb); p2 = SentenceMiningPipeline(n + 1, params, model, corpus=emb)
        s1, i1 = p1.search_tensors(emb[: min(n, 20)], None, k); s2, i2 = p2.search_tensors(emb[: min(n, 20)], None, k)
        ok = ok and torch.equal(i1, i2) and torch.equal(s1, s2)
    bad += not ok
    print(f"case {c:2d} n={n:3d} max|err|={err:.4f} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"fuzz_api: {cases - bad}/{cases} ok")
sys.exit(1 if bad else 0)
