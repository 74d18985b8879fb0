#!/usr/bin/env python3
"""Randomised sequences of add / delete / search / save+load on GpuFlatIndex against a plain Python model of the same
index (live rows in insertion order + the oracle's exact top-k of the reference's float32 cosine).  Usage: python tools/fuzz_index.py [rounds] [seed]"""
import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import search_ref
from text_similarity_amd.index import GpuFlatIndex

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, t0 = 0, time.time()
for rd in range(rounds):
    d = int(rng.choice([64, 100, 384, 768]))
    idx = GpuFlatIndex(space="cosine", dim=d, device="cuda:0")
    idx.init_index(max_elements=int(rng.choice([1, 10, 1000])))
    rows, labels, next_label = [], [], 1000
    for op in range(25):
        what = rng.choice(["add", "add", "del", "search", "search", "persist"])
        if what == "add":
            n = int(rng.choice([1, 2, 33, 500, 3000]))
            x = rng.standard_normal((n, d)).astype(np.float32)
            if rows and rng.random() < 0.3:
                x[0] = rows[int(rng.integers(len(rows)))]            # duplicate of a live row: tie by insertion order
            lab = np.arange(next_label, next_label + n) * 3 + 1      # labels are not row numbers
            next_label += n
            idx.add_items(torch.from_numpy(x) if rng.random() < 0.5 else x, lab)
            rows += list(x); labels += list(lab)
        elif what == "del" and rows:
            for _ in range(int(rng.choice([1, 5, 40]))):
                if not rows:
                    break
                j = int(rng.integers(len(rows)))
                idx.mark_deleted(int(labels[j]))
                del rows[j]; del labels[j]
        elif what == "persist" and rows:
            with tempfile.TemporaryDirectory() as tmp:
                idx.save_index(tmp)
                idx = GpuFlatIndex(space="cosine", dim=0, device="cuda:0")
                idx.load_index(tmp)
        else:
            Q, k = int(rng.choice([1, 7, 40])), int(rng.choice([1, 5, 10, 20]))
            q = rng.standard_normal((Q, d)).astype(np.float32)
            if rows and rng.random() < 0.5:
                q[0] = rows[int(rng.integers(len(rows)))]
            lab, sc = idx.search(q, k)
            lab, sc = lab.cpu().numpy(), sc.cpu().numpy()
            if not rows:
                ok = (lab == -1).all() and np.isneginf(sc).all()
            else:
                live, ll = np.stack(rows), np.asarray(labels)
                kk = min(k, len(rows))
                rs, ri = search_ref.cosine_topk_f32(q, live, kk)      # the index keeps the float32 rows: the reference's cosine
                ok = np.array_equal(lab[:, :kk], ll[ri]) and np.array_equal(sc[:, :kk], rs) and (lab[:, kk:] == -1).all()
            ok = ok and idx.num_live() == len(rows)
            bad += not ok
            if not ok:
                print(f"round {rd} op {op}: MISMATCH (d={d}, live={len(rows)}, Q={Q}, k={k})", flush=True)
    print(f"round {rd} d={d} live={len(rows)} ok so far: {bad == 0}  ({time.time() - t0:.0f} s)", flush=True)
print(f"fuzz_index: {'all searches exact' if bad == 0 else str(bad) + ' mismatches'}")
sys.exit(1 if bad else 0)
