"""ORACLE — test infrastructure only.  Never imported by the product package.

CPU restatements (numpy) of the callers right after the hot path:

* ``kmeans_lloyd``          what ``sklearn.cluster.KMeans(n_clusters, init=<array>, n_init=1, tol=0, algorithm="lloyd").fit(x)``
                            computes — the class /root/reference/src/pipeline/clustering.py:2,14 instantiates — on RAW rows:
                            Euclidean assignment (ties -> lower index), centre = mean of its points, until labels repeat
* ``retrieval_accuracy``    RetrievalAccuracyMeter.update  /root/reference/src/utils/metrics.py:466-507
* ``sts_correlations``      EmbeddingSimilarityMeter.update /root/reference/src/utils/metrics.py:357-381

Pinned by tests/test_oracle_golden.py against tests/golden/kmeans.npz and meters.npz, which tools/make_golden.py wrote by
running sklearn's KMeans and the reference's own meter classes in the build container.
"""
from __future__ import annotations

import numpy as np


def kmeans_lloyd(x: np.ndarray, init: np.ndarray, max_iter: int = 300):
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(init, dtype=np.float64).copy()
    labels = None
    for it in range(max_iter):
        d2 = (x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None, :]
        new = d2.argmin(1)
        if labels is not None and np.array_equal(new, labels):
            break
        labels = new
        for j in range(c.shape[0]):
            m = labels == j
            if m.any():
                c[j] = x[m].mean(0)
    inertia = float(((x - c[labels]) ** 2).sum())
    return labels.astype(np.int64), c, inertia


def retrieval_accuracy(src: np.ndarray, tgt: np.ndarray):
    a = src / np.linalg.norm(src, axis=1, keepdims=True)
    b = tgt / np.linalg.norm(tgt, axis=1, keepdims=True)
    sims = a.astype(np.float64) @ b.astype(np.float64).T
    n = sims.shape[0]
    fwd, bwd = sims.argmax(1), sims.T.argmax(1)
    wrong = np.array([[i, fwd[i]] for i in range(n) if fwd[i] != i], dtype=np.int64).reshape(-1, 2)
    s2t, t2s = float((fwd == np.arange(n)).mean()), float((bwd == np.arange(n)).mean())
    return s2t, t2s, (s2t + t2s) / 2, wrong


def sts_correlations(a: np.ndarray, b: np.ndarray, gold: np.ndarray):
    from scipy.stats import pearsonr, spearmanr
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    sims = {"cosine": (a64 * b64).sum(1) / (np.linalg.norm(a64, axis=1) * np.linalg.norm(b64, axis=1)),
            "manhattan": -np.abs(a64 - b64).sum(1), "euclidean": -np.sqrt(((a64 - b64) ** 2).sum(1)),
            "dot": (a64 * b64).sum(1)}
    corr = np.array([[pearsonr(gold, sims[k])[0], spearmanr(gold, sims[k])[0]] for k in ("cosine", "manhattan", "euclidean", "dot")])
    return corr, float(corr[:, 1].max())
