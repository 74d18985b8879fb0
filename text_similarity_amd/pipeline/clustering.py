"""``ClusteringPipeline`` (/root/reference/src/pipeline/clustering.py:8-31): k-means over corpus embeddings.

The reference delegates to ``sklearn.cluster.KMeans`` on the host, on the RAW (un-normalised) embeddings.  Here Lloyd's
iterations run on the device with the same semantics: Euclidean assignment
``argmin_j |x - c_j|^2 = argmax_j (x.c_j - |c_j|^2 / 2)`` (ties to the lower index, as sklearn's argmin), centre = mean
of its points, until the labels repeat.  The scores come from the dense cosine kernel of the search path (``tsim_cos_sim``:
``x.c = cos(x, c) |x| |c|``), the update step is a scatter-add.  Initial centres: ``init=<array>`` (what the parity
fixture uses: tests/golden/kmeans.npz pins labels / centres / inertia against sklearn with the same initial centres) or
k-means++ seeded from ``seed`` (sklearn's own k-means++ draws from its private RNG stream and cannot be reproduced).
An emptied cluster keeps its centre (sklearn relocates it to a far point; with k-means++ starts this does not occur on
the fixtures).  ``__call__`` returns ``{cluster_id: [row indices or texts]}`` (the reference builds that dict and forgets
to return it, clustering.py:20-23)."""
from __future__ import annotations

from collections import defaultdict
from typing import List, Optional, Union

import numpy as np
import torch

from .. import ops
from .search_pipeline import Pipeline


class ClusteringPipeline(Pipeline):
    def __init__(self, n_clusters, *args, method="k-means", max_iter: int = 300, seed: int = 0, init=None, **kwargs):
        super().__init__(*args, **kwargs)
        if method != "k-means":
            raise ValueError("only k-means is implemented (as in the reference)")
        self.method = method
        self.n_clusters = int(n_clusters)
        self.max_iter = max_iter
        self.seed = seed
        self.init = init
        self.labels_ = None
        self.cluster_centers_ = None
        self.inertia_ = None
        self.n_iter_ = 0

    def set_n_clusters(self, n: int):
        self.n_clusters = int(n)

    @staticmethod
    def _assign(x: torch.Tensor, xn: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
        """Euclidean assignment: argmax_j (x.c_j - |c_j|^2 / 2) with x.c from the dense cosine kernel."""
        cn = c.norm(dim=1)
        cos = torch.nan_to_num(ops.cos_sim_dense(x, c), nan=0.0)     # a zero row has no direction: x.c = 0
        score = cos * xn[:, None] * cn[None, :] - 0.5 * (cn * cn)[None, :]
        return score.argmax(dim=1)                                    # first maximum = lower index, like numpy / sklearn

    def _kmeanspp(self, x: torch.Tensor, k: int) -> torch.Tensor:
        """Greedy k-means++ (the variant sklearn uses: 2 + log k candidates per step, keep the one that lowers the potential
        most) on a sample of the points; seeded."""
        import math
        g = torch.Generator(device="cpu").manual_seed(self.seed)
        n = x.shape[0]
        samp = torch.randperm(n, generator=g)[:min(n, 4096)].to(x.device)
        xs = x[samp]
        trials = 2 + int(math.log(max(k, 1)))
        centers = [xs[int(torch.randint(len(samp), (1,), generator=g))]]
        dist = (xs - centers[0]).pow(2).sum(1)
        for _ in range(1, k):
            tot = float(dist.sum())
            if tot > 0:
                cand = torch.multinomial((dist / tot).cpu(), trials, replacement=True, generator=g).to(x.device)
            else:
                cand = torch.randint(len(samp), (trials,), generator=g).to(x.device)
            dc = torch.cdist(xs[cand], xs).pow(2)                     # [trials, m]
            pot = torch.minimum(dist[None, :], dc).sum(1)
            best = int(pot.argmin())
            centers.append(xs[cand[best]])
            dist = torch.minimum(dist, dc[best])
        return torch.stack(centers)

    def fit(self, emb: torch.Tensor, init: Optional[Union[torch.Tensor, np.ndarray]] = None):
        if not emb.is_cuda:
            emb = emb.cuda()
        x = emb.float().contiguous()
        n, d = x.shape
        k = min(self.n_clusters, n)
        init = self.init if init is None else init
        if init is not None:
            c = torch.as_tensor(np.asarray(init) if not isinstance(init, torch.Tensor) else init).to(x.device).float().contiguous()
            if c.shape != (k, d):
                raise ValueError(f"init must have shape ({k}, {d}), got {tuple(c.shape)}")
        else:
            c = self._kmeanspp(x, k)
        xn = x.norm(dim=1)
        labels = None
        self.n_iter_ = 0
        for it in range(self.max_iter):
            new = self._assign(x, xn, c)
            if labels is not None and torch.equal(new, labels):
                break
            labels = new
            self.n_iter_ = it + 1
            sums = torch.zeros((k, d), dtype=torch.float32, device=x.device).index_add_(0, labels, x)
            cnt = torch.bincount(labels, minlength=k).unsqueeze(1)
            c = torch.where(cnt > 0, sums / cnt.clamp(min=1), c).contiguous()      # an emptied cluster keeps its centre
        self.labels_ = labels
        self.cluster_centers_ = c
        self.inertia_ = float((x - c[labels]).pow(2).sum())
        return self

    def _cluster(self, corpus: Union[List[str], torch.Tensor, np.ndarray]):
        texts = corpus if isinstance(corpus, list) else None
        emb = self.encode_corpus(corpus) if isinstance(corpus, list) else corpus
        if isinstance(emb, np.ndarray):
            emb = torch.from_numpy(emb)
        self.fit(emb)
        results = defaultdict(list)
        for text_id, cluster_id in enumerate(self.labels_.tolist()):
            results[cluster_id].append(texts[text_id] if texts is not None else text_id)
        return dict(results)

    def __call__(self, embeddings, n_clusters=None):
        if n_clusters is not None:
            self.set_n_clusters(n_clusters)
        return self._cluster(embeddings)
