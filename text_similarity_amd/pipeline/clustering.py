"""``ClusteringPipeline`` (/root/reference/src/pipeline/clustering.py:8-31): k-means over corpus embeddings.

The reference delegates to ``sklearn.cluster.KMeans`` on the host.  Here Lloyd's iterations run on the device over
L2-normalised rows (spherical k-means: for unit rows the Euclidean assignment sklearn makes and the cosine assignment
coincide): the assignment step is ONE fused cosine top-1 call (points as queries, centroids as the corpus — the same
kernel as the search path), the update step a scatter-add.  Initialisation: k-means++ on a sample, seeded.
``__call__`` returns ``{cluster_id: [row indices or texts]}`` (the reference builds that dict and forgets to return it,
clustering.py:20-23)."""
from __future__ import annotations

from collections import defaultdict
from typing import List, Union

import numpy as np
import torch

from .. import ops
from .search_pipeline import Pipeline


class ClusteringPipeline(Pipeline):
    def __init__(self, n_clusters, *args, method="k-means", max_iter: int = 50, seed: int = 0, **kwargs):
        super().__init__(*args, **kwargs)
        if method != "k-means":
            raise ValueError("only k-means is implemented (as in the reference)")
        self.method = method
        self.n_clusters = int(n_clusters)
        self.max_iter = max_iter
        self.seed = seed
        self.labels_ = None
        self.cluster_centers_ = None

    def set_n_clusters(self, n: int):
        self.n_clusters = int(n)

    def fit(self, emb: torch.Tensor):
        if not emb.is_cuda:
            emb = emb.cuda()
        emb = emb.float().contiguous()
        n, d = emb.shape
        k = min(self.n_clusters, n)
        unit = ops.l2norm_rows(emb)                                   # float16 [n, ld]
        x = unit[:, :d].float()
        g = torch.Generator(device="cpu").manual_seed(self.seed)
        # k-means++ seeding on (a sample of) the points, cosine distance
        samp = torch.randperm(n, generator=g)[:min(n, 4096)].to(emb.device)
        xs = x[samp]
        first = int(torch.randint(len(samp), (1,), generator=g))
        centers = [xs[first]]
        dist = 1.0 - xs @ centers[0]
        for _ in range(1, k):
            p = torch.clamp(dist, min=0) ** 2
            tot = float(p.sum())
            nxt = int(torch.multinomial((p / tot).cpu(), 1, generator=g)) if tot > 0 else int(torch.randint(len(samp), (1,), generator=g))
            centers.append(xs[nxt])
            dist = torch.minimum(dist, 1.0 - xs @ centers[-1])
        c = torch.stack(centers)
        labels = None
        for _ in range(self.max_iter):
            cu = ops.l2norm_rows(c.contiguous())
            _, idx = ops.cosine_topk(unit, cu, d, 1)                  # assignment: fused cosine top-1
            new = idx[:, 0]
            if labels is not None and torch.equal(new, labels):
                break
            labels = new
            sums = torch.zeros((k, d), dtype=torch.float32, device=emb.device).index_add_(0, labels, x)
            cnt = torch.bincount(labels, minlength=k).unsqueeze(1)
            c = torch.where(cnt > 0, sums / cnt.clamp(min=1), c)      # an emptied cluster keeps its centre
        self.labels_ = labels
        self.cluster_centers_ = c
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
