"""``RankingPipeline`` (/root/reference/src/pipeline/ranking_pipeline.py:4-43): bi-encoder retrieval of ``top_k``
candidates per query (the fused cosine top-k of ``SentenceMiningPipeline``) followed by a caller-supplied cross-encoder
that re-scores each (query, candidate) pair.  The cross-encoder is outside the hot path and is used only through its
``predict(list of [query, text]) -> scores`` method, as in the reference.  Returns one dict per query:
``{'results': [{'corpus_id', 'text', 'score', 'cross-score'}...] sorted by cross-score, 'cross_scores', 'avg_score'}``
(the reference's loop reuses the first query's hits for every query, :27-28,33-38; here each query keeps its own)."""
from __future__ import annotations

from typing import List

from .search_pipeline import SentenceMiningPipeline


class RankingPipeline(SentenceMiningPipeline):
    def __init__(self, cross_encoder, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.cross_encoder = cross_encoder

    def _rank(self, queries: List[str], corpus: List[str], top_k=5, search_first=True) -> List[dict]:
        results = []
        if search_first:
            hits = self._search(queries, corpus, max_num_results=top_k)     # {query index: [(corpus id, text), ...]}
            scores = self.last_scores.tolist()
        for qi, query in enumerate(queries):
            if search_first:
                cand = [{"corpus_id": int(i), "text": t, "score": scores[qi][r]} for r, (i, t) in enumerate(hits[qi])]
            else:
                cand = [{"corpus_id": i, "text": t, "score": None} for i, t in enumerate(corpus)]
            cross_scores = list(self.cross_encoder.predict([[query, c["text"]] for c in cand]))
            for c, s in zip(cand, cross_scores):
                c["cross-score"] = float(s)
            cand.sort(key=lambda c: c["cross-score"], reverse=True)
            results.append({"results": cand, "cross_scores": [float(s) for s in cross_scores],
                            "avg_score": float(sum(cross_scores) / max(len(cross_scores), 1))})
        return results

    def __call__(self, queries: List[str], corpus: List[str], top_k=5, search_first=True):
        return self._rank(queries, corpus, top_k, search_first)
