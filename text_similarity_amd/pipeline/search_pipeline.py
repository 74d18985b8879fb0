"""Embed-and-search pipelines with the reference's class names and call signatures
(/root/reference/src/pipeline/search_pipeline.py:14-93).

``SentenceMiningPipeline`` is the brute-force searcher: in the reference a Python loop over queries doing
``expand_as`` + ``F.cosine_similarity`` + ``torch.topk`` per corpus chunk (:60-89).  Here each chunk is ONE call of
the fused MFMA cosine + top-k kernel and chunks are merged on the GPU.  Shipped bugs are not reproduced
(SURVEY.md §8 A5/A6): the chunk slice (:61) takes ``corpus[i : i + chunk]``, ``topk`` runs over the corpus axis,
``__call__`` passes the stored corpus, and k is clamped by the chunk size (``reference_k_clamp=True`` restores the
reference's clamp by ``len(queries)``, :78).  Ordering within a result list is (score desc, index asc); the
reference asks for ``sorted=False`` and leaves it undefined.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import torch

from .. import ops


class Pipeline:
    def __init__(self, params, model, name: Optional[str] = None):
        self.params = params
        self.model = model
        self.name = name

    def encode_corpus(self, documents: Union[List[str], torch.Tensor], convert_to_numpy: bool = False,
                      return_embeddings: bool = False):
        if isinstance(documents, list):
            return self.model.encode_text(documents, output_np=convert_to_numpy)
        return documents


class SearchPipeline(Pipeline):
    def __init__(self, *args, corpus: Optional[Union[List[str], torch.Tensor]] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.corpus = corpus

    def _index(self, corpus):
        raise NotImplementedError()

    def _search(self, queries, max_num_results: int):
        raise NotImplementedError()

    def __call__(self, queries, max_num_results):
        return self._search(queries, max_num_results)


class SentenceMiningPipeline(SearchPipeline):
    def __init__(self, corpus_chunk_size: int, *args, reference_k_clamp: bool = False, verbose: bool = False,
                 **kwargs):
        super().__init__(*args, **kwargs)
        self.corpus_chunk_size = int(corpus_chunk_size)
        self.reference_k_clamp = reference_k_clamp
        self.verbose = verbose
        self.last_scores = None      # [Q,k] float32 of the last search (the reference only prints indices)
        self.last_indices = None

    def search_tensors(self, query_embeddings: torch.Tensor, corpus=None, max_num_results: int = 10):
        """Device-level search: returns (scores [Q,k] f32, indices [Q,k] i64) over the whole corpus."""
        corpus = self.corpus if corpus is None else corpus
        n = len(corpus)
        d = query_embeddings.shape[1]
        qn = ops.l2norm_rows(query_embeddings.to(self.params.device))
        k = min(max_num_results, len(query_embeddings)) if self.reference_k_clamp else max_num_results
        k = max(1, min(k, n))
        scores, idxs = [], []
        for start in range(0, n, self.corpus_chunk_size):
            chunk = corpus[start:start + self.corpus_chunk_size]
            if isinstance(chunk, list):
                chunk = self.model.encode_text(chunk)
            cn = ops.l2norm_rows(chunk.to(self.params.device))
            s, i = ops.cosine_topk(qn, cn, d, min(k, cn.shape[0]), idx_offset=start)
            if s.shape[1] < k:   # short last chunk: pad so lists stack
                pad = k - s.shape[1]
                s = torch.cat([s, torch.full((s.shape[0], pad), float("-inf"), device=s.device)], 1)
                i = torch.cat([i, torch.full((i.shape[0], pad), -1, dtype=torch.int64, device=i.device)], 1)
            scores.append(s)
            idxs.append(i)
        if len(scores) == 1:
            return scores[0], idxs[0]
        return ops.topk_merge(scores, idxs, k)

    def _search(self, queries, corpus=None, max_num_results: int = 10, return_embeddings: bool = False
                ) -> Dict[int, Union[list, torch.Tensor]]:
        query_embeddings = self.encode_corpus(documents=queries, return_embeddings=return_embeddings)
        if corpus is not None:
            self.corpus = corpus
        scores, indices = self.search_tensors(query_embeddings, self.corpus, max_num_results)
        self.last_scores, self.last_indices = scores, indices
        top_candidates = {}
        idx_host = indices.cpu()
        for query_idx in range(idx_host.shape[0]):
            actual = idx_host[query_idx]
            actual = actual[actual >= 0]
            if self.verbose:
                print(f"Top candidates indexes: {actual}")
            if return_embeddings:
                assert isinstance(self.corpus, torch.Tensor)
                top_candidates[query_idx] = self.corpus[actual.to(self.corpus.device)]
            else:
                top_candidates[query_idx] = [(int(c), self.corpus[int(c)]) for c in actual]
        return top_candidates

    def __call__(self, queries, max_num_results: int, return_embeddings: bool = False):
        return self._search(queries, None, max_num_results, return_embeddings)
