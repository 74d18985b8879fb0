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

import os

from .. import ops
from ..index import GpuFlatIndex


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
        """Device-level search: returns (scores [Q,k] f32, indices [Q,k] i64) over the whole corpus.  Scores are the
        reference's ``F.cosine_similarity`` of the float32 embeddings (search_pipeline.py:76-77) and the order is exact for
        them: half-precision unit rows feed the MFMA kernel for candidate selection only.  1 <= max_num_results <= 64, width <= 768."""
        corpus = self.corpus if corpus is None else corpus
        n = len(corpus)
        d = query_embeddings.shape[1]
        qf = query_embeddings.to(self.params.device, dtype=torch.float32).contiguous()
        qn = ops.l2norm_rows(qf)
        k = min(max_num_results, len(query_embeddings)) if self.reference_k_clamp else max_num_results
        k = max(1, min(k, n))
        scores, idxs = [], []
        for start in range(0, n, self.corpus_chunk_size):
            chunk = corpus[start:start + self.corpus_chunk_size]
            if isinstance(chunk, list):
                chunk = self.model.encode_text(chunk)
            cf = chunk.to(self.params.device, dtype=torch.float32).contiguous()
            cn, rho = ops.l2norm_rows(cf, return_rho=True)     # rho: the chunk's rounding-residual maximum (guard bound)
            s, i = ops.cosine_topk(qn, cn, d, min(k, cn.shape[0]), idx_offset=start, eq_f32=qf, ec_f32=cf, rho_c=rho)
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


class SemanticSearchPipeline(SearchPipeline):
    """/root/reference/src/pipeline/search_pipeline.py:96-175 with the hnswlib ANN index replaced by an exact index in
    HBM (:class:`text_similarity_amd.index.GpuFlatIndex`): same constructor (``index_path`` first), ``_index``,
    ``_search`` / ``__call__`` returning ``{query_idx: [texts best-first]}``, ``add_to_index``, ``remove_from_index``,
    ``num_indexed``.  Differences: results are exact; ``ef`` / ``ef_construction`` / ``M`` are accepted and unused (the
    reference's ``assert max_num_results < ef`` has no meaning here); ``add_to_index`` also appends the texts to
    ``self.corpus`` — the reference only grows the index, so its new ids cannot be mapped back to text.
    ``max_num_results`` up to 64 (the reference's bound is ``ef`` = 50, search_pipeline.py:131); width <= 768."""

    def __init__(self, index_path, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.index_path = index_path
        hidden = getattr(self.params.model_parameters, "hidden_size", None) or self.model.get_sentence_embedding_dimension()
        self.index = GpuFlatIndex(space="cosine", dim=hidden, device=self.params.device)
        if os.path.exists(os.path.join(self.index_path, "index.bin")):
            self.index.load_index(self.index_path)
        else:
            self._index(self.corpus)

    def _index(self, corpus):
        os.makedirs(self.index_path, exist_ok=True)
        corpus_embeddings = self.encode_corpus(corpus)
        self.index.init_index(max_elements=len(corpus), ef_construction=getattr(self.params, "ef_construction", 0),
                              M=getattr(self.params, "M", 0))
        self.index.add_items(corpus_embeddings, list(range(corpus_embeddings.shape[0])))
        self.index.save_index(self.index_path)
        self.index.set_ef(getattr(self.params, "ef", 0))

    def _search(self, queries, max_num_results: int):
        query_embeddings = self.encode_corpus(queries)
        labels, scores = self.index.search(query_embeddings, max_num_results)
        self.last_labels, self.last_scores = labels, scores
        top_results = {}
        for qidx, row in enumerate(labels.cpu().tolist()):
            top_results[qidx] = [self.corpus[i] for i in row if i >= 0]
        return top_results

    def __call__(self, queries, max_num_results: int):
        return self._search(queries, max_num_results)

    def add_to_index(self, text):
        if isinstance(text, str):
            text = [text]
        embeddings = self.encode_corpus(list(text))
        first = len(self.corpus)
        self.index.resize_index(self.index.get_current_count() + embeddings.shape[0])
        self.index.add_items(embeddings, list(range(first, first + embeddings.shape[0])))
        self.corpus = list(self.corpus) + list(text)

    def remove_from_index(self, ids):
        for id in ids:
            try:
                self.index.mark_deleted(id)
            except RuntimeError:
                continue      # can't find id, continue (search_pipeline.py:167-169)

    def num_indexed(self):
        """current number of indexed embeddings"""
        return self.index.num_live()


class APISearchPipeline(SemanticSearchPipeline):
    """/root/reference/src/pipeline/search_pipeline.py:178-226: the serving variant of ``SemanticSearchPipeline`` whose
    query encoder is an ``onnxruntime.InferenceSession`` over ``params.model_path``.  Here the "session" is the native
    MI355X encoder the pipeline was built with (there is no ONNX runtime on this path): same constructor
    (``params, max_n_results, *args, inference_mode=True, session_options=None``), same ``__call__`` and the same
    ``encode_corpus(documents)`` contract — a list of per-sentence embedding rows in the caller's order (the reference
    sorts by length for batching and un-sorts, :200-226; ``encode_text`` does the same on the device).  The reference's loop
    reshapes every batch to ONE row before ``session.run`` (:217-220), which only works for batches of one sentence; the
    intended per-sentence embeddings are what is returned."""

    def __init__(self, params, max_n_results: int, *args, inference_mode: bool = True, session_options=None, **kwargs):
        # the reference forwards *args to SemanticSearchPipeline(index_path, params, model): its callers pass params a second
        # time there; (index_path, model) alone is accepted as well
        args = list(args)
        if len(args) == 2 and "model" not in kwargs:
            args.insert(1, params)
        super().__init__(*args, **kwargs)
        self.params = params
        self.inference_mode = inference_mode
        self.sess_options = session_options
        self.max_n_results = max_n_results
        self.session = self.model          # what runs the encoder forward

    def __call__(self, queries, max_num_results: Optional[int] = None):
        return self._search(queries, self.max_n_results if max_num_results is None else max_num_results)

    def encode_corpus(self, documents, convert_to_numpy: bool = False, return_embeddings: bool = False):
        if not isinstance(documents, list):
            return documents
        emb = self.model.encode_text(documents, output_np=False)
        return emb
