"""Corpus-sharded search over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm).  The reference has no multi-device code at all; this is the scale-out of
``SentenceMiningPipeline._search`` (/root/reference/src/pipeline/search_pipeline.py:60-89) for BASELINE.json config 4.

Partitioning: rank r owns corpus rows [offset_r, offset_r + n_r) — the float32 embeddings and their unit float16 rows,
resident in its HBM; the corpus never moves.  Per query batch there are exactly two exchanges, both tiny and
latency-bound:
  1. all-gather of the query embeddings   [Q_local, d] float32 per rank  -> [Q, d] everywhere (unit rows are made locally)
  2. all-gather of per-shard candidates   [Q, k] (score f32, global index i64) per rank, ONE buffer per rank that the search
     kernel writes directly ([scores | indices], ops.packed_result_views) and the merge kernel reads in place
followed by a k-way merge with the global tie rule (score desc, index asc).  On the GPU path ``finish`` issues exactly:
unit rows of the queries -> search -> all-gather -> merge; no torch arithmetic, no re-packing copies.  Every shard returns the exact top-k of its
rows with exact scores, so the result is bit-identical to a single-GPU search over the concatenated corpus
(tests/test_sharded_cpu.py, world_size 2 over gloo; tests/test_sharded_gpu.py with the HIP kernels).

``submit`` / ``finish`` split a search at the first exchange: ``submit`` starts the query all-gather of batch i+1 on a side
stream while the local search of batch i still runs on the main stream; ``search`` = ``finish(submit)``.

``local_search`` / ``merge`` are injectable so that the collective choreography can be exercised on CPU ranks
(gloo) with the oracle standing in for the kernels — in tests only; the defaults are the HIP ops and raise without
a GPU.
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _hip_local_search(q_f32, c_unit, c_f32, d, k, offset, rho_c=None, out=None):
    from .. import ops
    if c_f32 is None:                       # unit rows only: q_f32 holds unit float16 rows
        return ops.cosine_topk(q_f32, c_unit, d, k, idx_offset=offset, out=out)
    q_unit = ops.l2norm_rows(q_f32)
    return ops.cosine_topk(q_unit, c_unit, d, k, idx_offset=offset, eq_f32=q_f32, ec_f32=c_f32, rho_c=rho_c, out=out)


def _packed_bytes(Q: int, k: int) -> int:           # == ops.packed_result_bytes (kept here so CPU rehearsals need no GPU ops)
    return (Q * k * 4 + 7) // 8 * 8 + Q * k * 8


def _packed_views(buf: torch.Tensor, Q: int, k: int):
    so = (Q * k * 4 + 7) // 8 * 8
    s = buf[..., :Q * k * 4].view(torch.float32)
    i = buf[..., so:so + Q * k * 8].view(torch.int64)
    return s.unflatten(-1, (Q, k)), i.unflatten(-1, (Q, k))


def _hip_merge(scores: torch.Tensor, idx: torch.Tensor, k: int):
    from .. import ops
    return ops.topk_merge(scores, idx, k)


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Block partition of n_total rows: rank r owns [lo, hi)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _Ticket:
    __slots__ = ("q_all", "event", "counts")

    def __init__(self, q_all, event, counts=None):
        self.q_all, self.event, self.counts = q_all, event, counts


class ShardedCorpusSearch:
    def __init__(self, corpus_unit_local: torch.Tensor, d: int, row_offset: int,
                 group: Optional[dist.ProcessGroup] = None,
                 local_search: Callable = _hip_local_search, merge: Callable = _hip_merge,
                 corpus_f32_local: Optional[torch.Tensor] = None, corpus_rho: Optional[torch.Tensor] = None):
        """``corpus_f32_local`` [n_r, d] float32: the embeddings (scores are then the reference's cosines of float32 rows and
        queries are passed as float32 embeddings); without it queries are unit float16 rows and scores their inner products.
        ``corpus_rho``: the shard's rounding-residual maximum from ``ops.l2norm_rows(..., return_rho=True)`` (tightens the
        exactness guard's proven bound; results are exact with or without it)."""
        self.corpus_rho = corpus_rho
        self.corpus = corpus_unit_local
        self.corpus_f32 = corpus_f32_local
        self.d = int(d)
        self.row_offset = int(row_offset)
        self.group = group
        self.local_search = local_search
        self.merge = merge
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self._comm_stream = None

    def _all_gather(self, out: torch.Tensor, src: torch.Tensor) -> None:
        """all_gather_into_tensor; with the gloo backend (CPU rehearsals of the multi-GPU path, e.g. two ranks sharing
        one GPU) device tensors are staged through host memory, everything else is identical."""
        if src.is_cuda and dist.get_backend(self.group) == "gloo":
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, src.cpu(), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out, src, group=self.group)

    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        """all-gather of equally sized query slices: [Q_local, w] -> [world*Q_local, w] (rank-major); float32 embeddings or
        unit float16 rows."""
        if self.world == 1:
            return q_local
        # rows travel as raw bytes so that the gloo test backend (no half/bf16 support) runs the same code
        src = q_local.contiguous().view(torch.uint8)
        out = torch.empty((self.world * src.shape[0], src.shape[1]), dtype=torch.uint8, device=src.device)
        self._all_gather(out, src)
        return out.view(q_local.dtype)

    # ------------------------------------------------------------------ two-stage API
    def submit(self, q_local: torch.Tensor, counts: Optional[Sequence[int]] = None) -> _Ticket:
        """Start exchange 1 for a batch.  On a GPU it runs on a side stream ordered after the work already queued on the
        current stream (the encoder that produced ``q_local``), so it overlaps whatever the caller enqueues next.
        ``counts``: the number of query rows of EVERY rank when the batch does not split evenly (known without communication,
        e.g. ``shard_bounds(Q, world, r)``); slices are padded to the longest for the exchange and the padding is dropped
        from the result."""
        if counts is not None:
            counts = [int(c) for c in counts]
            if len(counts) != self.world or counts[self.rank] != q_local.shape[0]:
                raise ValueError(f"counts {counts} do not describe {self.world} ranks with {q_local.shape[0]} local rows")
            qmax = max(counts)
            if all(c == qmax for c in counts):
                counts = None
            elif q_local.shape[0] < qmax:      # pad with zero rows: a zero query scores 0 against everything, then is dropped
                pad = torch.zeros((qmax - q_local.shape[0], q_local.shape[1]), dtype=q_local.dtype, device=q_local.device)
                q_local = torch.cat([q_local, pad], 0)
        if self.world == 1 or not q_local.is_cuda:
            return _Ticket(self.gather_queries(q_local), None, counts)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=q_local.device)
        main = torch.cuda.current_stream(q_local.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ready)
            q_all = self.gather_queries(q_local)
            done = torch.cuda.Event()
            done.record(self._comm_stream)
        q_local.record_stream(self._comm_stream)
        return _Ticket(q_all, done, counts)

    def finish(self, ticket: _Ticket, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Local search of the gathered batch, exchange 2 (one packed buffer) and the merge: (scores [Q,k], global indices
        [Q,k]) for ALL queries (rank-major order), on every rank."""
        q_all = ticket.q_all
        if ticket.event is not None:
            torch.cuda.current_stream(q_all.device).wait_event(ticket.event)
            q_all.record_stream(torch.cuda.current_stream(q_all.device))
        Q = q_all.shape[0]
        hip = self.local_search is _hip_local_search
        if self.world == 1:
            s, i = (self.local_search(q_all, self.corpus, self.corpus_f32, self.d, k, self.row_offset, self.corpus_rho) if hip
                    else self.local_search(q_all, self.corpus, self.corpus_f32, self.d, k, self.row_offset))
            return s, i
        # one buffer per rank, [scores | indices]: the search writes it, the all-gather moves it, the merge reads it in place
        nb = _packed_bytes(Q, k)
        mine = torch.empty((nb,), dtype=torch.uint8, device=q_all.device)
        views = _packed_views(mine, Q, k)
        if hip:
            self.local_search(q_all, self.corpus, self.corpus_f32, self.d, k, self.row_offset, self.corpus_rho, out=views)
        else:                                   # injected stand-ins (CPU rehearsals) return fresh tensors
            s, i = self.local_search(q_all, self.corpus, self.corpus_f32, self.d, k, self.row_offset)
            views[0].copy_(s)
            views[1].copy_(i)
        gathered = torch.empty((self.world, nb), dtype=torch.uint8, device=mine.device)
        self._all_gather(gathered.view(-1), mine)
        s_all, i_all = _packed_views(gathered, Q, k)        # [world, Q, k] views, lists nb bytes apart
        s, i = self.merge(s_all, i_all, k)
        if ticket.counts is not None:           # uneven batch: drop the padding rows (rank-major order is kept)
            qmax = Q // self.world
            keep = [slice(r * qmax, r * qmax + c) for r, c in enumerate(ticket.counts)]
            s = torch.cat([s[sl] for sl in keep], 0)
            i = torch.cat([i[sl] for sl in keep], 0)
        return s, i

    def search(self, q_local: torch.Tensor, k: int, counts: Optional[Sequence[int]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Every rank passes its slice of the query batch and gets the merged result for ALL queries."""
        return self.finish(self.submit(q_local, counts), k)

    def search_stream(self, query_batches: Iterable[torch.Tensor], k: int) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """Software pipeline over a sequence of batches: the query all-gather of batch i+1 is in flight while batch i is
        searched.  ``query_batches`` may be a generator that encodes lazily."""
        prev = None
        for q in query_batches:
            t = self.submit(q)
            if prev is not None:
                yield self.finish(prev, k)
            prev = t
        if prev is not None:
            yield self.finish(prev, k)
