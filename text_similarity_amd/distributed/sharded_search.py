"""Corpus-sharded search over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm).  The reference has no multi-device code at all; this is the scale-out of
``SentenceMiningPipeline._search`` (/root/reference/src/pipeline/search_pipeline.py:60-89) for BASELINE.json config 4.

Partitioning: rank r owns corpus rows [offset_r, offset_r + n_r) as unit bf16 rows resident in its HBM; the
corpus never moves.  Per query batch there are exactly two exchanges, both tiny and latency-bound:
  1. all-gather of the query unit rows   [Q_local, ld] bf16 per rank  -> [Q, ld] everywhere
  2. all-gather of per-shard candidates  [Q, k] (score f32, global index i64) per rank
followed by a k-way merge with the global tie rule (score desc, index asc), so the result is bit-identical to a
single-GPU search over the concatenated corpus (tests/test_sharded_cpu.py, world_size 2 over gloo).

``local_search`` / ``merge`` are injectable so that the collective choreography can be exercised on CPU ranks
(gloo) with the oracle standing in for the kernels — in tests only; the defaults are the HIP ops and raise without
a GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def _hip_local_search(q_unit, c_unit, d, k, offset):
    from .. import ops
    return ops.cosine_topk(q_unit, c_unit, d, k, idx_offset=offset)


def _hip_merge(scores: List[torch.Tensor], idx: List[torch.Tensor], k: int):
    from .. import ops
    return ops.topk_merge(scores, idx, k)


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Block partition of n_total rows: rank r owns [lo, hi)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedCorpusSearch:
    def __init__(self, corpus_unit_local: torch.Tensor, d: int, row_offset: int,
                 group: Optional[dist.ProcessGroup] = None,
                 local_search: Callable = _hip_local_search, merge: Callable = _hip_merge):
        self.corpus = corpus_unit_local
        self.d = int(d)
        self.row_offset = int(row_offset)
        self.group = group
        self.local_search = local_search
        self.merge = merge
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0

    def _all_gather(self, out: torch.Tensor, src: torch.Tensor) -> None:
        """all_gather_into_tensor; with the gloo backend (CPU rehearsals of the multi-GPU path, e.g. two ranks sharing
        one GPU) device tensors are staged through host memory, everything else is identical."""
        if src.is_cuda and dist.get_backend(self.group) == "gloo":
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, src.cpu(), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out, src, group=self.group)

    def gather_queries(self, q_unit_local: torch.Tensor) -> torch.Tensor:
        """all-gather of equally sized query slices: [Q_local, ld] -> [world*Q_local, ld] (rank-major)."""
        if self.world == 1:
            return q_unit_local
        # bf16 travels as raw bytes so that the gloo test backend (no bf16/int16 support) runs the same code
        src = q_unit_local.contiguous().view(torch.uint8)
        out = torch.empty((self.world * src.shape[0], src.shape[1]), dtype=torch.uint8, device=src.device)
        self._all_gather(out, src)
        return out.view(torch.bfloat16)

    def search(self, q_unit_local: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Every rank passes its slice of the query batch and gets (scores [Q,k], global indices [Q,k]) for ALL
        queries (rank-major order)."""
        q_all = self.gather_queries(q_unit_local)
        s, i = self.local_search(q_all, self.corpus, self.d, k, self.row_offset)
        if self.world == 1:
            return s, i
        Q = s.shape[0]
        s_cat = torch.empty((self.world * Q, k), dtype=s.dtype, device=s.device)
        i_cat = torch.empty((self.world * Q, k), dtype=i.dtype, device=i.device)
        self._all_gather(s_cat, s.contiguous())
        self._all_gather(i_cat, i.contiguous())
        s_all, i_all = s_cat.view(self.world, Q, k), i_cat.view(self.world, Q, k)
        return self.merge([s_all[r] for r in range(self.world)], [i_all[r] for r in range(self.world)], k)
