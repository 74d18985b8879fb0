"""world_size-2 gloo test of the sharded-search choreography (ShardedCorpusSearch): the two collectives and the merge
give results bit-identical to an unsharded search.  CPU only: the oracle stands in for the HIP kernels here (tests may
do that; the product defaults are the HIP ops)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search_ref
from text_similarity_amd import presets
from text_similarity_amd.distributed.sharded_search import ShardedCorpusSearch, shard_bounds


def _pad(v, i, k):
    if v.shape[1] < k:
        pad = k - v.shape[1]
        v = np.concatenate([v, np.full((v.shape[0], pad), -np.inf, np.float32)], 1)
        i = np.concatenate([i, np.full((i.shape[0], pad), -1, np.int64)], 1)
    return torch.from_numpy(v), torch.from_numpy(i)


def _oracle_local(q, c_unit, c_f32, d, k, offset):
    if c_f32 is None:      # unit rows only: inner products of the stored rows
        v, i = search_ref.cosine_topk(q[:, :d].float().numpy(), c_unit[:, :d].float().numpy(), k, idx_offset=offset)
    else:                  # float32 embeddings: the reference's cosine
        v, i = search_ref.cosine_topk_f32(q.numpy(), c_f32.numpy(), k, idx_offset=offset)
    return _pad(v, i, k)


def _oracle_merge(scores, idx, k):
    vs = [s.numpy() for s in scores]
    ix = [i.numpy() for i in idx]
    keep = [np.where(i >= 0, v, -np.inf) for v, i in zip(vs, ix)]
    ix = [np.where(i >= 0, i, np.iinfo(np.int64).max) for i in ix]
    v, i = search_ref.merge_topk(keep, ix, k)
    return torch.from_numpy(v), torch.from_numpy(i)


def _data(n_total, world, q_total=None):
    corpus = presets.synthetic_embeddings(n_total, 384, "shard/c")
    corpus[n_total - 1] = corpus[3]          # a duplicate living on the LAST shard: tie must resolve to row 3
    queries = presets.synthetic_embeddings(8 * world if q_total is None else q_total, 384, "shard/q")
    queries[0] = corpus[3]
    return corpus, queries


def _uneven_worker(rank, world, port, n_total, q_total, k, out_dir):
    """world ranks, shards of unequal size (n_total % world != 0) and a query batch that does not split evenly."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = 384
        corpus, queries = _data(n_total, world, q_total)
        lo, hi = shard_bounds(n_total, world, rank)
        counts = [shard_bounds(q_total, world, r)[1] - shard_bounds(q_total, world, r)[0] for r in range(world)]
        qlo, qhi = shard_bounds(q_total, world, rank)
        eng = ShardedCorpusSearch(torch.from_numpy(search_ref.unit_rows(corpus[lo:hi])).to(torch.float16), d, lo,
                                  local_search=_oracle_local, merge=_oracle_merge, corpus_f32_local=torch.from_numpy(corpus[lo:hi]))
        s, i = eng.search(torch.from_numpy(queries[qlo:qhi]), k, counts=counts)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), s=s.numpy(), i=i.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,q_total", [(4, 1003, 35), (4, 1003, 3), (3, 100, 7)])
def test_uneven_shards_and_query_batches(tmp_path, world, n_total, q_total):
    """4 ranks, 1 003 rows (shards of 251/251/251/250) and 35 queries (9/9/9/8; 3 queries: one rank has none): the padded
    exchange and the dropped padding leave exactly the unsharded result, in query order, on every rank."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    k = 10
    mp.spawn(_uneven_worker, args=(world, port, n_total, q_total, k, str(tmp_path)), nprocs=world, join=True)
    corpus, queries = _data(n_total, world, q_total)
    ref_s, ref_i = search_ref.cosine_topk_f32(queries, corpus, k)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        np.testing.assert_array_equal(got["i"], ref_i)
        np.testing.assert_array_equal(got["s"], ref_s)
    assert ref_i[0, 0] == 3 and ref_i[0, 1] == n_total - 1


def _worker(rank, world, port, n_total, k, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = 384
        corpus, queries = _data(n_total, world)
        lo, hi = shard_bounds(n_total, world, rank)
        c_unit = torch.from_numpy(search_ref.unit_rows(corpus[lo:hi])).to(torch.float16)
        if mode == "unit":
            q_local = torch.from_numpy(search_ref.unit_rows(queries[rank * 8:(rank + 1) * 8])).to(torch.float16)
            eng = ShardedCorpusSearch(c_unit, d, lo, local_search=_oracle_local, merge=_oracle_merge)
        else:
            # un-normalised float32 embeddings (scaled rows: the cosine must not care)
            scale = (1.0 + np.arange(n_total, dtype=np.float32) % 7)[:, None]
            q_local = torch.from_numpy(queries[rank * 8:(rank + 1) * 8] * 3.0)
            eng = ShardedCorpusSearch(c_unit, d, lo, local_search=_oracle_local, merge=_oracle_merge,
                                      corpus_f32_local=torch.from_numpy((corpus * scale)[lo:hi]))
        if mode == "f32-stream":   # the pipelined form over two batches gives the same lists
            outs = list(eng.search_stream(iter([q_local, q_local.flip(0)]), k))
            s, i = outs[0]
            s2, i2 = outs[1]
            np.savez(os.path.join(out_dir, f"r{rank}.npz"), s=s.numpy(), i=i.numpy(), s2=s2.numpy(), i2=i2.numpy())
        else:
            s, i = eng.search(q_local, k)
            np.savez(os.path.join(out_dir, f"r{rank}.npz"), s=s.numpy(), i=i.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,k,mode", [(1001, 10, "unit"), (15, 10, "unit"), (1001, 10, "f32"), (15, 10, "f32"),
                                            (500, 5, "f32-stream")])
def test_two_rank_sharded_search_equals_unsharded(tmp_path, n_total, k, mode):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, n_total, k, mode, str(tmp_path)), nprocs=world, join=True)
    corpus, queries = _data(n_total, world)
    if mode == "unit":
        ref_s, ref_i = search_ref.cosine_topk(search_ref.unit_rows(queries), search_ref.unit_rows(corpus), k)
    else:
        scale = (1.0 + np.arange(n_total, dtype=np.float32) % 7)[:, None]
        ref_s, ref_i = search_ref.cosine_topk_f32(queries * 3.0, corpus * scale, k)
    ref_s, ref_i = (t.numpy() for t in _pad(ref_s, ref_i, k))
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        np.testing.assert_array_equal(got["i"], ref_i)
        np.testing.assert_array_equal(got["s"], ref_s)
        if mode == "f32-stream":   # second batch: every rank's slice reversed
            perm = np.concatenate([np.arange(8 * w, 8 * w + 8)[::-1] for w in range(world)])
            np.testing.assert_array_equal(got["i2"], ref_i[perm])
            np.testing.assert_array_equal(got["s2"], ref_s[perm])
    assert ref_i[0, 0] == 3 and ref_i[0, 1] == n_total - 1


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
