"""Two ranks, the real HIP kernels: ShardedCorpusSearch over torch.distributed must return, on every rank, exactly the
single-GPU result over the concatenated corpus (scores and indices bit for bit).  The GPU box has ONE GPU, so both ranks
share cuda:0 and the collectives run on the gloo backend (device tensors staged through host memory — the choreography,
the packed candidate buffer, the merge and the side-stream query gather are the ones RCCL runs at N > 1).

Ranks are started from multiprocessing's FORK SERVER, which conftest.py launches before anything touches the GPU: a
process that has initialised HIP must not fork+exec children on the GPU boxes."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(n_total, q_total, d):
    rng = np.random.default_rng(2024)
    corpus = (rng.standard_normal((n_total, d)) * np.exp(rng.uniform(-1, 1, (n_total, 1)))).astype(np.float32)
    queries = rng.standard_normal((q_total, d)).astype(np.float32)
    corpus[n_total - 1] = corpus[3] * 2.0        # same direction on the LAST shard: equal cosine, tie -> row 3 first
    queries[0] = corpus[3]
    return corpus, queries


def _rank_main(rank, world, port, n_total, q_total, d, k, mode, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from text_similarity_amd import ops
        from text_similarity_amd.distributed.sharded_search import ShardedCorpusSearch, shard_bounds
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        corpus, queries = _data(n_total, q_total, d)
        lo, hi = shard_bounds(n_total, world, rank)
        cf = torch.from_numpy(corpus[lo:hi]).to(dev)
        cu, rho = ops.l2norm_rows(cf, return_rho=True)
        eng = ShardedCorpusSearch(cu, d, lo, corpus_f32_local=cf, corpus_rho=rho)
        if mode == "uneven":      # 65 queries on 2 ranks: 33 / 32, padded for the exchange and dropped from the result
            qlo, qhi = shard_bounds(q_total, world, rank)
            counts = [shard_bounds(q_total, world, r)[1] - shard_bounds(q_total, world, r)[0] for r in range(world)]
            s, i = eng.search(torch.from_numpy(queries[qlo:qhi]).to(dev), k, counts=counts)
            torch.cuda.synchronize()
            np.savez(os.path.join(out_dir, f"r{rank}.npz"), s=s.cpu().numpy(), i=i.cpu().numpy())
            return
        ql = q_total // world
        q_local = torch.from_numpy(queries[rank * ql:(rank + 1) * ql]).to(dev)
        if mode == "stream":      # pipelined form: second batch = the local slice reversed
            (s, i), (s2, i2) = list(eng.search_stream(iter([q_local, q_local.flip(0).contiguous()]), k))
            torch.cuda.synchronize()
            np.savez(os.path.join(out_dir, f"r{rank}.npz"), s=s.cpu().numpy(), i=i.cpu().numpy(), s2=s2.cpu().numpy(),
                     i2=i2.cpu().numpy())
        else:
            s, i = eng.search(q_local, k)
            torch.cuda.synchronize()
            np.savez(os.path.join(out_dir, f"r{rank}.npz"), s=s.cpu().numpy(), i=i.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "stream", "uneven"])
def test_two_ranks_with_hip_kernels_equal_one_gpu(tmp_path, mode):
    import torch
    from text_similarity_amd import ops
    world, n_total, q_total, d, k = 2, 30001, (65 if mode == "uneven" else 64), 384, 10
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("forkserver")
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n_total, q_total, d, k, mode, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    corpus, queries = _data(n_total, q_total, d)
    cf, qf = torch.from_numpy(corpus).to("cuda:0"), torch.from_numpy(queries).to("cuda:0")
    ref_s, ref_i = ops.cosine_topk(ops.l2norm_rows(qf), ops.l2norm_rows(cf), d, k, eq_f32=qf, ec_f32=cf)
    ref_s, ref_i = ref_s.cpu().numpy(), ref_i.cpu().numpy()
    assert ref_i[0, 0] == 3 and ref_i[0, 1] == n_total - 1
    ql = q_total // world
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        np.testing.assert_array_equal(got["i"], ref_i)
        np.testing.assert_array_equal(got["s"], ref_s)
        if mode == "stream":
            perm = np.concatenate([np.arange(ql * w, ql * w + ql)[::-1] for w in range(world)])
            np.testing.assert_array_equal(got["i2"], ref_i[perm])
            np.testing.assert_array_equal(got["s2"], ref_s[perm])
