#!/usr/bin/env python3
"""bench.py — headline benchmark of the embed-and-search hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "sentences/sec encoded + Mpairs/sec cosine top-k @ N=1M d=384".

Workload (config.workload): all-MiniLM-L6-v2 architecture preset (synthetic weights), a corpus of 1 M x 384
float32 embeddings plus their L2-normalised IEEE-half (f16) unit rows RESIDENT IN HBM PER GPU (SURVEY.md §8(d) synthetic
embeddings), and per step one batch of Q synthetic sentences (pre-tokenised, resident in HBM) that is encoded (bf16 MFMA
encoder -> masked mean-pool -> float32 embeddings) and searched against the corpus: f16 MFMA cosine over the unit rows
selects candidates, which are re-scored exactly from the float32 rows — scores and (score desc, index asc) order are the
reference's F.cosine_similarity + topk of the float32 embeddings.
One step = one pass of the whole hot path over one query batch.  With N GPUs the corpus is sharded — `--scaling weak`
(default): 1 M rows per GPU; `--scaling strong`: `--total-rows` (8 M) split N ways — each rank encodes its slice of the
batch, query rows are all-gathered over RCCL, every rank searches all Q queries against its shard and the per-shard top-10
lists are all-gathered and merged.  `python bench.py --gpus N` without a launcher starts its N ranks itself
(torch.distributed.run as a child process, before this process touches a GPU) and relays rank 0's JSON line.

`value` = scored (query, corpus-row) pairs per second over the whole job, in Mpairs/s, from the wall time of the K
timed steps (encode INCLUDED).  `sentences_per_s` is the same time base.  Per-phase rates from HIP events are reported
under `phases`.  `roofline` is for the dominant kernel, cos_topk_partial: algorithmic FLOPs 2*Q*N_local*d per launch
(768 FLOP/pair) over its HIP-event duration on its own stream, against the dense bf16 MFMA peak; the HBM streaming
figure (ceil(Q/256) * N * d * 2 B per launch) is reported beside it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, Chip-level parameters)
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def measured_traffic(Q, n_local, d, k):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC passes (profiles/rNN_k1_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this bench, gfx950 x2 read correction).  Counters cannot be
    read live from Python, so the number is REPLAYED from that file, and only when this run's workload is the profiled
    one; returns (bytes or None, source file or None)."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        names = sorted(n for n in os.listdir(prof) if n.endswith("_k1_traffic.json"))
    except OSError:
        return None, None
    for name in reversed(names):
        try:
            with open(os.path.join(prof, name)) as f:
                t = json.load(f)
            w = t["workload"]
            if (w["queries_per_step"], w["corpus_rows_per_gpu"], w["d"], w["k"]) == (Q, n_local, d, k):
                return t["hbm_bytes_per_launch"], "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def encoder_traffic(preset, q_local):
    """HBM bytes per encoder layer (FETCH x 2 + WRITE over the layer's kernels) replayed from the newest committed PMC passes
    (profiles/rNN_encoder_traffic.json), when they were taken on this workload; else None."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        names = sorted(n for n in os.listdir(prof) if n.endswith("_encoder_traffic.json"))
    except OSError:
        return None
    for name in reversed(names):
        try:
            with open(os.path.join(prof, name)) as f:
                t = json.load(f)
            if t["workload"]["preset"] == preset and t["workload"]["sentences_per_step"] == q_local:
                return {"bytes": t["hbm_bytes_per_layer"], "source": "profiles/" + name}
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline(preset: str, d: int, n_rows: int, k: int):
    """The oracle (CPU port of the reference path) on this box's host cores, bounded to ~20-30 s:
    encode 2 048 synthetic sentences with the fp32 oracle encoder; cosine top-k of a query sample against n_rows x d fp32
    rows in BOTH forms — the reference's per-query loop (expand_as + F.cosine_similarity + topk,
    /root/reference/src/pipeline/search_pipeline.py:73-78) and its fastest CPU restatement (one fp32 GEMM over unit rows
    + topk).  `value` is the faster of the two."""
    import torch.nn.functional as F
    from oracle import encoder_ref
    from text_similarity_amd import presets
    # the box gives one GPU job a 16-core share; os.cpu_count() reports the whole host and oversubscribes torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    cfg = presets.PRESETS[preset]
    w = presets.synthetic_weights(preset)
    wt = {kk: torch.from_numpy(v) for kk, v in w.items()}
    n_sent = 2048
    flat, cu = presets.synthetic_token_batch(n_sent, seed="sent1234", vocab_size=cfg.vocab, max_len=256)
    encoder_ref.encode_packed(cfg, wt, flat[:int(cu[64])], cu[:65], batch_size=16)      # warm-up
    t0 = time.perf_counter()
    encoder_ref.encode_packed(cfg, wt, flat, cu, batch_size=16)
    t_enc = time.perf_counter() - t0
    g = torch.Generator().manual_seed(4321)
    corpus = torch.randn((n_rows, d), generator=g)
    q = torch.randn((1024, d), generator=g)
    # (a) the reference's loop form, on a sample of its 1 024 queries
    nl = 0
    t0 = time.perf_counter()
    while nl < 64 and (nl < 4 or time.perf_counter() - t0 < 8.0):
        sc = F.cosine_similarity(q[nl].unsqueeze(0).expand_as(corpus), corpus, dim=-1)
        torch.topk(sc, k, sorted=False, largest=True)
        nl += 1
    t_loop = (time.perf_counter() - t0) / nl
    # (b) GEMM form: unit rows once, then blocks of 128 queries (a [128, N] fp32 score block = 512 MB at N = 1 M)
    cn = F.normalize(corpus, dim=1)
    qn = F.normalize(q, dim=1)
    (qn[:8] @ cn.T).topk(k, dim=1)
    t0 = time.perf_counter()
    done = 0
    while done < 1024 and (done < 128 or time.perf_counter() - t0 < 8.0):
        (qn[done:done + 128] @ cn.T).topk(k, dim=1)
        done += 128
    t_gemm = (time.perf_counter() - t0) / done
    best = min(t_loop, t_gemm)
    return {"value": round(n_rows / best / 1e6, 1), "unit": "Mpairs/s", "cores": threads, "kind": "port",
            "sample": f"oracle on host cores, fp32: cosine top-{k} vs {n_rows} x {d} rows — reference loop form "
                      f"(expand_as + F.cosine_similarity + topk per query) {nl} queries, {t_loop * 1e3:.0f} ms each; GEMM form "
                      f"(unit rows @ + topk, blocks of 128) {done} of 1024 queries, {t_gemm * 1e3:.1f} ms per query; "
                      f"encode {n_sent} synthetic sentences with the fp32 oracle encoder in {t_enc:.1f} s",
            "loop_form_mpairs_per_s": round(n_rows / t_loop / 1e6, 1),
            "gemm_form_mpairs_per_s": round(n_rows / t_gemm / 1e6, 1),
            "encode_sentences_per_s": round(n_sent / t_enc, 1)}


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a CHILD torch.distributed.run before this
    process initialises a GPU (an exec from a GPU-initialised process is forbidden on the pool; counting devices is not an
    initialisation).  Fewer GPUs than ranks (the one-GPU development box): the gloo rehearsal, ranks share GPUs."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < args.gpus:
        env["TSIM_BENCH_BACKEND"] = "gloo"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--queries", type=int, default=4096, help="sentences per step over the whole job")
    ap.add_argument("--corpus-rows", type=int, default=1_000_000, help="corpus rows per GPU")
    ap.add_argument("--preset", default="all-MiniLM-L6-v2")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --corpus-rows per GPU; strong: --total-rows split over the GPUs")
    ap.add_argument("--total-rows", type=int, default=8_000_000, help="corpus rows over the whole job (--scaling strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of 4 result rows after the timed region")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    backend = os.environ.get("TSIM_BENCH_BACKEND", "nccl")   # "gloo": rehearsal with ranks sharing GPUs
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from text_similarity_amd import _lib, ops, presets
    from text_similarity_amd.distributed.sharded_search import ShardedCorpusSearch, shard_bounds
    from text_similarity_amd.native_encoder import NativeEncoder

    cfg = presets.PRESETS[args.preset]
    d, k = cfg.hidden, args.k
    Q = args.queries
    q_counts = [shard_bounds(Q, world, r)[1] - shard_bounds(Q, world, r)[0] for r in range(world)]   # uneven batches are fine
    q_lo = shard_bounds(Q, world, rank)[0]
    q_local = q_counts[rank]
    if args.scaling == "strong":
        n_total = args.total_rows
        row_lo, row_hi = shard_bounds(n_total, world, rank)
    else:
        n_total = args.corpus_rows * world
        row_lo, row_hi = rank * args.corpus_rows, (rank + 1) * args.corpus_rows
    n_local = row_hi - row_lo

    # ---- resident inputs (untimed): corpus shard, query-sentence batches, encoder weights
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    corpus_f32 = torch.randn((n_local, d), generator=g, device=dev)      # the embeddings (what the reference scores)
    corpus, corpus_rho = ops.l2norm_rows(corpus_f32, return_rho=True)    # their f16 unit rows (what the MFMA kernel streams)
    nb = 8  # distinct query batches, cycled
    flat_h, cu_h = presets.synthetic_token_batch(Q * nb, seed="sent1234", vocab_size=cfg.vocab, max_len=256)
    batches = []
    for b in range(nb):
        lo = b * Q + q_lo
        t0, t1 = int(cu_h[lo]), int(cu_h[lo + q_local])
        batches.append((torch.from_numpy(flat_h[t0:t1]).to(dev),
                        torch.from_numpy((cu_h[lo:lo + q_local + 1] - cu_h[lo]).astype(np.int32)).to(dev)))
    max_tok = max(int(b[0].numel()) for b in batches)
    enc = NativeEncoder.from_preset(args.preset, max_tokens=max(max_tok, 1), max_seqs=max(q_local, 1), device=dev)
    pos = [enc.positions(f, c) for f, c in batches]
    max_len = [int((c[1:] - c[:-1]).max().item()) if q_local else 0 for _, c in batches]
    engine = ShardedCorpusSearch(corpus, d, row_lo, corpus_f32_local=corpus_f32, corpus_rho=corpus_rho)
    mean_tokens = float(cu_h[-1]) / (Q * nb)

    L = _lib.lib()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream(dev)

    def step(i, rec=None):
        f, c = batches[i % nb]
        p, cols = pos[i % nb]
        if rec is not None:
            rec["e0"].record(stream)
        emb = enc.forward_packed(f, c, p, cols, max_len[i % nb], pooled=True, unit=False)["pooled"]
        if rec is not None:
            rec["e1"].record(stream)
            L.tsim_time_next_topk(rec["k0"].cuda_event, rec["k1"].cuda_event)
        s, idx = engine.search(emb, k, counts=q_counts)
        if rec is not None:
            rec["e2"].record(stream)
        return s, idx, emb

    for i in range(args.warmup):
        step(i)
    recs = [{n: ev() for n in ("e0", "e1", "e2", "k0", "k1")} for _ in range(args.steps)]
    for r in recs:      # hipEventCreate happens lazily at first record: do it outside the timed region
        for e in r.values():
            e.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i, recs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity of the last result (not timed): sorted lists, indices inside the global corpus
    s_last, i_last, emb_last = out
    assert s_last.shape == (Q, k) and bool((s_last[:, :-1] >= s_last[:, 1:]).all())
    assert int(i_last.min()) >= 0 and int(i_last.max()) < n_total
    # ... and four of its rows against the oracle's restatement of the reference search (F.cosine_similarity of the float32
    # embeddings + topk), bit for bit, at the full bench shape (single GPU: the oracle needs the whole corpus on the host)
    verified = None
    if world == 1 and rank == 0 and not args.no_verify and n_local <= 2_000_000:   # (the oracle scans the corpus on the host)
        from oracle import search_ref
        sel = [0, Q // 3, (2 * Q) // 3, Q - 1]
        rs, ri = search_ref.cosine_topk_f32(emb_last[sel].cpu().numpy(), corpus_f32.cpu().numpy(), k)
        ok = bool(np.array_equal(i_last[sel].cpu().numpy(), ri) and np.array_equal(s_last[sel].cpu().numpy(), rs))
        assert ok, "bench result differs from oracle/search_ref.cosine_topk_f32"
        verified = {"queries": sel, "vs": "oracle/search_ref.cosine_topk_f32", "indices_and_scores_bit_identical": ok}

    # the batched-query regime north_star's HBM target is defined on (Q_b <= 256): ONE query block, so
    # the main pass streams the shard exactly once — a real launch, not an as-if figure (untimed, after the K steps)
    q256_ms = q256_call_ms = None
    if rank == 0:
        q256 = torch.randn((256, d), generator=g, device=dev)
        u256 = ops.l2norm_rows(q256)
        e0, e1, c0, c1 = ev(), ev(), ev(), ev()
        for e in (e0, e1, c0, c1):
            e.record(stream)      # hipEventCreate happens at the first record
        ts, tc = [], []
        for _ in range(8):
            L.tsim_time_next_topk(e0.cuda_event, e1.cuda_event)
            c0.record(stream)
            ops.cosine_topk(u256, corpus, d, k, eq_f32=q256, ec_f32=corpus_f32, rho_c=corpus_rho)
            c1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
            tc.append(c0.elapsed_time(c1))
        q256_ms = float(np.mean(ts[2:]))
        q256_call_ms = float(np.mean(tc[2:]))      # the WHOLE call: threshold pre-pass, main pass, exact re-score, guard passes

    # the same encoder fed from STRINGS through the reference-named API (host tokenizer + H2D included,
    # tokenizer one chunk ahead on a host thread) — untimed extra, rank 0 only
    from_strings = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            from transformers import BertTokenizer
            from text_similarity_amd.configurations.config import Configuration, ModelParameters
            from text_similarity_amd.models.sentence_encoder import OnnxSentenceTransformerWrapper
            tok = BertTokenizer(vocab=presets.synthetic_vocab(cfg.vocab), do_lower_case=True)
            params = Configuration(model_parameters=ModelParameters(args.preset, hidden_size=d), model=args.preset, save_path="",
                                   tokenizer=tok, device=dev, batch_size=16, max_tokens_per_batch=max(max_tok, 1),
                                   max_seqs_per_batch=max(q_local, 1))
            wrap = OnnxSentenceTransformerWrapper(params=params, context_embedder=enc)
            sents = presets.synthetic_sentences(16384, seed="sent1234", vocab_size=cfg.vocab)
            wrap.encode_text(sents[:2048])
            wrap.encode_text(sents)
            st = wrap.last_encode_stats
            from_strings = {"sentences_per_s": round(st["sentences"] / st["wall_s"], 1), "sentences": st["sentences"],
                            "tokenizer_s": round(st["tokenizer_s"], 3), "wall_s": round(st["wall_s"], 3),
                            "note": "encode_text(List[str]): host tokenizer (one chunk ahead on a thread) + H2D + encode + D2D un-sort"}
        except Exception as exc:      # the measurement is an extra: never fail the bench line for it
            from_strings = {"error": repr(exc)}

    enc_ms = float(np.mean([r["e0"].elapsed_time(r["e1"]) for r in recs]))
    srch_ms = float(np.mean([r["e1"].elapsed_time(r["e2"]) for r in recs]))
    k1_ms = float(np.mean([r["k0"].elapsed_time(r["k1"]) for r in recs]))
    dev_counts = [torch.cuda.device_count()]
    if world > 1:
        t = torch.zeros(world, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        t[rank] = dev_counts[0]
        dist.all_reduce(t)
        dev_counts = [int(v) for v in t.tolist()]
    if rank == 0:
        pairs = float(Q) * n_total * args.steps
        k1_flops = 2.0 * Q * n_local * d                      # per launch (this rank's shard)
        k1_stream_bytes = -(-Q // 256) * n_local * d * 2.0      # corpus streamed once per 256-query block
        H, F, Ly = cfg.hidden, cfg.ffn, cfg.num_layers
        lens = np.diff(cu_h).astype(np.float64)
        sbar = float((lens ** 2).sum() / lens.sum())
        enc_flops = q_local * mean_tokens * Ly * (2 * (4 * H * H + 2 * H * F) + 4 * sbar * H)
        traffic, traffic_src = measured_traffic(Q, n_local, d, k)
        res = {
            "metric": "sentences/sec encoded + Mpairs/sec cosine top-k @ N=1M d=384",
            "value": round(pairs / elapsed / 1e6, 1), "unit": "Mpairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "bf16 encoder / f16 selection / fp32 scores", "data": "synthetic",
            "config": {"workload": f"{args.preset} preset (synthetic weights): encode {Q} synthetic sentences/step "
                                   f"(mean {mean_tokens:.1f} tokens, bf16 MFMA) + cosine top-{k} vs {n_local} x {d} corpus rows per GPU "
                                   f"resident in HBM as float32 embeddings + f16 unit rows (BASELINE configs[1] path at the metric's "
                                   f"N=1M, d=384" + (f"; strong scaling: {n_total} rows over {world} GPUs" if args.scaling == "strong" else "") + ")",
                       "queries_per_step": Q, "corpus_rows_per_gpu": n_local, "corpus_rows_total": n_total, "d": d, "k": k,
                       "scores": "reference cosine of the float32 embeddings (exact re-score of MFMA-selected candidates)",
                       "encoder_dtype": "bf16", "search_operand_dtype": "f16", "score_dtype": "fp32",
                       "parallelism": (f"corpus-sharded x{world}, queries all-gathered "
                                       f"({'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()})")
                       if world > 1 else "single GPU"},
            "distributed": {"world_size": dist.get_world_size() if world > 1 else 1,
                            "backend": dist.get_backend() if world > 1 else None,
                            "visible_devices_per_rank": dev_counts},
            "verified": verified,
            "sentences_per_s": round(Q * args.steps / elapsed, 1),
            "encode_text_from_strings": from_strings,
            "phases": {"encode_ms": round(enc_ms, 4), "search_ms": round(srch_ms, 4),
                       "encode_sentences_per_s_per_gpu": round(q_local / enc_ms * 1e3, 1),
                       "encode_tflops_per_gpu": round(enc_flops / enc_ms / 1e9, 1),
                       "search_mpairs_per_s_per_gpu": round(Q * n_local / srch_ms / 1e3, 1)},
            "roofline": {"bound": "mfma", "kernel": "cos_topk_partial_kernel<384,8,1,16> (main pass: phase A + phase B launches)",
                         "achieved": round(k1_flops / k1_ms / 1e9, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(k1_flops / k1_ms / 1e9 / PEAK_BF16_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": round(k1_ms, 4), "flops_per_launch": k1_flops,
                         # AS-IF figure of SURVEY.md §8(d) convention (B): ceil(Q/256) corpus streams per launch; the
                         # query blocks of a chunk share it through L2, real HBM traffic is `traffic`
                         "hbm_stream_asif_GBs": round(k1_stream_bytes / k1_ms / 1e6, 1),
                         "hbm_stream_asif_frac": round(k1_stream_bytes / k1_ms / 1e6 / PEAK_HBM_GBS, 4), "query_block": 256,
                         # a REAL one-query-block launch (Q = 256): the shard is streamed once, n_local*d*2 bytes
                         "q256_main_pass_ms": round(q256_ms, 4),
                         "hbm_frac_q256": round(n_local * d * 2.0 / q256_ms / 1e6 / PEAK_HBM_GBS, 4),
                         # the WHOLE Q = 256 call (pre-pass + main pass + exact re-score + guard launches): config 4's batches
                         "q256_call_ms": round(q256_call_ms, 4),
                         "hbm_frac_q256_call": round(n_local * d * 2.0 / q256_call_ms / 1e6 / PEAK_HBM_GBS, 4)},
            # the encoder (MFMA-bound): FLOPs of valid tokens only, L (2 (4 H^2 + 2 H F) + 4 S H) per token
            "roofline_encoder": {"bound": "mfma", "achieved": round(enc_flops / enc_ms / 1e9, 1), "peak": PEAK_BF16_TFLOPS,
                                 "unit": "TFLOP/s", "frac": round(enc_flops / enc_ms / 1e9 / PEAK_BF16_TFLOPS, 4),
                                 "encode_ms": round(enc_ms, 4), "flops_per_step": enc_flops,
                                 "hbm_bytes_per_layer": encoder_traffic(args.preset, q_local)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.preset, d, n_local, k)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
