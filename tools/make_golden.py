#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference in the build container.

Run from the repo root (needs /root/reference, so only here, never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 HF_HUB_OFFLINE=1 python tools/make_golden.py

What is executed from the reference:
  * ``OnnxSentenceTransformerWrapper.forward``   src/models/sentence_encoder.py:32-39
  * ``AvgPoolingStrategy.forward``               src/modules/modules.py:158-171
  * ``cos_sim``                                  src/utils/metrics.py:81-101
  * ``EmbeddingsFeatures`` / ``Configuration``    src/dataset/dataset.py:213-251, src/configurations/config.py:23-37
around the third-party arithmetic they delegate to (HF ``BertModel``/``MPNetModel`` built from local config
objects, torch ``cosine_similarity`` / ``topk``).  Four third-party packages the reference imports at module
load but never touches on this path (nltk, sentence_transformers, hnswlib, onnxruntime) are absent here; empty
module objects are registered for them so that the import statements pass.  Nothing of theirs is called.

Outputs are data only (inputs + expected outputs); weights are regenerated from
``text_similarity_amd.presets.synthetic_weights`` and pinned by a checksum.
"""
import os
import sys
import types
import zlib
from importlib.machinery import ModuleSpec
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
os.environ.setdefault("HF_HUB_OFFLINE", "1")


def _absent(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m


_absent("nltk", tag=MagicMock())
_absent("nltk.corpus", wordnet=MagicMock())
_absent("sentence_transformers")
_absent("sentence_transformers.SentenceTransformer", SentenceTransformer=MagicMock())
_absent("hnswlib", Index=MagicMock())
_absent("onnxruntime", SessionOptions=MagicMock(), InferenceSession=MagicMock())

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

os.chdir(REF)
from src.configurations.config import Configuration, ModelParameters  # noqa: E402
from src.dataset.dataset import EmbeddingsFeatures  # noqa: E402
from src.models.sentence_encoder import OnnxSentenceTransformerWrapper  # noqa: E402
from src.modules.modules import AvgPoolingStrategy  # noqa: E402
from src.utils.metrics import cos_sim  # noqa: E402
os.chdir(REPO)

from transformers import BertConfig, BertModel, BertTokenizer, MPNetConfig, MPNetModel  # noqa: E402

from text_similarity_amd import presets  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def weights_crc(w):
    c = 0
    for k in sorted(w):
        c = zlib.crc32(np.ascontiguousarray(w[k]).tobytes(), c)
    return np.uint32(c)


def hf_model(preset):
    cfg = presets.PRESETS[preset]
    w = presets.synthetic_weights(preset)
    if cfg.arch == "bert":
        hc = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers,
                        num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                        max_position_embeddings=cfg.max_pos, layer_norm_eps=cfg.ln_eps,
                        type_vocab_size=cfg.type_vocab, hidden_act="gelu")
        m = BertModel(hc, add_pooling_layer=False)
    else:
        hc = MPNetConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers,
                         num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                         max_position_embeddings=cfg.max_pos, layer_norm_eps=cfg.ln_eps,
                         relative_attention_num_buckets=cfg.rel_buckets, hidden_act="gelu")
        m = MPNetModel(hc, add_pooling_layer=False)
    sd = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    m.eval()
    return cfg, w, m


def ref_wrapper(model, preset):
    params = Configuration(model_parameters=ModelParameters(preset), model=preset, save_path="",
                           device=torch.device("cpu"))
    return params, OnnxSentenceTransformerWrapper(params=params, context_embedder=model)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrs.items()))


# ------------------------------------------------------------------ G1 tiny encoders (full tensors)
def g1_tiny():
    for preset in ("tiny-bert", "tiny-mpnet"):
        cfg, w, m = hf_model(preset)
        params, wrap = ref_wrapper(m, preset)
        B, S = 6, 12
        ids = presets.randint(preset + "/g1ids", B * S, 5, cfg.vocab).reshape(B, S)
        mask = np.ones((B, S), dtype=np.int64)
        mask[1, 7:] = 0          # ragged
        mask[2, 1:] = 0          # single valid token
        mask[3, :] = 0           # all-zero mask row -> 0 / 1e-9 clamp path
        mask[4, 3:9] = 0         # hole in the middle (not a prefix mask)
        mask[5, 10:] = 0
        if cfg.arch == "mpnet":  # padded positions carry the pad id like a real tokenizer would
            ids = np.where(mask == 1, ids, cfg.pad_id)
            ids[4, 3:9] = presets.randint(preset + "/g1hole", 6, 5, cfg.vocab)  # masked but not pad tokens
        tid, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
        hidden = m(input_ids=tid, attention_mask=tmask)[0]
        pooled = wrap.forward(tid, tmask)
        pooled2 = AvgPoolingStrategy(params).forward(hidden, EmbeddingsFeatures(tid, tmask))
        assert torch.equal(pooled, pooled2)
        save(f"encoder_{preset}.npz", input_ids=ids, attention_mask=mask,
             last_hidden_state=hidden.numpy(), pooled=pooled.numpy(), weights_crc=weights_crc(w))


# ------------------------------------------------------------------ G2 preset-shape encoders (32 pooled rows)
def g2_presets():
    for preset in ("all-MiniLM-L6-v2", "all-mpnet-base-v2", "bert-base-uncased"):
        cfg, w, m = hf_model(preset)
        params, wrap = ref_wrapper(m, preset)
        n = 32
        flat, cu = presets.synthetic_token_batch(n, seed="g2/" + preset, vocab_size=cfg.vocab, max_len=64)
        if cfg.arch == "mpnet":   # <s>=0 </s>=2 pad=1 in the MPNet vocab
            flat = flat.copy()
            flat[cu[:-1]] = 0
            flat[cu[1:] - 1] = 2
        lens = np.diff(cu)
        order = np.argsort(lens, kind="stable")
        pooled = np.zeros((n, cfg.hidden), dtype=np.float32)
        for s in range(0, n, 16):
            rows = order[s:s + 16]
            S = int(lens[rows].max())
            ids = np.full((len(rows), S), cfg.pad_id, dtype=np.int64)
            mask = np.zeros((len(rows), S), dtype=np.int64)
            for i, r in enumerate(rows):
                ids[i, :lens[r]] = flat[cu[r]:cu[r + 1]]
                mask[i, :lens[r]] = 1
            pooled[rows] = wrap.forward(torch.from_numpy(ids), torch.from_numpy(mask)).numpy()
        save(f"encoder_{preset}.npz", flat_ids=flat, cu_seqlens=cu, pooled=pooled, weights_crc=weights_crc(w))


# ------------------------------------------------------------------ G3 pooling edge cases
def g3_pool():
    params = Configuration(model_parameters=ModelParameters("p"), model="p", save_path="", device=torch.device("cpu"))
    B, S, H = 5, 9, 48
    h = presets.normal("g3/h", B * S * H).reshape(B, S, H)
    mask = np.ones((B, S), dtype=np.int64)
    mask[1, 1:] = 0
    mask[2, :] = 0
    mask[3, 4:] = 0
    mask[4, 2:5] = 0
    ids = np.zeros((B, S), dtype=np.int64)
    out = AvgPoolingStrategy(params).forward(torch.from_numpy(h), EmbeddingsFeatures(torch.from_numpy(ids), torch.from_numpy(mask)))
    save("pool_edge.npz", hidden=h, attention_mask=mask, pooled=out.numpy())


# ------------------------------------------------------------------ G4 cosine
def g4_cos():
    a = presets.normal("g4/a", 37 * 384).reshape(37, 384)
    b = presets.normal("g4/b", 101 * 384).reshape(101, 384)
    b[17] = b[3]                      # duplicate row
    cs = cos_sim(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    cs1d = cos_sim(torch.from_numpy(a[0]), torch.from_numpy(b)).numpy()
    bz = b.copy()
    bz[5] = 0.0                       # zero row: cos_sim -> NaN (no eps), cosine_similarity -> 0
    cs_zero = cos_sim(torch.from_numpy(a), torch.from_numpy(bz)).numpy()
    rows = []
    for q in range(4):
        qe = torch.from_numpy(a[q]).unsqueeze(0).expand(bz.shape[0], -1)
        rows.append(F.cosine_similarity(qe, torch.from_numpy(bz), dim=-1).numpy())
    save("search_cos.npz", a=a, b=b, cos_sim=cs, cos_sim_1d=cs1d, b_zero=bz, cos_sim_zero=cs_zero,
         cosine_similarity_rows=np.stack(rows))


# ------------------------------------------------------------------ G5 top-k
def g5_topk():
    e = presets.synthetic_embeddings(300, 384, "g5/c")
    e[40] = e[7]
    e[41] = e[7]
    e[200] = e[7]                     # duplicates -> exact score ties
    q = presets.synthetic_embeddings(9, 384, "g5/q")
    q[0] = e[7]
    sc = torch.from_numpy(q) @ torch.from_numpy(e).T
    out = {"corpus": e, "queries": q, "scores_mm": sc.numpy()}
    for k in (1, 3, 10, 300):
        v, i = torch.topk(sc, k, dim=1, largest=True, sorted=True)
        out[f"topk{k}_values"] = v.numpy()
        out[f"topk{k}_indices"] = i.numpy()
    # the reference's per-query form: expand + cosine_similarity + topk (search_pipeline.py:76-78, dim fixed)
    vals, idxs = [], []
    for qi in range(q.shape[0]):
        qe = torch.from_numpy(q[qi]).unsqueeze(0).expand(e.shape[0], -1)
        s = F.cosine_similarity(qe, torch.from_numpy(e), dim=-1)
        v, i = torch.topk(s, 10, sorted=True, largest=True)
        vals.append(v.numpy())
        idxs.append(i.numpy())
    out["loop_top10_values"] = np.stack(vals)
    out["loop_top10_indices"] = np.stack(idxs)
    save("search_topk.npz", **out)


# ------------------------------------------------------------------ G6 end-to-end config 1
def g6_e2e():
    preset = "all-MiniLM-L6-v2"
    cfg, w, m = hf_model(preset)
    params, wrap = ref_wrapper(m, preset)
    n = 1000
    sents = presets.synthetic_sentences(n, seed="sent1234", vocab_size=cfg.vocab)
    vocab = presets.synthetic_vocab(cfg.vocab)
    tok = BertTokenizer(vocab=vocab, do_lower_case=True)   # transformers 5.x: vocab dict, no files
    assert tok.cls_token_id == 101 and tok.pad_token_id == 0
    params.tokenizer = tok
    # the encode_text loop of sentence_encoder.py:136-173 (its body is what runs; the broken
    # self.encode(features, parallel_mode=False) call is replaced by the wrapper forward it intends)
    order = np.argsort([len(s) for s in sents])
    docs = [sents[i] for i in order]
    enc = []
    flat, lens = [], np.zeros(n, dtype=np.int64)
    for s in range(0, n, params.batch_size):
        batch = docs[s:s + params.batch_size]
        d = tok(text=batch, add_special_tokens=True, padding='longest', truncation=True,
                max_length=params.sequence_max_len, return_attention_mask=True,
                return_token_type_ids=False, return_tensors='pt')
        emb = wrap.forward(d["input_ids"], d["attention_mask"])
        enc.extend(emb)
        for i in range(len(batch)):
            L = int(d["attention_mask"][i].sum())
            lens[order[s + i]] = L
    enc = [enc[i] for i in np.argsort(order)]
    E = torch.stack(enc)
    # token ids in original order, packed
    cu = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=cu[1:])
    flat = np.zeros(int(cu[-1]), dtype=np.int32)
    for i, s in enumerate(sents):
        ids = tok(text=[s], add_special_tokens=True, truncation=True, max_length=params.sequence_max_len)["input_ids"][0]
        flat[cu[i]:cu[i + 1]] = ids
    # search: every sentence queries the whole corpus (search_pipeline.py:73-78 with the dim bug fixed)
    vals, idxs = [], []
    for qi in range(n):
        qe = E[qi].unsqueeze(0).expand_as(E)
        s = F.cosine_similarity(qe, E, dim=-1)
        v, i = torch.topk(s, 10, sorted=True, largest=True)
        vals.append(v.numpy())
        idxs.append(i.numpy())
    save("e2e_config1.npz", flat_ids=flat, cu_seqlens=cu.astype(np.int32), embeddings=E.numpy(),
         top10_values=np.stack(vals), top10_indices=np.stack(idxs), weights_crc=weights_crc(w),
         first_sentences=np.array(sents[:4]))


# ------------------------------------------------------------------ G7 meters (§8(f) N4)
def g7_meters():
    """The reference's own RetrievalAccuracyMeter / EmbeddingSimilarityMeter (src/utils/metrics.py:317-381, 450-507)."""
    import contextlib
    import io
    from src.utils.metrics import EmbeddingSimilarityMeter, RetrievalAccuracyMeter
    n, d = 300, 96
    src = presets.normal("g7/src", n * d).reshape(n, d).astype(np.float32)
    tgt = (src * 1.7 + 1.1 * presets.normal("g7/noise", n * d).reshape(n, d)).astype(np.float32)
    tgt[5] = tgt[200]
    tgt[17] = tgt[16]          # exact duplicate rows: argmax ties resolve to the lower index
    m = RetrievalAccuracyMeter(print_wrong_matches=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m.update(torch.from_numpy(src), torch.from_numpy(tgt), [f"s{i}" for i in range(n)], [f"t{i}" for i in range(n)])
    wrong = np.array([[int(x) for x in ln.split(",")[0].replace("i:", "").replace("j:", "").split()] for ln in m.lines], dtype=np.int64)
    k, dd = 250, 64
    a = presets.normal("g7/a", k * dd).reshape(k, dd).astype(np.float32)
    b = (a + presets.normal("g7/b", k * dd).reshape(k, dd) * np.linspace(0.1, 3, k)[:, None]).astype(np.float32)
    gold = np.linspace(5, 0, k).astype(np.float32)
    e = EmbeddingSimilarityMeter()
    e.update((a, b), gold, k)
    # the reference never stores the eight correlations it computes (locals of update(), metrics.py:364-371): recompute them
    # with the functions it calls, and keep its own outputs val / avg
    from scipy.stats import pearsonr, spearmanr
    from sklearn.metrics.pairwise import paired_cosine_distances, paired_euclidean_distances, paired_manhattan_distances
    sims = {"cosine": 1 - paired_cosine_distances(a, b), "manhattan": -paired_manhattan_distances(a, b),
            "euclidean": -paired_euclidean_distances(a, b), "dot": np.array([np.dot(x, y) for x, y in zip(a, b)])}
    corr = np.array([[pearsonr(gold, sims[kk])[0], spearmanr(gold, sims[kk])[0]] for kk in ("cosine", "manhattan", "euclidean", "dot")])
    save("meters.npz", src=src, tgt=tgt, src2tgt=np.float64(m.src2tgt), tgt2src=np.float64(m.tgt2src), avg=np.float64(m.avg),
         wrong_pairs=wrong, sts_a=a, sts_b=b, sts_gold=gold, sts_val=np.float64(e.val), sts_avg=np.float64(e.avg),
         sts_corr=corr)


# ------------------------------------------------------------------ G8 k-means (§8(f) N3)
def g8_kmeans():
    """sklearn.cluster.KMeans, the class the reference's ClusteringPipeline instantiates (src/pipeline/clustering.py:2,14),
    on RAW (un-normalised) rows.  The reference's default initialisation (k-means++ from sklearn's own RNG stream) cannot
    be reproduced outside sklearn, so the fixture fixes the initial centres (init=array, n_init=1) and pins what Lloyd's
    iterations make of them: labels, centres, inertia."""
    from sklearn.cluster import KMeans
    out = {}
    rng = np.random.default_rng(77)
    for name, k, per, d, spread, scale in (("separated", 7, 120, 64, 4.0, 1.0), ("overlap", 5, 200, 32, 1.2, 1.0),
                                           ("norms", 6, 150, 48, 2.0, 6.0)):
        centers = rng.standard_normal((k, d)).astype(np.float32) * spread
        if name == "norms":    # clusters at very different distances from the origin: cosine and Euclidean assignment disagree
            centers *= np.linspace(0.2, scale, k, dtype=np.float32)[:, None]
        x = (np.repeat(centers, per, 0) + rng.standard_normal((k * per, d))).astype(np.float32)
        x = x[rng.permutation(k * per)]
        init = x[rng.choice(k * per, k, replace=False)].copy()
        km = KMeans(n_clusters=k, init=init, n_init=1, max_iter=300, tol=0.0, algorithm="lloyd").fit(x)
        out[f"{name}_x"], out[f"{name}_init"] = x, init
        out[f"{name}_labels"], out[f"{name}_centers"] = km.labels_.astype(np.int64), km.cluster_centers_.astype(np.float32)
        out[f"{name}_inertia"], out[f"{name}_n_iter"] = np.float64(km.inertia_), np.int64(km.n_iter_)
    save("kmeans.npz", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]
    for g in which:
        {"g1": g1_tiny, "g2": g2_presets, "g3": g3_pool, "g4": g4_cos, "g5": g5_topk, "g6": g6_e2e, "g7": g7_meters,
         "g8": g8_kmeans}[g]()
