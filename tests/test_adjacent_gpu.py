"""consumers of the fused cosine top-k — retrieval/STS meters and the clustering / ranking
pipelines — checked against plain numpy restatements of the reference's host code."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import search_ref
from text_similarity_amd import presets
from text_similarity_amd.pipeline.clustering import ClusteringPipeline
from text_similarity_amd.pipeline.ranking_pipeline import RankingPipeline
from text_similarity_amd.utils.metrics import EmbeddingSimilarityMeter, RetrievalAccuracyMeter

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_retrieval_accuracy_meter_matches_dense_argmax():
    n, d = 700, 384
    src = presets.normal("n4/src", n * d).reshape(n, d)
    tgt = (src + 0.9 * presets.normal("n4/noise", n * d).reshape(n, d)).astype(np.float32)
    tgt[5] = tgt[400]                                   # some wrong matches
    m = RetrievalAccuracyMeter(print_wrong_matches=True)
    m.update(torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV), [f"s{i}" for i in range(n)], [f"t{i}" for i in range(n)])
    # metrics.py:466-500 restated: dense cosine matrix (canonical scores of the same unit rows), argmax per row / column
    sims = search_ref.canonical_scores(search_ref.unit_rows(src), search_ref.unit_rows(tgt))
    s2t = float((sims.argmax(1) == np.arange(n)).mean())
    t2s = float((sims.T.argmax(1) == np.arange(n)).mean())
    assert m.src2tgt == pytest.approx(s2t, abs=1e-12) and m.tgt2src == pytest.approx(t2s, abs=1e-12)
    assert m.avg == pytest.approx((s2t + t2s) / 2) and 0.5 < m.avg < 1.0
    assert len(m.lines) == int(round((1 - s2t) * n)) and "INCORRECT" in str(m)


def test_embedding_similarity_meter_matches_scipy():
    from scipy.stats import pearsonr, spearmanr
    n, d = 200, 64
    a = presets.normal("n4/a", n * d).reshape(n, d)
    b = (a + presets.normal("n4/b", n * d).reshape(n, d) * np.linspace(0.1, 3, n)[:, None]).astype(np.float32)
    gold = np.linspace(5, 0, n)
    m = EmbeddingSimilarityMeter()
    m.update((a, b), gold, n)
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert m.eval_pearson_cosine == pytest.approx(pearsonr(gold, cos)[0], abs=1e-5)
    assert m.eval_spearman_euclidean == pytest.approx(spearmanr(gold, -np.linalg.norm(a - b, axis=1))[0], abs=1e-5)
    assert m.val == pytest.approx(max(m.eval_spearman_cosine, m.eval_spearman_manhattan, m.eval_spearman_euclidean,
                                      m.eval_spearman_dot)) and m.avg == m.val


def test_clustering_pipeline_recovers_separated_clusters():
    k, per, d = 7, 300, 384
    centers = presets.normal("n3/c", k * d).reshape(k, d) * 4
    x = (np.repeat(centers, per, 0) + presets.normal("n3/x", k * per * d).reshape(k * per, d)).astype(np.float32)
    truth = np.repeat(np.arange(k), per)
    perm = np.random.default_rng(0).permutation(k * per)
    pipe = ClusteringPipeline(k, SimpleNamespace(device=DEV), None)
    out = pipe(torch.from_numpy(x[perm]).to(DEV), k)
    assert sorted(len(v) for v in out.values()) == [per] * k
    lab = pipe.labels_.cpu().numpy()
    for c in range(k):                                   # every found cluster is one true cluster
        assert len(set(truth[perm][lab == c])) == 1
    # assignment step == oracle top-1 against the final centres
    _, ref = search_ref.cosine_topk(search_ref.unit_rows(x[perm]), search_ref.unit_rows(pipe.cluster_centers_.cpu().numpy()), 1)
    np.testing.assert_array_equal(lab, ref[:, 0])


class _FakeModel:
    """Stands in for the sentence encoder: text 'w<i>' -> row i of a fixed embedding table."""

    def __init__(self, table):
        self.table = torch.from_numpy(table).to(DEV)

    def encode_text(self, documents, output_np=False):
        return self.table[[int(t[1:]) for t in documents]]


class _FakeCross:
    def predict(self, pairs):
        return [float(int(t[1:]) % 7) for _, t in pairs]     # an arbitrary re-ranking signal


def test_ranking_pipeline_retrieve_then_rerank():
    n, d = 500, 384
    table = presets.normal("n3/rank", n * d).reshape(n, d)
    corpus = [f"w{i}" for i in range(100, 400)]
    queries = ["w3", "w250", "w77"]
    pipe = RankingPipeline(_FakeCross(), 128, SimpleNamespace(device=DEV), _FakeModel(table))
    out = pipe(queries, corpus, top_k=6)
    sc, ix = search_ref.cosine_topk_f32(table[[3, 250, 77]], table[100:400], 6)      # the reference's float32 cosine
    for qi, res in enumerate(out):
        assert sorted(r["corpus_id"] for r in res["results"]) == sorted(ix[qi].tolist())
        cs = [r["cross-score"] for r in res["results"]]
        assert cs == sorted(cs, reverse=True) and res["avg_score"] == pytest.approx(sum(cs) / 6)
        by_id = {r["corpus_id"]: r["score"] for r in res["results"]}
        for r, i in enumerate(ix[qi]):
            assert by_id[int(i)] == sc[qi][r]
    assert out[1]["results"][0]["text"] in corpus
