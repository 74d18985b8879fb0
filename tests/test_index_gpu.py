"""§8(f) N1: exact GPU index with the hnswlib-shaped surface + SemanticSearchPipeline, against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import search_ref
from text_similarity_amd import presets
from text_similarity_amd.index import GpuFlatIndex

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle(q, live_rows, live_labels, k):
    s, i = search_ref.cosine_topk_f32(q, live_rows, k)      # the reference's cosine of the float32 rows
    return live_labels[i], s


def test_add_query_delete_persist(tmp_path):
    d = 384
    x = presets.normal("idx/x", 3000 * d).reshape(3000, d)
    q = presets.normal("idx/q", 20 * d).reshape(20, d)
    x[2500] = x[17]                                     # duplicate: tie -> earlier row first
    labels = np.arange(1000, 4000, dtype=np.int64)      # labels are not row numbers
    idx = GpuFlatIndex(space="cosine", dim=d, device=DEV)
    idx.init_index(max_elements=100)
    idx.add_items(x[:2000], labels[:2000])
    idx.add_items(torch.from_numpy(x[2000:]), labels[2000:])     # growth past the reserved capacity
    assert idx.get_current_count() == 3000
    lab, sc = idx.search(q, 10)
    rl, rs = _oracle(q, x, labels, 10)
    np.testing.assert_array_equal(lab.cpu().numpy(), rl)
    np.testing.assert_array_equal(sc.cpu().numpy(), rs)
    hl, hd = idx.knn_query(q[:3], k=5)                  # hnswlib convention: distances = 1 - cosine
    np.testing.assert_array_equal(hl, rl[:3, :5])
    np.testing.assert_allclose(hd, 1.0 - rs[:3, :5], rtol=0, atol=1e-7)
    # delete the best hit of every query, plus some labels that do not exist
    victims = set(int(v) for v in rl[:, 0])
    for v in victims:
        idx.mark_deleted(v)
    with pytest.raises(RuntimeError):
        idx.mark_deleted(999999)
    with pytest.raises(RuntimeError):
        idx.mark_deleted(next(iter(victims)))           # already gone
    assert idx.num_live() == 3000 - len(victims)
    keep = ~np.isin(labels, list(victims))
    lab2, sc2 = idx.search(q, 10)
    rl2, rs2 = _oracle(q, x[keep], labels[keep], 10)
    np.testing.assert_array_equal(lab2.cpu().numpy(), rl2)
    np.testing.assert_array_equal(sc2.cpu().numpy(), rs2)
    assert not (np.isin(lab2.cpu().numpy(), list(victims))).any()
    # persist / reload: identical answers
    idx.save_index(str(tmp_path))
    assert os.path.exists(tmp_path / "index.bin")
    idx2 = GpuFlatIndex(space="cosine", dim=0, device=DEV)
    idx2.load_index(str(tmp_path))
    lab3, sc3 = idx2.search(q, 10)
    assert torch.equal(lab3, lab2) and torch.equal(sc3, sc2)
    # k = 50 (the reference's bound is ef = 50, search_pipeline.py:131)
    lab50, sc50 = idx2.search(q[:4], 50)
    rl50, rs50 = _oracle(q[:4], x[keep], labels[keep], 50)
    np.testing.assert_array_equal(lab50.cpu().numpy(), rl50)
    np.testing.assert_array_equal(sc50.cpu().numpy(), rs50)
    # fewer live rows than k
    small = GpuFlatIndex(dim=d, device=DEV)
    small.add_items(x[:4])
    l4, s4 = small.search(q[:2], 6)
    assert (l4[:, 4:] == -1).all() and torch.isinf(s4[:, 4:]).all() and (l4[:, :4] >= 0).all()


def test_semantic_search_pipeline_surface(tmp_path):
    from transformers import BertTokenizer
    from text_similarity_amd.configurations.config import ModelParameters, SearchConfiguration
    from text_similarity_amd.models.sentence_encoder import SentenceTransformerWrapper
    from text_similarity_amd.pipeline.search_pipeline import SemanticSearchPipeline
    preset = "all-MiniLM-L6-v2"
    tok = BertTokenizer(vocab=presets.synthetic_vocab(30522), do_lower_case=True)
    params = SearchConfiguration(model_parameters=ModelParameters(preset, hidden_size=384), model=preset, save_path="",
                                 tokenizer=tok, device=torch.device(DEV), max_tokens_per_batch=8192, max_seqs_per_batch=512)
    model = SentenceTransformerWrapper.from_preset(preset, params, parallel_mode=False)
    sents = presets.synthetic_sentences(300, seed="sem/s", vocab_size=30522)
    path = str(tmp_path / "index")
    pipe = SemanticSearchPipeline(path, params, model, corpus=list(sents[:200]))
    assert os.path.exists(os.path.join(path, "index.bin")) and pipe.num_indexed() == 200
    res = pipe(sents[:5], 3)
    assert all(res[q][0] == sents[q] and len(res[q]) == 3 for q in range(5))       # a sentence finds itself first
    pipe.add_to_index(sents[200:300])
    assert pipe.num_indexed() == 300 and pipe(sents[250:251], 1)[0] == [sents[250]]
    pipe.remove_from_index([250, 123456])                                            # unknown ids are skipped
    assert pipe.num_indexed() == 299 and pipe(sents[250:251], 1)[0] != [sents[250]]
    # a second pipeline on the same directory loads the saved index (the first 200 sentences)
    pipe2 = SemanticSearchPipeline(path, params, model, corpus=list(sents[:200]))
    assert pipe2.num_indexed() == 200 and pipe2(sents[7:8], 1)[0] == [sents[7]]


def test_api_search_pipeline_surface(tmp_path):
    """search_pipeline.py:178-226 (the ONNX-runtime serving variant): same constructor and call shape on the native encoder."""
    from transformers import BertTokenizer
    from text_similarity_amd.configurations.config import ModelParameters, SearchConfiguration
    from text_similarity_amd.models.sentence_encoder import SentenceTransformerWrapper
    from text_similarity_amd.pipeline.search_pipeline import APISearchPipeline
    preset = "all-MiniLM-L6-v2"
    tok = BertTokenizer(vocab=presets.synthetic_vocab(30522), do_lower_case=True)
    params = SearchConfiguration(model_parameters=ModelParameters(preset, hidden_size=384), model=preset, save_path="",
                                 tokenizer=tok, device=torch.device(DEV), max_tokens_per_batch=8192, max_seqs_per_batch=512)
    model = SentenceTransformerWrapper.from_preset(preset, params, parallel_mode=False)
    sents = presets.synthetic_sentences(120, seed="api/s", vocab_size=30522)
    for args in ((str(tmp_path / "a"), params, model), (str(tmp_path / "b"), model)):       # reference style / short style
        pipe = APISearchPipeline(params, 3, *args, corpus=list(sents[:100]))
        res = pipe(sents[:4])                       # max_n_results from the constructor
        assert all(res[q][0] == sents[q] and len(res[q]) == 3 for q in range(4))
        assert len(pipe(sents[5:6], 1)[0]) == 1 and pipe.session is model
        emb = pipe.encode_corpus(list(sents[:7]))
        assert emb.shape == (7, 384) and torch.equal(emb, model.encode_text(list(sents[:7])))
