"""CPU checks of the boundary: libtsim.so loads, exports every symbol include/tsim.h declares, rejects bad arguments
with the documented error codes (no kernel is launched), and the product refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from text_similarity_amd import _lib, ops, presets

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from text_similarity_amd.build import build
        build(verbose=False)
    return _lib.lib()


def test_header_symbols_are_exported_and_bound(lib):
    hdr = open(os.path.join(REPO, "include", "tsim.h")).read()
    declared = set(re.findall(r"\b(tsim_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tsim_pad_dim"} - {"tsim_pad_dim"}  # keep all
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(cdll, name), f"{name} declared in include/tsim.h but not exported"
    assert declared == set(_lib.DECLARED_SYMBOLS), declared ^ set(_lib.DECLARED_SYMBOLS)


def test_pure_host_entry_points(lib):
    assert lib.tsim_version() >= 100
    assert [lib.tsim_pad_dim(d) for d in (1, 64, 128, 129, 384, 385, 768, 769)] == [128, 128, 128, 256, 384, 512, 768, 0]
    assert lib.tsim_cosine_topk_workspace_bytes(256, 1_000_000, 10) > 0
    assert lib.tsim_cosine_topk_workspace_bytes(256, 1_000_000, 64) > 0       # k > 28: no list kernel, still served
    assert lib.tsim_cosine_topk_workspace_bytes(256, 1_000_000, 65) == 0      # k <= 64
    with pytest.raises(ValueError):
        ops.pad_dim(1000)


def test_argument_validation_returns_error_codes(lib):
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    # k out of range, empty corpus, wrong row stride, null pointers: all rejected before any launch
    assert lib.tsim_cosine_topk(p, 4, p, 100, 384, 384, 0, p, p, 0, p, 4096, None) == 1
    assert lib.tsim_cosine_topk(p, 4, p, 100, 384, 384, 65, p, p, 0, p, 4096, None) == 1
    # float32 matrices: both or neither; strides at least d
    assert lib.tsim_cosine_topk_ex(p, p, 384, 4, p, None, 384, None, 100, 384, 384, 10, p, p, None, 0, p, 4096, None) == 1
    assert lib.tsim_cosine_topk_ex(p, p, 100, 4, p, p, 384, None, 100, 384, 384, 10, p, p, None, 0, p, 4096, None) == 1
    assert lib.tsim_cosine_topk_ex(p, p, 384, 4, p, p, 384, None, 100, 384, 384, 10, p, p, None, 0, p, 4096, None) == 3   # workspace
    assert lib.tsim_cosine_topk(p, 4, p, 0, 384, 384, 10, p, p, 0, p, 4096, None) == 1
    assert lib.tsim_cosine_topk(p, 4, p, 100, 384, 400, 10, p, p, 0, p, 4096, None) == 1
    assert b"tsim_pad_dim" in lib.tsim_last_error()
    assert lib.tsim_cosine_topk(None, 4, p, 100, 384, 384, 10, p, p, 0, p, 4096, None) == 1
    assert lib.tsim_l2norm_rows(p, 7, 4, 384, 384, p, 384, 1e-8, None, None) == 1
    assert lib.tsim_mean_pool(None, 0, p, 1, 1, 1, p, None) == 1
    with pytest.raises(ValueError):
        _lib.check(1, "x")


def test_product_has_no_cpu_path():
    x = torch.zeros(4, 384)
    for call in (lambda: ops.l2norm_rows(x), lambda: ops.cos_sim_dense(x, x),
                 lambda: ops.cosine_topk(x.half(), x.half(), 384, 2),
                 lambda: ops.mean_pool(torch.zeros(1, 2, 3), torch.ones(1, 2))):
        with pytest.raises(_lib.TsimError):
            call()
    if not torch.cuda.is_available():
        from text_similarity_amd.native_encoder import NativeEncoder
        with pytest.raises(_lib.TsimError):
            NativeEncoder.from_preset("tiny-bert")


def test_reference_named_api_imports_and_containers():
    from text_similarity_amd.configurations.config import Configuration, ModelParameters, SearchConfiguration
    from text_similarity_amd.dataset.dataset import EmbeddingsFeatures
    from text_similarity_amd.models.sentence_encoder import OnnxSentenceTransformerWrapper, SentenceTransformerWrapper  # noqa
    from text_similarity_amd.modules.modules import AvgPoolingStrategy  # noqa
    from text_similarity_amd.pipeline.search_pipeline import Pipeline, SearchPipeline, SentenceMiningPipeline
    from text_similarity_amd.utils.metrics import cos_sim  # noqa
    c = Configuration(ModelParameters("m"), "m", "/tmp")
    assert (c.sequence_max_len, c.batch_size) == (256, 16)           # config.py:29,32
    s = SearchConfiguration(ModelParameters("m"), "m", "/tmp")
    assert (s.ef, s.ef_construction, s.M) == (50, 400, 64)
    f = EmbeddingsFeatures(torch.ones(1, 2), torch.ones(1, 2))
    assert set(f.to_dict()) == {"input_ids", "attention_mask"}         # dataset.py:230-240
    assert set(EmbeddingsFeatures.from_dict({**f.to_dict(), "token_type_ids": torch.zeros(1, 2)}).to_dict()) == \
        {"input_ids", "attention_mask", "token_type_ids"}
    p = SentenceMiningPipeline(100, c, model=None, corpus=["a", "b"])
    assert isinstance(p, SearchPipeline) and isinstance(p, Pipeline) and p.corpus_chunk_size == 100
    t = torch.zeros(2, 3)
    assert p.encode_corpus(t) is t                                    # tensors pass through (search_pipeline.py:19-22)


def test_synthetic_generators_are_deterministic():
    from text_similarity_amd import presets
    a = presets.uniform01("s", 5)
    np.testing.assert_array_equal(a, presets.uniform01("s", 10)[:5])
    np.testing.assert_array_equal(presets.uniform01("s", 5, offset=5), presets.uniform01("s", 10)[5:])
    x = presets.normal("n", 1000)
    assert abs(float(x.mean())) < 0.15 and 0.85 < float(x.std()) < 1.15
    e = presets.synthetic_embeddings(4, 384, "e")
    np.testing.assert_array_equal(e, presets.bf16_round(e))
    assert presets.to_bf16_bits(np.array([1.0, -2.0], np.float32)).tolist() == [0x3F80, 0xC000]
    w = presets.synthetic_weights("tiny-mpnet")
    assert "encoder.relative_attention_bias.weight" in w and "embeddings.token_type_embeddings.weight" not in w


def test_hf_directory_round_trip_and_config_validation(tmp_path):
    from text_similarity_amd import presets
    from text_similarity_amd.weights import config_from_hf, load_hf_dir, save_hf_dir
    cfg = presets.PRESETS["tiny-bert"]
    w = presets.synthetic_weights("tiny-bert")
    save_hf_dir(str(tmp_path), cfg, w)
    cfg2, w2 = load_hf_dir(str(tmp_path))
    assert cfg2 == cfg and all(np.array_equal(w[k], w2[k]) for k in w)
    with pytest.raises(ValueError):
        config_from_hf({"model_type": "gpt2"})
    with pytest.raises(ValueError):
        config_from_hf({"model_type": "bert", "hidden_act": "relu", "num_hidden_layers": 1, "hidden_size": 64,
                        "num_attention_heads": 4, "intermediate_size": 128, "vocab_size": 10, "max_position_embeddings": 8})
    with pytest.raises(FileNotFoundError):
        load_hf_dir(str(tmp_path / "nope"))


def test_position_table_guard_matches_hf_index_error():
    """HF raises IndexError when a sequence needs a position row beyond the table: BERT rows 0..len-1, MPNet
    rows pad_id+1..pad_id+len, so max_pos 514 holds 512 tokens and a 513- or 514-token sequence must be refused before
    any launch.  Out-of-range token ids are clamped on the device and reported by NativeEncoder.check() (GPU test)."""
    from text_similarity_amd.native_encoder import NativeEncoder
    mp_cfg, bert_cfg = presets.PRESETS["all-mpnet-base-v2"], presets.PRESETS["bert-base-uncased"]
    assert (mp_cfg.max_pos, mp_cfg.pad_id, bert_cfg.max_pos) == (514, 1, 512)
    NativeEncoder.check_lengths(mp_cfg, 512)
    NativeEncoder.check_lengths(bert_cfg, 512)
    for cfg, n in ((mp_cfg, 513), (mp_cfg, 514), (bert_cfg, 513)):
        with pytest.raises(ValueError):
            NativeEncoder.check_lengths(cfg, n)


def test_main_pass_plan_balances_the_xcds(lib):
    """plan_topk (csrc/k1_topk.h): workgroup b runs on XCD b % 8 and one workgroup is resident per CU, so a pass runs in rounds
    of 32 workgroups per XCD.  The chunk count must not leave some XCDs with a nearly empty extra round (round 2 shipped
    ceil(256 / query blocks) chunks at first: Q = 1 280 -> 52 chunks -> 35 workgroups on four XCDs, 1.36 ms instead of 0.77)."""
    plan = (ctypes.c_int32 * 4)()
    for Q in list(range(1, 8193, 97)) + [256, 768, 1280, 2304, 4096, 16384, 40000, 100000]:
        assert lib.tsim_cosine_topk_plan(Q, 1_000_000, 384, 10, plan) == 0
        nqb, nch, rpc, per_xcd = plan[0], plan[1], plan[2], plan[3]
        assert nqb == -(-Q // 256) and nch >= 1 and nch * rpc >= 1_000_000 and rpc % 32 == 0
        rounds = -(-per_xcd // 32)
        ideal = nqb * nch / 256.0                       # rounds if the 256 CUs could be filled exactly
        # time ~ rounds / nch; compare with the best achievable for this many query blocks (all CUs busy, no remainder)
        assert rounds / nch <= 1.35 * nqb / 256.0 + 1e-9 or rounds == 1, (Q, nqb, nch, per_xcd, rounds, ideal)
        if rounds == 1 and nqb <= 256:
            assert per_xcd * 8 >= 0.6 * 256 or nch * rpc < 2 * 256 * 32, (Q, nqb, nch, per_xcd)   # a single round is mostly full
    # small corpora: at least 256 rows per chunk
    assert lib.tsim_cosine_topk_plan(64, 1000, 128, 5, plan) == 0 and plan[1] <= 4
    assert lib.tsim_cosine_topk_plan(0, 1000, 128, 5, plan) == 1


def test_backend_tokenizer_path_gives_the_wrapper_ids():
    """encode_text drives fast tokenizers through their backend encode_batch (one call per chunk, all host cores): the ids
    must be those of the reference's tokenizer call (/root/reference/src/models/sentence_encoder.py:144-153, padding aside) —
    truncation to sequence_max_len including [CLS]/[SEP], empty strings, punctuation, accents, long inputs — and the
    backend's truncation / padding settings must be left as they were."""
    import numpy as np
    from transformers import BertTokenizer
    from text_similarity_amd import presets
    from text_similarity_amd.models.sentence_encoder import _tokenize_packed
    tok = BertTokenizer(vocab=presets.synthetic_vocab(30522), do_lower_case=True)
    sents = presets.synthetic_sentences(300, seed="tokpath", vocab_size=30522)
    sents[5] = " ".join(["w00017"] * 700)
    sents[6] = ""
    sents[7] = "Hello, WORLD! w00001-w00002 héllo 中文 w00003"
    before = (tok.backend_tokenizer.truncation, tok.backend_tokenizer.padding)
    for max_len in (256, 16, 3):
        flat, lens = _tokenize_packed(tok, sents, max_len, 64)
        ref = tok(text=sents, add_special_tokens=True, padding=False, truncation=True, max_length=max_len,
                  return_attention_mask=False, return_token_type_ids=False)["input_ids"]
        assert lens.tolist() == [len(x) for x in ref] and int(lens.max()) <= max_len
        assert flat.tolist() == [t for x in ref for t in x]
    tok.backend_tokenizer.no_truncation()
    flat, lens = _tokenize_packed(tok, sents[:10], 32, 64)
    assert tok.backend_tokenizer.truncation is None and tok.backend_tokenizer.padding == before[1]


def test_build_refuses_diagnostic_defines_for_the_product_library():
    import pytest
    from text_similarity_amd import build as b
    if b.TAG:
        pytest.skip("a variant build is selected")
    with pytest.raises(RuntimeError, match="diagnostic"):
        b.build(extra_flags=["-DTSIM_K1_NOSEL"])
    with pytest.raises(RuntimeError, match="diagnostic"):
        b.build(extra_flags=["-DTSIM_LN_DIAG=3"])
