/*
 * tsim.h — C ABI of libtsim.so, the MI355X (gfx950) embed-and-search engine.
 *
 * The reference (cr1m5onk1ng/text_similarity) has no FFI of its own: its boundary for this path is a
 * Python API whose arithmetic lives in torch / transformers.  Each entry point below replaces one of
 * those call sites; the Python classes in text_similarity_amd/ keep the reference's names and
 * signatures and reach these functions through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the caller owns all buffers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); calls only enqueue
 *     work on it, they never synchronise and never allocate device memory (except tsim_encoder_create);
 *   - return value 0 = ok, otherwise a TSIM_E* code; tsim_last_error() gives the message for the calling
 *     thread;
 *   - "unit rows" are row-major IEEE half (float16) L2-normalised rows with a row stride of `ld` elements, ld a multiple
 *     of 8 and every row 16-byte aligned; tsim_pad_dim(d) is the stride the engine itself produces.  Half, not bf16:
 *     the elements of a unit row are <= 1, the f16 MFMA runs at the bf16 rate, and 11 significand bits keep the MFMA
 *     selection scores within ~1e-4 of the exact cosine (bf16: ~1e-3), which keeps the exactness guard quiet.
 */
#ifndef TSIM_H
#define TSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSIM_OK 0
#define TSIM_EINVAL 1      /* bad argument (shape, alignment, unsupported size) */
#define TSIM_EHIP 2        /* a HIP runtime call failed */
#define TSIM_ENOMEM 3      /* workspace too small / allocation failed */
#define TSIM_EUNSUPPORTED 4

#define TSIM_F32 0
#define TSIM_BF16 1

#define TSIM_ARCH_BERT 0
#define TSIM_ARCH_MPNET 1

int tsim_version(void);
const char *tsim_last_error(void);

/* Row stride (elements) of the engine's internal unit-row matrices for an embedding width d:
 * the smallest supported kernel width >= d (128, 256, 384, 512 or 768); 0 if d > 768. */
int tsim_pad_dim(int d);

/* ---------------------------------------------------------------------------------------------
 * A7  F.cosine_similarity operand preparation   /root/reference/src/pipeline/search_pipeline.py:77
 * out[r, :d] = half( x[r, :] / max(||x[r, :]||_2, eps) ), out[r, d:ld_out] = 0, evaluated canonically: float64 sum of
 * squares in a fixed order, float64 reciprocal, one rounding float64 -> half (oracle/search_ref.unit_rows is bit-identical).
 * torch divides each operand by max(norm, eps) with eps = 1e-8; a zero row stays zero, so its score
 * against anything is 0.  x_dtype is TSIM_F32 or TSIM_BF16, ld_in its row stride in elements.
 * rho_max (device float, may be NULL): atomically raised to the largest rounding residual of the rows written,
 * rho_r = || out[r] - x[r] / max(||x[r]||, eps) ||_2 (rounded up).  The caller zeroes the word once and may let several calls
 * accumulate into it (a corpus that grows); tsim_cosine_topk_ex turns it into a PROVEN bound on |MFMA score - exact score|. */
int tsim_l2norm_rows(const void *x, int x_dtype, int64_t rows, int d, int64_t ld_in,
                     void *out_f16, int ld_out, float eps, float *rho_max, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A6/A7/A9  the per-query loop `expand_as -> F.cosine_similarity -> torch.topk`
 *           /root/reference/src/pipeline/search_pipeline.py:73-78, fused.
 * eq_unit [Q, ld] and ec_unit [N, ld] are L2-normalised half rows (tsim_l2norm_rows).  MFMA inner products of those rows
 * (scores live in registers, survivors go through per-lane queues in LDS; the N x Q matrix is never written) SELECT
 * candidates; every score that is returned, and the final order, comes from an exact re-score:
 *   - eq_f32 / ec_f32 given (float32 embeddings, row strides ldq_f32 / ldc_f32 elements): the reference's value
 *     x.y / (max(|x|, 1e-8) * max(|y|, 1e-8)) of the float32 rows, evaluated in float64 in a fixed order and rounded once
 *     to float32 (oracle/search_ref.exact_cosine) — torch's own float32 evaluation differs from it by rounding only;
 *   - eq_f32 == ec_f32 == NULL: the inner product of the unit rows as stored (oracle/search_ref.canonical_scores).
 * Results are ordered by (score descending, index ascending) — the tie rule torch.topk leaves undefined.
 * Exactness guard (a proof, not an estimate).  A stored half row is u^ = u + delta with u the exact unit row and
 * rho = |delta|_2; for any two rows |u^q.u^c - u_q.u_c| <= rho_q + rho_c + rho_q rho_c (Cauchy-Schwarz), u_q.u_c is the
 * reference's cosine, and the MFMA's float32 accumulation adds at most ld 2^-23 (1+rho_q)(1+rho_c).  eps_q = that bound with
 * rho_q measured from the query's own two rows and rho_c = *ec_rho_max, the largest residual of the shard's unit rows as
 * reported by tsim_l2norm_rows / tsim_encoder_forward (NULL: the a-priori bound 2^-11 + sqrt(ld) 2^-25 of a correctly rounded
 * unit row, about twice as loose).  Every row that was NOT re-scored has an MFMA score <= cut, hence an exact score
 * <= cut + eps_q: if that is below the k-th exact score the list stands (status 0).  Otherwise EVERY row whose MFMA score
 * exceeds (k-th exact score - eps_q) is collected and re-scored (status 1); if more than 1 024 rows qualify, or the bound is
 * seen to fail on a re-scored row (unit rows that are not the images of the float32 rows), the whole shard is scored exactly
 * for that query (status 2).  With unit rows only (no float32 matrices) the two scores differ by float32 accumulation alone
 * and eps = max(4 x the largest difference seen, ld 2^-23).  out_status [Q] int32 (may be NULL) reports the pass per query.
 * out_scores [Q, k] float32, out_idx [Q, k] int64 = shard row index + idx_offset (-1 / -inf when the shard has fewer
 * than k rows).  1 <= k <= 64 (k > 28 skips the list kernel: block maxima -> collect -> re-score).
 * workspace: tsim_cosine_topk_workspace_bytes(Q, N, k).
 * tsim_cosine_topk(...) == tsim_cosine_topk_ex with NULL float32 matrices and NULL status. */
size_t tsim_cosine_topk_workspace_bytes(int64_t Q, int64_t N, int k);
/* Introspection (host only, no launch): how the main pass of a search of Q queries against N rows of padded width ld is cut.
 * plan[0] = query blocks, plan[1] = corpus chunks, plan[2] = rows per chunk, plan[3] = the largest number of workgroups any one
 * XCD receives (workgroup b runs on XCD b % 8; a round is 32 of them).  Returns TSIM_OK or TSIM_EINVAL. */
int tsim_cosine_topk_plan(int64_t Q, int64_t N, int ld, int k, int32_t plan[4]);
int tsim_cosine_topk_ex(const void *eq_unit, const float *eq_f32, int64_t ldq_f32, int64_t Q,
                        const void *ec_unit, const float *ec_f32, int64_t ldc_f32, const float *ec_rho_max, int64_t N,
                        int d, int ld, int k, float *out_scores, int64_t *out_idx, int32_t *out_status,
                        int64_t idx_offset, void *workspace, size_t workspace_bytes, void *stream);
int tsim_cosine_topk(const void *eq_unit, int64_t Q, const void *ec_unit, int64_t N, int d, int ld,
                     int k, float *out_scores, int64_t *out_idx, int64_t idx_offset,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Measurement hook (bench.py): the NEXT tsim_cosine_topk call of the calling thread records `start` right before
 * and `stop` right after the launch of its dominant kernel (cos_topk_partial) on the call's stream.  Both are
 * hipEvent_t handles passed as void*; the hook is cleared by that call.  Pass NULLs to cancel. */
void tsim_time_next_topk(void *start_event, void *stop_event);

/* Merge `nlists` sorted candidate lists per query (the per-shard results of tsim_cosine_topk on the
 * shards of a partitioned corpus, or the per-chunk results of search_pipeline.py:60 `corpus_chunk_size`
 * chunking): scores/idx are [nlists, Q, k_in]; output [Q, k_out] by (score desc, index asc);
 * entries with idx < 0 are ignored. */
int tsim_topk_merge(const float *scores, const int64_t *idx, int nlists, int64_t Q, int k_in,
                    int k_out, float *out_scores, int64_t *out_idx, void *stream);
/* The same with the lists `list_stride_scores` / `list_stride_idx` ELEMENTS apart (>= Q * k_in): merges the per-rank
 * [scores | indices] buffers of an all-gather in place, without re-stacking them (distributed/sharded_search.py). */
int tsim_topk_merge_strided(const float *scores, const int64_t *idx, int nlists, int64_t Q, int k_in, int k_out,
                            int64_t list_stride_scores, int64_t list_stride_idx, float *out_scores,
                            int64_t *out_idx, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A8  cos_sim(a, b)   /root/reference/src/utils/metrics.py:81-101
 * out[i, j] = <a_i/||a_i||, b_j/||b_j||> in float32, dense [Na, Nb]; no eps (a zero row gives NaN,
 * as in the reference).  For evaluation-sized inputs; the search path never materialises this. */
int tsim_cos_sim(const float *a, int64_t na, const float *b, int64_t nb, int d, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A4  AvgPoolingStrategy.forward   /root/reference/src/modules/modules.py:158-171
 *     (== OnnxSentenceTransformerWrapper.forward, src/models/sentence_encoder.py:35-38)
 * out[b, :] = sum_s hidden[b, s, :] * mask[b, s] / max(sum_s mask[b, s], 1e-9); hidden is float32 or bf16
 * [B, S, H] contiguous, mask int32 [B, S]; out float32 [B, H]. */
int tsim_mean_pool(const void *hidden, int hidden_dtype, const int32_t *mask, int64_t B, int S, int H,
                   float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A3  context_embedder(**features)[0]  — the HF AutoModel forward the reference calls at
 *     /root/reference/src/models/sentence_encoder.py:33,107-108,118 (layer arithmetic:
 *     src/models/bert_of_theseus.py:185-211, 244-336, 346-350, 411-414, 424-428), followed by A4.
 */
typedef struct tsim_encoder tsim_encoder;

/* Projection operand formats.  TSIM_W_MXFP8: weights AND the activations entering each projection are OCP MXFP8
 * (e4m3 elements, one E8M0 power-of-two scale per 32 consecutive elements of the contraction axis), multiplied by
 * v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 MFMA rate); needs hidden % 256 == 0 and ffn % 256 == 0.
 * The reference has no fp8 path (HF fp32 forward, src/models/sentence_encoder.py:33): this is north_star's config 5. */
#define TSIM_W_BF16 0
#define TSIM_W_MXFP8 1

/* bf16 [rows, K] (device) -> MXFP8: q_out uint8 [rows, K] e4m3 bytes, scale_out uint8 [rows, K/32] E8M0 bytes.
 * K % 32 == 0.  Bit-exact restatement: oracle/fp8_ref.mx_quantize. */
int tsim_quantize_mxfp8(const void *x_bf16, int64_t rows, int K, void *q_out, void *scale_out, void *stream);

/* out_f32 [M, N] = dequant(xq, xs) [M, K] @ dequant(wq, ws) [N, K]^T + bias  on the block-scaled fp8 MFMA (the projection
 * kernel of the MXFP8 encoder, exposed for parity tests).  N % 256 == 0, K % 128 == 0, K >= 256; xq, xs and out_f32 must be
 * allocated for M rounded up to a multiple of 256 rows (the kernel works on whole 256-row tiles). */
int tsim_gemm_mxfp8(const void *xq, const void *xs, const void *wq, const void *ws, const float *bias, float *out_f32,
                    int M, int N, int K, void *stream);

typedef struct tsim_encoder_config {
    int32_t arch;          /* TSIM_ARCH_BERT | TSIM_ARCH_MPNET */
    int32_t num_layers, hidden, heads, ffn, vocab, max_pos;
    int32_t pad_id;        /* MPNet: position ids skip tokens equal to pad_id */
    int32_t rel_buckets;   /* MPNet relative-position buckets (32) */
    float ln_eps;
    int32_t max_tokens;    /* capacity of the activation workspace, in packed tokens per call */
    int32_t max_seqs;      /* capacity in sequences per call */
    int32_t weight_dtype;  /* TSIM_W_BF16 | TSIM_W_MXFP8 (projection operands; BASELINE.json configs[4]) */
} tsim_encoder_config;

/* Per-layer weights, all HOST pointers to float32 in torch nn.Linear layout [out, in]; the engine
 * converts to bf16 and uploads.  rel_bias_host is [rel_buckets, heads] or NULL for BERT. */
typedef struct tsim_layer_weights_host {
    const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *ln1_g, *ln1_b;
    const float *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
} tsim_layer_weights_host;

typedef struct tsim_encoder_weights_host {
    const float *word_emb, *pos_emb, *type_emb /* row 0 is added; NULL for MPNet */, *emb_ln_g, *emb_ln_b;
    const float *rel_bias;
    const tsim_layer_weights_host *layers;
} tsim_encoder_weights_host;

int tsim_encoder_create(const tsim_encoder_config *cfg, const tsim_encoder_weights_host *w,
                        tsim_encoder **out);
void tsim_encoder_destroy(tsim_encoder *enc);

/* Forward on PACKED tokens (no padding work): token t of sequence b lives at cu_seqlens[b] <= t <
 * cu_seqlens[b+1]; tok_ids/tok_pos int32 [T] (tok_pos = position-embedding row, tok_col = column of the
 * token in the padded batch, used for MPNet's relative bias; pass tok_col = NULL to use tok_pos).
 * max_len = the longest sequence of the batch (sizes the attention grid); max_len (+ pad_id + 1 for MPNet, whose position
 * rows start there) must not exceed max_pos, else TSIM_EINVAL.
 * Outputs (either may be NULL): pooled_f32 [B, hidden] = masked mean-pool (A4), un-normalised like the
 * reference's encode_text; unit_f16 [B, ld_unit] = L2-normalised half rows ready for tsim_cosine_topk, with
 * unit_rho_max (device float, may be NULL) raised to their largest rounding residual exactly as tsim_l2norm_rows does;
 * last_hidden_bf16 [T, hidden] for tests. */
int tsim_encoder_forward(tsim_encoder *enc, const int32_t *tok_ids, const int32_t *tok_pos,
                         const int32_t *tok_col, const int32_t *cu_seqlens, int32_t T, int32_t B,
                         int32_t max_len, float *pooled_f32, void *unit_f16, int ld_unit, float *unit_rho_max,
                         void *last_hidden_bf16, void *stream);

/* Kernels cannot raise HF's IndexError: a token id outside [0, vocab), a position row outside [0, max_pos) or a token whose
 * column is >= the max_len passed to tsim_encoder_forward is clamped / computed anyway and leaves a bit in a per-encoder
 * flag word.  This call copies the word to *flags_host, clears it and SYNCHRONISES `stream` (the only entry point that
 * does): 0 = every forward since the last call was clean.  The Python wrappers call it at the end of encode_text. */
#define TSIM_ENC_ERR_TOKEN_ID 1
#define TSIM_ENC_ERR_POSITION 2
#define TSIM_ENC_ERR_MAX_LEN 4
int tsim_encoder_error_flags(tsim_encoder *enc, int32_t *flags_host, void *stream);

/* ---- tokenizer (host code, no GPU): BERT WordPiece for pure-ASCII sentences -------------------------------------------------
 * Replaces, for the sentences it handles, the host tokenizer call of the reference's encode_text
 * (/root/reference/src/models/sentence_encoder.py:144-153: tokenizer(text=batch, padding=True, truncation=True,
 * max_length=...)) with the same ids: BertNormalizer (clean_text, lowercase) -> BertPreTokenizer -> WordPiece -> prefix /
 * suffix special ids -> truncation on the right to max_len, as the `tokenizers` library the reference depends on does it.
 * The vocabulary is one blob of UTF-8 keys + vocab_size + 1 byte offsets; the id of a key is its position.  `added_*`: the
 * contents of the library's added / special tokens: a sentence containing one of them, or any non-ASCII byte, is NOT
 * handled.  Thread-safe after creation. */
int tsim_wordpiece_create(const char *vocab_text, const int64_t *vocab_offsets, int32_t vocab_size,
                          const char *continuing_prefix, int32_t unk_id, const int32_t *prefix_ids, int32_t n_prefix,
                          const int32_t *suffix_ids, int32_t n_suffix, int32_t lowercase, int32_t max_input_chars_per_word,
                          const char *added_text, const int64_t *added_offsets, int32_t n_added, void **handle);
void tsim_wordpiece_destroy(void *handle);
/* n sentences (one blob + n + 1 byte offsets) on n_threads host threads (<= 0: all cores).  out_ids receives the ids of the
 * handled sentences back to back (sentence order); out_capacity (in ids) must be at least sum_i min(max_len, bytes_i +
 * specials), e.g. total bytes + n * specials, else TSIM_ENOMEM.  out_lens[i] = ids of sentence i (0 when not handled),
 * handled[i] = 1 / 0: the caller tokenises the others with the library itself. */
int tsim_wordpiece_encode(void *handle, const char *text, const int64_t *text_offsets, int64_t n, int32_t max_len,
                          int32_t n_threads, int32_t *out_ids, int64_t out_capacity, int32_t *out_lens, uint8_t *handled);

#ifdef __cplusplus
}
#endif
#endif /* TSIM_H */
