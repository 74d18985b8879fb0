// Transformer encoder forward for gfx950 (MI355X) on PACKED tokens: BERT and MPNet families.
//
// Replaces `context_embedder(**features)[0]` + mean-pool of the reference
// (/root/reference/src/models/sentence_encoder.py:33-38; layer arithmetic as stated in
// /root/reference/src/models/bert_of_theseus.py:185-211 embeddings, :244-336 attention,
// :346-350 / :411-414 / :424-428 projections + residual LayerNorm + GELU).
//
// Layout in HBM (per encoder handle): activations are [tokens, features] bf16 row-major over the packed
// token axis (no padding rows between sequences; the token axis is rounded up to 128 rows of slack):
//   x0, x1 [Tp, H]   residual stream (LayerNorm outputs)      qkv [Tp, 3H]   fused Q|K|V projections
//   ctx    [Tp, H]   attention output                         h1  [Tp, F]    GELU(FFN1)
// Weights bf16 in nn.Linear layout [out, in] (K contiguous for both GEMM operands); embedding tables,
// biases and LayerNorm parameters float32.
#include <math.h>
#include <cmath>

#include <stdlib.h>
#include <string.h>

#include <vector>

#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_pp.h"

namespace tsim {

// =====================================================================================================
// BLOCK-PACKED activation layout (round 3) for the wide intermediates of the hidden-384 path, qkv [T, 3H] and h1 [T, F]:
//     element (token t, feature f)  ->  ((t >> 5) * (W / 8) + (f >> 3)) * 256 + (t & 31) * 8 + (f & 7)      (W = row width)
// i.e. per block of 32 tokens and group of 8 features one contiguous 512-byte run of [token][8 features].  Why: a wave of the
// projection kernels owns 32 tokens x 32 features and stores 16 bytes per lane; in the row-major layout one store instruction
// touches 32 rows x 32 B, and such a store costs its wave ~400 cycles of issue (tools/microbench/mfma_loop: two of them per
// step 2 598 cycles against 1 780 without and 1 913 with two CONTIGUOUS 1-KiB stores; in-kernel stamps 650-750 per step).  In
// the packed layout the same instruction writes 1 KiB of contiguous memory.  The readers gain too: attention's q / k / v loads
// (16 B per lane of 32 consecutive tokens: two runs instead of 32 pieces) and the LDS-DMA pieces of the FFN2 operand (256-byte
// runs instead of 64-byte ones).  A block of 32 tokens occupies 32 * W elements in both layouts, so block-aligned row offsets
// are the same.  Token counts are padded to 128 (+128) rows by the encoder's buffers.
// =====================================================================================================
__device__ __forceinline__ int64_t packed_off(int64_t t, int f, int W) {   // element offset of (t, f), f a multiple of 8 here
    return ((t >> 5) * (W >> 3) + (f >> 3)) * 256 + (t & 31) * 8 + (f & 7);
}

// =====================================================================================================
// embeddings: x[t] = LayerNorm(word[id] + pos[pos_id] (+ type[0]))        one wave per token, fp32 math
// HBM-bound: reads 2-3 rows of H floats, writes H bf16.
// =====================================================================================================
template <int VPL>  // values per lane = H / 64
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t *__restrict__ ids,
                                                       const int32_t *__restrict__ pos,
                                                       const float *__restrict__ word,
                                                       const float *__restrict__ pos_emb,
                                                       const float *__restrict__ type0,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float eps, int T, int H,
                                                       bf16_t *__restrict__ out, int vocab, int max_pos,
                                                       const int32_t *__restrict__ col, int max_len,
                                                       int *__restrict__ err_flags) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    // HF raises IndexError on an id / position outside its table; a kernel cannot, so the index is clamped (no read
    // outside the tables) and an error bit is left for tsim_encoder_error_flags.  A token whose column is >= the max_len
    // the caller promised would be skipped by the attention grid: flagged as well.
    int id = ids[t], ps = pos[t];
    int bad = 0;
    if (id < 0 || id >= vocab) { bad |= TSIM_ENC_ERR_TOKEN_ID; id = id < 0 ? 0 : vocab - 1; }
    if (ps < 0 || ps >= max_pos) { bad |= TSIM_ENC_ERR_POSITION; ps = ps < 0 ? 0 : max_pos - 1; }
    if (col[t] >= max_len || col[t] < 0) bad |= TSIM_ENC_ERR_MAX_LEN;
    if (bad && lane == 0) atomicOr(err_flags, bad);
    const float *w = word + (int64_t)id * H;
    const float *p = pos_emb + (int64_t)ps * H;
    // all row loads first, the sums after them: with `if (type0)` inside the loop every iteration was a branch with its own
    // s_waitcnt vmcnt(0) — six dependent memory round trips per token (50 us per forward for a kernel that moves 155 MB)
    float v[VPL], pv[VPL], tv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = w[lane + i * 64];
        pv[i] = p[lane + i * 64];
    }
    if (type0) {   // wave-uniform
#pragma unroll
        for (int i = 0; i < VPL; ++i) tv[i] = type0[lane + i * 64];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i];
        if (type0) a += tv[i];      // HF order: (word + token_type) + position
        a += pv[i];
        v[i] = a;
        s += a;
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float dlt = v[i] - mean;
        q = fmaf(dlt, dlt, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int j = lane + i * 64;
        out[(int64_t)t * H + j] = f32_to_bf16((v[i] - mean) * rstd * gamma[j] + beta[j]);
    }
}

// =====================================================================================================
// GEMM  out[m, n] = epilogue( sum_k X[m,k] * W[n,k] + bias[n] )          bf16 in, fp32 accumulate (MFMA)
//
// MFMA orientation: A = W tile (rows = output features), B = X tile (columns = tokens), so a lane owns
// ONE token (column lane&31) and 16 features per 32x32 tile.  Row-wise epilogues (LayerNorm statistics,
// packed stores of 4 consecutive features) then need almost no cross-lane traffic.
// Staging: both operands are K-contiguous; tiles stream HBM/L2 -> LDS with global_load_lds_dwordx4 into
// a 2-stage ring (one barrier per K-step: wait -> barrier -> issue next -> compute).  LDS image: rows of
// BK*2 bytes packed into 256-byte super-rows, 16-byte slot index XORed with (super-row & 15) so that the
// ds_read_b128 of a fragment (32 lanes = 32 rows, same K chunk) is bank-conflict free; the permutation is
// applied on the SOURCE address because LDS-DMA writes lane-linearly.
// Epilogues: BIAS | BIAS+GELU(erf) | BIAS+RESIDUAL+LAYERNORM (requires BN == N: a workgroup owns whole rows).
// Roofline: MFMA-bound when K is large; at K = 384 (MiniLM) 288 FLOP per byte moved, i.e. at the
// HBM/MFMA balance point, so operand reuse through L2 (XCD-aware tile order) matters.
// =====================================================================================================
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RES_LN = 2 };
#ifndef TSIM_LN_EPI_DIRECT
#define TSIM_LN_EPI_DIRECT 1   // LayerNorm epilogue: residual / result as 16-byte accesses per lane (0: both tiles through LDS, the first form)
#endif

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int NST = 2>
constexpr int gemm_lds_bytes() {
    return NST * (BM + BN) * BK * 2;
}

// NST = ring slots.  2: wait vmcnt(0) -> barrier -> issue next -> compute (prefetch distance one k-tile).  > 2: counted
// vmcnt, prefetch distance NST-1 k-tiles (needs the stage's 1-KiB pieces to split evenly over the waves).
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int EPI, int NST = 2>
__global__ __launch_bounds__(WAVES_M *WAVES_N * 64) void gemm_bf16_kernel(
    const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const float *__restrict__ bias,
    const bf16_t *__restrict__ res, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
    bf16_t *__restrict__ out, int M, int N, int K, int mtiles, int ntiles, const bf16_t *__restrict__ Wimg, int xpacked) {
    // xpacked: X is in the block-packed layout (packed_off; BM is a multiple of 32, so a tile starts on a block)
    // Wimg (optional, BN == N): W re-laid at load time as the LDS images of its k-tiles (pack_gemm_w_kernel), so that every
    // DMA piece of the W operand is 1 KiB of CONTIGUOUS memory.  From row-major W a piece gathers 8 rows x 128 B, and that
    // shape streams from L2 at half the rate (tools/microbench/dma_stream: 60 vs 115-128 GB/s per CU); W is 3/4 of the bytes
    // this kernel stages at BM = 128, BN = 384.
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N, MT = TM / 32, NT = TN / 32;
    constexpr int RB = BK * 2;       // bytes per tile row
    constexpr int CPR = RB / 16;     // 16-byte chunks per row
    constexpr int RPS = 256 / RB;    // rows per 256-byte super-row
    constexpr int X_BYTES = BM * RB, W_BYTES = BN * RB, STAGE = X_BYTES + W_BYTES;
    constexpr int PIECES = STAGE / 1024, XPIECES = X_BYTES / 1024;
    constexpr int KSTEPS = BK / 16;
    static_assert(TM % 32 == 0 && TN % 32 == 0 && X_BYTES % 1024 == 0 && W_BYTES % 1024 == 0, "tile shape");
    static_assert(NST == 2 || PIECES % NW == 0, "a deep ring needs equal DMA counts per wave");
    constexpr int PPW = (PIECES + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int r = lane & 31, h = lane >> 5;

    // tile order: the ntiles feature tiles of one token tile share blockIdx%8, i.e. one XCD's L2 (speed only)
    const int b = blockIdx.x, xcd = b & 7, jj = b >> 3;
    const int nt_id = jj % ntiles;
    const int mt_id = (jj / ntiles) * 8 + xcd;
    if (mt_id >= mtiles) return;
    const int m0 = mt_id * BM, n0 = nt_id * BN;

    // ---- LDS-DMA sources: per piece handled by this wave, the byte offset of this lane's 16 B inside the operand
    // (computed once: per k-tile only kt * BK * 2 is added)
    constexpr int PPW_MAX = (PIECES + NW - 1) / NW;
    int src_off[PPW_MAX];
#pragma unroll
    for (int i = 0; i < PPW_MAX; ++i) {
        const int p = wave + i * NW;
        const bool isx = p < XPIECES;
        const int sl = (isx ? p : p - XPIECES) * 64 + lane;
        const int sr = sl >> 4, chp = sl & 15;
        const int ch = chp ^ (sr & 15);
        const int row = sr * RPS + ch / CPR, c = ch % CPR;
        src_off[i] = (isx && xpacked) ? (row >> 5) * K * 64 + (row & 31) * 16 + c * 512 : row * K * 2 + c * 16;
    }
    const int xkstride = xpacked ? (BK / 8) * 512 : BK * 2;   // bytes per k-tile along a row of X
    const char *xbase = reinterpret_cast<const char *>(X + (int64_t)m0 * K);
    const char *wbase = reinterpret_cast<const char *>(W + (int64_t)n0 * K);
    const char *wimg = reinterpret_cast<const char *>(Wimg);
    // source of piece p (this wave's i-th) for k-tile kt
    auto piece_src = [&](int i, int p, int kt) __attribute__((always_inline)) -> const char * {
        if (p < XPIECES) return xbase + src_off[i] + kt * xkstride;
        if (wimg) return wimg + (int64_t)kt * W_BYTES + (p - XPIECES) * 1024 + lane * 16;
        return wbase + src_off[i] + kt * (BK * 2);
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW_MAX; ++i) {
            const int p = wave + i * NW;
            if (p < PIECES) glds16(piece_src(i, p, kt), smem + stage * STAGE + p * 1024);
        }
    };
    // fragment address inside a region for tile row `row`, k-step s, lane half h
    auto frag_off = [&](int row, int s) {
        const int sr = row / RPS;
        const int ch = (row % RPS) * CPR + 2 * s + h;
        return sr * 256 + ((ch ^ (sr & 15)) << 4);
    };

    // accumulators START from the bias (acc[i][j][g]: feature n0 + wn*TN + i*32 + (g&3) + 8*(g>>2) + 4*h): no bias pass in the
    // epilogue, and the same arithmetic as ln_rows_gemm_kernel below, whose rows must carry the same bits
    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + n0 + wn * TN + 4 * h + i * 32 + 8 * gq);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                acc[i][j][4 * gq + 0] = bv.x;
                acc[i][j][4 * gq + 1] = bv.y;
                acc[i][j][4 * gq + 2] = bv.z;
                acc[i][j][4 * gq + 3] = bv.w;
            }
        }
#if defined(TSIM_LN_DIAG) && TSIM_LN_DIAG == 3
    const int nk = 1;   // TIMING-ONLY: prologue + epilogue alone
#else
    const int nk = K / BK;
#endif
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));   // retire the bias loads before any LDS-DMA is in flight
    // (Reading the residual tile here, under the first tiles' LDS-DMA latency, instead of in the epilogue measured +0.15 ms per
    // forward: reverted.)
    if constexpr (NST == 2) {
        issue(0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) issue(i < nk ? i : nk - 1, i);   // past-the-end: re-read (uniform vmcnt)
    }
    for (int kt = 0; kt < nk; ++kt) {
        const char *xs = smem + (kt % NST) * STAGE;
        const char *ws = xs + X_BYTES;
        auto load_frags = [&](int s, bf16x8 (&bx)[MT], bf16x8 (&aw)[NT]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
                bx[j] = *reinterpret_cast<const bf16x8 *>(xs + frag_off(wm * TM + j * 32 + r, s));
#pragma unroll
            for (int i = 0; i < NT; ++i)
                aw[i] = *reinterpret_cast<const bf16x8 *>(ws + frag_off(wn * TN + i * 32 + r, s));
        };
        if constexpr (NST == 2) {
            // tile kt landed for everyone; everyone finished reading tile kt-1.  The next tile's DMA pieces are issued
            // a k-step's worth at a time BEHIND that k-step's MFMAs (a piece costs its wave ~100 issue cycles: all of
            // them up front would leave the matrix pipe idle), and the fragments of k-step s+1 are read before the
            // MFMAs of k-step s.
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            bf16x8 bx[2][MT], aw[2][NT];
#if defined(TSIM_LN_DIAG) && (TSIM_LN_DIAG == 1 || TSIM_LN_DIAG == 4 || TSIM_LN_DIAG == 5)   // TIMING-ONLY: staging alone (no fragment reads, no MFMAs); 4: X pieces only, 5: W pieces only
            for (int s = 0; s < KSTEPS; ++s)
                if (kt + 1 < nk) {
#pragma unroll
                    for (int i = s * PPW_MAX / KSTEPS; i < (s + 1) * PPW_MAX / KSTEPS; ++i) {
                        const int p = wave + i * NW;
                        if (TSIM_LN_DIAG == 4 && p >= XPIECES) continue;
                        if (TSIM_LN_DIAG == 5 && p < XPIECES) continue;
                        if (p < PIECES) glds16(piece_src(i, p, kt + 1), smem + ((kt + 1) & 1) * STAGE + p * 1024);
                    }
                }
            continue;
#endif
            load_frags(0, bx[0], aw[0]);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                if (s + 1 < KSTEPS) load_frags(s + 1, bx[(s + 1) & 1], aw[(s + 1) & 1]);
                // (issuing the next k-tile's pieces behind the first one or two k-steps only, instead of all four: measured equal)
#if defined(TSIM_LN_DIAG) && TSIM_LN_DIAG == 2   // TIMING-ONLY: no staging inside the loop (stale tiles)
                if (false) {
#else
                if (kt + 1 < nk) {
#endif
#pragma unroll
                    for (int i = s * PPW_MAX / KSTEPS; i < (s + 1) * PPW_MAX / KSTEPS; ++i) {
                        const int p = wave + i * NW;
                        if (p < PIECES) glds16(piece_src(i, p, kt + 1), smem + ((kt + 1) & 1) * STAGE + p * 1024);
                    }
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[s & 1][i], bx[s & 1][j], acc[i][j], 0, 0, 0);
            }
        } else {
            wait_vmcnt<(NST - 2) * PPW>();  // tile kt landed (NST-2 younger tiles may be in flight)
            __builtin_amdgcn_s_barrier();
            const int nx = kt + NST - 1;
            issue(nx < nk ? nx : nk - 1, nx % NST);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                bf16x8 bx[MT], aw[NT];
                load_frags(s, bx, aw);
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[i], bx[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    if constexpr (NST > 2) wait_vmcnt<0>();   // the re-read tiles of the tail must land before LDS is reused / freed
    // ---- epilogue.  acc[i][j][g]: feature n = n0 + wn*TN + i*32 + (g&3) + 8*(g>>2) + 4*h, token m = m0 + wm*TM + j*32 + r
    const int nbase = n0 + wn * TN + 4 * h;
    const int mbase = m0 + wm * TM + r;

    if constexpr (EPI == EPI_RES_LN) {
        // Residual add + LayerNorm over the N = BN features of each token (two-pass statistics, fp32), with the
        // residual tile and the result tile moved THROUGH LDS: a lane owns scattered 8-byte groups of 32 different
        // token rows, so direct global loads/stores touch 32 cache lines per instruction (measured: 18 us of a 55 us
        // tile).  The [BM x N] bf16 tile is contiguous in memory: it comes in by LDS-DMA in whole 1-KiB pieces and
        // goes out as 16-byte row-contiguous stores; the scattered accesses hit LDS instead (16-byte slots XORed with
        // the row so that the 32 rows of a wave spread over the banks).
#if TSIM_LN_EPI_DIRECT
        // DIRECT form (end of round 2): the residual comes in and the result goes out as 16-byte accesses per lane with a
        // v_permlane32_swap between the half-waves (lane (r, h) touches features 8 (gq + h) .. + 7 of its token's row: 32
        // contiguous bytes per row and instruction pair), the way gemm_xres2 and the fused FFN kernel store.  No residual tile
        // DMA, no output tile in LDS, two workgroup barriers less; only the row statistics still cross the four feature waves
        // through LDS.  (The first form, below, moved both tiles through LDS because 8-byte groups per lane touched 32 cache
        // lines per instruction; timing-only builds put prologue + epilogue at 43 % of this kernel.)
        float *red = reinterpret_cast<float *>(smem);            // [2][WAVES_N][BM] partial sums
        __builtin_amdgcn_s_barrier();                            // every wave is past its last fragment read
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int64_t m = m0 + wm * TM + j * 32 + r;
            const int64_t mr = m < M ? m : M - 1;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; gq += 2) {
                    const uint4 o = *reinterpret_cast<const uint4 *>(res + mr * N + n0 + wn * TN + i * 32 + 8 * gq + 8 * h);
                    auto s0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
                    const uint32_t a0 = s0[0], c0 = s0[1], a1 = s1[0], c1 = s1[1];
                    acc[i][j][4 * gq + 0] += __uint_as_float(a0 << 16);
                    acc[i][j][4 * gq + 1] += __uint_as_float(a0 & 0xffff0000u);
                    acc[i][j][4 * gq + 2] += __uint_as_float(a1 << 16);
                    acc[i][j][4 * gq + 3] += __uint_as_float(a1 & 0xffff0000u);
                    acc[i][j][4 * gq + 4] += __uint_as_float(c0 << 16);
                    acc[i][j][4 * gq + 5] += __uint_as_float(c0 & 0xffff0000u);
                    acc[i][j][4 * gq + 6] += __uint_as_float(c1 << 16);
                    acc[i][j][4 * gq + 7] += __uint_as_float(c1 & 0xffff0000u);
                }
        }
        float mean[MT], rstd[MT];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        // explicit operations, nothing left to contraction: ln_rows_gemm_kernel repeats them literally
                        if (pass == 0) {
                            s += acc[i][j][g];
                        } else {
                            const float dlt = acc[i][j][g] - mean[j];
                            s = fmaf(dlt, dlt, s);
                        }
                    }
                s += __shfl_xor(s, 32, 64);
                if (h == 0) red[(pass * WAVES_N + wn) * BM + wm * TM + j * 32 + r] = s;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES_N; ++w) s += red[(pass * WAVES_N + w) * BM + wm * TM + j * 32 + r];
                if (pass == 0)
                    mean[j] = s / (float)N;
                else
                    rstd[j] = 1.0f / sqrtf(s / (float)N + eps);
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            float4 gv[4], be[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                gv[gq] = *reinterpret_cast<const float4 *>(gamma + nbase + i * 32 + 8 * gq);
                be[gq] = *reinterpret_cast<const float4 *>(beta + nbase + i * 32 + 8 * gq);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                uint32_t pk[8];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float y0 = fmaf((acc[i][j][4 * gq + 0] - mean[j]) * rstd[j], gv[gq].x, be[gq].x);
                    const float y1 = fmaf((acc[i][j][4 * gq + 1] - mean[j]) * rstd[j], gv[gq].y, be[gq].y);
                    const float y2 = fmaf((acc[i][j][4 * gq + 2] - mean[j]) * rstd[j], gv[gq].z, be[gq].z);
                    const float y3 = fmaf((acc[i][j][4 * gq + 3] - mean[j]) * rstd[j], gv[gq].w, be[gq].w);
                    pk[2 * gq] = pack_bf16x2(y0, y1);
                    pk[2 * gq + 1] = pack_bf16x2(y2, y3);
                }
                const int64_t m = m0 + wm * TM + j * 32 + r;
#pragma unroll
                for (int gq = 0; gq < 4; gq += 2) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gq], pk[2 * gq + 2], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gq + 1], pk[2 * gq + 3], false, false);
                    if (m < M)
                        *reinterpret_cast<uint4 *>(out + m * N + n0 + wn * TN + i * 32 + 8 * gq + 8 * h) =
                            make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        }
#else
        constexpr int SPR = BN / 8;                       // 16-byte slots per tile row
        constexpr int SWZ = SPR >= 16 ? 15 : SPR - 1;
        constexpr int TILE_B = BM * BN * 2;
        constexpr int TPIECES = TILE_B / 1024;
        static_assert(TILE_B % 1024 == 0 && TILE_B + 2 * WAVES_N * BM * 4 <= NST * STAGE, "epilogue tile must fit the staging LDS");
        float *red = reinterpret_cast<float *>(smem + TILE_B);   // [2][WAVES_N][BM] partial sums
        __builtin_amdgcn_s_barrier();                     // every wave is past its last fragment read
        {
            const char *rbase = reinterpret_cast<const char *>(res + (int64_t)m0 * N);
            for (int p = wave; p < TPIECES; p += NW) {
                const int sl = p * 64 + lane;
                const int row = sl / SPR, cp = sl % SPR;
                glds16(rbase + (int64_t)row * (BN * 2) + ((cp ^ (row & SWZ)) << 4), smem + p * 1024);
            }
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        auto tile_addr = [&](int row, int nloc) __attribute__((always_inline)) {   // 8-byte group of 4 features
            return smem + row * (BN * 2) + ((((nloc >> 3) ^ (row & SWZ))) << 4) + ((nloc & 4) << 1);
        };
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int row = wm * TM + j * 32 + r;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const uint2 rv = *reinterpret_cast<const uint2 *>(tile_addr(row, wn * TN + i * 32 + 8 * gq + 4 * h));
                    acc[i][j][4 * gq + 0] += __uint_as_float(rv.x << 16);
                    acc[i][j][4 * gq + 1] += __uint_as_float(rv.x & 0xffff0000u);
                    acc[i][j][4 * gq + 2] += __uint_as_float(rv.y << 16);
                    acc[i][j][4 * gq + 3] += __uint_as_float(rv.y & 0xffff0000u);
                }
        }
        float mean[MT], rstd[MT];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        // explicit operations, nothing left to contraction: ln_rows_gemm_kernel repeats them literally
                        if (pass == 0) {
                            s += acc[i][j][g];
                        } else {
                            const float dlt = acc[i][j][g] - mean[j];
                            s = fmaf(dlt, dlt, s);
                        }
                    }
                s += __shfl_xor(s, 32, 64);
                if (h == 0) red[(pass * WAVES_N + wn) * BM + wm * TM + j * 32 + r] = s;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES_N; ++w) s += red[(pass * WAVES_N + w) * BM + wm * TM + j * 32 + r];
                if (pass == 0)
                    mean[j] = s / (float)N;
                else
                    rstd[j] = 1.0f / sqrtf(s / (float)N + eps);
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = nbase + i * 32 + 8 * gq;
                const float4 gv = *reinterpret_cast<const float4 *>(gamma + n);
                const float4 be = *reinterpret_cast<const float4 *>(beta + n);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const float y0 = fmaf((acc[i][j][4 * gq + 0] - mean[j]) * rstd[j], gv.x, be.x);
                    const float y1 = fmaf((acc[i][j][4 * gq + 1] - mean[j]) * rstd[j], gv.y, be.y);
                    const float y2 = fmaf((acc[i][j][4 * gq + 2] - mean[j]) * rstd[j], gv.z, be.z);
                    const float y3 = fmaf((acc[i][j][4 * gq + 3] - mean[j]) * rstd[j], gv.w, be.w);
                    uint2 o;
                    o.x = pack_bf16x2(y0, y1);
                    o.y = pack_bf16x2(y2, y3);
                    // each lane overwrites exactly the residual group it read
                    *reinterpret_cast<uint2 *>(tile_addr(wm * TM + j * 32 + r, wn * TN + i * 32 + 8 * gq + 4 * h)) = o;
                }
            }
        __syncthreads();
        {
            char *obase = reinterpret_cast<char *>(out + (int64_t)m0 * N);
            for (int sl = threadIdx.x; sl < BM * SPR; sl += NW * 64) {
                const int row = sl / SPR, cp = sl % SPR;
                if (m0 + row < M)
                    *reinterpret_cast<uint4 *>(obase + (int64_t)row * (BN * 2) + ((cp ^ (row & SWZ)) << 4)) =
                        *reinterpret_cast<const uint4 *>(smem + sl * 16);
            }
        }
#endif
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = nbase + i * 32 + 8 * gq;
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int64_t m = mbase + j * 32;
                    float y0 = acc[i][j][4 * gq + 0], y1 = acc[i][j][4 * gq + 1];
                    float y2 = acc[i][j][4 * gq + 2], y3 = acc[i][j][4 * gq + 3];
                    if constexpr (EPI == EPI_GELU) {
                        gelu2(y0, y1);
                        gelu2(y2, y3);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(y0, y1);
                    o.y = pack_bf16x2(y2, y3);
                    *reinterpret_cast<uint2 *>(out + m * N + n) = o;
                }
            }
    }
}

// =====================================================================================================
// GEMM with the activation operand RESIDENT IN REGISTERS (K = 384 layers of MiniLM: QKV and FFN1).
//
// At K = 384 a tiled GEMM spends its time in prologues and epilogues: six K-steps per output tile.  Here a workgroup
// (8 waves) owns 256 tokens for the WHOLE layer: each wave loads its 32 token rows once as MFMA B fragments (K/16 x 4 =
// 96 VGPRs) and then sweeps every output feature while W streams through LDS exactly like the corpus does in the
// cosine kernel: 192-feature x 64-k tiles (24 KiB), 3-stage LDS-DMA ring, counted vmcnt, one raw s_barrier per tile, 24
// MFMAs per tile and wave, XOR-swizzled image read with conflict-free ds_read_b128.  The W stream never restarts
// between output tiles, so there is one prologue per 256 tokens instead of one per 128x128 tile.
// Orientation as in gemm_bf16_kernel (token on the lane); the epilogue pairs the two half-waves with
// v_permlane32_swap so every lane stores 16 contiguous bytes.
// =====================================================================================================
constexpr int XR_BN = 192, XR_BK = 64, XR_NSTAGE = 3;
constexpr int XR_STG_ROW = 208;   // bytes per token row of the output staging image (192 + 16 pad: spreads rows over banks)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#ifdef TSIM_PP_STAMPS
// DIAGNOSTIC build only (python -m text_similarity_amd.build --stamps; tools/pp_stamps.py --xres): cycles of wave 0 per
// workgroup: [0] tile-steps, [1] wait (vmcnt + barrier), [2] DMA issue, [3] fragment reads + MFMAs, [4] epilogues,
// [5] activation reloads, [6] items.
__device__ unsigned long long g_xr_stamps[8];
#define XR_T() __builtin_amdgcn_s_memtime()
#ifndef TSIM_XR_STAMP_TID
#define TSIM_XR_STAMP_TID 0   // 256: wave 4, the SIMD partner of wave 0
#endif
#define XR_ACC(i, v) do { if (threadIdx.x == TSIM_XR_STAMP_TID) atomicAdd(&g_xr_stamps[i], (unsigned long long)(v)); } while (0)
#else
#define XR_T() 0ull
#define XR_ACC(i, v) do { } while (0)
#endif

template <int K, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_xres_kernel(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W,
                                                        const float *__restrict__ bias, bf16_t *__restrict__ out,
                                                        int M, int N, int items_total) {
    constexpr int KSTEPS = K / 16, KG = K / XR_BK;          // 24 k-steps, 6 k-groups
    constexpr int STAGE = XR_BN * XR_BK * 2;                 // 24 KiB
    constexpr int PIECES = STAGE / 1024, PPW = PIECES / NW;  // 24 pieces, 3 (8 waves) or 6 (4 waves) per wave
    constexpr int NSUB = XR_BN / 32;                         // 6 feature sub-tiles of 32
    constexpr int BMX = NW * 32;                             // tokens per block
    static_assert(PIECES % NW == 0 && K % XR_BK == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ntiles = N / XR_BN;
    // persistent workgroups: a contiguous range of work items (token block of 32*NW, feature tile of 192), feature
    // tile fastest, so a workgroup changes token block at most a couple of times and the load is balanced to one item
    const int it0 = (int)((int64_t)items_total * blockIdx.x / gridDim.x);
    const int it1 = (int)((int64_t)items_total * (blockIdx.x + 1) / gridDim.x);
    const int total = (it1 - it0) * KG;                      // W tiles this workgroup streams
    if (total <= 0) return;

    int src_off[PPW];   // byte offset inside a W tile's source of this lane's 16 B
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int sl = (wave * PPW + i) * 64 + lane;
        const int sr = sl >> 4, ch = (sl & 15) ^ (sr & 15);
        const int row = sr * 2 + (ch >> 3), c = ch & 7;
        src_off[i] = row * K * 2 + c * 16;
    }
    auto issue = [&](int tt, int stage) {
        const int t2 = tt < total ? tt : total - 1;           // past-the-end: re-read the last tile (uniform vmcnt)
        const int j = (it0 + t2 / KG) % ntiles, g = t2 % KG;
        const char *base = reinterpret_cast<const char *>(W) + ((int64_t)j * XR_BN * K + g * XR_BK) * 2;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            glds16(base + src_off[i], smem + stage * STAGE + (wave * PPW + i) * 1024);
    };
    int aoff[NSUB];
#pragma unroll
    for (int i = 0; i < NSUB; ++i) aoff[i] = ((i * 32 + r) >> 1) * 256;   // super-row base; chunk XOR added per ks
    const int rodd = (r & 1) * 8;

    bf16x8 bx[KSTEPS];   // resident activation fragments: B[k = 8h + j][col r] of k-step s = X[m0 + r][16 s + 8 h + j]
    int cur_mb = -1, m0 = 0;
    f32x16 acc[NSUB];
#pragma unroll
    for (int i = 0; i < NSUB; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;

    // The whole bias vector goes into LDS once per (persistent) workgroup: an ordinary global load inside the loop would make
    // hipcc drain the W ring (s_waitcnt vmcnt(0)) at its first use, in every epilogue.  Read back with inline-asm ds_read.
    const uint32_t bias_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + XR_NSTAGE * STAGE +
                              NW * 32 * XR_STG_ROW;
    for (int p = wave; p * 256 < N; p += NW)
        if (p * 256 + lane * 4 < N) glds16(bias + p * 256 + lane * 4, smem + XR_NSTAGE * STAGE + NW * 32 * XR_STG_ROW + p * 1024);
    wait_vmcnt<0>();
    // bias[n .. n+3] for the four 8-feature groups of sub-tile i: four reads in flight, one wait
    auto bias16 = [&](int n, f32x4 (&bv)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            asm volatile("ds_read_b128 %0, %1" : "=v"(bv[gq]) : "v"(bias_lds + (n + 8 * gq) * 4) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3])::"memory");
    };
    [[maybe_unused]] unsigned long long xs_n = 0, xs_x = 0, xs_w = 0, xs_i = 0, xs_c = 0, xs_e = 0, xs_it = 0;   // (diagnostic build only)
    // vmcnt bookkeeping across an epilogue: its NS global stores are YOUNGER than the two W tiles in flight, and vmcnt
    // retires in issue order, so "all but my PPW youngest operations" (the plain ring wait) would drain every store of the
    // epilogue before the next MFMA could start — for the next TWO steps (the tile waited for in step s was issued in step
    // s-2).  Those two waits leave NS more operations outstanding instead.  Only when every store was issued for certain
    // (all 64 lanes active: no exec-zero skip) — otherwise the plain, draining wait.
    constexpr int NS = 12;   // global store instructions per wave and epilogue (both epilogue forms)
    static_assert(BMX * 12 / (NW * 64) == 6, "store count of the staged epilogue");
    int stores_younger = 0;
    issue(0, 0);
    issue(1, 1);
    for (int tt = 0; tt < total; ++tt) {
        const int stage = tt % XR_NSTAGE;
        const int g = tt % KG;
        const int item = it0 + tt / KG;
        const unsigned long long xt0 = XR_T();
        if (g == 0 && item / ntiles != cur_mb) {              // wave-uniform: new token block -> reload fragments
            cur_mb = item / ntiles;
            m0 = cur_mb * BMX + wave * 32;
            const bf16_t *xp = X + (int64_t)(m0 + r) * K + 8 * h;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) bx[s] = *reinterpret_cast<const bf16x8 *>(xp + 16 * s);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(bx[s]));   // retire these ordinary loads here
            stores_younger = 0;   // ... and with them (vmcnt(0)) everything older
        }
        const unsigned long long xt1 = XR_T();
#ifdef TSIM_XRES_DRAIN   // A/B: the plain ring wait everywhere (drains the epilogue's stores)
        stores_younger = 0;
#endif
        if (stores_younger > 0) {
            wait_vmcnt<PPW + NS>();
            --stores_younger;
        } else {
            wait_vmcnt<PPW>();
        }
        __builtin_amdgcn_s_barrier();
        const unsigned long long xt2 = XR_T();
        issue(tt + 2, (stage + 2) % XR_NSTAGE);
        const unsigned long long xt3 = XR_T();
        const char *ws = smem + stage * STAGE;
        // the k-group index selects which resident fragments to use: unrolled switch keeps bx[] in registers
#pragma unroll
        for (int gg = 0; gg < KG; ++gg) {
            if (g == gg) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int i = 0; i < NSUB; ++i) {
                        const int sr = (i * 32 + r) >> 1;
                        const int ch = (rodd + 2 * ks + h) ^ (sr & 15);
                        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ws + aoff[i] + (ch << 4));
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx[gg * 4 + ks], acc[i], 0, 0, 0);
                    }
                }
            }
        }
#ifdef TSIM_PP_STAMPS
#pragma unroll
        for (int i = 0; i < NSUB; ++i) asm volatile("" : "+v"(acc[i]));
#endif
        const unsigned long long xt4 = XR_T();
        xs_n += 1; xs_x += xt1 - xt0; xs_w += xt2 - xt1; xs_i += xt3 - xt2; xs_c += xt4 - xt3;
        if (g == KG - 1) {
            if constexpr (EPI == EPI_GELU) {
                // VALU-bound epilogue: direct 16-byte stores (the LDS-staged form below costs four more barriers
                // per item and measured 6 % slower here, 7 % faster for the bias-only epilogue)
                // epilogue of output tile j: acc[i][q]: feature n0 + i*32 + (q&3) + 8*(q>>2) + 4*h, token m0 + r
                const int n0 = (item % ntiles) * XR_BN;
                const int64_t m = m0 + r;
    #pragma unroll
                for (int i = 0; i < NSUB; ++i) {
                    uint32_t pk[8];   // pk[2*gq], pk[2*gq+1]: this lane's 4 features of group gq as packed bf16
                    f32x4 bvs[4];
                    bias16(n0 + i * 32 + 4 * h, bvs);
    #pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 bv = bvs[gq];
                        float y0 = acc[i][4 * gq] + bv[0], y1 = acc[i][4 * gq + 1] + bv[1];
                        float y2 = acc[i][4 * gq + 2] + bv[2], y3 = acc[i][4 * gq + 3] + bv[3];
                        if constexpr (EPI == EPI_GELU) {
                            gelu2(y0, y1); gelu2(y2, y3);
                        }
                        pk[2 * gq] = pack_bf16x2(y0, y1);
                        pk[2 * gq + 1] = pack_bf16x2(y2, y3);
                    }
                    // lane (r,0) holds features 8gq+0..3, lane (r,1) features 8gq+4..7.  Swap so that the low half-wave
                    // owns features 8gq..8gq+7 of group gq (even gq) and the high half-wave those of group gq+1: one
                    // 16-byte store per lane and group pair (cdna_hip_programming.md T21).
    #pragma unroll
                    for (int gq = 0; gq < 4; gq += 2) {
                        uint32_t a0 = pk[2 * gq], a1 = pk[2 * gq + 1], b0 = pk[2 * gq + 2], b1 = pk[2 * gq + 3];
                        auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                        auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                        a0 = s0[0]; b0 = s0[1]; a1 = s1[0]; b1 = s1[1];
                        if (m < M) {
                            uint4 o = make_uint4(a0, a1, b0, b1);
                            *reinterpret_cast<uint4 *>(out + m * N + n0 + i * 32 + 8 * gq + 8 * h) = o;
                        }
                    }
    #pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
                }
                if (m0 + 32 <= M) stores_younger = 2;   // wave-uniform: all 12 stores of this wave were issued
            } else {
                // epilogue of output tile j: acc[i][q]: feature n0 + i*32 + (q&3) + 8*(q>>2) + 4*h, token m0 + r.
                // A lane owns 32-byte pieces of 32 different token rows, so direct stores touch 32 cache lines per
                // instruction (measured: ~11 of 15.7 us per item).  The tile goes out THROUGH LDS instead, half a tile (96
                // features) at a time: 16-byte pieces into a padded [256 tokens][208 B] image, then every thread stores
                // row-contiguous 16-byte chunks (192 B per token row).  LDS accesses are inline asm so that hipcc does not
                // drain the W ring (vmcnt(0)) in front of them; the image lies behind the ring.
                const int n0 = (item % ntiles) * XR_BN;
                const int mb0 = cur_mb * BMX;
                const uint32_t stg = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + XR_NSTAGE * STAGE;
    #pragma unroll
                for (int half = 0; half < 2; ++half) {
    #pragma unroll
                    for (int il = 0; il < 3; ++il) {
                        const int i = half * 3 + il;
                        uint32_t pk[8];   // pk[2*gq], pk[2*gq+1]: this lane's 4 features of group gq as packed bf16
                        f32x4 bvs[4];
                        bias16(n0 + i * 32 + 4 * h, bvs);
    #pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const f32x4 bv = bvs[gq];
                            float y0 = acc[i][4 * gq] + bv[0], y1 = acc[i][4 * gq + 1] + bv[1];
                            float y2 = acc[i][4 * gq + 2] + bv[2], y3 = acc[i][4 * gq + 3] + bv[3];
                            if constexpr (EPI == EPI_GELU) {
                                gelu2(y0, y1); gelu2(y2, y3);
                            }
                            pk[2 * gq] = pack_bf16x2(y0, y1);
                            pk[2 * gq + 1] = pack_bf16x2(y2, y3);
                        }
                        // lane (r,0) holds features 8gq+0..3, lane (r,1) features 8gq+4..7: swap so that the low half-wave
                        // owns all 8 features of group gq (even gq) and the high half-wave those of group gq+1 (T21)
    #pragma unroll
                        for (int gq = 0; gq < 4; gq += 2) {
                            uint32_t a0 = pk[2 * gq], a1 = pk[2 * gq + 1], b0 = pk[2 * gq + 2], b1 = pk[2 * gq + 3];
                            auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                            auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                            u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                            const uint32_t addr = stg + (wave * 32 + r) * XR_STG_ROW + (il * 32 + 8 * gq + 8 * h) * 2;
                            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(o) : "memory");
                        }
    #pragma unroll
                        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    // 256 rows x 12 chunks of 16 B = 3072 chunks over NW*64 threads
                    constexpr int NCP = BMX * 12 / (NW * 64);   // 16-byte chunks per thread: all reads in flight, one wait
                    u32x4 cv[NCP];
    #pragma unroll
                    for (int c = 0; c < NCP; ++c) {
                        const int idx = c * NW * 64 + (int)threadIdx.x;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(cv[c]) : "v"(stg + (idx / 12) * XR_STG_ROW + (idx % 12) * 16) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    #pragma unroll
                    for (int c = 0; c < NCP; ++c) {
                        asm volatile("" : "+v"(cv[c]));
                        const int idx = c * NW * 64 + (int)threadIdx.x;
                        const int row = idx / 12, ch = idx % 12;
                        if (mb0 + row < M)
                            *reinterpret_cast<u32x4 *>(out + (int64_t)(mb0 + row) * N + n0 + half * 96 + ch * 8) = cv[c];
                    }
                    __builtin_amdgcn_s_barrier();   // image may be overwritten
                }
                if (mb0 + BMX <= M) stores_younger = 2;   // every thread issued its 12 stores
            }
            xs_e += XR_T() - xt4; xs_it += 1;
        }
    }
    wait_vmcnt<0>();
    XR_ACC(0, xs_n); XR_ACC(5, xs_x); XR_ACC(1, xs_w); XR_ACC(2, xs_i); XR_ACC(3, xs_c); XR_ACC(4, xs_e); XR_ACC(6, xs_it);
}

template <int... I, class Fn>
__device__ __forceinline__ void ff_static_for(std::integer_sequence<int, I...>, Fn &&f) {   // f(integral_constant<I>) for each I
    (f(std::integral_constant<int, I>{}), ...);
}

// =====================================================================================================
// gemm_xres2: the register-resident K = 384 projection with the EPILOGUE OVERLAPPED.
//
// In gemm_xres_kernel every sixth tile-step all eight waves stop feeding the matrix pipe and run the epilogue of a
// 192-feature item together (bias, GELU, packing, stores): 28 % of the kernel's time with the MFMA pipe idle, plus the
// activation reloads (profiles/README.md).  Here an item is 96 features (three 32-feature sub-tiles, W tiles of 96 x 128 k:
// the same 24 KiB per step, three steps per item) and there are TWO accumulator sets (2 x 48 VGPRs beside the 96 of the
// resident activations): while the 24 MFMAs of a step accumulate item i, the wave finishes one sub-tile of item i-1 in
// their shadow — two accumulator registers (GELU, bf16 pack) after every k-step, the half-wave swap and the two 16-byte
// stores at the end of the step.  The bias is not added in the epilogue: the accumulators START from it.
// Same k order per output as gemm_xres_kernel (ascending), so rows stay batch-invariant bit for bit.
// =====================================================================================================
// TWO STEPS PER BARRIER (TSIM_X2_PAIR).  Stamps (tools/x2_stamps.py): of a step's ~3 200 cycles a wave spends ~1 000 in the ring wait
// + workgroup barrier and both waves of a SIMD sit there together — a cost per BARRIER, not per MFMA.  With six ring slots the
// tiles are synchronised in pairs: one vmcnt wait + barrier in front of every even step covers the tiles of steps st and st + 1
// (both issued four steps earlier), the odd step runs straight on; a step issues the tile of step st + 4 into the slot the pair
// before the current one has left (everyone is past that pair: they passed this pair's barrier).
// MEASURED: QKV 73.1 -> 71.2 us, FFN1 119.6 -> 115.3 us (-3 %): the wait is mostly not a per-barrier constant.
#ifndef TSIM_X2_PAIR
#define TSIM_X2_PAIR 1
#endif
constexpr int X2_BN = 96, X2_BK = 128, X2_NSTAGE = TSIM_X2_PAIR ? 6 : 3, X2_PD = TSIM_X2_PAIR ? 4 : 2, X2_KG = 3, X2_NSUB = 3;

template <int EPI, bool PK>   // PK: the output goes out in the block-packed layout (see packed_off)
__global__ __launch_bounds__(512) void gemm_xres2_kernel(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W,
                                                         const float *__restrict__ bias, bf16_t *__restrict__ out,
                                                         int M, int N, int items_total) {
    constexpr int K = 384, KSTEPS = K / 16, NW = 8, BMX = NW * 32;
    constexpr int STAGE = X2_BN * X2_BK * 2;                 // 24 KiB
    constexpr int PIECES = STAGE / 1024, PPW = PIECES / NW;  // 24 pieces, 3 per wave
    static_assert(PIECES % NW == 0 && X2_KG * X2_BK == K, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ntiles = N / X2_BN;
    const int it0 = (int)((int64_t)items_total * blockIdx.x / gridDim.x);
    const int it1 = (int)((int64_t)items_total * (blockIdx.x + 1) / gridDim.x);
    const int total = (it1 - it0) * X2_KG;                   // W tiles (steps) of this workgroup
    if (total <= 0) return;

    // LDS image of a W tile: 96 rows of 256 B (128 k), 16-byte slot c of row rr holds source chunk c ^ (rr & 15)
    int src_off[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int sl = (wave * PPW + i) * 64 + lane;
        const int row = sl >> 4, ch = (sl & 15) ^ (row & 15);
        src_off[i] = row * K * 2 + ch * 16;
    }
    auto issue_pieces = [&](int st, int stage, int i0, int i1) __attribute__((always_inline)) {
        const int s2 = st < total ? st : total - 1;           // past-the-end: re-read the last tile (uniform vmcnt)
        const int j = (it0 + s2 / X2_KG) % ntiles, g = s2 % X2_KG;
        const char *base = reinterpret_cast<const char *>(W) + ((int64_t)j * X2_BN * K + g * X2_BK) * 2;
#if defined(TSIM_X2_DIAG) && (TSIM_X2_DIAG & 4)   // TIMING-ONLY: no W stream (stale tiles)
        (void)base;
#else
        for (int i = i0; i < i1; ++i)
            glds16(base + src_off[i], smem + stage * STAGE + (wave * PPW + i) * 1024);
#endif
    };
    auto issue = [&](int st, int stage) __attribute__((always_inline)) { issue_pieces(st, stage, 0, PPW); };
    // fragment of sub-tile i, k-step ks (of 8 in a tile): row i*32 + r, source chunk 2 ks + h -> slot (2 ks + h) ^ (r & 15)
    int cks[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) cks[ks] = r * 256 + (((2 * ks + h) ^ (r & 15)) << 4);

    // bias -> LDS once per workgroup (an ordinary global load inside the loop would drain the W ring at its first use)
    const uint32_t bias_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + X2_NSTAGE * STAGE;
    for (int p = wave; p * 256 < N; p += NW)
        if (p * 256 + lane * 4 < N) glds16(bias + p * 256 + lane * 4, smem + X2_NSTAGE * STAGE + p * 1024);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    bf16x8 bx[KSTEPS];
    int cur_mb = -1, m0 = 0;
    f32x16 accA[X2_NSUB], accB[X2_NSUB];
    int old_m0 = 0, old_n0 = 0;
    int st = 0;                       // steps done
    int ring = 0;                     // st % X2_NSTAGE, kept as a counter
    int young1 = 0, young2 = 0;       // global stores issued in the previous step / the one before (-1: unknown)
    [[maybe_unused]] int young3 = 0;  // ... and the one before that (pairs)

    // epilogue of ONE finished sub-tile whose 16 registers have been packed into pk[8] (pk[2gq], pk[2gq+1] = the lane's four
    // features 8gq + 4h .. +3 of group gq): swap half-waves so that every lane owns 8 contiguous features, two 16-byte stores
    // EARLY STORES (round 3).  Stamps (tools/x2_stamps.py): the two 16-byte stores of a sub-tile, issued between a step's MFMA
    // stream and the next barrier, cost their wave 650-750 cycles per step (a store instruction touches 32 rows x 32 B and is
    // issue-bound), during which it feeds nothing to the matrix pipe — and both partners of a SIMD sit in that segment or at the
    // barrier for ~30 % of a step.  Here the shadow epilogue runs at double density over the first four k-steps (all of pk[] is
    // ready after k-step 3) and the two stores go out INSIDE the stream, behind k-steps 4 and 6: while a wave queues at the
    // address path its partner has the matrix pipe.  Same values, same addresses; only the issue position moves.
#ifndef TSIM_X2_EARLY_STORE
#define TSIM_X2_EARLY_STORE 0   // measured EQUAL to stores at the end of the step (2.61-2.62 ms per forward either way; the
#endif                          // micro-benchmark agrees: 2 480 vs 2 598 cycles per step): what costs is the store's PATTERN
    // (the store is inline asm in the SGPR-base + 32-bit-offset form: inside the stream there is no room for a 64-bit address
    // per store; the output is < 4 GiB, checked by the launcher.  All vmcnt bookkeeping of this kernel is manual anyway.)
    uint32_t old_rowoff = 0;   // byte offset of this lane's 16-byte column group in its token row of the PREVIOUS item's block
    const uint64_t out_base = (uint64_t)(uintptr_t)out;
    auto store_half = [&](const uint32_t (&pk)[8], int mrow0, int ncol, auto gqc) __attribute__((always_inline)) {
        constexpr int gq = decltype(gqc)::value;
        const bool full = mrow0 + 32 <= M;    // wave-uniform
        uint32_t a0 = pk[2 * gq], a1 = pk[2 * gq + 1], b0 = pk[2 * gq + 2], b1 = pk[2 * gq + 3];
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        const uint32_t voff = old_rowoff + (uint32_t)(ncol + 8 * gq) * (PK ? 64u : 2u);
        const uint64_t ob = out_base;   // (an asm operand alone does not make a generic lambda capture the variable)
#if !(defined(TSIM_X2_DIAG) && (TSIM_X2_DIAG & 1))   // (bit 1: TIMING-ONLY, no output stores)
        if (full || mrow0 + r < M)
            // (s_nop: a store of more than 64 bits needs one wait state before a VALU may overwrite its data registers; hipcc
            // pads its own stores but cannot see into an asm statement — without it the next instruction now and then replaced the
            // data under the store: a corrupted 16-byte group, non-finite rows after the LayerNorm, the whole sequence after the
            // next attention, in ~20 % of the forwards on some boxes and none on others)
#ifdef TSIM_X2_DIAG_NO_STORE_NOP   // (control build for tools/nan_probe.py: the fault as it was)
            asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(o), "s"(ob) : "memory");
#else
            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(o), "s"(ob) : "memory");
#endif
#else
        asm volatile("" ::"v"(voff), "v"(o));
#endif
    };
    auto finish2 = [&](float y0, float y1) __attribute__((always_inline)) -> uint32_t {
#if !(defined(TSIM_X2_DIAG) && (TSIM_X2_DIAG & 2))   // (bit 2: TIMING-ONLY, no activation function)
        // (two independent scalar chains, gelu_n<2>, instead of the packed one: measured equal, 2.633 vs 2.632 ms per forward)
        if constexpr (EPI == EPI_GELU) gelu2(y0, y1);
#endif
        return pack_bf16x2(y0, y1);
    };

#pragma unroll
    for (int i = 0; i < X2_PD; ++i) issue(i, i);
    [[maybe_unused]] unsigned x2_n = 0, x2_w = 0, x2_c = 0, x2_e = 0;   // (diagnostic build only; 32-bit sums: a workgroup runs < 2^32 cycles)
    auto run_item = [&](f32x16 (&cur)[X2_NSUB], f32x16 (&old)[X2_NSUB], int item, auto with_old) __attribute__((always_inline)) {
        constexpr bool OLD = decltype(with_old)::value;
        if (item / ntiles != cur_mb) {                        // wave-uniform: new token block -> reload fragments
            cur_mb = item / ntiles;
            m0 = cur_mb * BMX + wave * 32;
            const bf16_t *xp = X + (int64_t)(m0 + r) * K + 8 * h;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) bx[s] = *reinterpret_cast<const bf16x8 *>(xp + 16 * s);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(bx[s]));   // retire these ordinary loads here
            young1 = young2 = young3 = 0;                     // ... and with them (vmcnt(0)) everything older
        }
        const int n0 = (item % ntiles) * X2_BN;
        // accumulators start from the bias: cur[i][q] belongs to feature n0 + 32 i + (q & 3) + 8 (q >> 2) + 4 h
        // (ONE address register + immediates: twelve separately computed addresses were hoisted out of the loop and spilled)
        const uint32_t baddr = bias_lds + (n0 + 4 * h) * 4;
        ff_static_for(std::make_integer_sequence<int, X2_NSUB>{}, [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            f32x4 bv[4];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[0]) : "v"(baddr), "n"((i * 32 + 0) * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[1]) : "v"(baddr), "n"((i * 32 + 8) * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[2]) : "v"(baddr), "n"((i * 32 + 16) * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[3]) : "v"(baddr), "n"((i * 32 + 24) * 4) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3])::"memory");
#pragma unroll
            for (int q = 0; q < 16; ++q) cur[i][q] = bv[q >> 2][q & 3];
        });
#pragma unroll
        for (int g = 0; g < X2_KG; ++g) {
            // (opaque on purpose: with six steps per loop trip and six slots hipcc knows the slot of every unrolled step, hoists the
            // 6 x 8 fragment addresses out of the loop as invariants and spills all 48 of them)
            if constexpr (TSIM_X2_PAIR) asm volatile("" : "+s"(ring));
            const int stage = TSIM_X2_PAIR ? ring : st % X2_NSTAGE;
            [[maybe_unused]] const unsigned long long xs0 = XR_T();
            // ring wait: the tile of this step was issued two steps ago; the stores of the last two steps are younger than it
            if constexpr (TSIM_X2_EARLY_STORE) {
                // issue order inside a step: piece a (k-step 1), piece b (3), store 1 (4), piece c (5), store 2 (6).  Younger than
                // the pieces of step st-2: its store 2, and everything of step st-1.
                const int extra = (young1 >= 0 && young2 >= 0) ? young1 + (young2 >> 1) : 0;   // 0 (or unknown: drain), 1, 2, 3
                if (extra == 3) wait_vmcnt<PPW + 3>();
                else if (extra == 2) wait_vmcnt<PPW + 2>();
                else if (extra == 1) wait_vmcnt<PPW + 1>();
                else wait_vmcnt<PPW>();
            } else if constexpr (TSIM_X2_PAIR) {
                // even step: the tiles of this step and the next must have landed.  Younger than the next step's pieces: the pieces
                // of steps st + 2 and st + 3 and the stores of the last three steps
                if ((st & 1) == 0) {
#if defined(TSIM_X2_PAIR_DBG) && (TSIM_X2_PAIR_DBG & 4)
                    const int ys = 0;
#else
                    const int ys = (young1 >= 0 && young2 >= 0 && young3 >= 0) ? young1 + young2 + young3 : 0;
#endif
                    if (ys == 6) wait_vmcnt<2 * PPW + 6>();
                    else if (ys == 4) wait_vmcnt<2 * PPW + 4>();
                    else if (ys == 2) wait_vmcnt<2 * PPW + 2>();
                    else wait_vmcnt<2 * PPW>();               // none, or unknown store count: always safe
#if defined(TSIM_X2_PAIR_DBG) && (TSIM_X2_PAIR_DBG & 2)
                    wait_vmcnt<0>();
#endif
                    __builtin_amdgcn_s_barrier();
                }
#if defined(TSIM_X2_PAIR_DBG) && (TSIM_X2_PAIR_DBG & 1)
                else __builtin_amdgcn_s_barrier();
#endif
                __builtin_amdgcn_sched_barrier(0);            // (the odd step has no barrier to fence hipcc's scheduler either)
            } else {
            if (young1 == 0 && young2 == 0) wait_vmcnt<PPW>();
            else if (young1 >= 0 && young2 >= 0 && young1 + young2 == 2) wait_vmcnt<PPW + 2>();
            else if (young1 >= 0 && young2 >= 0 && young1 + young2 == 4) wait_vmcnt<PPW + 4>();
            else wait_vmcnt<PPW>();                           // unknown store count: drain (always safe)
            }
            if constexpr (!TSIM_X2_PAIR) __builtin_amdgcn_s_barrier();
            [[maybe_unused]] const unsigned long long xs1 = XR_T();
            // Right behind the barrier all eight waves have LDS-DMA to issue and queue at the CU's one address path while the
            // matrix pipe idles; the tile is not needed for two steps, so its three pieces are dropped between the k-steps'
            // MFMAs instead (TSIM_X2_SPREAD=0: all three at the head of the step).
#ifndef TSIM_X2_SPREAD
#define TSIM_X2_SPREAD 1
#endif
            static_assert(PPW == 3, "issue schedule below places three pieces");
            static_assert(!(TSIM_X2_PAIR && TSIM_X2_EARLY_STORE), "the pair form counts stores at the end of a step");
            if constexpr (!TSIM_X2_SPREAD) issue(st + X2_PD, (stage + X2_PD) % X2_NSTAGE);
            const char *ws = smem + stage * STAGE;
            uint32_t pk[8];
#ifndef TSIM_X2_ASMPIPE
#define TSIM_X2_ASMPIPE 1
#endif
#if TSIM_X2_ASMPIPE
            // The step's 24 fragment reads (n = 3 ks + i) roll PF reads ahead of their MFMAs with COUNTED lgkmcnt waits, as in the
            // search kernel's tile loop: hipcc's own schedule of the plain loads below waits lgkmcnt(0) in front of every k-step,
            // i.e. for the three reads it has just issued (timing-only builds: the bare loop without stores, activation and W
            // stream ran at 43 % of the MFMA rate).  No other LDS instruction is issued between these reads (LDS-DMA counts in
            // vmcnt), so the counts are exact.
            {
                constexpr int PF = 3, NRD = 8 * X2_NSUB;
                lds_u32x4 fr[PF + 1];
                const uint32_t wsl = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + stage * STAGE;
                auto rd = [&](auto nc) __attribute__((always_inline)) {
                    constexpr int n = decltype(nc)::value;
                    lds_read_b128_imm<(n % X2_NSUB) * 8192>(fr[n % (PF + 1)], wsl + cks[n / X2_NSUB]);
                };
                ff_static_for(std::make_integer_sequence<int, PF>{}, rd);
                ff_static_for(std::make_integer_sequence<int, NRD>{}, [&](auto nc) __attribute__((always_inline)) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int ks = n / X2_NSUB, i = n % X2_NSUB;
                    if constexpr (n + PF < NRD) rd(std::integral_constant<int, n + PF>{});
                    constexpr int younger = n + PF < NRD ? PF : NRD - 1 - n;
                    lgkm_wait_counted<younger>(fr[n % (PF + 1)]);
                    cur[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[n % (PF + 1)]), bx[g * 8 + ks], cur[i], 0, 0, 0);
                    if constexpr (i == X2_NSUB - 1) {
                        if constexpr (TSIM_X2_SPREAD && (ks == 1 || ks == 3 || ks == 5))
                            issue_pieces(st + X2_PD, (stage + X2_PD) % X2_NSTAGE, ks >> 1, (ks >> 1) + 1);
                        if constexpr (OLD && !TSIM_X2_EARLY_STORE) pk[ks] = finish2(old[g][2 * ks], old[g][2 * ks + 1]);   // in the MFMAs' shadow
                        if constexpr (OLD && TSIM_X2_EARLY_STORE) {
                            if constexpr (ks < 4) {
                                pk[2 * ks] = finish2(old[g][4 * ks], old[g][4 * ks + 1]);
                                pk[2 * ks + 1] = finish2(old[g][4 * ks + 2], old[g][4 * ks + 3]);
                            }
                            if constexpr (ks == 4) store_half(pk, old_m0, old_n0 + g * 32, std::integral_constant<int, 0>{});
                            if constexpr (ks == 6) store_half(pk, old_m0, old_n0 + g * 32, std::integral_constant<int, 2>{});
                        }
                    }
                });
            }
#else
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int i = 0; i < X2_NSUB; ++i) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ws + i * 8192 + cks[ks]);
                    cur[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx[g * 8 + ks], cur[i], 0, 0, 0);
                }
                if constexpr (TSIM_X2_SPREAD) {
                    if (ks == 1 || ks == 3 || ks == 5) issue_pieces(st + X2_PD, (stage + X2_PD) % X2_NSTAGE, ks >> 1, (ks >> 1) + 1);
                }
                if constexpr (OLD) pk[ks] = finish2(old[g][2 * ks], old[g][2 * ks + 1]);   // in the MFMAs' shadow
            }
#endif
            (void)ws;
#ifdef TSIM_PP_STAMPS
#pragma unroll
            for (int i = 0; i < X2_NSUB; ++i) asm volatile("" : "+v"(cur[i]));
#endif
            [[maybe_unused]] const unsigned long long xs2 = XR_T();
            young3 = young2;
            young2 = young1;
            young1 = 0;
            if constexpr (OLD) {
                if constexpr (TSIM_X2_EARLY_STORE) {
#if defined(TSIM_X2_DIAG) && (TSIM_X2_DIAG & 1)
                    young1 = 0;
#else
                    young1 = old_m0 + 32 <= M ? 2 : -1;      // both stores of this step were issued for certain / unknown
#endif
                } else {
                    store_half(pk, old_m0, old_n0 + g * 32, std::integral_constant<int, 0>{});
                    store_half(pk, old_m0, old_n0 + g * 32, std::integral_constant<int, 2>{});
#if defined(TSIM_X2_DIAG) && (TSIM_X2_DIAG & 1)
                    young1 = 0;
#else
                    young1 = old_m0 + 32 <= M ? 2 : -1;
#endif
                }
            }
            ++st;
            ring = ring + 1 == X2_NSTAGE ? 0 : ring + 1;
#ifdef TSIM_PP_STAMPS
            { const unsigned long long xs3 = XR_T(); x2_n += 1; x2_w += (unsigned)(xs1 - xs0); x2_c += (unsigned)(xs2 - xs1); x2_e += (unsigned)(xs3 - xs2); }
#endif
        }
        old_m0 = m0;
        old_n0 = n0;
        old_rowoff = PK ? (uint32_t)(m0 >> 5) * (uint32_t)(N * 64) + 16u * r + 512u * h     // m0 is a multiple of 32
                        : (uint32_t)(m0 + r) * (uint32_t)(N * 2) + 16u * h;
    };
    int item = it0;
    run_item(accA, accB, item++, std::false_type{});
    for (; item + 2 <= it1; item += 2) {
        run_item(accB, accA, item, std::true_type{});
        run_item(accA, accB, item + 1, std::true_type{});
    }
    const bool lastA = item >= it1;          // the last finished item sits in accA, unless one more item goes into accB
    if (!lastA) run_item(accB, accA, item, std::true_type{});
    wait_vmcnt<0>();                         // no LDS-DMA may outlive the workgroup (past-the-end tiles)
    // flush: the last item's three sub-tiles
    auto flush = [&](f32x16 (&acc)[X2_NSUB]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < X2_NSUB; ++g) {
            uint32_t pk[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) pk[ks] = finish2(acc[g][2 * ks], acc[g][2 * ks + 1]);
            store_half(pk, old_m0, old_n0 + g * 32, std::integral_constant<int, 0>{});
            store_half(pk, old_m0, old_n0 + g * 32, std::integral_constant<int, 2>{});
        }
    };
    if (lastA) flush(accA); else flush(accB);
    XR_ACC(0, x2_n); XR_ACC(1, x2_w); XR_ACC(3, x2_c); XR_ACC(4, x2_e); XR_ACC(6, x2_n / 3);
}

// =====================================================================================================
// ffn_fused: FFN1 -> GELU -> FFN2 -> +residual -> LayerNorm in ONE kernel (hidden 384; the [T, F] intermediate never leaves
// the CU).  Unfused, a MiniLM layer writes and re-reads 412 MB of h = GELU(x W1^T + b1) through HBM: more traffic than all its
// other tensors together, and the two projections around it sit at a third of the HBM roofline and a quarter of the MFMA
// roofline at once (profiles/README.md).
//
// Workgroup = 128 tokens = 4 token groups of 32, TWO waves per group with different ROLES (partners w and w+4 share a SIMD):
//   producer (waves 0-3): holds the group's x rows as MFMA B fragments (96 VGPRs); per phase s it scores ONE 32-feature chunk
//          h^T[32 features x 32 tokens] = W1[chunk s] . x^T (24 MFMAs, accumulator starts from b1), applies GELU, rounds to
//          bf16 and writes the tile to a 2-KiB LDS slot — in the accumulator's own layout, which IS the B-operand layout of
//          the second product (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand");
//          GELU + rounding of chunk s-1 run in the shadow of chunk s's MFMA chain (two accumulators);
//   consumer (waves 4-7): holds the group's whole output row block y^T[384 features x 32 tokens] (12 accumulator tiles = 192
//          VGPRs, starting from b2); per phase it adds chunk s-2: y^T[t] += W2[rows of t, chunk] . h (2 MFMAs per tile, 24).
// The producer runs two chunks ahead of the consumer (two h slots per group), so on every SIMD one wave's GELU / LDS traffic
// runs beside the other's MFMAs, and neither role needs more than ~230 registers: 96 + 192 in one wave would not fit.
// The k order inside a k-step of the second product is the accumulator's row order (16 s + 8 (j >> 2) + 4 h + (j & 3)): W2 is
// re-laid at load time (pack_ffn_w2_kernel) so that its A fragments are plain 16-byte reads.
// W streams through a ring of five 24-KiB LDS slots by LDS-DMA, unit 2s = W1[chunk s], unit 2s+1 = W2[.., chunk s-2]: both
// matrices are stored as the exact LDS images (pack_ffn_w*_kernel), so every DMA piece is 1 KiB of contiguous memory.  One
// raw s_barrier per phase (it also publishes the h slot); phase s waits for its two units with vmcnt(3) (one younger unit
// stays in flight) and issues units 2s+3, 2s+4.
// Staging: 48 KiB per phase for 48 MFMAs per SIMD = 31 B/clk/CU at the full MFMA rate — the order of the measured L2 -> LDS
// rates, so this kernel runs near the staging bound; what it removes is the HBM round trip of h.
// Epilogue (consumers): residual added in the accumulator layout (16-byte loads + v_permlane32_swap, the store path
// backwards), two-pass LayerNorm statistics inside the wave, 16-byte row stores.
// Summation order per output: chunks ascending, the permuted k order inside: independent of what else is in the batch.
// =====================================================================================================
constexpr int FF_UNIT = 24576, FF_NSLOT = 5, FF_XB = 16384;   // ring unit, ring slots, h fragments (4 groups x 2 x 2 KiB)
#ifdef TSIM_PP_STAMPS
// DIAGNOSTIC build: [0..3] producer wave 0: phases, wait (vmcnt + barrier), DMA issue, work; [4..7] the same for consumer wave 4
__device__ unsigned long long g_ff_stamps[8];
#define FF_T() __builtin_amdgcn_s_memtime()
#else
#define FF_T() 0ull
#endif

// W [BN rows, K] -> per k-tile of BK the W-region LDS image of gemm_bf16_kernel<.., BN, BK, ..>: 16-byte slot sl of the image
// (super-row sr = sl >> 4 of 256 B = 256 / (2 BK) tile rows, slot chp = sl & 15) holds chunk ch = chp ^ (sr & 15) of the super-row.
__global__ __launch_bounds__(256) void pack_gemm_w_kernel(const uint4 *__restrict__ W, uint4 *__restrict__ img, int BN, int BK, int K) {
    const int slots = BN * BK * 2 / 16;                             // 16-byte slots per k-tile image
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= (int64_t)slots * (K / BK)) return;
    const int kt = (int)(u / slots), sl = (int)(u % slots);
    const int cpr = BK * 2 / 16, rps = 256 / (BK * 2);
    const int sr = sl >> 4, ch = (sl & 15) ^ (sr & 15);
    const int row = sr * rps + ch / cpr, c = ch % cpr;
    img[u] = W[((int64_t)row * K + (int64_t)kt * BK) * 2 / 16 + c];
}
// W [N rows, K] -> MFMA A-fragment images: 1-KiB block (k-step s, 32-feature sub-tile j) at (s * (N / 32) + j) * 1024, lane (r, h)'s
// 16 bytes = W[32 j + r][16 s + 8 h .. + 8).  Two consecutive k-steps (24 KiB at N = 384) are ln_rows_gemm_kernel's W region of a
// 32-k tile — LDS-DMA copies it verbatim, a fragment read is lane-linear (conflict-free, one address register) — and
// ln_tail_gemm_kernel loads the same blocks straight into registers, one contiguous KiB per load.
__global__ __launch_bounds__(256) void pack_frag_w_kernel(const uint4 *__restrict__ W, uint4 *__restrict__ img, int N, int K) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;      // 16-byte unit of the image
    if (u >= (int64_t)N * K / 8) return;
    const int lane = (int)(u & 63), blk = (int)(u >> 6);
    const int nt = N / 32, s = blk / nt, j = blk % nt, r = lane & 31, h = lane >> 5;
    img[u] = W[((int64_t)(32 * j + r) * K + 16 * s + 8 * h) / 8];
}
// W1 [F, 384] -> per 32-row chunk the K1-style LDS image: row rr, 16-byte slot c holds source chunk c ^ (rr & 15) (low 4 bits)
__global__ __launch_bounds__(256) void pack_ffn_w1_kernel(const uint4 *__restrict__ W, uint4 *__restrict__ img, int F) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;      // 16-byte unit of the image
    if (u >= (int64_t)F * 48) return;
    const int c = (int)(u / 1536), sl = (int)(u % 1536);            // chunk, slot in the chunk image (32 rows x 48 slots)
    const int rr = sl / 48, cc = sl % 48;
    img[u] = W[((int64_t)(c * 32 + rr) * 48) + (cc ^ (rr & 15))];
}
// W2 [384, F] -> per 32-column chunk c: image [384 rows][4 slots of 16 B]; slot s' of row ro holds k-step s, half hh with
// (2 s + hh) = s' ^ ((ro >> 2) & 3), its 8 elements in the accumulator's k order: W2[ro][32 c + 16 s + 8 (j >> 2) + 4 hh + (j & 3)]
__global__ __launch_bounds__(256) void pack_ffn_w2_kernel(const bf16_t *__restrict__ W, bf16_t *__restrict__ img, int F) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // element of the image
    if (e >= (int64_t)384 * F) return;
    const int c = (int)(e / (384 * 32)), o = (int)(e % (384 * 32));
    const int ro = o / 32, sp = (o % 32) / 8, j = o % 8;
    const int sh = sp ^ ((ro >> 2) & 3), s = sh >> 1, hh = sh & 1;
    img[e] = W[(int64_t)ro * F + 32 * c + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)];
}

__global__ __launch_bounds__(512) void ffn_fused_kernel(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W1img,
                                                        const bf16_t *__restrict__ W2img, const float *__restrict__ b1,
                                                        const float *__restrict__ b2, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float eps, bf16_t *__restrict__ out,
                                                        int M, int F) {
    constexpr int K = 384, KSTEPS = 24, PPU = 3;             // pieces of a 24-KiB unit per wave (24 / 8)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tg = wave & 3, consumer = wave >> 2;
    const int r = lane & 31, hh = lane >> 5;
    const int nchunks = F / 32, nphases = nchunks + 2, nunits = 2 * nphases;
    const int m0 = blockIdx.x * 128 + tg * 32;
    char *ring = smem;
    char *xbuf = smem + FF_NSLOT * FF_UNIT;                   // h fragments of (group, chunk parity): xbuf + (tg * 2 + par) * 2048
    const uint32_t b1_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + FF_NSLOT * FF_UNIT + FF_XB;

    // unit u of phase u/2: even = W1 image of chunk u/2 (producer); odd = W2 image of chunk u/2 - 2 (consumer, two phases behind)
    // pieces [i0, i1) of this wave's PPU pieces of unit u
    auto issue_pieces = [&](int u, int i0, int i1) __attribute__((always_inline)) {
        const int uu = u < nunits ? u : nunits - 1;           // past-the-end: re-read the last unit (uniform vmcnt)
        int c = (uu & 1) ? (uu >> 1) - 2 : (uu >> 1);
        c = c < 0 ? 0 : (c >= nchunks ? nchunks - 1 : c);     // units outside the chunk range are never consumed: any valid source
        const char *src = reinterpret_cast<const char *>((uu & 1) ? W2img : W1img) + (int64_t)c * FF_UNIT;
        char *dst = ring + (uu % FF_NSLOT) * FF_UNIT;
#ifdef TSIM_FF_DIAG_NODMA   // DIAGNOSTIC (results wrong): the kernel without its W stream
        (void)src; (void)dst;
#else
        for (int i = i0; i < i1; ++i)
            glds16(src + (wave * PPU + i) * 1024 + lane * 16, dst + (wave * PPU + i) * 1024);
#endif
    };
    auto issue_unit = [&](int u) __attribute__((always_inline)) { issue_pieces(u, 0, PPU); };
    // Issue schedule inside a phase.  Right behind the barrier every wave of the CU has LDS-DMA to issue, the CU's address path
    // takes one wave-instruction at a time (~16 cycles per KiB) and a wave whose next instruction is such a load waits for
    // its turn: with all six pieces up front every wave sat ~500-600 cycles in that queue while the matrix pipe idled (stamps).
    // So only the unit that is needed NEXT phase goes out at once; the pieces of the unit after it are dropped between the
    // phase's MFMAs (TSIM_FF_SPREAD=0: the first form).
#ifndef TSIM_FF_SPREAD
#define TSIM_FF_SPREAD 1
#endif
    // b1 -> LDS (read per chunk inside the loop: an ordinary global load there would drain the ring)
    for (int p = wave; p * 256 < F; p += 8)
        if (p * 256 + lane * 4 < F) glds16(b1 + p * 256 + lane * 4, smem + FF_NSLOT * FF_UNIT + FF_XB + p * 1024);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    if (!consumer) {
        // ================================================================ producer: FFN1 + GELU, one chunk per phase
        bf16x8 bx[KSTEPS];   // B[k = 8 hh + j][col r] of k-step s = X[m0 + r][16 s + 8 hh + j]; rows past M repeat the last one
        {
            const int64_t row = m0 + r < M ? m0 + r : M - 1;
            const bf16_t *xp = X + row * K + 8 * hh;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) bx[s] = *reinterpret_cast<const bf16x8 *>(xp + 16 * s);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(bx[s]));   // retired before any LDS-DMA is in flight
        }
        issue_unit(0);
        issue_unit(1);
        issue_unit(2);
        const int w1off = r * 768;                             // + (s >> 3) * 256 + (((2 (s & 7) + hh) ^ (r & 15)) << 4)
        int w1x[8];
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) w1x[bb] = ((2 * bb + hh) ^ (r & 15)) << 4;
        const uint32_t slot0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)xbuf) + tg * 4096 + lane * 32;
        // Phase ph: the 24 MFMAs of chunk ph, and in their shadow GELU + bf16 of chunk ph-1 (whose accumulator was finished in
        // the phase before) as sixteen independent polynomial chains (gelu_n): the accumulation chain of a chunk and the
        // activation of the previous one overlap instead of following each other.  h of chunk c is published at the end of
        // phase c+1 and read by the consumer in phase c+2.
        f32x16 hprev;
#pragma unroll
        for (int q = 0; q < 16; ++q) hprev[q] = 0.f;
        [[maybe_unused]] unsigned long long fs_w = 0, fs_i = 0, fs_k = 0;
        auto phase = [&](int ph, auto has_mfma, auto has_prev) __attribute__((always_inline)) {
            constexpr bool MM = decltype(has_mfma)::value, PV = decltype(has_prev)::value;
            [[maybe_unused]] const unsigned long long t0 = FF_T();
            wait_vmcnt<PPU>();                                 // units 2ph, 2ph+1 landed (2ph+2 may be in flight)
            __builtin_amdgcn_s_barrier();                      // ... for everyone; everyone is past phase ph-1
            [[maybe_unused]] const unsigned long long t1 = FF_T();
            issue_unit(2 * ph + 3);
            if constexpr (!MM || !TSIM_FF_SPREAD) issue_unit(2 * ph + 4);
            [[maybe_unused]] const unsigned long long t2 = FF_T();
            f32x16 hacc;
            [[maybe_unused]] f32x16 hacc1;   // FF_NACC = 2: odd k-steps accumulate here (two independent chains), added at the end
            if constexpr (MM) {
#pragma unroll
                for (int q = 0; q < 16; ++q) hacc1[q] = 0.f;
                f32x4 bv[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bv[gq]) : "v"(b1_lds + (32 * ph + 8 * gq + 4 * hh) * 4) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3])::"memory");
#pragma unroll
                for (int q = 0; q < 16; ++q) hacc[q] = bv[q >> 2][q & 3];
            }
            // the accumulation chain is ONE dependent MFMA after the other: fragment reads roll PF k-steps ahead (counted waits)
            constexpr int PF = 6;
            lds_u32x4 fr[PF + 1];
            const uint32_t tb = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)(ring + ((2 * ph) % FF_NSLOT) * FF_UNIT)) + w1off;
            auto rd = [&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                lds_read_b128_imm<(n >> 3) * 256>(fr[n % (PF + 1)], tb + w1x[n & 7]);
            };
            if constexpr (MM) ff_static_for(std::make_integer_sequence<int, PF>{}, rd);
            uint32_t hw[8];   // registers 8s .. 8s+7 of the finished tile, GELU'd and packed: the B fragment of k-step s
            // GELU of chunk ph-1 in 11 stages, each cut into two halves of 8 independent operations: ONE half behind EVERY MFMA of
            // the chain (32 issue cycles: the time the next, dependent MFMA has to wait for its accumulator anyway).  The first
            // form put a whole 16-operation stage behind every second MFMA: that pair then cost 64 cycles of VALU issue plus the
            // full dependency stall of the back-to-back MFMA behind it (stamps: 1 989 cycles of work per phase for 768 of MFMA).
            float gv[2][8], gxc[2][8], gu[2][8], gp[2][8];
            if constexpr (PV) {
#pragma unroll
                for (int q = 0; q < 16; ++q) gv[q >> 3][q & 7] = hprev[q];
            }
            ff_static_for(std::make_integer_sequence<int, KSTEPS>{}, [&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                if constexpr (MM) {
                    if constexpr (s + PF < KSTEPS) rd(std::integral_constant<int, s + PF>{});
                    constexpr int younger = s + PF < KSTEPS ? PF : KSTEPS - 1 - s;
                    lgkm_wait_counted<younger>(fr[s % (PF + 1)]);
#ifndef TSIM_FF_NACC
#define TSIM_FF_NACC 2
#endif
                    if constexpr (TSIM_FF_NACC == 2 && (s & 1))
                        hacc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[s % (PF + 1)]), bx[s], hacc1, 0, 0, 0);
                    else
                        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[s % (PF + 1)]), bx[s], hacc, 0, 0, 0);
                    if constexpr (TSIM_FF_SPREAD && (s == 5 || s == 11 || s == 17)) issue_pieces(2 * ph + 4, s / 6, s / 6 + 1);
                }
#ifdef TSIM_FF_GELU_PAIRED   // A/B: the first form
                if constexpr (PV && (s & 1) && s / 2 <= 10) {
                    gelu_stage<s / 2, 8>(gv[0], gxc[0], gu[0], gp[0]);
                    gelu_stage<s / 2, 8>(gv[1], gxc[1], gu[1], gp[1]);
                }
#else
#ifndef TSIM_FF_DIAG_NOGELU   // (defined: TIMING-ONLY, wrong results: the producer without its activation VALU work)
                if constexpr (PV && s / 2 <= 10) gelu_stage<s / 2, 8>(gv[s & 1], gxc[s & 1], gu[s & 1], gp[s & 1]);
#endif
#endif
                if constexpr (PV && s == 23) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) hw[e] = pack_bf16x2(gv[(2 * e) >> 3][(2 * e) & 7], gv[(2 * e + 1) >> 3][(2 * e + 1) & 7]);
                }
            });
            if constexpr (PV) {
                const u32x4 f0 = {hw[0], hw[1], hw[2], hw[3]}, f1 = {hw[4], hw[5], hw[6], hw[7]};
                // slot of chunk ph-1's parity: its previous content (chunk ph-3) was read by the consumer in phase ph-1
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                             ::"v"(slot0 + ((ph - 1) & 1) * 2048), "v"(f0), "v"(f1) : "memory");
            }
            if constexpr (MM) {
                if constexpr (TSIM_FF_NACC == 2) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) hprev[q] = hacc[q] + hacc1[q];
                } else {
                    hprev = hacc;
                }
            }
#ifdef TSIM_PP_STAMPS
            { const unsigned long long t3 = FF_T(); fs_w += t1 - t0; fs_i += t2 - t1; fs_k += t3 - t2; }
#endif
        };
        phase(0, std::true_type{}, std::false_type{});
        for (int ph = 1; ph < nchunks; ++ph) phase(ph, std::true_type{}, std::true_type{});
        phase(nchunks, std::false_type{}, std::true_type{});
        phase(nchunks + 1, std::false_type{}, std::false_type{});      // the consumer's last phase: ring + barrier only
        wait_vmcnt<0>();                                       // past-the-end units must not outlive the workgroup
#ifdef TSIM_PP_STAMPS
        if (threadIdx.x == 0) {
            atomicAdd(&g_ff_stamps[0], (unsigned long long)nphases); atomicAdd(&g_ff_stamps[1], fs_w);
            atomicAdd(&g_ff_stamps[2], fs_i); atomicAdd(&g_ff_stamps[3], fs_k);
        }
#endif
        return;
    }
    // ==================================================================== consumer: FFN2 into the whole row block
    f32x16 y[12];    // y[t][q]: output feature 32 t + (q & 3) + 8 (q >> 2) + 4 hh of token m0 + r; starts from b2
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(b2 + 32 * t + 8 * gq + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[t][4 * gq + e] = bv[e];
        }
#pragma unroll
    for (int t = 0; t < 12; ++t) asm volatile("" : "+v"(y[t]));   // retire the b2 loads before any LDS-DMA is in flight
    issue_unit(0);
    issue_unit(1);
    issue_unit(2);
    const int w2row = r * 64;                                  // + t * 2048 + (((2 s + hh) ^ ((r >> 2) & 3)) << 4)
    const int w2x0 = ((0 + hh) ^ ((r >> 2) & 3)) << 4, w2x1 = ((2 + hh) ^ ((r >> 2) & 3)) << 4;
    const char *hslot0 = xbuf + tg * 4096 + lane * 32;
    [[maybe_unused]] unsigned long long cs_w = 0, cs_i = 0, cs_k = 0;
    for (int ph = 0; ph < nphases; ++ph) {
        [[maybe_unused]] const unsigned long long t0 = FF_T();
        wait_vmcnt<PPU>();
        __builtin_amdgcn_s_barrier();                          // units landed; the producer's h of chunk ph-2 is written
        [[maybe_unused]] const unsigned long long t1 = FF_T();
        issue_unit(2 * ph + 3);
        if (ph < 2 || !TSIM_FF_SPREAD) issue_unit(2 * ph + 4);
        [[maybe_unused]] const unsigned long long t2 = FF_T();
        if (ph >= 2) {
            const uint32_t hs = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)(hslot0 + ((ph - 2) & 1) * 2048));
            lds_u32x4 hv0, hv1;
            lds_read_b128_imm<0>(hv0, hs);
            lds_read_b128_imm<16>(hv1, hs);
            constexpr int PF = 4;
            lds_u32x4 fr[PF + 1];
            const uint32_t un = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)(ring + ((2 * ph + 1) % FF_NSLOT) * FF_UNIT)) + w2row;
            const uint32_t ua0 = un + w2x0, ua1 = un + w2x1;
            auto rd = [&](auto nc) __attribute__((always_inline)) {    // read n: tile n >> 1, k-step n & 1
                constexpr int n = decltype(nc)::value;
                lds_read_b128_imm<(n >> 1) * 2048>(fr[n % (PF + 1)], (n & 1) ? ua1 : ua0);
            };
            ff_static_for(std::make_integer_sequence<int, PF>{}, rd);
            lgkm_wait_counted<PF>(hv0);                        // the two h reads are older than the PF fragment reads
            asm volatile("" : "+v"(hv1));
            const bf16x8 h0 = __builtin_bit_cast(bf16x8, hv0), h1 = __builtin_bit_cast(bf16x8, hv1);
            ff_static_for(std::make_integer_sequence<int, 24>{}, [&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                if constexpr (n + PF < 24) rd(std::integral_constant<int, n + PF>{});
                constexpr int younger = n + PF < 24 ? PF : 23 - n;
                lgkm_wait_counted<younger>(fr[n % (PF + 1)]);
                y[n >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[n % (PF + 1)]), (n & 1) ? h1 : h0,
                                                                   y[n >> 1], 0, 0, 0);
                if constexpr (TSIM_FF_SPREAD && (n == 5 || n == 11 || n == 17)) issue_pieces(2 * ph + 4, n / 6, n / 6 + 1);
            });
        }
#ifdef TSIM_PP_STAMPS
        {
#pragma unroll
            for (int t = 0; t < 12; ++t) asm volatile("" : "+v"(y[t]));
            const unsigned long long t3 = FF_T();
            cs_w += t1 - t0; cs_i += t2 - t1; cs_k += t3 - t2;
        }
#endif
    }
    wait_vmcnt<0>();
#ifdef TSIM_PP_STAMPS
    if (threadIdx.x == 256) {
        atomicAdd(&g_ff_stamps[4], (unsigned long long)nphases); atomicAdd(&g_ff_stamps[5], cs_w);
        atomicAdd(&g_ff_stamps[6], cs_i); atomicAdd(&g_ff_stamps[7], cs_k);
    }
#endif

    // ---------------------------------------------------------------- epilogue: + residual, LayerNorm, store
    const int64_t m = m0 + r;
    const bool live = m < M;
    const int64_t mr = live ? m : M - 1;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2) {
            // the lane's 16 bytes: the 8 features of group gq + hh; the swap turns them back into the accumulator layout
            const uint4 o = *reinterpret_cast<const uint4 *>(X + mr * K + 32 * t + 8 * gq + 8 * hh);
            auto s0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
            const uint32_t a0 = s0[0], c0 = s0[1], a1 = s1[0], c1 = s1[1];
            y[t][4 * gq + 0] += __uint_as_float(a0 << 16);
            y[t][4 * gq + 1] += __uint_as_float(a0 & 0xffff0000u);
            y[t][4 * gq + 2] += __uint_as_float(a1 << 16);
            y[t][4 * gq + 3] += __uint_as_float(a1 & 0xffff0000u);
            y[t][4 * gq + 4] += __uint_as_float(c0 << 16);
            y[t][4 * gq + 5] += __uint_as_float(c0 & 0xffff0000u);
            y[t][4 * gq + 6] += __uint_as_float(c1 << 16);
            y[t][4 * gq + 7] += __uint_as_float(c1 & 0xffff0000u);
        }
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) s1 += y[t][q];
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / 384.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float dlt = y[t][q] - mean;
            s2 = fmaf(dlt, dlt, s2);
        }
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / 384.0f) + eps);
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        uint32_t pk[8];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = 32 * t + 8 * gq + 4 * hh;
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(gamma + n), bv = *reinterpret_cast<const f32x4 *>(beta + n);
            const float o0 = (y[t][4 * gq + 0] - mean) * rstd * gv[0] + bv[0], o1 = (y[t][4 * gq + 1] - mean) * rstd * gv[1] + bv[1];
            const float o2 = (y[t][4 * gq + 2] - mean) * rstd * gv[2] + bv[2], o3 = (y[t][4 * gq + 3] - mean) * rstd * gv[3] + bv[3];
            pk[2 * gq] = pack_bf16x2(o0, o1);
            pk[2 * gq + 1] = pack_bf16x2(o2, o3);
        }
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2) {
            auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gq], pk[2 * gq + 2], false, false);
            auto s1w = __builtin_amdgcn_permlane32_swap(pk[2 * gq + 1], pk[2 * gq + 3], false, false);
            if (live) *reinterpret_cast<uint4 *>(out + m * K + 32 * t + 8 * gq + 8 * hh) = make_uint4(s0[0], s1w[0], s0[1], s1w[1]);
        }
    }
}

// =====================================================================================================
// ln_rows_gemm: out = LayerNorm(X W^T + bias + res) for hidden 384 with 256 tokens per workgroup — the O-projection and FFN2 of
// a MiniLM layer wherever whole rounds of 256-token tiles exist (the rest of the tokens goes through gemm_bf16_kernel<128 / 32,
// 384, ..>, whose rows carry the same bits: same k order, same MFMA shape, accumulators from the bias, the same statistics).
//
// Why (round 3).  LDS is the binding resource of the LayerNorm GEMM: an LDS-DMA write moves 64 B per LDS cycle, a ds_read_b128
// 256 B, and at BM = 128, BK = 64 a k-tile stages 64 KiB (1 024 LDS cycles) and is read 160 KiB (640) for 1 536 cycles of MFMA
// per SIMD — measured: staging alone 71 of 80 us (profiles/README.md).  The W tile is 3/4 of the staged bytes and is re-staged for
// every token tile, so the tile must hold more tokens; gemm_bf16_kernel's 2 x 4 wave grid at BM = 256 needs 192 accumulator
// registers beside ~110 others and spills (measured 115 us per launch instead of 80).  Here every WAVE owns 32 token rows and ALL
// 384 features (12 accumulator tiles = 192 VGPRs, the layout of ffn_fused_kernel's consumer): LayerNorm statistics never leave the
// wave — no barrier, no LDS round trip, no parameter staging in the epilogue — and the k loop is the asm-pipelined form of the
// other kernels: 32-k tiles (16 KiB of X + 24 KiB of W as packed images) through a three-slot LDS-DMA ring with counted vmcnt (two
// tiles in flight across the barrier), fragment reads rolling four ahead with counted lgkmcnt, 24 MFMAs per wave and tile.
// Per k-tile: 40 KiB staged (640 LDS cycles) + 208 KiB read (832) for 1 536 MFMA cycles per SIMD.
// =====================================================================================================
// NW = waves = 32-token row blocks per workgroup, NST = ring slots.  <8, 3>: the main form (256 tokens, 120 KiB).  <2, 5>: the
// REMAINDER form — 64 tokens per workgroup, and since such a launch has far fewer workgroups than CUs, a five-slot ring (140 KiB)
// that keeps three tiles in flight: a remainder's cost is its k loop's LATENCY (k-tiles x time per tile, whatever the token
// count), and an LDS-DMA tile takes ~1.1 us from issue to landing.  Every wave runs the same instruction stream on its rows in
// both forms: same bits.
constexpr int LR_BK = 32, LR_WBYTES = 384 * LR_BK * 2;

// STAG (stagger): an LDS-DMA instruction costs the ISSUING wave 60-185 cycles (five pieces per wave and tile: ~650 cycles
// against 768 of MFMA), and when all eight waves issue right behind the barrier no MFMA runs anywhere on the CU meanwhile
// (measured: 2 700 cycles per tile for 1 536 of matrix work).  With STAG the second half of the waves — the SIMD partners of
// the first half — issue their pieces AFTER their MFMAs: on every SIMD one wave feeds the matrix pipe while the other queues
// at the address path.  Their tile then has one tile period less to land, hence one ring slot more (four: exactly 160 KiB).
// Measured EQUAL to the plain three-slot form (profiles/README.md, round 3) and left as an opt-in (TSIM_LN_ROWS_STAG=1).
template <int NW, int LR_NST, bool STAG = false>
__global__ __launch_bounds__(NW * 64) void ln_rows_gemm_kernel(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wimg,
                                                               const float *__restrict__ bias, const bf16_t *__restrict__ res,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               float eps, bf16_t *__restrict__ out, int M, int K, int xpacked) {
    // xpacked: X is in the block-packed layout (packed_off): FFN2's operand h1
    constexpr int N = 384, XBYTES = NW * 32 * LR_BK * 2, STAGE = XBYTES + LR_WBYTES;
    constexpr int XP = XBYTES / 1024, PIECES = STAGE / 1024, PPW = PIECES / NW;   // pieces 0..XP-1: X, the rest: W
    static_assert(PIECES % NW == 0 && XP == 2 * NW, "every wave issues two X pieces and PPW - 2 W pieces per k-tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * (NW * 32);
    const int nk = K / LR_BK;

    // accumulators start from the bias: acc[t][q] = output feature 32 t + (q & 3) + 8 (q >> 2) + 4 h of token m0 + 32 wave + r
    f32x16 acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + 32 * t + 8 * gq + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * gq + e] = bv[e];
        }
    // LDS image of the X region (as gemm_bf16_kernel at BK = 32): rows of 64 B, four to a 256-byte super-row; 16-byte slot (sr, chp)
    // holds chunk ch = chp ^ (sr & 15) of the super-row = chunk ch & 3 of row 4 sr + (ch >> 2).  The permutation goes on the SOURCE
    // address (LDS-DMA writes lane-linearly).  The W region is the tile's 24 fragment blocks (pack_frag_w_kernel), copied verbatim.
    const char *xsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = (wave + i * NW) * 64 + lane;
        const int sr = sl >> 4, ch = (sl & 15) ^ (sr & 15);
        const int row = sr * 4 + (ch >> 2);
        const int64_t m = m0 + row < M ? m0 + row : M - 1;     // rows past M repeat the last one (never stored)
        xsrc[i] = xpacked ? reinterpret_cast<const char *>(X + packed_off(m, (ch & 3) * 8, K))
                          : reinterpret_cast<const char *>(X) + m * K * 2 + (ch & 3) * 16;
    }
    const int xkstride = xpacked ? (LR_BK / 8) * 512 : LR_BK * 2;   // bytes per k-tile along a row
    const char *wsrc = reinterpret_cast<const char *>(Wimg) + lane * 16;
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
        const int k2 = kt < nk ? kt : nk - 1;                  // past-the-end: re-read the last tile (uniform vmcnt)
        char *dst = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(xsrc[i] + k2 * xkstride, dst + (wave + i * NW) * 1024);
#pragma unroll
        for (int i = 2; i < PPW; ++i)
            glds16(wsrc + (int64_t)k2 * LR_WBYTES + (wave + i * NW - XP) * 1024, dst + (wave + i * NW) * 1024);
    };
    // fragment read addresses inside a stage.  X: row rho = 32 wave + r, k-step s, lane half h -> super-row rho >> 2, chunk
    // (rho & 3) * 4 + 2 s + h.  W: block 12 s + t of the region, this lane's 16 bytes: one address register and an immediate.
    const uint32_t lbase = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
    uint32_t xo[2];
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
        const int ch = (r & 3) * 4 + 2 * sk + h;
        xo[sk] = lbase + wave * 2048 + (r >> 2) * 256 + ((ch ^ ((((wave & 1) << 3) + (r >> 2)) & 15)) << 4);
    }
    const uint32_t wl = lbase + XBYTES + lane * 16;

    static_assert(LR_NST >= 3 && (LR_NST - 2) * PPW <= 63, "ring depth");
#pragma unroll
    for (int i = 0; i < LR_NST - 1; ++i) issue(i, i);
#pragma unroll
    for (int t = 0; t < 12; ++t) asm volatile("" : "+v"(acc[t]));   // retire the bias loads here (and with them the first tiles)
    auto do_tile = [&](int kt, auto stc) __attribute__((always_inline)) {
        constexpr int stage = decltype(stc)::value;
        wait_vmcnt<(LR_NST - 2) * PPW>();  // my pieces of tile kt (those of kt + 1 .. kt + NST - 2 stay in flight)
        __builtin_amdgcn_s_barrier();      // everyone's landed; everyone is past tile kt - 1
        const bool late = STAG && wave >= NW / 2;
        if (!late) issue(kt + LR_NST - 1, (stage + LR_NST - 1) % LR_NST);
        constexpr int PF = 4, NRD = 24;
        lds_u32x4 bfr[2], fr[PF + 1];
        const uint32_t so = stage * STAGE;
        lds_read_b128_imm<0>(bfr[0], xo[0] + so);
        lds_read_b128_imm<0>(bfr[1], xo[1] + so);
        const uint32_t wa = wl + so;
        auto rd = [&](auto nc) __attribute__((always_inline)) {            // read n: k-step n / 12, tile n % 12 = block n
            constexpr int n = decltype(nc)::value;
            lds_read_b128_imm<n * 1024>(fr[n % (PF + 1)], wa);
        };
        ff_static_for(std::make_integer_sequence<int, PF>{}, rd);
        ff_static_for(std::make_integer_sequence<int, NRD>{}, [&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n + PF < NRD) rd(std::integral_constant<int, n + PF>{});
            constexpr int younger = n + PF < NRD ? PF : NRD - 1 - n;
            lgkm_wait_counted<younger>(fr[n % (PF + 1)]);                  // ... and everything older: both X fragments
            if constexpr (n == 0) { asm volatile("" : "+v"(bfr[0])); asm volatile("" : "+v"(bfr[1])); }
            acc[n % 12] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[n % (PF + 1)]),
                                                                   __builtin_bit_cast(bf16x8, bfr[n / 12]), acc[n % 12], 0, 0, 0);
        });
        if (late) {
#pragma unroll
            for (int t = 0; t < 12; ++t) asm volatile("" : "+v"(acc[t]));   // keep the issue behind the MFMA stream
            issue(kt + LR_NST - 1, (stage + LR_NST - 1) % LR_NST);
        }
    };
    int kt = 0;
    for (; kt + LR_NST <= nk; kt += LR_NST)
        ff_static_for(std::make_integer_sequence<int, LR_NST>{}, [&](auto sc) __attribute__((always_inline)) {
            do_tile(kt + decltype(sc)::value, sc);
        });
    ff_static_for(std::make_integer_sequence<int, LR_NST - 1>{}, [&](auto sc) __attribute__((always_inline)) {
        if (kt + decltype(sc)::value < nk) do_tile(kt + decltype(sc)::value, sc);
    });
    wait_vmcnt<0>();   // past-the-end tiles must not outlive the workgroup

    // ---------------------------------------------------------------- epilogue: + residual, LayerNorm, store (wave-local)
    // + residual: the lane's 16 bytes are the 8 features of group gq + h; the swap turns them into the accumulator layout
    const int64_t m = m0 + wave * 32 + r;
    const bool live = m < M;
    const int64_t mr = live ? m : M - 1;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2) {
            const uint4 o = *reinterpret_cast<const uint4 *>(res + mr * N + 32 * t + 8 * gq + 8 * h);
            auto s0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
            const uint32_t a0 = s0[0], c0 = s0[1], a1 = s1[0], c1 = s1[1];
            acc[t][4 * gq + 0] += __uint_as_float(a0 << 16);
            acc[t][4 * gq + 1] += __uint_as_float(a0 & 0xffff0000u);
            acc[t][4 * gq + 2] += __uint_as_float(a1 << 16);
            acc[t][4 * gq + 3] += __uint_as_float(a1 & 0xffff0000u);
            acc[t][4 * gq + 4] += __uint_as_float(c0 << 16);
            acc[t][4 * gq + 5] += __uint_as_float(c0 & 0xffff0000u);
            acc[t][4 * gq + 6] += __uint_as_float(c1 << 16);
            acc[t][4 * gq + 7] += __uint_as_float(c1 & 0xffff0000u);
        }
    // two-pass statistics in the association order of gemm_bf16_kernel<.., 384, .., WAVES_N = 4, ..>: four groups of three
    // sub-tiles (there: one wave each), each summed tile by tile, register by register, then across the half-waves, then
    // ((0 + g0) + g1) + g2) + g3
    float mean, rstd;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float sg = 0.f;
#pragma unroll
            for (int t = 3 * w; t < 3 * w + 3; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (pass == 0) {
                        sg += acc[t][q];
                    } else {
                        const float dlt = acc[t][q] - mean;
                        sg = fmaf(dlt, dlt, sg);
                    }
                }
            sg += __shfl_xor(sg, 32, 64);
            tot += sg;
        }
        if (pass == 0)
            mean = tot / (float)N;
        else
            rstd = 1.0f / sqrtf(tot / (float)N + eps);
    }
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        uint32_t pk[8];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = 32 * t + 8 * gq + 4 * h;
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(gamma + n), bv = *reinterpret_cast<const f32x4 *>(beta + n);
            const float o0 = fmaf((acc[t][4 * gq + 0] - mean) * rstd, gv[0], bv[0]);
            const float o1 = fmaf((acc[t][4 * gq + 1] - mean) * rstd, gv[1], bv[1]);
            const float o2 = fmaf((acc[t][4 * gq + 2] - mean) * rstd, gv[2], bv[2]);
            const float o3 = fmaf((acc[t][4 * gq + 3] - mean) * rstd, gv[3], bv[3]);
            pk[2 * gq] = pack_bf16x2(o0, o1);
            pk[2 * gq + 1] = pack_bf16x2(o2, o3);
        }
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2) {
            auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gq], pk[2 * gq + 2], false, false);
            auto s1w = __builtin_amdgcn_permlane32_swap(pk[2 * gq + 1], pk[2 * gq + 3], false, false);
            if (live) *reinterpret_cast<uint4 *>(out + m * N + 32 * t + 8 * gq + 8 * h) = make_uint4(s0[0], s1w[0], s0[1], s1w[1]);
        }
    }
}

// =====================================================================================================
// attention on packed tokens (MFMA).  One wave per (sequence, head, block of 32 queries); a workgroup = 4 heads.
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16 with A = K rows, B = Q rows: the QUERY lands on the lane (column), 16 keys
//                 in the accumulator registers -> softmax statistics are lane-local plus one cross-half shuffle;
//   O^T = V^T P^T the score accumulator, converted pairwise to bf16, IS the B operand of the second product
//                 (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand"); the k order inside a
//                 k-step is then key = 16s + 8(j>>2) + 4h + (j&3), and the V^T fragment is gathered in that order.
// fp32 online softmax over key blocks of 32 (scores = q.k/sqrt(dh) [+ MPNet relative-position bias]).  No mask is
// needed: the packed layout holds valid tokens only, which equals the reference's additive (1-m)*-10000 mask in fp32
// (exp underflows to exactly 0).  Operands come straight from HBM/L2 (each byte of qkv is read once per query block);
// FLOPs are ~S/(6H) of the layer's (<1 % at the benchmark's 16-token mean length): HBM/latency-bound.
// =====================================================================================================
template <int DH, bool REL, bool PK = false>   // PK: qkv is in the block-packed layout (packed_off); ctx stays row-major
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH <= 32 ? 6 : 4))) void attention_kernel(const bf16_t *__restrict__ qkv,
                                                        const int32_t *__restrict__ cu, const int32_t *__restrict__ col,
                                                        const float *__restrict__ relb, int relw, int H,
                                                        int heads, float scale, bf16_t *__restrict__ ctx, int B, int xcd_map) {
    constexpr int KS = DH / 16;            // k-steps of the QK^T product
    constexpr int OT = (DH + 31) / 32;     // 32-row blocks of O^T
    constexpr int ROWB = DH * 2 < 64 ? 64 : DH * 2;   // bytes per key row of the V image (>= 32 dims, so block reads stay inside)
    __shared__ __attribute__((aligned(16))) char vimg[4 * 32 * ROWB];
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: workgroup b runs on XCD b % 8, so sequence = (b % 8) * ceil(B / 8) + b / 8 gives every XCD one contiguous
    // range of sequences: neighbours in the packed token axis share cache lines (a sequence's slice of a 512-byte chunk or of a
    // row rarely starts on a line boundary), and with the plain order the two halves of such a line were fetched by two L2s
    int seq = blockIdx.x;
    if (xcd_map) {
        seq = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (seq >= B) return;
    }
    const int head = blockIdx.y * 4 + wave, qb = blockIdx.z;
    const int t0 = cu[seq], S = cu[seq + 1] - t0;
    if (qb * 32 >= S || head >= heads) return;  // wave-uniform
    const int r = lane & 31, h = lane >> 5;
    const int64_t H3 = 3 * (int64_t)H;
    const int qi = qb * 32 + r;
    const int tq = t0 + (qi < S ? qi : S - 1);

    bf16x8 qf[KS];
    {
        // element (token, feature) of qkv; every access below is 8 features (16 bytes) at a feature offset that is a multiple of 8
        const bf16_t *qp = PK ? qkv + packed_off(tq, head * DH + 8 * h, (int)H3) : qkv + tq * H3 + head * DH + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8 *>(qp + (PK ? 2 * 256 : 16) * s);   // +16 features = two groups of 8
        }
    }
    const int qcol = REL ? col[tq] : 0;
    f32x16 o[OT];
#pragma unroll
    for (int ob = 0; ob < OT; ++ob)
#pragma unroll
        for (int g = 0; g < 16; ++g) o[ob][g] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < S; k0 += 32) {
        const int kr = k0 + r;
        const int tk = t0 + (kr < S ? kr : S - 1);
        f32x16 sc;
#pragma unroll
        for (int g = 0; g < 16; ++g) sc[g] = 0.f;
        // the key block's V rows are requested together with its K rows (measured equal to requesting them after the softmax)
        constexpr int CH = DH / 16;                          // 16-byte chunks per half row of V
        uint4 vr[CH];
        {
            const bf16_t *kp = PK ? qkv + packed_off(tk, H + head * DH + 8 * h, (int)H3) : qkv + tk * H3 + H + head * DH + 8 * h;
            const bf16_t *vsrc = PK ? qkv + packed_off(tk, 2 * H + head * DH + h * (DH / 2), (int)H3)
                                    : qkv + tk * H3 + 2 * H + head * DH + h * (DH / 2);
            bf16x8 kf[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const bf16x8 *>(kp + (PK ? 2 * 256 : 16) * s);
#pragma unroll
            for (int c = 0; c < CH; ++c) vr[c] = *reinterpret_cast<const uint4 *>(vsrc + (PK ? 256 : 8) * c);
#pragma unroll
            for (int s = 0; s < KS; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sc, 0, 0, 0);
        }
        // sc[g]: key k0 + (g&3) + 8(g>>2) + 4h, query qi
        float bm = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int key = k0 + (g & 3) + 8 * (g >> 2) + 4 * h;
            float v = sc[g] * scale;
            if (REL) {
                const int kc = col[t0 + (key < S ? key : S - 1)];
                v += relb[head * relw + (kc - qcol) + (relw >> 1)];
            }
            v = key < S ? v : -INFINITY;
            sc[g] = v;
            bm = fmaxf(bm, v);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        const float corr = __expf(m - mn);     // first block: exp(-inf) = 0
        float ps = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            sc[g] = __expf(sc[g] - mn);
            ps += sc[g];
        }
        ps += __shfl_xor(ps, 32, 64);
        l = l * corr + ps;
        m = mn;
#pragma unroll
        for (int ob = 0; ob < OT; ++ob)
#pragma unroll
            for (int g = 0; g < 16; ++g) o[ob][g] *= corr;
        // P^T fragments: k-step s takes accumulator registers 8s..8s+7
        bf16x8 pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s][j] = (__bf16)sc[8 * s + j];
        // V^T fragments in the matching k order.  V rows (keys) are row-major in memory; the MFMA wants, per lane, 8 keys of one
        // head dimension.  Each lane loads half a V row with 16-byte loads into a wave-private [32 keys][ROWB] LDS image and
        // ds_read_b64_tr_b16 hands every lane 4 keys of its dimension (a 16-lane group reads a 4-key x 16-dim block column-
        // major; lane 4q+p of the group supplies the address of key q, dims 4p..4p+3): 2-byte gathers from global memory
        // (16 load instructions per k-step pair) were the kernel's largest cost.  EXEC is full here (out-of-range keys and
        // queries are clamped, not masked), as the instruction requires.
        {
            char *vrow = vimg + wave * (32 * ROWB) + r * ROWB + h * DH;
#pragma unroll
            for (int c = 0; c < CH; ++c) *reinterpret_cast<uint4 *>(vrow + 16 * c) = vr[c];
        }
        __builtin_amdgcn_wave_barrier();                     // wave-private image: LDS ops of one wave execute in order
#pragma unroll
        for (int ob = 0; ob < OT; ++ob) {
            const bool iv = ob * 32 + r < DH;
            const int grp16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 vf;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int key = 16 * s + 8 * half + 4 * h + q4;              // row of the block this lane addresses
                    const char *ad = vimg + wave * (32 * ROWB) + key * ROWB + (ob * 32 + grp16 * 16 + 4 * p4) * 2;
                    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)ad);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        vf[4 * half + e] = __builtin_bit_cast(__bf16, (bf16_t)(iv ? (bf16_t)t[e] : (bf16_t)0));
                }
                o[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], o[ob], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();                     // the next key block overwrites the image
    }
    if (qi < S) {
        const float inv = 1.0f / l;
        bf16_t *op = ctx + (int64_t)(t0 + qi) * H + head * DH + 4 * h;
#pragma unroll
        for (int ob = 0; ob < OT; ++ob)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int i = ob * 32 + 8 * gq;  // + 4h + (0..3)
                if (i + 4 * h < DH) {
                    uint2 w;
                    w.x = pack_bf16x2(o[ob][4 * gq] * inv, o[ob][4 * gq + 1] * inv);
                    w.y = pack_bf16x2(o[ob][4 * gq + 2] * inv, o[ob][4 * gq + 3] * inv);
                    *reinterpret_cast<uint2 *>(op + i) = w;
                }
            }
    }
}

// =====================================================================================================
// masked mean-pool over the packed layout (A4) + optional fused L2-normalise to half precision (search operand).
// One wave per sequence: pooled[b] = sum_t x[t] / max(len, 1e-9).  HBM-bound: T*H*2 B in.
// A lane owns 16-byte pieces of the row (features 8c .. 8c+7 for c = lane, lane + 64): one 16-byte load per token and
// piece (the first form read 2 bytes per lane and load: 48 us for 52 MB).  Per feature the sum still runs over the
// tokens in ascending order in float32, and the squared norm is taken in the canonical element order of
// tsim_l2norm_rows (element j on lane j % 64, ascending, then the xor butterfly) from a wave-private LDS copy of the
// pooled row, so pooled and unit rows keep their bits.
// =====================================================================================================
template <int NP>   // 16-byte pieces per lane = ceil(H / 512)
__global__ __launch_bounds__(256) void pool_packed_kernel(const bf16_t *__restrict__ x,
                                                          const int32_t *__restrict__ cu, int B, int H,
                                                          float *__restrict__ pooled, unit_t *__restrict__ unit,
                                                          int ld_unit, float *__restrict__ rho_max) {
    __shared__ __attribute__((aligned(16))) float rows[4][NP * 512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;
    const int t0 = cu[b], t1 = cu[b + 1];
    float s[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[p][e] = 0.f;
    // the loop is latency-bound (one dependent-free load per token): four tokens' loads are issued together, the adds stay in
    // token order
    auto add_row = [&](int p, const uint4 &v) __attribute__((always_inline)) {
        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[p][2 * q] += __uint_as_float(wv[q] << 16);
            s[p][2 * q + 1] += __uint_as_float(wv[q] & 0xffff0000u);
        }
    };
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int f0 = (lane + 64 * p) * 8;
        if (f0 >= H) continue;
        const bf16_t *xp = x + f0;
        // eight tokens per round trip, the ragged end included: a token past the end re-reads the last row (no branch around a
        // load: that would put a full wait behind it) and adds zeros — most sequences of the benchmark are one or two round trips
        for (int t = t0; t < t1; t += 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4 *>(xp + (int64_t)(t + u < t1 ? t + u : t1 - 1) * H);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t + u >= t1) v[u] = make_uint4(0u, 0u, 0u, 0u);
                add_row(p, v[u]);
            }
        }
    }
    const float den = fmaxf((float)(t1 - t0), 1e-9f);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int f0 = (lane + 64 * p) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[p][e] = s[p][e] / den;
        if (f0 < H) {
            const float4 lo = make_float4(s[p][0], s[p][1], s[p][2], s[p][3]), hi = make_float4(s[p][4], s[p][5], s[p][6], s[p][7]);
            if (pooled) {
                *reinterpret_cast<float4 *>(pooled + (int64_t)b * H + f0) = lo;
                *reinterpret_cast<float4 *>(pooled + (int64_t)b * H + f0 + 4) = hi;
            }
            *reinterpret_cast<float4 *>(&rows[w][f0]) = lo;
            *reinterpret_cast<float4 *>(&rows[w][f0 + 4]) = hi;
        }
    }
    if (unit) {  // identical bits to tsim_l2norm_rows(pooled): same element order, same float64 scale
        __builtin_amdgcn_wave_barrier();   // wave-private row: LDS operations of one wave execute in order
        double ss = 0.0;
        for (int jx = lane; jx < H; jx += 64) {
            const float v = rows[w][jx];
            ss = fma((double)v, (double)v, ss);
        }
        const double inv = canonical_inv_norm(ss, 1e-8f);
        double r2 = 0.0;   // squared rounding residual of the row (the search guard's rho, see tsim_l2norm_rows)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int f0 = (lane + 64 * p) * 8;
            if (f0 < H) {
                f16x8 u;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    u[e] = canonical_unit_elem(s[p][e], inv);
                    const double de = (double)(float)u[e] - (double)s[p][e] * inv;
                    r2 = fma(de, de, r2);
                }
                *reinterpret_cast<f16x8 *>(unit + (int64_t)b * ld_unit + f0) = u;
            }
        }
        for (int jx = H + lane; jx < ld_unit; jx += 64) unit[(int64_t)b * ld_unit + jx] = 0;
        if (rho_max) {
            const float rho = rho_round_up(sqrt(wave_sum_f64(r2)));
            if (lane == 0) rho_publish(rho_max, rho);
        }
    }
}

// =====================================================================================================
// host side
// =====================================================================================================
static uint16_t host_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace tsim

using namespace tsim;

struct tsim_encoder {
    tsim_encoder_config cfg;
    int Tp = 0;
    // weights
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr, *relb = nullptr;
    int relw = 0;
    struct Layer {
        bf16_t *wqkv, *wo, *w1, *w2;
        bf16_t *pqkv = nullptr, *po = nullptr, *p1 = nullptr, *p2 = nullptr;     // tile-major copies for the ping-pong GEMM
        bf16_t *lo = nullptr, *l2 = nullptr;   // H = 384: O-proj / FFN2 weights as the k-tile LDS images of the LayerNorm GEMM
        bf16_t *lo32 = nullptr, *l232 = nullptr;   // ... and as MFMA fragment images (pack_frag_w_kernel) for ln_rows_gemm / ln_tail_gemm
        uint8_t *qqkv = nullptr, *qo = nullptr, *q1 = nullptr, *q2 = nullptr;   // MXFP8 weights: e4m3 bytes (tile-major) ...
        uint8_t *sqkv = nullptr, *so = nullptr, *s1 = nullptr, *s2 = nullptr;   // ... and E8M0 block scales [out, in/32]
        float *bqkv, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
    };
    std::vector<Layer> layers;
    std::vector<void *> allocs;
    // activations
    bf16_t *x0 = nullptr, *x1 = nullptr, *qkv = nullptr, *ctx = nullptr, *h1 = nullptr;
    float *ybuf = nullptr;   // fp32 pre-LayerNorm sums (wide models only)
    uint8_t *aq = nullptr, *as = nullptr, *hq = nullptr, *hs = nullptr;   // MXFP8 images of a projection's input ([Tp,H] / [Tp,F])
    int *err_flags = nullptr;   // TSIM_ENC_ERR_* bits raised by kernels since the last tsim_encoder_error_flags
};

namespace tsim {

static int dev_alloc(tsim_encoder *e, size_t bytes, void **out) {
    void *p = nullptr;
    hipError_t err = hipMalloc(&p, bytes);
    if (err != hipSuccess) return fail(TSIM_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    e->allocs.push_back(p);
    *out = p;
    return TSIM_OK;
}

static int upload_f32(tsim_encoder *e, const float *h, size_t n, float **out) {
    int rc = dev_alloc(e, n * 4, (void **)out);
    if (rc) return rc;
    TSIM_HIP_CHECK(hipMemcpy(*out, h, n * 4, hipMemcpyHostToDevice));
    return TSIM_OK;
}

static int upload_bf16(tsim_encoder *e, const std::vector<const float *> &parts, size_t n_each, bf16_t **out) {
    std::vector<uint16_t> tmp(n_each * parts.size());
    for (size_t p = 0; p < parts.size(); ++p)
        for (size_t i = 0; i < n_each; ++i) tmp[p * n_each + i] = host_f32_to_bf16(parts[p][i]);
    int rc = dev_alloc(e, tmp.size() * 2, (void **)out);
    if (rc) return rc;
    TSIM_HIP_CHECK(hipMemcpy(*out, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    return TSIM_OK;
}

// MPNet relative_position_bucket (transformers mpnet/modeling_mpnet.py relative_position_bucket):
// rel = key_col - query_col, n = -rel.
// fp32 -> e4m3 byte, nearest even, |v| <= 448 (oracle/fp8_ref.f32_to_e4m3)
static uint8_t host_f32_to_e4m3(float v) {
    const uint8_t sign = std::signbit(v) ? 0x80 : 0;
    const double a = std::fabs((double)v);
    if (a == 0.0) return sign;
    int exp;
    const double mant = std::frexp(a, &exp);          // a = mant * 2^exp, mant in [0.5, 1)
    long byte;
    if (a >= std::ldexp(1.0, -6))
        byte = ((long)(exp - 1 + 7) << 3) + std::lrint((mant * 2.0 - 1.0) * 8.0);   // a carry of 8 bumps the exponent
    else
        byte = std::lrint(std::ldexp(a, 9));
    if (byte > 0x7e) byte = 0x7e;
    return (uint8_t)byte | sign;
}

// parts: row blocks of n_rows x K fp32 (nn.Linear [out, in]) -> MXFP8 along K (oracle/fp8_ref.mx_quantize)
static int upload_mxfp8(tsim_encoder *e, const std::vector<const float *> &parts, size_t n_rows, size_t K, uint8_t **q,
                        uint8_t **sc) {
    const size_t rows = n_rows * parts.size(), nb = K / 32;
    std::vector<uint8_t> hq(rows * K), hs(rows * nb);
    for (size_t p = 0; p < parts.size(); ++p)
        for (size_t r = 0; r < n_rows; ++r)
            for (size_t b = 0; b < nb; ++b) {
                const float *src = parts[p] + r * K + b * 32;
                float amax = 0.f;
                for (int i = 0; i < 32; ++i) amax = std::fmax(amax, std::fabs(src[i]));
                int sexp = 0;
                if (amax > 0.f) {
                    int ex;
                    (void)std::frexp((double)amax, &ex);
                    sexp = ex - 1 - 8;
                    sexp = sexp < -127 ? -127 : (sexp > 127 ? 127 : sexp);
                }
                uint8_t *dst = hq.data() + (p * n_rows + r) * K + b * 32;
                for (int i = 0; i < 32; ++i) {
                    double y = std::ldexp((double)src[i], -sexp);
                    y = y > 448.0 ? 448.0 : (y < -448.0 ? -448.0 : y);
                    dst[i] = host_f32_to_e4m3((float)y);
                }
                hs[(p * n_rows + r) * nb + b] = (uint8_t)(sexp + 127);
            }
    int rc = dev_alloc(e, hq.size(), (void **)q);
    if (rc) return rc;
    if ((rc = dev_alloc(e, hs.size(), (void **)sc))) return rc;
    TSIM_HIP_CHECK(hipMemcpy(*q, hq.data(), hq.size(), hipMemcpyHostToDevice));
    TSIM_HIP_CHECK(hipMemcpy(*sc, hs.data(), hs.size(), hipMemcpyHostToDevice));
    return TSIM_OK;
}

static int mpnet_bucket(int rel, int num_buckets) {
    int ret = 0;
    int n = -rel;
    const int nb = num_buckets / 2;
    if (n < 0) {
        ret += nb;
        n = -n;
    }
    const int max_exact = nb / 2;
    if (n < max_exact) return ret + n;
    // float32 arithmetic like torch: log(n / max_exact) / log(128 / max_exact) * (nb - max_exact)
    const float v = logf((float)n / (float)max_exact) / (float)log(128.0 / max_exact) * (float)(nb - max_exact);
    int large = max_exact + (int)v;
    if (large > nb - 1) large = nb - 1;
    return ret + large;
}

template <int BM, int BN, int BK, int WM, int WN, int EPI, int NST = 2>
static int launch_gemm(const bf16_t *X, const bf16_t *W, const float *bias, const bf16_t *res, const float *gamma,
                       const float *beta, float eps, bf16_t *out, int M, int N, int K, hipStream_t st,
                       const bf16_t *Wimg = nullptr, bool xpacked = false) {
    constexpr int lds = gemm_lds_bytes<BM, BN, BK, WM, WN, NST>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = gemm_bf16_kernel<BM, BN, BK, WM, WN, EPI, NST>;
    static DevOnce lds_once;
    TSIM_MAX_LDS(lds_once, kern, lds);
    if (N % BN != 0 || K % BK != 0) return fail(TSIM_EUNSUPPORTED, "gemm: N=%d K=%d not tileable by %dx%d", N, K, BN, BK);
    const int mtiles = (M + BM - 1) / BM, ntiles = N / BN;
    const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
    if (Wimg && ntiles != 1) return fail(TSIM_EINVAL, "gemm: a packed W image needs BN == N");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, st, X, W, bias, res, gamma, beta, eps, out, M, N, K,
                       mtiles, ntiles, Wimg, xpacked ? 1 : 0);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

template <int EPI, int NW>
static int gemm_xres_nw(const bf16_t *X, const bf16_t *W, const float *bias, bf16_t *out, int M, int N, hipStream_t st) {
    constexpr int lds = XR_NSTAGE * XR_BN * XR_BK * 2 + NW * 32 * XR_STG_ROW + 8192;   // ring | output image | bias (N <= 2048)
    if (N > 2048) return fail(TSIM_EUNSUPPORTED, "gemm_xres: N=%d > 2048", N);
    auto kern = gemm_xres_kernel<384, EPI, NW>;
    static DevOnce lds_once;
    TSIM_MAX_LDS(lds_once, kern, lds);
    const int items = ((M + NW * 32 - 1) / (NW * 32)) * (N / XR_BN);
    const int slots = 256 * (8 / NW);             // persistent workgroups: one (8 waves) or two (4 waves) per CU
    const int grid = items < slots ? items : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, X, W, bias, out, M, N, items);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

template <int EPI, bool PK>
static int gemm_xres2(const bf16_t *X, const bf16_t *W, const float *bias, bf16_t *out, int M, int N, hipStream_t st) {
    constexpr int lds = X2_NSTAGE * X2_BN * X2_BK * 2 + 8192;   // ring | bias (N <= 2048)
    if (N > 2048) return fail(TSIM_EUNSUPPORTED, "gemm_xres2: N=%d > 2048", N);
    if (((int64_t)M + 256) * N * 2 >= (1ll << 32)) return fail(TSIM_EUNSUPPORTED, "gemm_xres2: output of %d x %d exceeds 4 GiB", M, N);
    auto kern = gemm_xres2_kernel<EPI, PK>;
    static DevOnce lds_once;
    TSIM_MAX_LDS(lds_once, kern, lds);
    const int items = ((M + 255) / 256) * (N / X2_BN);
    const int grid = items < 256 ? items : 256;             // persistent workgroups, one per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, X, W, bias, out, M, N, items);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

template <int EPI>
static int gemm_xres(const bf16_t *X, const bf16_t *W, const float *bias, bf16_t *out, int M, int N, hipStream_t st) {
    static int v2 = -1;
    if (v2 < 0) { const char *e = getenv("TSIM_XRES2"); v2 = e ? atoi(e) : 1; }
    if (v2 && N % X2_BN == 0) return gemm_xres2<EPI, false>(X, W, bias, out, M, N, st);
    return gemm_xres_nw<EPI, 8>(X, W, bias, out, M, N, st);   // 8 waves = 256 tokens per workgroup (4-wave groups measured slower)
}

static int ffn_fused(const bf16_t *X, const bf16_t *W1img, const bf16_t *W2img, const float *b1, const float *b2,
                     const float *gamma, const float *beta, float eps, bf16_t *out, int M, int F, hipStream_t st) {
    const int lds = FF_NSLOT * FF_UNIT + FF_XB + ((F * 4 + 1023) / 1024) * 1024;   // ring | h exchange | b1
    if (lds > 160 * 1024) return fail(TSIM_EUNSUPPORTED, "ffn_fused: F=%d needs %d B of LDS", F, lds);
    static DevOnce lds_once;
    TSIM_MAX_LDS(lds_once, ffn_fused_kernel, 160 * 1024);
    hipLaunchKernelGGL(ffn_fused_kernel, dim3((unsigned)((M + 127) / 128)), dim3(512), lds, st, X, W1img, W2img, b1, b2, gamma,
                       beta, eps, out, M, F);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

// =====================================================================================================
// ln_tail_gemm: the LayerNorm GEMM of width 384 for FEW rows (the remainder behind ln_rows_gemm's whole rounds, and small batches).
//
// A remainder launch has far fewer workgroups than CUs, and every workgroup has to see ALL of W whatever its token count: its cost is
// the latency of its k loop.  Through an LDS ring that loop paid, per 64-k tile, a workgroup barrier, seven LDS-DMA issues per wave
// (~100-150 cycles each) and the DMA's own latency for twelve MFMAs per wave: 1.7 us per tile, 40 us for FFN2's 24 tiles and 1 657
// rows (against 90 us for the 65 536 rows of the main launch) — 13 % of a MiniLM forward for 2.5 % of its rows.
// Here nothing is shared, so nothing goes through LDS: a workgroup is 32 token rows x 4 waves, wave w owns features 96 w .. + 95
// (three accumulator tiles) and streams ITS slice of W straight from L2 into MFMA A fragments, and the token rows likewise into B
// fragments (the four waves read the same rows: L1 hits).  W is stored as MFMA fragment blocks (pack_frag_w_kernel): a wave's load
// is one contiguous KiB — and the image is the one ln_rows_gemm has just streamed through every XCD's L2.  (Reading a swizzled
// LDS image instead, 16-byte chunks scattered over 2 KiB per load, measured 38 us against 27 for FFN2's remainder.)  Loads roll PF
// k-steps ahead of their MFMAs with counted vmcnt waits; no barrier until the LayerNorm statistics cross the four feature waves (gemm_bf16_kernel's direct
// epilogue, operation for operation).
// Same arithmetic as the other two forms — accumulators start from the bias, k ascending in 32x32x16 steps, statistics in the order
// (wave's 48 values, half-wave partner, waves 0..3) — so a row has the same bits whichever kernel produced it.
// Bound: 12 KiB of W + 4 KiB of rows per k-step and workgroup at the CU's 64 B/clk L2 path = 256 cycles per k-step.
// =====================================================================================================
typedef __attribute__((ext_vector_type(4))) uint32_t lt_u32x4;
constexpr int LT_PF = 12;   // k-steps in flight (4 loads each: vmcnt <= 63 allows 16; K / 16 must be a multiple)
template <int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ln_tail_gemm_kernel(
    const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wimg32, const float *__restrict__ bias, const bf16_t *__restrict__ res,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps, bf16_t *__restrict__ out, int M, int K, int xpacked) {
    constexpr int N = 384, NT = 3, WAVES_N = 4, BM = 32;
    __shared__ float red[2 * WAVES_N * BM];
    const int lane = threadIdx.x & 63;
    const int wn = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int ksteps = K / 16;                                   // a multiple of PF (launcher)

    // per-lane source addresses at k-step 0; per k-step the W pointer advances 12 KiB, the row pointer 1 KiB (packed) or 32 B
    const int64_t mrow = (int64_t)(m0 + r < M ? m0 + r : M - 1);
    const char *xp = reinterpret_cast<const char *>(X) +
                     (xpacked ? (int64_t)(m0 >> 5) * K * 64 + h * 512 + r * 16          // (rows of a partial block exist: padding)
                              : (mrow * K + 8 * h) * 2);
    const int xstep = xpacked ? 1024 : 32;
    const char *wp = reinterpret_cast<const char *>(Wimg32) + (int64_t)(NT * wn) * 1024 + lane * 16;
    constexpr int WSTEP = (N / 32) * 1024;                       // fragment blocks of one k-step

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + wn * 96 + 4 * h + i * 32 + 8 * gq);
            acc[i][4 * gq + 0] = bv.x;
            acc[i][4 * gq + 1] = bv.y;
            acc[i][4 * gq + 2] = bv.z;
            acc[i][4 * gq + 3] = bv.w;
        }
#pragma unroll
    for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(acc[i]));   // retire the bias loads: the vmcnt counts below are exact

    lt_u32x4 fw[PF][NT], fx[PF];
    // (the three W loads of a step use immediate offsets 0 / 1024 / 2048 from one address register)
    auto issue = [&](auto slotc, int s) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value;
        const int sc = s < ksteps ? s : ksteps - 1;              // past the end: re-read the last step (uniform vmcnt)
        const char *w = wp + (int64_t)sc * WSTEP;
        const char *x = xp + (int64_t)sc * xstep;
        // (references: an asm operand alone does not make a generic lambda capture the arrays)
        lt_u32x4 &f0 = fw[slot][0], &f1 = fw[slot][1], &f2 = fw[slot][2], &f3 = fx[slot];
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f0) : "v"(w) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(f1) : "v"(w) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(f2) : "v"(w) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f3) : "v"(x) : "memory");
    };
    ff_static_for(std::make_integer_sequence<int, PF>{}, [&](auto sc) __attribute__((always_inline)) { issue(sc, decltype(sc)::value); });
    for (int s0 = 0; s0 < ksteps; s0 += PF) {
        ff_static_for(std::make_integer_sequence<int, PF>{}, [&](auto sc) __attribute__((always_inline)) {
            constexpr int slot = decltype(sc)::value;
            // the 4 (PF - 1) loads of the PF - 1 younger steps may stay in flight
            lt_u32x4 &f0 = fw[slot][0], &f1 = fw[slot][1], &f2 = fw[slot][2], &f3 = fx[slot];
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "n"(4 * (PF - 1)) : "memory");
            const bf16x8 bx = __builtin_bit_cast(bf16x8, f3);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f0), bx, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1), bx, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f2), bx, acc[2], 0, 0, 0);
            // the slot's registers are free once the MFMAs have READ them: the reload below is ordered behind them by its
            // dependency on the same registers
            asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
            issue(sc, s0 + slot + PF);
        });
    }
    wait_vmcnt<0>();

    // ---- epilogue: gemm_bf16_kernel<.., WAVES_N = 4, ..>'s direct form with MT = 1, operation for operation
    const int nbase = wn * 96 + 4 * h;
    {
        const int64_t m = m0 + r;
        const int64_t mr = m < M ? m : M - 1;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; gq += 2) {
                const uint4 o = *reinterpret_cast<const uint4 *>(res + mr * N + wn * 96 + i * 32 + 8 * gq + 8 * h);
                auto s0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
                const uint32_t a0 = s0[0], c0 = s0[1], a1 = s1[0], c1 = s1[1];
                acc[i][4 * gq + 0] += __uint_as_float(a0 << 16);
                acc[i][4 * gq + 1] += __uint_as_float(a0 & 0xffff0000u);
                acc[i][4 * gq + 2] += __uint_as_float(a1 << 16);
                acc[i][4 * gq + 3] += __uint_as_float(a1 & 0xffff0000u);
                acc[i][4 * gq + 4] += __uint_as_float(c0 << 16);
                acc[i][4 * gq + 5] += __uint_as_float(c0 & 0xffff0000u);
                acc[i][4 * gq + 6] += __uint_as_float(c1 << 16);
                acc[i][4 * gq + 7] += __uint_as_float(c1 & 0xffff0000u);
            }
    }
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (pass == 0) {
                    s += acc[i][g];
                } else {
                    const float dlt = acc[i][g] - mean;
                    s = fmaf(dlt, dlt, s);
                }
            }
        s += __shfl_xor(s, 32, 64);
        if (h == 0) red[(pass * WAVES_N + wn) * BM + r] = s;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_N; ++w) t += red[(pass * WAVES_N + w) * BM + r];
        if (pass == 0)
            mean = t / (float)N;
        else
            rstd = 1.0f / sqrtf(t / (float)N + eps);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        float4 gv[4], be[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            gv[gq] = *reinterpret_cast<const float4 *>(gamma + nbase + i * 32 + 8 * gq);
            be[gq] = *reinterpret_cast<const float4 *>(beta + nbase + i * 32 + 8 * gq);
        }
        uint32_t pk[8];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float y0 = fmaf((acc[i][4 * gq + 0] - mean) * rstd, gv[gq].x, be[gq].x);
            const float y1 = fmaf((acc[i][4 * gq + 1] - mean) * rstd, gv[gq].y, be[gq].y);
            const float y2 = fmaf((acc[i][4 * gq + 2] - mean) * rstd, gv[gq].z, be[gq].z);
            const float y3 = fmaf((acc[i][4 * gq + 3] - mean) * rstd, gv[gq].w, be[gq].w);
            pk[2 * gq] = pack_bf16x2(y0, y1);
            pk[2 * gq + 1] = pack_bf16x2(y2, y3);
        }
        const int64_t m = m0 + r;
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2) {
            auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gq], pk[2 * gq + 2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gq + 1], pk[2 * gq + 3], false, false);
            if (m < M)
                *reinterpret_cast<uint4 *>(out + m * N + wn * 96 + i * 32 + 8 * gq + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
}

static int ln_tail_gemm(const bf16_t *X, const bf16_t *Wimg32, const float *bias, const bf16_t *res, const float *gamma,
                        const float *beta, float eps, bf16_t *out, int M, int K, hipStream_t st, bool xpacked) {
    if (K % (16 * LT_PF) != 0) return fail(TSIM_EUNSUPPORTED, "ln_tail_gemm: K=%d", K);
    hipLaunchKernelGGL(ln_tail_gemm_kernel<LT_PF>, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, st, X, Wimg32, bias, res, gamma, beta,
                       eps, out, M, K, xpacked ? 1 : 0);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

// LayerNorm GEMM of width 384 over rows [0, M) in workgroups of NW * 32 token rows (the last one may be partial)
template <int NW, int NST, bool STAG = false>
static int ln_rows_gemm(const bf16_t *X, const bf16_t *Wimg32, const float *bias, const bf16_t *res, const float *gamma,
                        const float *beta, float eps, bf16_t *out, int M, int K, hipStream_t st, bool xpacked) {
    constexpr int lds = NST * (NW * 32 * LR_BK * 2 + LR_WBYTES);   // <8, 3>: 120 KiB, <2, 5>: 140 KiB
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = ln_rows_gemm_kernel<NW, NST, STAG>;
    static DevOnce lds_once;
    TSIM_MAX_LDS(lds_once, kern, lds);
    if (K % LR_BK != 0 || K < NST * LR_BK) return fail(TSIM_EUNSUPPORTED, "ln_rows_gemm: K=%d", K);
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + NW * 32 - 1) / (NW * 32))), dim3(NW * 64), lds, st, X, Wimg32, bias, res, gamma,
                       beta, eps, out, M, K, xpacked ? 1 : 0);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

// The block-packed qkv / h1 layout (packed_off) needs gemm_xres2 as the producer on both projections
static bool use_packed_layout(int H, int F) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("TSIM_PACKED_ACT"), *x2 = getenv("TSIM_XRES2"), *xr = getenv("TSIM_GEMM_XRES"), *bg = getenv("TSIM_GEMM_BIG");
        on = (e ? atoi(e) : 1) && (x2 ? atoi(x2) : 1) && (xr ? atoi(xr) : 1) && (bg ? atoi(bg) != 2 : 1);
    }
    return on && H == 384 && (3 * H) % X2_BN == 0 && F % X2_BN == 0 && F <= 2048;
}

template <int EPI>
static int gemm_plain(const bf16_t *X, const bf16_t *W, const bf16_t *Wp, const float *bias, bf16_t *out, int M, int N,
                      int K, hipStream_t st) {
    static int use_xres = -1, big = -1;
    if (use_xres < 0) { const char *e = getenv("TSIM_GEMM_XRES"); use_xres = e ? atoi(e) : 1; }
    if (big < 0) { const char *e = getenv("TSIM_GEMM_BIG"); big = e ? atoi(e) : 1; }
    if (use_xres && big != 2 && K == 384 && N % XR_BN == 0) return gemm_xres<EPI>(X, W, bias, out, M, N, st);
    if (big && (K >= 768 || big == 2) && gemm_pp_supported(N, K))
        return gemm_pp(EPI == EPI_GELU ? PP_EPI_GELU : PP_EPI_BIAS, X, Wp ? Wp : W, Wp != nullptr, bias, out, M, N, K, st);
    if (N % 128 == 0)
        return launch_gemm<128, 128, 64, 2, 2, EPI>(X, W, bias, nullptr, nullptr, nullptr, 0.f, out, M, N, K, st);
    return launch_gemm<128, 64, 64, 2, 2, EPI>(X, W, bias, nullptr, nullptr, nullptr, 0.f, out, M, N, K, st);
}

static int gemm_res_ln(const bf16_t *X, const bf16_t *W, const bf16_t *Wp, const float *bias, const bf16_t *res,
                       const float *gamma, const float *beta, float eps, bf16_t *out, int M, int N, int K, float *ybuf,
                       hipStream_t st, const bf16_t *Wimg = nullptr, const bf16_t *Wimg32 = nullptr, bool xpacked = false) {
    if (xpacked && N != 384) return fail(TSIM_EINVAL, "gemm_res_ln: the packed operand layout is a hidden-384 form");
    static int use_img = -1;
    if (use_img < 0) { const char *e = getenv("TSIM_LN_WIMG"); use_img = e ? atoi(e) : 1; }
    if (!use_img) Wimg = nullptr;
    static int big = -1;
    if (big < 0) { const char *e = getenv("TSIM_GEMM_BIG"); big = e ? atoi(e) : 1; }
    if (big && ybuf && N >= 512 && N % 256 == 0 && gemm_pp_supported(N, K)) {
        // wide rows: a workgroup cannot own whole 768-feature rows at a 256-token tile, so the projection writes
        // fp32 sums and a row kernel adds the residual and normalises (HBM-bound, 8 B per element)
        int rc = gemm_pp(PP_EPI_F32, X, Wp ? Wp : W, Wp != nullptr, bias, ybuf, M, N, K, st);
        if (rc) return rc;
        return res_ln_rows(ybuf, res, gamma, beta, eps, out, nullptr, nullptr, M, N, st);
    }
    switch (N) {
        case 384: {
            // 128-token tiles, one workgroup per CU: mt tiles take ceil(mt/256) rounds and a nearly empty last round
            // costs a full one.  A small remainder is launched separately with 32-token tiles (4x more, 4x shorter
            // workgroups), e.g. 525 tiles = 2 rounds + 13 tiles -> 2 rounds + a quarter round.
            static int split = -1;
            if (split < 0) { const char *e = getenv("TSIM_LN_TAIL"); split = e ? atoi(e) : 1; }
            // (64-token tiles measured slower: 64 x 384 x 32-k with two workgroups per CU 2.96 ms per forward, 64 x 384 x 64-k 3.08,
            // against 2.74: the W tile is re-staged for half as many tokens)
            // 256-token tiles, one 32-token row block per wave (ln_rows_gemm_kernel), for whole rounds of 256 tiles and for a last
            // round that is at least half full; everything else — and every small batch — goes through the 128- / 32-token
            // forms below.  All forms give a row the same bits (bit-wise large-vs-small-batch test).
            static int rows256 = -1;
            if (rows256 < 0) { const char *e = getenv("TSIM_LN_ROWS256"); rows256 = e ? atoi(e) : 1; }
            bool after_rows = false;   // the rest below is the remainder of a ln_rows launch: a true tail
            if (rows256 && Wimg32) {
                const int t256 = M / 256;
                const int use = (t256 % 256) >= 128 ? t256 : (t256 / 256) * 256;
                if (use > 0) {
                    // (a four-slot ring with the second half of the waves issuing behind their MFMAs — STAG = true — measured equal,
                    // 68.5 vs 66-68 us per launch; not instantiated)
                    int rc = ln_rows_gemm<8, 3>(X, Wimg32, bias, res, gamma, beta, eps, out, use * 256, K, st, xpacked);
                    const int m1 = use * 256;
                    if (rc || m1 == M) return rc;
                    X += (int64_t)m1 * K; res += (int64_t)m1 * N; out += (int64_t)m1 * N; M -= m1;
                    after_rows = true;
                }
                // few rows (a remainder, or a small batch): one round of 32-row workgroups that stream W straight into registers
                static int frag_tail = -1;
                if (frag_tail < 0) { const char *e = getenv("TSIM_LN_TAIL_FRAG"); frag_tail = e ? atoi(e) : 1; }
                if (frag_tail && M <= 256 * 32 && K % (16 * LT_PF) == 0)
                    return ln_tail_gemm(X, Wimg32, bias, res, gamma, beta, eps, out, M, K, st, xpacked);
                // a remainder of at most 128 workgroups of 64 rows: the deep-ring form (TSIM_LN_ROWS_TAIL=0: the 64- / 32-token
                // tiles of gemm_bf16_kernel below)
                static int rows_tail = -1;
                if (rows_tail < 0) { const char *e = getenv("TSIM_LN_ROWS_TAIL"); rows_tail = e ? atoi(e) : 0; }
                if (rows_tail && after_rows && M <= 128 * 64 && K >= 5 * LR_BK)
                    return ln_rows_gemm<2, 5>(X, Wimg32, bias, res, gamma, beta, eps, out, M, K, st, xpacked);
            }
            const int mt = (M + 127) / 128, full = (mt / 256) * 256, rem = mt - full;
            // (four 32-k ring slots instead of two 64-k ones — prefetch distance 3 — measured 2.5 % SLOWER per forward: the deep
            // path issues a k-tile's DMA pieces in one burst ahead of the MFMAs instead of behind each k-step's)
            auto main_launch = [&](int rows) {
                return launch_gemm<128, 384, 64, 2, 4, EPI_RES_LN>(X, W, bias, res, gamma, beta, eps, out, rows, N, K, st, Wimg, xpacked);
            };
            if (split && rem > 0 && rem <= 96 && (full > 0 || after_rows)) {
                const int m_main = full * 128;
                if (full > 0) {
                    int rc = main_launch(m_main);
                    if (rc) return rc;
                }
                // (a three-slot ring for the remainder launch measured no different: 2.744-2.751 vs 2.751-2.752 ms per forward)
                // TSIM_LN_TAIL_BM: token rows per workgroup of the remainder launch (32: four waves, 64: eight waves = 7 instead of
                // 13 DMA pieces per wave and k-tile)
                static int tail_bm = -1;
                if (tail_bm < 0) { const char *e = getenv("TSIM_LN_TAIL_BM"); tail_bm = e ? atoi(e) : 64; }
                if (tail_bm == 64)
                    return launch_gemm<64, 384, 64, 2, 4, EPI_RES_LN>(X + (int64_t)m_main * K, W, bias, res + (int64_t)m_main * N,
                                                                      gamma, beta, eps, out + (int64_t)m_main * N, M - m_main, N, K, st, Wimg, xpacked);
                return launch_gemm<32, 384, 64, 1, 4, EPI_RES_LN>(X + (int64_t)m_main * K, W, bias, res + (int64_t)m_main * N,
                                                                  gamma, beta, eps, out + (int64_t)m_main * N, M - m_main, N, K, st, Wimg, xpacked);
            }
            return main_launch(M);
        }
        case 768: return launch_gemm<64, 768, 32, 1, 8, EPI_RES_LN>(X, W, bias, res, gamma, beta, eps, out, M, N, K, st);
        case 64: return launch_gemm<128, 64, 64, 4, 2, EPI_RES_LN>(X, W, bias, res, gamma, beta, eps, out, M, N, K, st);
        default: return fail(TSIM_EUNSUPPORTED, "encoder: hidden size %d has no fused LayerNorm GEMM (64, 384, 768)", N);
    }
}

}  // namespace tsim

extern "C" int tsim_encoder_create(const tsim_encoder_config *cfg, const tsim_encoder_weights_host *w,
                                   tsim_encoder **out) {
    TSIM_REQUIRE(cfg && w && out, "encoder_create: null pointer");
    const int H = cfg->hidden, F = cfg->ffn, L = cfg->num_layers;
    TSIM_REQUIRE(H == 64 || H == 384 || H == 768, "encoder_create: hidden=%d unsupported (64, 384, 768)", H);
    TSIM_REQUIRE(cfg->heads > 0 && H % cfg->heads == 0, "encoder_create: heads=%d does not divide hidden=%d", cfg->heads, H);
    const int dh = H / cfg->heads;
    TSIM_REQUIRE(dh == 16 || dh == 32 || dh == 64, "encoder_create: head_dim=%d unsupported (16, 32, 64)", dh);
    TSIM_REQUIRE(F % 64 == 0 && L > 0 && cfg->max_tokens > 0 && cfg->max_seqs > 0, "encoder_create: bad ffn/layers/capacity");
    TSIM_REQUIRE(cfg->arch == TSIM_ARCH_BERT || cfg->arch == TSIM_ARCH_MPNET, "encoder_create: unknown arch");
    TSIM_REQUIRE(w->word_emb && w->pos_emb && w->emb_ln_g && w->emb_ln_b && w->layers, "encoder_create: missing weights");
    TSIM_REQUIRE(cfg->arch != TSIM_ARCH_MPNET || w->rel_bias, "encoder_create: MPNet needs rel_bias");
    const bool mx = cfg->weight_dtype == TSIM_W_MXFP8;
    TSIM_REQUIRE(cfg->weight_dtype == TSIM_W_BF16 || mx, "encoder_create: unknown weight_dtype %d", cfg->weight_dtype);
    if (mx && !(gemm_pp_mx_supported(H, H) && gemm_pp_mx_supported(3 * H, H) && gemm_pp_mx_supported(F, H) && gemm_pp_mx_supported(H, F)))
        return fail(TSIM_EUNSUPPORTED, "encoder_create: MXFP8 projections need hidden and ffn to be multiples of 256 (got %d, %d)", H, F);
    tsim_encoder *e = new tsim_encoder();
    e->cfg = *cfg;
    e->Tp = (cfg->max_tokens + 127) / 128 * 128 + 128;
    int rc = TSIM_OK;
    auto bail = [&](int code) {
        tsim_encoder_destroy(e);
        return code;
    };
    if ((rc = upload_f32(e, w->word_emb, (size_t)cfg->vocab * H, &e->word))) return bail(rc);
    if ((rc = upload_f32(e, w->pos_emb, (size_t)cfg->max_pos * H, &e->pos))) return bail(rc);
    if (w->type_emb && (rc = upload_f32(e, w->type_emb, H, &e->type0))) return bail(rc);
    if ((rc = upload_f32(e, w->emb_ln_g, H, &e->emb_g))) return bail(rc);
    if ((rc = upload_f32(e, w->emb_ln_b, H, &e->emb_b))) return bail(rc);
    if (cfg->arch == TSIM_ARCH_MPNET) {
        const int maxp = cfg->max_pos;
        e->relw = 2 * maxp - 1;
        std::vector<float> tab((size_t)cfg->heads * e->relw);
        for (int d = -(maxp - 1); d <= maxp - 1; ++d) {
            const int bk = mpnet_bucket(d, cfg->rel_buckets);
            for (int hh = 0; hh < cfg->heads; ++hh) tab[(size_t)hh * e->relw + d + maxp - 1] = w->rel_bias[bk * cfg->heads + hh];
        }
        if ((rc = upload_f32(e, tab.data(), tab.size(), &e->relb))) return bail(rc);
    }
    e->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        const tsim_layer_weights_host &lw = w->layers[l];
        tsim_encoder::Layer &d = e->layers[l];
        if ((rc = upload_bf16(e, {lw.wq, lw.wk, lw.wv}, (size_t)H * H, &d.wqkv))) return bail(rc);
        if ((rc = upload_bf16(e, {lw.wo}, (size_t)H * H, &d.wo))) return bail(rc);
        if ((rc = upload_bf16(e, {lw.w1}, (size_t)F * H, &d.w1))) return bail(rc);
        if ((rc = upload_bf16(e, {lw.w2}, (size_t)H * F, &d.w2))) return bail(rc);
        if (mx) {
            if ((rc = upload_mxfp8(e, {lw.wq, lw.wk, lw.wv}, H, H, &d.qqkv, &d.sqkv))) return bail(rc);
            if ((rc = upload_mxfp8(e, {lw.wo}, H, H, &d.qo, &d.so))) return bail(rc);
            if ((rc = upload_mxfp8(e, {lw.w1}, F, H, &d.q1, &d.s1))) return bail(rc);
            if ((rc = upload_mxfp8(e, {lw.w2}, H, F, &d.q2, &d.s2))) return bail(rc);
        }
        // tile-major copies for the ping-pong projections (contiguous 1-KiB DMA pieces): [feature tile][k-tile][LDS image]
        auto repack = [&](const void *src, int n_rows, int kbytes, void **dst) {
            int r2 = dev_alloc(e, (size_t)n_rows * kbytes, dst);
            return r2 ? r2 : pack_w(src, *dst, n_rows, kbytes, gemm_pp_tile_width(n_rows), nullptr);
        };
        if (mx) {
            void *t;
            if ((rc = repack(d.qqkv, 3 * H, H, &t))) return bail(rc); d.qqkv = (uint8_t *)t;
            if ((rc = repack(d.qo, H, H, &t))) return bail(rc); d.qo = (uint8_t *)t;
            if ((rc = repack(d.q1, F, H, &t))) return bail(rc); d.q1 = (uint8_t *)t;
            if ((rc = repack(d.q2, H, F, &t))) return bail(rc); d.q2 = (uint8_t *)t;
        } else if (H >= 512 && H % 256 == 0 && gemm_pp_supported(H, H) && gemm_pp_supported(F, H) && gemm_pp_supported(H, F)) {
            if ((rc = repack(d.wqkv, 3 * H, H * 2, (void **)&d.pqkv))) return bail(rc);
            if ((rc = repack(d.wo, H, H * 2, (void **)&d.po))) return bail(rc);
            if ((rc = repack(d.w1, F, H * 2, (void **)&d.p1))) return bail(rc);
            if ((rc = repack(d.w2, H, F * 2, (void **)&d.p2))) return bail(rc);
        }
        if (!mx && H == 384 && F % 64 == 0) {   // LayerNorm GEMM (BN = 384, BK = 64): W as contiguous k-tile images
            if ((rc = dev_alloc(e, (size_t)H * H * 2, (void **)&d.lo))) return bail(rc);
            if ((rc = dev_alloc(e, (size_t)H * F * 2, (void **)&d.l2))) return bail(rc);
            hipLaunchKernelGGL(pack_gemm_w_kernel, dim3((unsigned)((H * H / 8 + 255) / 256)), dim3(256), 0, 0,
                               reinterpret_cast<const uint4 *>(d.wo), reinterpret_cast<uint4 *>(d.lo), 384, 64, H);
            hipLaunchKernelGGL(pack_gemm_w_kernel, dim3((unsigned)((H * F / 8 + 255) / 256)), dim3(256), 0, 0,
                               reinterpret_cast<const uint4 *>(d.w2), reinterpret_cast<uint4 *>(d.l2), 384, 64, F);
            if ((rc = dev_alloc(e, (size_t)H * H * 2, (void **)&d.lo32))) return bail(rc);
            if ((rc = dev_alloc(e, (size_t)H * F * 2, (void **)&d.l232))) return bail(rc);
            hipLaunchKernelGGL(pack_frag_w_kernel, dim3((unsigned)((H * H / 8 + 255) / 256)), dim3(256), 0, 0,
                               reinterpret_cast<const uint4 *>(d.wo), reinterpret_cast<uint4 *>(d.lo32), 384, H);
            hipLaunchKernelGGL(pack_frag_w_kernel, dim3((unsigned)((H * F / 8 + 255) / 256)), dim3(256), 0, 0,
                               reinterpret_cast<const uint4 *>(d.w2), reinterpret_cast<uint4 *>(d.l232), 384, F);
            if (hipGetLastError() != hipSuccess) return bail(fail(TSIM_EHIP, "LayerNorm GEMM weight packing failed"));
        }
        if (!mx && H == 384 && F % 64 == 0 && F <= 4096) {   // fused FFN (ffn_fused_kernel): both matrices as LDS images
            if ((rc = dev_alloc(e, (size_t)F * H * 2, (void **)&d.p1))) return bail(rc);
            if ((rc = dev_alloc(e, (size_t)F * H * 2, (void **)&d.p2))) return bail(rc);
            hipLaunchKernelGGL(pack_ffn_w1_kernel, dim3((unsigned)((F * 48 + 255) / 256)), dim3(256), 0, 0,
                               reinterpret_cast<const uint4 *>(d.w1), reinterpret_cast<uint4 *>(d.p1), F);
            hipLaunchKernelGGL(pack_ffn_w2_kernel, dim3((unsigned)((384 * F + 255) / 256)), dim3(256), 0, 0, d.w2, d.p2, F);
            if (hipGetLastError() != hipSuccess) return bail(fail(TSIM_EHIP, "FFN weight packing failed"));
        }
        std::vector<float> bq(3 * (size_t)H);
        memcpy(bq.data(), lw.bq, H * 4);
        memcpy(bq.data() + H, lw.bk, H * 4);
        memcpy(bq.data() + 2 * H, lw.bv, H * 4);
        if ((rc = upload_f32(e, bq.data(), bq.size(), &d.bqkv))) return bail(rc);
        if ((rc = upload_f32(e, lw.bo, H, &d.bo))) return bail(rc);
        if ((rc = upload_f32(e, lw.b1, F, &d.b1))) return bail(rc);
        if ((rc = upload_f32(e, lw.b2, H, &d.b2))) return bail(rc);
        if ((rc = upload_f32(e, lw.ln1_g, H, &d.g1))) return bail(rc);
        if ((rc = upload_f32(e, lw.ln1_b, H, &d.be1))) return bail(rc);
        if ((rc = upload_f32(e, lw.ln2_g, H, &d.g2))) return bail(rc);
        if ((rc = upload_f32(e, lw.ln2_b, H, &d.be2))) return bail(rc);
    }
    const size_t Tp = e->Tp;
    struct { bf16_t **p; size_t n; } acts[] = {{&e->x0, Tp * H}, {&e->x1, Tp * H}, {&e->qkv, Tp * 3 * H},
                                               {&e->ctx, Tp * H}, {&e->h1, Tp * F}};
    for (auto &a : acts) {
        if ((rc = dev_alloc(e, a.n * 2, (void **)a.p))) return bail(rc);
        if (hipMemset(*a.p, 0, a.n * 2) != hipSuccess) return bail(fail(TSIM_EHIP, "hipMemset failed"));
    }
    if (H >= 512 && H % 256 == 0 && gemm_pp_supported(H, H)) {
        if ((rc = dev_alloc(e, Tp * H * 4, (void **)&e->ybuf))) return bail(rc);
        if (hipMemset(e->ybuf, 0, Tp * H * 4) != hipSuccess) return bail(fail(TSIM_EHIP, "hipMemset failed"));
    }
    if (mx) {
        struct { uint8_t **p; size_t n; } qb[] = {{&e->aq, Tp * H}, {&e->as, Tp * H / 32}, {&e->hq, Tp * F}, {&e->hs, Tp * F / 32}};
        for (auto &a : qb) {
            if ((rc = dev_alloc(e, a.n, (void **)a.p))) return bail(rc);
            if (hipMemset(*a.p, 0, a.n) != hipSuccess) return bail(fail(TSIM_EHIP, "hipMemset failed"));
        }
    }
    if ((rc = dev_alloc(e, 256, (void **)&e->err_flags))) return bail(rc);
    if (hipMemset(e->err_flags, 0, 256) != hipSuccess) return bail(fail(TSIM_EHIP, "hipMemset failed"));
    if (hipDeviceSynchronize() != hipSuccess) return bail(fail(TSIM_EHIP, "sync after upload failed"));
    *out = e;
    return TSIM_OK;
}

extern "C" int tsim_encoder_error_flags(tsim_encoder *e, int32_t *flags_host, void *stream) {
    TSIM_REQUIRE(e && flags_host, "encoder_error_flags: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    TSIM_HIP_CHECK(hipMemcpyAsync(flags_host, e->err_flags, 4, hipMemcpyDeviceToHost, st));
    TSIM_HIP_CHECK(hipMemsetAsync(e->err_flags, 0, 4, st));
    TSIM_HIP_CHECK(hipStreamSynchronize(st));
    return TSIM_OK;
}

extern "C" int tsim_quantize_mxfp8(const void *x_bf16, int64_t rows, int K, void *q_out, void *scale_out, void *stream) {
    TSIM_REQUIRE(x_bf16 && q_out && scale_out, "quantize_mxfp8: null pointer");
    TSIM_REQUIRE(rows >= 0 && K > 0, "quantize_mxfp8: bad shape rows=%lld K=%d", (long long)rows, K);
    return quant_mx(static_cast<const bf16_t *>(x_bf16), rows, K, static_cast<uint8_t *>(q_out),
                    static_cast<uint8_t *>(scale_out), reinterpret_cast<hipStream_t>(stream));
}

extern "C" int tsim_gemm_mxfp8(const void *xq, const void *xs, const void *wq, const void *ws, const float *bias,
                               float *out_f32, int M, int N, int K, void *stream) {
    TSIM_REQUIRE(xq && xs && wq && ws && bias && out_f32, "gemm_mxfp8: null pointer");
    TSIM_REQUIRE(M >= 0 && N > 0 && K > 0, "gemm_mxfp8: bad shape M=%d N=%d K=%d", M, N, K);
    return gemm_pp_mx(PP_EPI_F32, static_cast<const uint8_t *>(xq), static_cast<const uint8_t *>(xs),
                      static_cast<const uint8_t *>(wq), 0, static_cast<const uint8_t *>(ws), bias, out_f32, nullptr, M, N, K,
                      reinterpret_cast<hipStream_t>(stream));
}

#ifdef TSIM_PP_STAMPS
extern "C" int tsim_debug_ff_stamps(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsim::g_ff_stamps), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(tsim::g_ff_stamps), z, 64) != hipSuccess) return 1; }
    return 0;
}
extern "C" int tsim_debug_xr_stamps(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsim::g_xr_stamps), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(tsim::g_xr_stamps), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" void tsim_encoder_destroy(tsim_encoder *e) {
    if (!e) return;
    for (void *p : e->allocs) (void)hipFree(p);
    delete e;
}

extern "C" int tsim_encoder_forward(tsim_encoder *e, const int32_t *tok_ids, const int32_t *tok_pos,
                                    const int32_t *tok_col, const int32_t *cu_seqlens, int32_t T, int32_t B,
                                    int32_t max_len, float *pooled_f32, void *unit_bf16, int ld_unit, float *unit_rho_max,
                                    void *last_hidden_bf16, void *stream) {
    TSIM_REQUIRE(e && cu_seqlens && (T == 0 || (tok_ids && tok_pos)), "encoder_forward: null pointer");   // T = 0: only empty sequences
    TSIM_REQUIRE(T >= 0 && B >= 0 && T <= e->cfg.max_tokens && B <= e->cfg.max_seqs,
                 "encoder_forward: T=%d B=%d exceed capacity (%d tokens, %d sequences)", T, B, e->cfg.max_tokens, e->cfg.max_seqs);
    // position rows: BERT uses 0 .. len-1, MPNet pad_id+1 .. pad_id+len (modeling_mpnet create_position_ids_from_input_ids)
    const int pos_span = max_len + (e->cfg.arch == TSIM_ARCH_MPNET ? e->cfg.pad_id + 1 : 0);
    TSIM_REQUIRE(max_len >= 0 && pos_span <= e->cfg.max_pos,
                 "encoder_forward: sequences of %d tokens need position rows up to %d, the table has %d", max_len, pos_span - 1,
                 e->cfg.max_pos);
    TSIM_REQUIRE(!unit_bf16 || ld_unit >= e->cfg.hidden, "encoder_forward: ld_unit < hidden");
    if (B == 0) return TSIM_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const tsim_encoder_config &c = e->cfg;
    const int H = c.hidden, F = c.ffn, dh = H / c.heads;
    const bool mx = c.weight_dtype == TSIM_W_MXFP8;
    int rc;
    if (T > 0) {
        const unsigned g = (unsigned)((T + 3) / 4);
        // column of a token inside its sequence: tok_col when given (MPNet), else tok_pos (BERT: position == column)
        const int32_t *colp = tok_col ? tok_col : tok_pos;
        const int col_limit = tok_col || c.arch == TSIM_ARCH_BERT ? max_len : c.max_pos;   // MPNet without tok_col: pos is not a column
#define EMBED(V) hipLaunchKernelGGL(embed_ln_kernel<V>, dim3(g), dim3(256), 0, st, tok_ids, tok_pos, e->word, e->pos, e->type0, e->emb_g, e->emb_b, c.ln_eps, T, H, e->x0, c.vocab, c.max_pos, colp, col_limit, e->err_flags)
        if (H == 64) EMBED(1); else if (H == 384) EMBED(6); else EMBED(12);
#undef EMBED
        TSIM_HIP_CHECK(hipGetLastError());
        const float scale = 1.0f / sqrtf((float)dh);
        const bool rel = c.arch == TSIM_ARCH_MPNET;
        // qkv and h1 in the block-packed layout (packed_off) when gemm_xres2 produces them: hidden 384, BERT family
        const bool pk = !mx && !rel && use_packed_layout(H, F);
        const int32_t *col = rel ? (tok_col ? tok_col : tok_pos) : nullptr;
        const int qblocks = (max_len + 31) / 32 > 0 ? (max_len + 31) / 32 : 1;
        static int att_xcd = -1;
        if (att_xcd < 0) { const char *ev = getenv("TSIM_ATT_XCD"); att_xcd = ev ? atoi(ev) : 1; }
        const dim3 agrid((unsigned)(att_xcd ? ((B + 7) / 8) * 8 : B), (unsigned)((c.heads + 3) / 4), (unsigned)qblocks);
        for (int l = 0; l < c.num_layers; ++l) {
            const tsim_encoder::Layer &L = e->layers[l];
            if (mx) {   // projections on MXFP8 operands (v_mfma_scale_f32_32x32x64_f8f6f4); x0's image comes fused from the
                        // previous layer's LayerNorm kernel, for layer 0 from the stand-alone quantiser
                if (l == 0 && (rc = quant_mx(e->x0, T, H, e->aq, e->as, st))) return rc;
                if ((rc = gemm_pp_mx(PP_EPI_BIAS, e->aq, e->as, L.qqkv, 1, L.sqkv, L.bqkv, e->qkv, nullptr, T, 3 * H, H, st))) return rc;
            } else if (pk) {
                if ((rc = gemm_xres2<EPI_BIAS, true>(e->x0, L.wqkv, L.bqkv, e->qkv, T, 3 * H, st))) return rc;
            } else if ((rc = gemm_plain<EPI_BIAS>(e->x0, L.wqkv, L.pqkv, L.bqkv, e->qkv, T, 3 * H, H, st))) return rc;
#define ATT(D)                                                                                                 \
    do {                                                                                                       \
        if (rel)                                                                                               \
            hipLaunchKernelGGL((attention_kernel<D, true>), agrid, dim3(256), 0, st, e->qkv, cu_seqlens, col,  \
                               e->relb, e->relw, H, c.heads, scale, e->ctx, B, att_xcd);                                \
        else if (pk)                                                                                           \
            hipLaunchKernelGGL((attention_kernel<D, false, true>), agrid, dim3(256), 0, st, e->qkv, cu_seqlens, col, \
                               e->relb, e->relw, H, c.heads, scale, e->ctx, B, att_xcd);                                \
        else                                                                                                   \
            hipLaunchKernelGGL((attention_kernel<D, false>), agrid, dim3(256), 0, st, e->qkv, cu_seqlens, col, \
                               e->relb, e->relw, H, c.heads, scale, e->ctx, B, att_xcd);                                \
    } while (0)
            if (dh == 16) ATT(16); else if (dh == 32) ATT(32); else ATT(64);
#undef ATT
            TSIM_HIP_CHECK(hipGetLastError());
            if (mx) {
                if ((rc = quant_mx(e->ctx, T, H, e->aq, e->as, st))) return rc;
                if ((rc = gemm_pp_mx(PP_EPI_F32, e->aq, e->as, L.qo, 1, L.so, L.bo, e->ybuf, nullptr, T, H, H, st))) return rc;
                if ((rc = res_ln_rows(e->ybuf, e->x0, L.g1, L.be1, c.ln_eps, e->x1, e->aq, e->as, T, H, st))) return rc;
                if ((rc = gemm_pp_mx(PP_EPI_GELU_MX, e->aq, e->as, L.q1, 1, L.s1, L.b1, e->hq, e->hs, T, F, H, st))) return rc;
                if ((rc = gemm_pp_mx(PP_EPI_F32, e->hq, e->hs, L.q2, 1, L.s2, L.b2, e->ybuf, nullptr, T, H, F, st))) return rc;
                if ((rc = res_ln_rows(e->ybuf, e->x1, L.g2, L.be2, c.ln_eps, e->x0, e->aq, e->as, T, H, st))) return rc;
                continue;
            }
            if ((rc = gemm_res_ln(e->ctx, L.wo, L.po, L.bo, e->x0, L.g1, L.be1, c.ln_eps, e->x1, T, H, H, e->ybuf, st, L.lo, L.lo32))) return rc;
            static int fused = -1;
            // OFF by default: at the bench shape (67 k tokens = 525 blocks of 128 on 256 CUs: three rounds) the fused kernel
            // takes 256 us per layer against 245 us for FFN1 + FFN2 + tail (profiles/README.md, round 2); kept for shapes that
            // fill whole rounds and as the starting point of the next round
            if (fused < 0) { const char *ev = getenv("TSIM_FFN_FUSED"); fused = ev ? atoi(ev) : 0; }
            if (fused && H == 384 && L.p1 && L.p2) {
                if ((rc = ffn_fused(e->x1, L.p1, L.p2, L.b1, L.b2, L.g2, L.be2, c.ln_eps, e->x0, T, F, st))) return rc;
                continue;
            }
            if (pk) {
                if ((rc = gemm_xres2<EPI_GELU, true>(e->x1, L.w1, L.b1, e->h1, T, F, st))) return rc;
            } else if ((rc = gemm_plain<EPI_GELU>(e->x1, L.w1, nullptr, L.b1, e->h1, T, F, H, st))) return rc;
            if ((rc = gemm_res_ln(e->h1, L.w2, nullptr, L.b2, e->x1, L.g2, L.be2, c.ln_eps, e->x0, T, H, F, e->ybuf, st, L.l2, L.l232, pk))) return rc;
        }
        if (last_hidden_bf16)
            TSIM_HIP_CHECK(hipMemcpyAsync(last_hidden_bf16, e->x0, (size_t)T * H * 2, hipMemcpyDeviceToDevice, st));
    }
    if (pooled_f32 || unit_bf16) {
        const unsigned g = (unsigned)((B + 3) / 4);
#define POOL(V) hipLaunchKernelGGL(pool_packed_kernel<V>, dim3(g), dim3(256), 0, st, e->x0, cu_seqlens, B, H, pooled_f32, (unit_t *)unit_bf16, ld_unit, unit_bf16 ? unit_rho_max : nullptr)
        if (H % 8 != 0 || H > 1024 || (unit_bf16 && ld_unit % 8 != 0))
            return fail(TSIM_EUNSUPPORTED, "encoder: pooling needs a hidden size that is a multiple of 8, at most 1024 (got %d)", H);
        if (((uintptr_t)pooled_f32 | (uintptr_t)unit_bf16) & 15)
            return fail(TSIM_EINVAL, "encoder: pooled / unit outputs must be 16-byte aligned");
        if (H <= 512) POOL(1); else POOL(2);
#undef POOL
        TSIM_HIP_CHECK(hipGetLastError());
    }
    return TSIM_OK;
}
