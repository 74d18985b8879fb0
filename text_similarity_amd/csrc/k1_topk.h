// K1 of the cosine top-k path: fused MFMA scoring + per-lane top-k selection (gfx950).
// Included by k1_kl16.hip / k1_kl32.hip (one translation unit per list length so they build in parallel).
#pragma once
#include <math.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace tsim {

// =====================================================================================================
// K1: cos_topk_partial — the hot kernel.
//
// Work decomposition: workgroup = (query block of NWAVES*32 queries) x (corpus chunk of rows_per_chunk
// rows).  Each wave keeps its 32 query rows resident in VGPRs as the MFMA B operand (D/16 k-steps x 4
// VGPRs = 96 VGPRs at D = 384), so the only operand that moves in the main loop is the corpus.  Corpus
// tiles of 32 rows stream HBM -> LDS with LDS-DMA (global_load_lds_dwordx4, a 3-stage ring, counted vmcnt,
// one raw s_barrier per tile) and are shared by all waves of the workgroup.  Per tile and wave: D/16
// v_mfma_f32_32x32x16_f16 with A = corpus tile (ds_read_b128 from an XOR-swizzled row-major image), the
// 32x32 score tile stays in 16 accumulator VGPRs: lane (r, h) holds query r against corpus rows
// {(reg&3) + 8*(reg>>2) + 4*h}.  Because the query sits on the lane, top-k selection needs no cross-lane
// traffic: one threshold VGPR (the lane's current KL-th best), a max3 tree over the 16 accumulators and a
// wave-uniform branch reject almost every tile; survivors are appended to a per-lane queue in LDS and
// drained in bulk (all 64 lanes insert into their sorted register lists together), which amortises the
// insertion network over up to 64 candidates.
//
// Output: per (query, chunk, lane-half) a sorted list of KL (score, local row) pairs; K2 merges them.
// Roofline: 2*32*32*D FLOP per tile and wave on the MFMA pipe; D*2 bytes per corpus row from HBM once
// per query block; nothing proportional to Q*N is ever written.
// =====================================================================================================
constexpr int K1_TILE_ROWS = 32;
constexpr int K1_NSTAGE = 3;   // ring slots (tiles); the PAIR variant uses 4 slots = two 2-tile stages
#ifndef TSIM_K1_QCAP
#define TSIM_K1_QCAP 8
#endif
constexpr int K1_QCAP = TSIM_K1_QCAP;  // per-lane candidate queue depth (entries)

template <int KL>
__device__ __forceinline__ void list_insert(float (&ls)[KL], int (&li)[KL], float s, int i) {
    // bubble (s,i) down a list sorted by score descending.  Strict '>' keeps the element that was seen
    // earlier ahead of a later one with an equal score; a lane sees corpus rows in increasing order, so
    // ties resolve to the lower index.  s = -inf is a no-op.
#pragma unroll
    for (int j = 0; j < KL; ++j) {
        const bool gt = s > ls[j];
        const float ns = gt ? ls[j] : s;
        const int ni = gt ? li[j] : i;
        ls[j] = gt ? s : ls[j];
        li[j] = gt ? i : li[j];
        s = ns;
        i = ni;
    }
}

// float <-> int whose signed order equals the float order (for atomic max on scores)
__device__ __forceinline__ int float_to_ordered(float f) {
    const int k = __float_as_int(f);
    return k >= 0 ? k : k ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }
constexpr int K1_GTHR_INIT = (int)0x80808080;  // memset byte 0x80: below every finite score
// bound published by another lane -> strict '>' filter value: the next float BELOW it (keeps equal scores eligible)
// A bound of +0.0 or -0.0 (ordered keys 0 and -1: e.g. an all-zero query scores exactly 0 against everything) must admit
// both zeros, and the float "below" +0.0 in key order is -0.0, which compares EQUAL to it: use the smallest normal
// negative number there (a weaker filter is always safe; it also keeps the test independent of the denormal mode).
__device__ __forceinline__ float import_threshold(int key) {
    if (key <= K1_GTHR_INIT) return -INFINITY;
    if (key == 0 || key == -1) return -1.17549435e-38f;
    return ordered_to_float(key - 1);
}

template <int D, int NWAVES, int QW, bool PAIR = false, int NST = K1_NSTAGE>
constexpr int k1_lds_bytes() {
    return (PAIR ? 4 : NST) * K1_TILE_ROWS * D * 2 + NWAVES * QW * K1_QCAP * 64 * 8;
}

// Measured on MI355X (gpurun r01, N = 1 M, d = 384, k = 10; profiles/README.md):
//   * the same kernel with the selection removed runs Q = 4096 in 2.46 ms and Q = 16384 in 9.2 ms (1.28-1.37 PFLOP/s,
//     ~54 % of the dense bf16 peak): that is the ceiling of this streaming structure;
//   * the fast check (max3 tree + one compare per tile) costs 2-4 %; the slow path (queueing + drains, amplified by
//     the per-tile workgroup barrier: one draining wave stalls eight) costs +33 % at Q = 4096, +24 % at Q = 16384 and
//     3.9x at Q = 256, where per-lane streams are only ~1 k rows;
//   * dropped variants: 4 waves x 64 queries at one wave per SIMD (1.5x slower: nothing hides LDS->MFMA latency);
//     two independent 4-wave workgroups per CU (no gain at Q = 4096, 2x slower at Q = 16384); deferring tile t's
//     filter into tile t+1's MFMA stream with ping-pong accumulators (1.2x slower: splits the stream into basic
//     blocks); refreshing thresholds from the shared word every 16 tiles (no gain).
// QW = query sets (of 32) resident per wave.  QW = 1: 8 waves x 32 queries, two waves per SIMD.  QW = 2: 4 waves x
// 64 queries, one wave per SIMD with the whole 512-register file: every corpus fragment read from LDS feeds two
// MFMAs, halving LDS traffic and per-tile fixed costs.  Both serve 256 queries per workgroup.
// MAXONLY = threshold pre-pass: no lists and no queues, every lane just keeps the maximum score of its sub-stream and
// writes it to part_s[query][partition]; the KL-th largest of a query's block maxima (distinct rows by construction)
// is a valid lower bound of its final KL-th best score (thr_select_kernel), with which the main pass starts.
// PAIR = two tiles per barrier (ring of two 2-tile stages) instead of one tile per barrier (ring of three tiles).
#ifdef TSIM_PP_STAMPS
// DIAGNOSTIC build only (python -m text_similarity_amd.build --stamps; tools/pp_stamps.py --search): cycles of wave 0 of every
// workgroup of the main pass: [0] tile pairs, [1] wait for the pair's DMA, [2] barrier, [3] DMA issue, [4] fragment reads +
// MFMAs + selection, [5] prologue (query fragments, thresholds), [6] epilogue (drain, list write-out), [7] workgroups.
// Ping-pong schedule: [8..15] wave 0 (group 0) and [16..23] wave 4 (group 1) of every workgroup: +0 tiles, +1 L sections
// that carry the filter and the DMA issue, +2 the other L sections, +3 barrier after L, +4 M sections, +5 barrier after M,
// +6 whole loop.
#define K1_NSTAMPS 24
static __device__ unsigned long long g_k1_stamps[K1_NSTAMPS];
#define K1_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define K1_STAMP(v) do { } while (0)
#define K1_NSTAMPS 0
#endif

template <int... I, class F>
__device__ __forceinline__ void k1_static_for(std::integer_sequence<int, I...>, F &&f) {   // f(integral_constant<I>) for each I
    (f(std::integral_constant<int, I>{}), ...);
}
typedef __attribute__((ext_vector_type(4))) uint32_t k1_u32x4;
template <int OFF>
__device__ __forceinline__ void k1_lds_read(k1_u32x4 &dst, uint32_t addr) {   // ds_read_b128 at addr + immediate offset
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void k1_lgkm_wait(k1_u32x4 &consumed) {   // all but the N youngest LDS operations are done
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(consumed) : "n"(N) : "memory");
}

// COLLECT = the widening pass behind K2's exactness guard (search.hip): queries are the FLAGGED ones only (slot s of the
// launch = query coll.qmap[s], *coll.qcount slots in all, known on the device only), every lane filters with the FIXED
// threshold gthr[slot] (no lists, nothing published) and appends every (score, shard row) above it to the slot's buffer
// coll.buf[slot * cap ..] through an atomic counter coll.cnt[slot]; a counter beyond cap means the slot overflowed.
struct K1Collect {
    const int *qcount;           // device: number of slots (<= Q)
    const int *qmap;             // device: slot -> query row
    unsigned long long *buf;     // [Q][cap] entries: score bits | (uint64)(shard row) << 32
    int *cnt;                    // [Q] entries appended per slot (may exceed cap)
    int cap;
    // list kernels launched over a ROW RANGE of the shard (two-phase main pass, search.hip): this launch's lists are
    // p2_base .. of the query's p2_total lists, and its rows start at shard row row_base.  All zero: one launch, whole shard.
    int p2_base, p2_total, row_base;
};

// PP = two-group ping-pong schedule of the tile loop (8 waves, QW = 1): the waves form two groups of four, one wave of each
// group per SIMD, running the SAME program one barrier interval apart.  A wave alternates between M — the tile's 24-MFMA
// stream with its own fragment reads rolling four k-steps ahead — and L — everything else: the selection filter of the tile
// it has just scored and the LDS-DMA issue of the tile after next:
//   group 0 : M(t) | L(t) | M(t+1) | L(t+1) ...
//   group 1 :  -   | M(t) |  L(t)  | M(t+1) ...
// so in every interval exactly one wave per SIMD feeds the matrix pipe and its partner's VALU / branches / DMA issue run
// beside it.  Under the common-barrier schedules the two waves of a SIMD raced instead: the older one ran its MFMAs nearly
// alone, then idled at the barrier while the younger ran with its LDS latencies exposed (5 725 cycles per tile pair for
// 3 072 of MFMA work: profiles/README.md).  (A first form that cut the tile into 12-MFMA sections with register-held
// fragments — L = 12 ds_read_b128 + side work, M = 12 MFMAs — measured 8 % SLOWER than PAIR: its L sections took ~600
// cycles against 384 of MFMA.)
// Ring: three tile slots; tile t+2 is issued in L(t): its slot held tile t-1, last read in group 1's M(t-1), two barriers
// earlier for either group.  Tile t+1 must have landed when group 0 starts M(t+1): group 0 waits for its pieces at the end
// of L(t) (vmcnt(PPW): t+2 stays in flight), group 1 at the end of M(t) (vmcnt(0): it has not issued t+2 yet).
// M16 = the score tile is computed with v_mfma_f32_16x16x32_f16 (four 16x16 accumulators: query set qs x row set rs) instead
// of one v_mfma_f32_32x32x16_f16: the same matrix-pipe time and the same LDS fragment traffic (one ds_read_b128 per two
// instructions), but 9-10 % faster in this kernel (measured with the selection disabled: 2.72 -> 2.47 ms at Q = 4096; the
// 16x16 shape draws less power per FLOP and the chip holds a higher clock, MI355X_MICROARCH.md).  In the 16x16 result layout a
// lane holds TWO queries (lane&15 of each set) against 8 rows each; eight v_permlane32_swap per tile hand the upper half-wave's
// set-0 values to the lower half-wave and the lower's set-1 values to the upper, after which every lane again owns ONE query
// (set lane>>5, column lane&15) against 16 rows, in exactly the register -> row pattern of the 32x32 layout with the
// row-half bit h = (lane>>4)&1.  Selection, queues, lists and the output format are unchanged.
// NST = ring slots of the one-tile-per-barrier schedule (tiles t+1 .. t+NST-2 are in flight while tile t is scored).  Three
// slots keep 48 KiB per CU in flight, which bounds a latency-limited stream (few query blocks: one 256-query block reaches
// ~4.2 TB/s with the matrix pipe half idle); five slots double that.
template <int D, int NWAVES, int QW, int KL, bool MAXONLY, bool PAIR, bool COLLECT = false, bool PP = false, bool M16 = false,
          int NST = K1_NSTAGE>
__global__ __launch_bounds__(NWAVES * 64) void cos_topk_partial_kernel(
    const unit_t *__restrict__ eq, int Q, const unit_t *__restrict__ ec, int64_t N, int rows_per_chunk,
    int nchunks, int nqb, int *__restrict__ gthr, float *__restrict__ part_s,
    int *__restrict__ part_i, K1Collect coll) {
#ifdef TSIM_PP_STAMPS
    const unsigned long long ks_tk = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;                        // bytes per corpus row
    constexpr int STAGE_BYTES = K1_TILE_ROWS * ROWB;   // 24 KiB at D = 384
    constexpr int KSTEPS = D / 16;
    constexpr int CH16 = D / 8;                        // 16-byte chunks per row
    constexpr int PIECES = STAGE_BYTES / 1024;         // 1-KiB LDS-DMA wave-instructions per stage
    static_assert(PIECES % NWAVES == 0, "stage must split evenly over the waves");
    constexpr int PPW = PIECES / NWAVES;
    constexpr int QPW = 32 * QW;                       // queries per wave

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(!M16 || (KL <= 16 && !PP && D % 32 == 0), "16x16x32 form: asm-pipelined tile loop");
    // r = the wave's query this lane selects for, h = which half of the tile's 8-row groups it sees (its list = partition h
    // of the chunk); c16 / g16 = column and k-group of the lane in the 16x16x32 operand layout
    const int c16 = lane & 15, g16 = lane >> 4;
    const int r = M16 ? c16 + 16 * (lane >> 5) : lane & 31, h = M16 ? (g16 & 1) : lane >> 5, x = lane & 15;

    // blockIdx -> (query block, corpus chunk).  Blocks b and b+8 share an XCD (round-robin dispatch), so
    // the nqb query blocks of one chunk are given the same b%8: they stream the same corpus rows at about
    // the same time and share them through that XCD's L2.  Speed only; any placement is correct.
    const int b = blockIdx.x;
    const int xcd = b & 7, jj = b >> 3;
    int qb, chunk;
    if (nchunks >= 8) {
        qb = jj % nqb;
        chunk = (jj / nqb) * 8 + xcd;
    } else {   // fewer chunks than XCDs (many query blocks): 8/nchunks XCDs share a chunk and split the query blocks
        const int per = 8 / nchunks;            // plan_topk makes nchunks 1, 2 or 4; any value < 8 stays correct
        chunk = xcd < per * nchunks ? xcd % nchunks : nchunks;   // (XCD labels beyond per*nchunks idle)
        qb = jj * per + xcd / nchunks;
    }
    if (chunk >= nchunks || qb >= nqb) return;
    if constexpr (COLLECT) {   // the slot count lives on the device: query blocks past it leave (workgroup-uniform)
        const int qe = *coll.qcount;
        Q = qe < Q ? qe : Q;
        if (qb * (NWAVES * 32 * QW) >= Q) return;
    }

    const int64_t row0 = (int64_t)chunk * rows_per_chunk;
    const int crows = (int)(((N - row0) < (int64_t)rows_per_chunk) ? (N - row0) : (int64_t)rows_per_chunk);
    const int ntiles = (crows + K1_TILE_ROWS - 1) / K1_TILE_ROWS;
    const int q0 = qb * (NWAVES * QPW) + wave * QPW;
    const bool wave_on = q0 < Q;  // wave-uniform: waves past the last query only help staging

    // ---- resident query fragments: B[k = 8h + j][col r] of k-step s = eq[q0 + 32u + r][16 s + 8 h + j]
    f16x8 bq[QW][KSTEPS];
    float thr[QW];
    int *gt[QW];   // this lane's query's shared threshold word
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        const int qrow = (q0 + 32 * u + r < Q) ? (q0 + 32 * u + r) : (Q - 1);
        int qsrc = qrow;
        if constexpr (COLLECT) qsrc = wave_on ? coll.qmap[qrow] : 0;
        if constexpr (M16) {
            // B operand of 16x16x32, query set qs, k-step s (32 wide): column c16 = query q0 + 16 qs + c16, k = 32 s + 8 g16 + j.
            // Every lane carries fragments of BOTH sets: bq[0][qs * KSTEPS/2 + s].  (qrow / thresholds: the lane's own query r.)
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                int qf = q0 + 32 * u + 16 * qs + c16;
                qf = qf < Q ? qf : Q - 1;
                if constexpr (COLLECT) qf = wave_on ? coll.qmap[qf] : 0;
                const unit_t *qp = eq + (int64_t)qf * D + 8 * g16;
#pragma unroll
                for (int s = 0; s < KSTEPS / 2; ++s) bq[u][qs * (KSTEPS / 2) + s] = *reinterpret_cast<const f16x8 *>(qp + 32 * s);
            }
        } else {
            const unit_t *qp = eq + (int64_t)qsrc * D + 8 * h;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) bq[u][s] = *reinterpret_cast<const f16x8 *>(qp + 16 * s);
        }
        if constexpr (MAXONLY) {   // pre-pass: no thresholds (gthr is null)
            gt[u] = nullptr;
            thr[u] = -INFINITY;
        } else {
            gt[u] = gthr + qrow;
            thr[u] = import_threshold(__hip_atomic_load(gt[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if constexpr (COLLECT)   // lanes past the last slot hold a copy of it: they must never queue anything
                if (q0 + 32 * u + r >= Q) thr[u] = INFINITY;
#ifdef TSIM_K1_NOSEL
            thr[u] = INFINITY;   // TIMING-ONLY diagnostic: the filter runs but nothing ever passes it
#endif
        }
    }
    // Make the compiler retire these ordinary loads HERE: inside the main loop only LDS-DMA is in flight
    // and is waited for with counted vmcnt (cdna_hip_programming.md §5, "Three .s-level traps" (b)).
#pragma unroll
    for (int u = 0; u < QW; ++u) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(bq[u][s]));
        asm volatile("" : "+v"(thr[u]));
    }

    // ---- LDS-DMA source offsets.  LDS image of a stage = the 32 rows back to back (row-major, 16-byte
    // slots), slot c of row rr holding source chunk c ^ (rr & 15): the XOR makes the ds_read_b128 of the
    // A fragment (32 lanes = 32 rows, same chunk) conflict-free while every 256-byte source segment is
    // still read whole.  LDS-DMA writes lane-linearly, so the permutation goes on the SOURCE address.
    int src_row[PPW], src_off[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int slot = (wave * PPW + i) * 64 + lane;
        const int rr = slot / CH16, cc = slot % CH16;
        src_row[i] = rr;
        src_off[i] = (cc ^ (rr & 15)) * 16;
    }
    const char *src_base[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        src_base[i] = reinterpret_cast<const char *>(ec) + (row0 + src_row[i]) * ROWB + src_off[i];
    const int full_tiles = crows / K1_TILE_ROWS;   // tiles whose 32 rows all exist
    auto issue_tile = [&](int t, int stage) __attribute__((always_inline))  {
        const int tt = t < ntiles ? t : ntiles - 1;  // past-the-end tiles re-read the last one: keeps vmcnt uniform
        if (tt < full_tiles) {
            // tiles are contiguous in the corpus: per-lane base pointer + t * STAGE_BYTES
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                glds16(src_base[i] + (int64_t)tt * STAGE_BYTES, smem + stage * STAGE_BYTES + (wave * PPW + i) * 1024);
            return;
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int64_t gr = row0 + (int64_t)tt * K1_TILE_ROWS + src_row[i];
            gr = gr < N ? gr : N - 1;
            const char *src = reinterpret_cast<const char *>(ec) + gr * ROWB + src_off[i];
            glds16(src, smem + stage * STAGE_BYTES + (wave * PPW + i) * 1024);
        }
    };

    // A-fragment read offsets: lane (r,h), k-step s reads chunk (2s+h) of row r -> slot (2s+h) ^ x.
    // (2s+h) ^ x = 16*(s>>3) + ((2*(s&7)+h) ^ x): eight base offsets + an immediate.
    // 16x16x32: lane (c16, g16), row set rs, k-step s reads chunk 4s + g16 of row 16 rs + c16 -> slot (4s + g16) ^ c16
    // = 16*(s>>2) + ((4*(s&3) + g16) ^ c16): four base offsets + an immediate (conflict-free in every ds_read_b128 lane group).
    int aoff[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb)
        aoff[bb] = M16 ? c16 * ROWB + (((4 * (bb & 3) + g16) ^ c16) << 4) : (lane & 31) * ROWB + (((2 * bb + (lane >> 5)) ^ x) << 4);

    float ls[QW][KL];
    int li[QW][KL];
    int cnt[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        cnt[u] = 0;
#pragma unroll
        for (int j = 0; j < KL; ++j) {
            ls[u][j] = -INFINITY;
            li[u][j] = -1;
        }
    }
    // Per-lane candidate queues in LDS (one per query set): entry p of this lane at qaddr + p*512 (lanes
    // interleaved, 8 B each).  Accessed with inline asm: hipcc would otherwise drain the LDS-DMA ring
    // (s_waitcnt vmcnt(0)) before every ordinary LDS access that might alias it; the queues never overlap the
    // staging buffers.
    const uint32_t qaddr0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) +
                            (PAIR ? 4 : NST) * STAGE_BYTES + wave * (QW * K1_QCAP * 64 * 8) + lane * 8;

    auto drain = [&](auto uc) __attribute__((always_inline))  {
        constexpr int u = decltype(uc)::value;
        const uint32_t qaddr = qaddr0 + u * (K1_QCAP * 64 * 8);
        if constexpr (COLLECT) {
            const int slot = q0 + 32 * u + r;   // lanes past Q never queue (thr = +inf): cnt stays 0 for them
            const int n = cnt[u];
            int pos = 0;
            if (n > 0) pos = atomicAdd(coll.cnt + slot, n);
#pragma unroll 1
            for (int p = 0; p < K1_QCAP; ++p) {
                if (!__any(p < cnt[u])) break;
                uint64_t e;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(qaddr + p * 512) : "memory");
                if (p < n && pos + p < coll.cap)   // queue rows are chunk-relative: make them shard rows
                    coll.buf[(int64_t)slot * coll.cap + pos + p] = e + ((uint64_t)(uint32_t)row0 << 32);
            }
            cnt[u] = 0;
            return;
        }
#pragma unroll 1
        for (int p = 0; p < K1_QCAP; ++p) {
            if (!__any(p < cnt[u])) break;
            uint64_t e;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(qaddr + p * 512) : "memory");
            const float s = (p < cnt[u]) ? __uint_as_float((uint32_t)e) : -INFINITY;
            // entries were queued against an older (lower) threshold: most no longer beat the list tail
            if (__any(s > ls[u][KL - 1])) list_insert<KL>(ls[u], li[u], s, (int)(e >> 32));
        }
        cnt[u] = 0;
        thr[u] = fmaxf(thr[u], ls[u][KL - 1]);
        // Publish this lane's KL-th best and adopt the best bound any workgroup has published for the query:
        // KL elements of the corpus score >= that bound, so nothing strictly below it can reach the query's final
        // list.  A foreign bound is imported one ulp lower: rows of other lanes/workgroups are not "later in index
        // order", so an equal score may still win its tie on the index.  Stale reads only make the filter weaker.
        if (ls[u][KL - 1] > -INFINITY)
            (void)__hip_atomic_fetch_max(gt[u], float_to_ordered(ls[u][KL - 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        thr[u] = fmaxf(thr[u], import_threshold(__hip_atomic_load(gt[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    };

    float bmax[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) bmax[u] = -INFINITY;

    auto filter = [&](auto uc, f32x16 &acc, int t) __attribute__((always_inline))  {
        constexpr int u = decltype(uc)::value;
        const uint32_t qaddr = qaddr0 + u * (K1_QCAP * 64 * 8);
        const int trow = t * K1_TILE_ROWS + 4 * h;  // local (chunk-relative) row of acc[0]
        if ((t + 1) * K1_TILE_ROWS > crows) {       // ragged last tile: rows past the chunk never compete
#pragma unroll
            for (int g = 0; g < 16; ++g)
                if (trow + (g & 3) + 8 * (g >> 2) >= crows) acc[g] = -INFINITY;
        }
        // maxima of the four 4-register groups first (four independent chains), then their maximum: the rare candidate path
        // below only scans the groups that hold one (a wave ballot per group, then per register of a hit group, instead of
        // one per register)
        float mg[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            mg[gq] = fmaxf(fmaxf(fmaxf(acc[4 * gq], acc[4 * gq + 1]), acc[4 * gq + 2]), acc[4 * gq + 3]);
        const float m = fmaxf(fmaxf(fmaxf(mg[0], mg[1]), mg[2]), mg[3]);
        if constexpr (MAXONLY) {
            bmax[u] = fmaxf(bmax[u], m);
            return;
        }
        if (__any(m > thr[u])) {
#ifdef TSIM_PP_STAMPS
            if (threadIdx.x == 0) atomicAdd(&g_k1_stamps[7 + 8], 1ull);   // [15]: candidate events seen by wave 0
#endif
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#ifndef TSIM_K1_FLAT_SCAN
                if (!__any(mg[gq] > thr[u])) continue;
#endif
#pragma unroll
                for (int g = 4 * gq; g < 4 * gq + 4; ++g) {
                    const bool p = acc[g] > thr[u];
                    if (__any(p)) {
                        if (p) {
                            const uint64_t e = (uint64_t)__float_as_uint(acc[g]) |
                                               ((uint64_t)(uint32_t)(trow + (g & 3) + 8 * (g >> 2)) << 32);
                            asm volatile("ds_write_b64 %0, %1" ::"v"(qaddr + cnt[u] * 512), "v"(e) : "memory");
                            cnt[u]++;
                        }
                        if (__any(cnt[u] == K1_QCAP)) drain(uc);
                    }
                }
            }
        }
    };

    issue_tile(0, 0);
    issue_tile(1, 1);
    if constexpr (!PAIR && !PP)
#pragma unroll
        for (int i = 2; i < NST - 1; ++i) issue_tile(i, i);

    auto compute_tile = [&](int t, int stage) __attribute__((always_inline))  {
        if (!wave_on) return;
        const char *abase = smem + stage * STAGE_BYTES;
        f32x16 acc[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[u][g] = 0.f;
        if constexpr (M16) {
            // 24 fragment reads per tile (k-step s = n>>1, row set rs = n&1), each feeding 2 QW 16x16x32 instructions (the wave's
            // 2 QW query sets of 16); reads roll PF ahead with counted lgkmcnt waits, as in the 32x32 form below.  With QW = 2
            // (four waves, one per SIMD) a wave owns the SIMD's matrix pipe: 8 independent accumulation chains, 4 instructions per
            // fragment read.
#ifndef TSIM_K1_PF16
#define TSIM_K1_PF16 4
#endif
            constexpr int PF = TSIM_K1_PF16;
            constexpr int NRD = KSTEPS;   // (D/32) k-steps x 2 row sets
            k1_u32x4 fr[PF + 1];
            f32x4 a16[2 * QW][2];         // [query set][row set]
#pragma unroll
            for (int i = 0; i < 4 * QW; ++i) a16[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint32_t lbase = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + stage * STAGE_BYTES;
            auto rd = [&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                constexpr int ks = n >> 1, rs = n & 1;
                k1_lds_read<(ks >> 2) * 256 + rs * 16 * ROWB>(fr[n % (PF + 1)], lbase + aoff[ks & 3]);
            };
            auto step = [&](auto sc) __attribute__((always_inline)) {
                constexpr int n = decltype(sc)::value;
                constexpr int ks = n >> 1, rs = n & 1;
                if constexpr (n + PF < NRD) rd(std::integral_constant<int, n + PF>{});
                constexpr int younger = n + PF < NRD ? PF : NRD - 1 - n;
                k1_lgkm_wait<younger>(fr[n % (PF + 1)]);
#pragma unroll
                for (int qs = 0; qs < 2 * QW; ++qs)
                    a16[qs][rs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fr[n % (PF + 1)]),
                                                                          bq[qs >> 1][(qs & 1) * (KSTEPS / 2) + ks], a16[qs][rs], 0, 0, 0);
            };
            k1_static_for(std::make_integer_sequence<int, PF>{}, rd);
            k1_static_for(std::make_integer_sequence<int, NRD>{}, step);
            // a16[qs][rs][j] = query (16 qs + c16) x tile row (16 rs + 4 g16 + j).  Swap within each pair of sets: lanes 0-31 keep
            // their even-set values and receive the even-set values of lane+32 (rows 4 (g16+2) + j); lanes 32-63 receive the
            // odd-set values of lane-32 (rows 4 (g16-2) + j) and keep their own.  For every lane the first result is then the
            // 8m = 0 group and the second the 8m = 8 group of its query: acc[4 (2 rs + m) + j] = row 16 rs + 8 m + 4 h + j, the
            // 32x32 layout.
#pragma unroll
            for (int u = 0; u < QW; ++u)
#pragma unroll
                for (int rs = 0; rs < 2; ++rs)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a16[2 * u][rs][j]),
                                                                   __float_as_uint(a16[2 * u + 1][rs][j]), false, false);
                        acc[u][8 * rs + j] = __uint_as_float(sw[0]);
                        acc[u][8 * rs + 4 + j] = __uint_as_float(sw[1]);
                    }
        } else if constexpr (QW == 1) {
            if constexpr (KL <= 16) {
                // rolling software pipeline: the fragment read of k-step s+PF is issued right before the MFMA of k-step s, so an
                // LDS read has PF MFMAs (PF*32 pipe cycles) to land.  Reads and waits are inline asm with COUNTED waits
                // (lgkmcnt(PF): the PF younger reads stay in flight): hipcc's own schedule waited lgkmcnt(0) every fifth MFMA,
                // i.e. for the read it had just issued.
#ifndef TSIM_K1_PF
#define TSIM_K1_PF 4
#endif
                constexpr int PF = TSIM_K1_PF;
                k1_u32x4 fr[PF + 1];
                const uint32_t lbase = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + stage * STAGE_BYTES;
                auto rd = [&](auto nc) __attribute__((always_inline)) {
                    constexpr int n = decltype(nc)::value;
                    k1_lds_read<(n >> 3) * 256>(fr[n % (PF + 1)], lbase + aoff[n & 7]);
                };
                auto step = [&](auto sc) __attribute__((always_inline)) {
                    constexpr int sidx = decltype(sc)::value;
                    if constexpr (sidx + PF < KSTEPS) rd(std::integral_constant<int, sidx + PF>{});
                    constexpr int younger = sidx + PF < KSTEPS ? PF : KSTEPS - 1 - sidx;
                    k1_lgkm_wait<younger>(fr[sidx % (PF + 1)]);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[sidx % (PF + 1)]), bq[0][sidx],
                                                                     acc[0], 0, 0, 0);
                };
                k1_static_for(std::make_integer_sequence<int, PF>{}, rd);
                k1_static_for(std::make_integer_sequence<int, KSTEPS>{}, step);
            } else {   // KL = 32: 64 list registers leave no room for the asm pipeline without spills; compiler-scheduled form
                // rolling software pipeline: the fragment read of k-step s+PF is issued right before the MFMA of k-step
                // s, so an LDS read has PF MFMAs (PF*32 pipe cycles) to land; sched_group_barrier pins the interleave and
                // the compiler's counted lgkmcnt waits follow from it.
                constexpr int PF = 4;
                f16x8 fr[PF + 1];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    fr[i] = *reinterpret_cast<const f16x8 *>(abase + aoff[i & 7] + (i >> 3) * 256);
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    if (s + PF < KSTEPS) {
                        const int n = s + PF;
                        fr[n % (PF + 1)] = *reinterpret_cast<const f16x8 *>(abase + aoff[n & 7] + (n >> 3) * 256);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[s % (PF + 1)], bq[0][s], acc[0], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const f16x8 a = *reinterpret_cast<const f16x8 *>(abase + aoff[s & 7] + (s >> 3) * 256);
#pragma unroll
                for (int u = 0; u < QW; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[u][s], acc[u], 0, 0, 0);
            }
        }
        filter(std::integral_constant<int, 0>{}, acc[0], t);
        if constexpr (QW > 1) filter(std::integral_constant<int, 1>{}, acc[QW - 1], t);
    };

#ifdef TSIM_PP_STAMPS
    unsigned long long ks_n = 0, ks_dma = 0, ks_bar = 0, ks_iss = 0, ks_cmp = 0;
    const unsigned long long ks_t0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (PP) {
        static_assert(!PAIR && QW == 1 && NWAVES == 8 && KL <= 16, "ping-pong schedule: 8 waves, one query set per wave");
        const int grp = wave >> 2;
        const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
        uint32_t abase[8];
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) abase[bb] = lds0 + aoff[bb];
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        issue_tile(2, 2);
        wait_vmcnt<2 * PPW>();          // my pieces of tile 0 (tiles 1 and 2 stay in flight)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one interval behind group 0
#ifdef TSIM_PP_STAMPS
        unsigned long long pl0 = 0, pl1 = 0, pb1 = 0, pm = 0, pb2 = 0;
        const unsigned long long pp_t0 = __builtin_amdgcn_s_memtime();
#endif
        auto do_tile = [&](int t, auto stc) __attribute__((always_inline)) {
            constexpr int stage = decltype(stc)::value;
            K1_STAMP(q0);
            // ---------------- M: the tile's MFMA stream with its own fragment reads rolling PF k-steps ahead (counted lgkmcnt)
            if (wave_on) {
#ifndef TSIM_K1_PP_PF
#define TSIM_K1_PP_PF 8
#endif
                // alone on the SIMD's matrix pipe, the stream is paced by LDS latency / PF: 4 reads in flight gave 42 cycles
                // per k-step (1 000 per tile for 768 of MFMA work), so the reads roll 8 k-steps (256 pipe cycles) ahead here
                constexpr int PF = TSIM_K1_PP_PF;
                k1_u32x4 fr[PF + 1];
                auto rd = [&](auto nc) __attribute__((always_inline)) {
                    constexpr int n = decltype(nc)::value;
                    k1_lds_read<stage * STAGE_BYTES + (n >> 3) * 256>(fr[n % (PF + 1)], abase[n & 7]);
                };
                __builtin_amdgcn_s_setprio(1);
                k1_static_for(std::make_integer_sequence<int, PF>{}, rd);
                k1_static_for(std::make_integer_sequence<int, KSTEPS>{}, [&](auto sc) __attribute__((always_inline)) {
                    constexpr int sidx = decltype(sc)::value;
                    if constexpr (sidx + PF < KSTEPS) rd(std::integral_constant<int, sidx + PF>{});
                    constexpr int younger = sidx + PF < KSTEPS ? PF : KSTEPS - 1 - sidx;
                    k1_lgkm_wait<younger>(fr[sidx % (PF + 1)]);
                    if constexpr (sidx == 0)   // srcC = inline constant 0: no register clear per tile
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[0]), bq[0][0], f32x16{}, 0, 0, 0);
                    else
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[sidx % (PF + 1)]), bq[0][sidx],
                                                                     acc, 0, 0, 0);
                });
                asm volatile("" : "+v"(acc));      // keep the stream inside its barrier interval
                __builtin_amdgcn_s_setprio(0);
            }
            if (grp == 1) wait_vmcnt<0>();         // group 1: its pieces of tile t+1 are all it has in flight here
            K1_STAMP(q1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            K1_STAMP(q2);
            // ---------------- L: selection filter of the tile just scored + LDS-DMA issue of tile t+2, beside the partner's M.
            // Slot (stage+2)%3 held tile t-1: last read in group 1's M(t-1), two barriers ago for either group.
            if (wave_on) filter(std::integral_constant<int, 0>{}, acc, t);
            K1_STAMP(q2f);
            if (t >= 1) issue_tile(t + 2, (stage + 2) % 3);
            if (grp == 0) wait_vmcnt<PPW>();       // group 0: my pieces of tile t+1 (t+2 stays in flight)
            K1_STAMP(q3);
#ifdef TSIM_PP_STAMPS
            pl1 += q2f - q2;
#endif
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef TSIM_PP_STAMPS
            {
                const unsigned long long q4 = __builtin_amdgcn_s_memtime();
                pm += q1 - q0; pb1 += q2 - q1; pl0 += q3 - q2; pb2 += q4 - q3;
            }
#endif
        };
        int t = 0;
        for (; t + 3 <= ntiles; t += 3) {
            do_tile(t, std::integral_constant<int, 0>{});
            do_tile(t + 1, std::integral_constant<int, 1>{});
            do_tile(t + 2, std::integral_constant<int, 2>{});
        }
        if (t < ntiles) do_tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntiles) do_tile(t + 1, std::integral_constant<int, 1>{});
        if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's first barrier
#ifdef TSIM_PP_STAMPS
        (void)pl1;
        if (!MAXONLY && !COLLECT && (threadIdx.x == 0 || threadIdx.x == 256)) {
            const int o = threadIdx.x == 0 ? 8 : 16;
            atomicAdd(&g_k1_stamps[o + 0], (unsigned long long)ntiles); atomicAdd(&g_k1_stamps[o + 1], pl0);
            atomicAdd(&g_k1_stamps[o + 2], pl1); atomicAdd(&g_k1_stamps[o + 3], pb1); atomicAdd(&g_k1_stamps[o + 4], pm);
            atomicAdd(&g_k1_stamps[o + 5], pb2); atomicAdd(&g_k1_stamps[o + 6], __builtin_amdgcn_s_memtime() - pp_t0);
        }
#endif
    } else if constexpr (PAIR) {
        // stage = two tiles; stages alternate between slots {0,1} and {2,3}.  At the barrier of pair p everyone has
        // finished pair p-1, whose slots are exactly those of pair p+1, which is then issued and has one pair-time
        // (~48 MFMAs per wave) to land.
        auto do_pair = [&](int t, int slot) __attribute__((always_inline))  {
            K1_STAMP(s0);
            wait_vmcnt<0>();
            K1_STAMP(s1);
            __builtin_amdgcn_s_barrier();
            K1_STAMP(s2);
            // STAGGER (MI355X_MICROARCH.md "Two waves per SIMD" item 9): the two waves of a SIMD run the same program and would
            // reach their MFMA streams, their LDS bursts and the barrier together; waves 4-7 (the partners of 0-3) issue the
            // block's LDS-DMA BETWEEN its two tiles instead of at its head, which shifts their MFMA streams by the length of
            // the issue (~440 cycles) against their partners'.
#ifndef TSIM_K1_STAGGER
#define TSIM_K1_STAGGER 1
#endif
            const bool late = TSIM_K1_STAGGER != 0 && wave >= 4;
            if (!late) {
                issue_tile(t + 2, (slot + 2) & 3);
                issue_tile(t + 3, (slot + 3) & 3);
            }
            K1_STAMP(s3);
            compute_tile(t, slot);
            if (late) {
                issue_tile(t + 2, (slot + 2) & 3);
                issue_tile(t + 3, (slot + 3) & 3);
            }
            if (t + 1 < ntiles) compute_tile(t + 1, slot + 1);
#ifdef TSIM_PP_STAMPS
            {
                const unsigned long long s4 = __builtin_amdgcn_s_memtime();
                ks_n += 1; ks_dma += s1 - s0; ks_bar += s2 - s1; ks_iss += s3 - s2; ks_cmp += s4 - s3;
            }
#endif
        };
        int t = 0;
        for (; t + 4 <= ntiles; t += 4) {
            do_pair(t, 0);
            do_pair(t + 2, 2);
        }
        if (t < ntiles) do_pair(t, 0);
        if (t + 2 < ntiles) do_pair(t + 2, 2);
    } else {
        static_assert(NST >= 3 && (NST - 2) * PPW <= 63, "ring depth");
        auto do_tile = [&](int t, int stage) __attribute__((always_inline))  {
            wait_vmcnt<(NST - 2) * PPW>();     // my pieces of tile t have landed (tiles t+1 .. t+NST-2 may be in flight)
            __builtin_amdgcn_s_barrier();      // everyone's pieces landed; everyone is done reading tile t-1
            issue_tile(t + NST - 1, (stage + NST - 1) % NST);   // into the slot of tile t-1
            compute_tile(t, stage);
        };
        int t = 0;
        for (; t + NST <= ntiles; t += NST)
            k1_static_for(std::make_integer_sequence<int, NST>{}, [&](auto sc) __attribute__((always_inline)) {
                do_tile(t + decltype(sc)::value, decltype(sc)::value);
            });
        k1_static_for(std::make_integer_sequence<int, NST - 1>{}, [&](auto sc) __attribute__((always_inline)) {
            if (t + decltype(sc)::value < ntiles) do_tile(t + decltype(sc)::value, decltype(sc)::value);
        });
    }
    wait_vmcnt<0>();  // no LDS-DMA may outlive the workgroup
#ifdef TSIM_PP_STAMPS
    const unsigned long long ks_t1 = __builtin_amdgcn_s_memtime();
#endif

    if (wave_on) {
        const int P2 = coll.p2_total > 0 ? coll.p2_total : nchunks * 2;
        const int pfirst = coll.p2_base;
        const int base = (int)row0 + coll.row_base;  // local -> shard row index (N < 2^31 enforced by the host)
        auto flush = [&](auto uc) __attribute__((always_inline))  {
            constexpr int u = decltype(uc)::value;
            if constexpr (MAXONLY) {
                if (q0 + 32 * u + r < Q) part_s[(int64_t)(q0 + 32 * u + r) * P2 + pfirst + chunk * 2 + h] = bmax[u];
                return;
            }
            drain(uc);
            if constexpr (COLLECT) return;
            if (q0 + 32 * u + r < Q) {
                const int64_t o = ((int64_t)(q0 + 32 * u + r) * P2 + pfirst + chunk * 2 + h) * KL;
#pragma unroll
                for (int j = 0; j < KL; j += 4) {
                    *reinterpret_cast<float4 *>(part_s + o + j) =
                        make_float4(ls[u][j], ls[u][j + 1], ls[u][j + 2], ls[u][j + 3]);
                    int4 iv;
                    iv.x = li[u][j] < 0 ? -1 : li[u][j] + base;
                    iv.y = li[u][j + 1] < 0 ? -1 : li[u][j + 1] + base;
                    iv.z = li[u][j + 2] < 0 ? -1 : li[u][j + 2] + base;
                    iv.w = li[u][j + 3] < 0 ? -1 : li[u][j + 3] + base;
                    *reinterpret_cast<int4 *>(part_i + o + j) = iv;
                }
            }
        };
        flush(std::integral_constant<int, 0>{});
        if constexpr (QW > 1) flush(std::integral_constant<int, 1>{});
    }
#ifdef TSIM_PP_STAMPS
    if constexpr (PAIR && !MAXONLY) {
        if (threadIdx.x == 0) {
            const unsigned long long ks_t2 = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_k1_stamps[0], ks_n); atomicAdd(&g_k1_stamps[1], ks_dma); atomicAdd(&g_k1_stamps[2], ks_bar);
            atomicAdd(&g_k1_stamps[3], ks_iss); atomicAdd(&g_k1_stamps[4], ks_cmp);
            atomicAdd(&g_k1_stamps[5], ks_t0 - ks_tk); atomicAdd(&g_k1_stamps[6], ks_t2 - ks_t1); atomicAdd(&g_k1_stamps[7], 1ull);
        }
    }
#endif
}

struct TopkPlan {
    int nqb, nchunks, rows_per_chunk, P2, KL, qpb /* queries per workgroup */, variant;
    size_t part_elems;
};

// Pre-pass plan: block maxima over the first S rows.  S = 64 K rows (32 K when there are >= 4 query blocks), cut into
// enough chunks to fill the chip; returns false when the corpus is too small for a pre-pass to pay.
static inline bool plan_prepass(int64_t Q, int64_t N, const TopkPlan &mainp, TopkPlan *p) {
    if (N < 262144) return false;
#ifndef TSIM_K1_PREPASS_ROWS
#define TSIM_K1_PREPASS_ROWS 32768
#endif
    const int64_t S = mainp.nqb >= 4 ? TSIM_K1_PREPASS_ROWS : 65536;
    int nch = (512 / mainp.nqb) & ~7;   // a multiple of 8: the same number of chunks on every XCD (block mapping of the kernel)
    if (nch < 16) nch = 16;
    if (nch > S / 64) nch = (int)(S / 64);
    *p = mainp;
    p->nchunks = nch;
    p->rows_per_chunk = (int)(S / nch);
    p->P2 = 2 * nch;
    p->part_elems = (size_t)Q * p->P2;
    return p->P2 >= 2 * mainp.KL;
}
constexpr int K1_PREPASS_MAX_P2 = 2048;

static inline int plan_topk(int64_t Q, int64_t N, int D, int k, TopkPlan *p) {
    p->KL = k <= 12 ? 16 : 32;
    p->variant = 1;
    p->qpb = D <= 384 ? 256 : 128;
    const int qpb = p->qpb;
    p->nqb = (int)((Q + qpb - 1) / qpb);
    // One workgroup is resident per CU (128 KiB of LDS) and workgroup b runs on XCD b % 8 (32 CUs), so a pass runs in rounds
    // of 32 workgroups PER XCD and a nearly empty last round costs a full one.  The block mapping of the kernel gives XCD x the
    // chunks x, x + 8, ... with all their query blocks (>= 8 chunks), or lets 8 / nchunks XCDs share a chunk and split its
    // query blocks (< 8 chunks).  Choose the number of chunks that minimises rounds / chunks (time in units of one workgroup's
    // pass over the whole shard), and among near-equal choices the one with the fewest chunks and rounds: the per-lane selection
    // cost falls with stream length (candidates ~ KL*ln(n/KL); 512 workgroups measured 3-9 % slower than 256 at Q <= 1 024).
    // Examples: 16 query blocks -> 16 chunks (2 per XCD x 16 = 32 workgroups per XCD); 5 -> 48 (the old ceil(256 / 5) = 52 put
    // 7 x 5 = 35 workgroups on XCDs 0-3: two rounds, 1.36 ms instead of 0.85 at Q = 1 280); 9 -> 56 (63 per XCD, two rounds of
    // 56 chunks instead of two rounds of 29); 157 -> 8 (five rounds for 4.9 of work).
    int64_t per_xcd = 32;
    {   // A/B knob: workgroups per round the plan assumes (256 = 32 per XCD)
        static int env_target = -1;
        if (env_target < 0) { const char *e = getenv("TSIM_K1_TARGET_WGS"); env_target = e ? atoi(e) : 0; }
        if (env_target >= 8) per_xcd = env_target / 8;
    }
    const int64_t max_ch = (N + 255) / 256;  // at least 256 rows per chunk
    int64_t nch = 1;
    double best = 1e30;
    for (int64_t cand = 1; cand <= 256 && cand <= max_ch; ++cand) {
        if (cand < 8 && (cand & (cand - 1)) != 0) continue;   // below 8 chunks whole XCDs share a chunk: powers of two only
        const int64_t wg_xcd = cand >= 8 ? ((cand + 7) / 8) * p->nqb : (p->nqb + (8 / cand) - 1) / (8 / cand);
        const int64_t rounds = (wg_xcd + per_xcd - 1) / per_xcd;
        // more rounds of shorter streams must pay for themselves: +6 % per doubling of the round count
        const double cost = (double)rounds / (double)cand * (1.0 + 0.06 * log2((double)rounds));
        if (cost < best * 0.985) {
            best = cost;
            nch = cand;
        }
    }
    int64_t rpc = (N + nch - 1) / nch;
    rpc = (rpc + K1_TILE_ROWS - 1) / K1_TILE_ROWS * K1_TILE_ROWS;
    p->rows_per_chunk = (int)rpc;
    p->nchunks = (int)((N + rpc - 1) / rpc);
    p->P2 = p->nchunks * 2;
    p->part_elems = (size_t)Q * p->P2 * p->KL;
    return 0;
}

template <int D, int NWAVES, int QW, int KL, bool MAXONLY = false, bool PAIR = false, bool COLLECT = false, bool PP = false,
          bool M16 = false, int NST = K1_NSTAGE>
static int launch_k1(const TopkPlan &p, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                     float *part_s, int *part_i, int *gthr, hipStream_t st, K1Collect coll = K1Collect{}) {
    constexpr int lds = k1_lds_bytes<D, NWAVES, QW, PAIR, NST>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = cos_topk_partial_kernel<D, NWAVES, QW, KL, MAXONLY, PAIR, COLLECT, PP, M16, NST>;
    // the > 64 KiB dynamic-LDS opt-in is per device (a process may drive several GPUs): set it once per device
    static DevOnce lds_once;
    TSIM_MAX_LDS(lds_once, kern, lds);
    const int grid = p.nchunks >= 8 ? ((p.nchunks + 7) / 8) * 8 * p.nqb
                                    : 8 * ((p.nqb + (8 / p.nchunks) - 1) / (8 / p.nchunks));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, st, eq, (int)Q, ec, N, p.rows_per_chunk,
                       p.nchunks, p.nqb, gthr, part_s, part_i, coll);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

// the headline shape (D = 384, KL = 16 main pass) lives in its own translation unit, k1_d384.hip: it is the kernel under
// tuning, and one instantiation rebuilds in a minute instead of five
int k1_launch_d384_kl16(const TopkPlan &p, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N, float *part_s,
                        int *part_i, int *gthr, hipStream_t st, K1Collect coll);

template <int KL, bool MAXONLY = false, bool COLLECT = false>
static int launch_k1_kl(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                        float *part_s, int *part_i, int *gthr, hipStream_t st, K1Collect coll = K1Collect{}) {
    constexpr bool M = KL <= 16;   // the 16x16x32 form wherever the asm-pipelined tile loop is used (lists of 16, pre-pass, collect)
    constexpr bool P = KL <= 16 && !MAXONLY && !COLLECT;   // two tiles per barrier for the list-of-16 main pass (as at D = 384)
    switch (D) {
        case 128: return launch_k1<128, 8, 1, KL, MAXONLY, P, COLLECT, false, M>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
        case 256: return launch_k1<256, 8, 1, KL, MAXONLY, P, COLLECT, false, M>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
        case 384: {
            if constexpr (!MAXONLY && !COLLECT && KL == 16)
                return k1_launch_d384_kl16(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
            else
                return launch_k1<384, 8, 1, KL, MAXONLY, false, COLLECT, false, M>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
        }
        case 512: return launch_k1<512, 4, 1, KL, MAXONLY, P, COLLECT, false, M>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
        case 768: return launch_k1<768, 4, 1, KL, MAXONLY, false, COLLECT, false, M>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
        default: return fail(TSIM_EUNSUPPORTED, "cosine_topk: unsupported padded width %d", D);
    }
}

// threshold pre-pass over the first rows of the corpus (block maxima only); defined in k1_kl16.hip
int k1_launch_blockmax(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                       float *bmax, hipStream_t st);
int k1_launch_kl16(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                   float *part_s, int *part_i, int *gthr, hipStream_t st, K1Collect range = K1Collect{});
int k1_launch_kl32(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                   float *part_s, int *part_i, int *gthr, hipStream_t st, K1Collect range = K1Collect{});
// widening pass (COLLECT mode) over at most Q slots; defined in k1_collect.hip
int k1_launch_collect(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N, int *gthr_slots,
                      K1Collect coll, hipStream_t st);

}  // namespace tsim
