// Large-K projection GEMM for the base-size encoders (K >= 768: all-mpnet-base-v2, bert-base-uncased) and the
// residual + LayerNorm row kernel that follows the two projections feeding a LayerNorm.
//
//   out[M, N] = X[M, K] @ W[N, K]^T + bias      X: bf16 token rows, W: bf16 [out, in] (nn.Linear layout)
//
// gemm_pp_kernel: 256-token x 256-feature output tile per workgroup (8 waves), BK = 64, two 64-KiB LDS slots filled by
// LDS-DMA (global_load_lds 16 B/lane; the XOR swizzle sits on the SOURCE address, the LDS image is lane-linear).
// The 8 waves form two groups of four (one wave of each group per SIMD) that run the same program ONE BARRIER APART:
// while group 0 issues its ds_read_b128 fragment loads (and the next tile's DMA) group 1 runs its MFMA cluster, and
// vice versa ("ping-pong"): every barrier interval has exactly one wave per SIMD feeding the matrix pipe, and the LDS
// latency of the other wave is hidden behind it instead of behind the compiler's guess at a schedule.
//   group 0 : L00 | M00 | L01 | M01 | ...            L = fragment loads (+ DMA issue on the first section of a k-tile)
//   group 1 :  -  | L00 | M00 | L01 | M01 | ...      M = KSEC*8 MFMAs v_mfma_f32_32x32x16_bf16
// DMA protocol (one k-tile in flight): tile t+1 is issued into slot (t+1)&1 in each wave's first L section of tile t —
// its previous occupant t-1 was last read by group 1 in the interval before, and that read is retired (lgkmcnt(0))
// before the closing barrier; every wave waits vmcnt(0) for its pieces of t+1 in the last interval of tile t, one
// barrier before group 0's first read of it.  All LDS traffic inside the loop is inline asm so that hipcc neither
// drains the DMA (s_waitcnt vmcnt(0)) before "aliasing" LDS reads nor re-schedules loads across the raw s_barrier.
// Roofline: MFMA.  Per k-tile a workgroup stages 64 KiB for 8.4 MFLOP = 128 FLOP/B from L2: 32 B/clk/CU at the full
// MFMA rate (32 cycles per 32x32x16 MFMA), against a measured LDS-DMA ceiling of 40-56 B/clk/CU.
// Persistent workgroups walk the tile list with stride gridDim.x; where the epilogue has LDS of its own (fp32 output, or
// 128-wide feature tiles) the next tile's first k-tile is fetched during the epilogue.
//
// Epilogues: bias | bias + GELU(erf) -> bf16, staged through LDS so that every global store is a 16-byte piece of a
// 512-byte row segment;  bias -> fp32 (pre-LayerNorm sums; the MFMA operands are swapped so that the feature runs
// along the lanes and each store instruction writes two full 128-byte lines).
#include "common.h"
#include "gemm_pp.h"

namespace tsim {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int PP_BM = 256, PP_BK = 64;          // token tile, k-tile (128 bytes per tile row); feature tile BN = 256 or 128
constexpr int PP_XB = PP_BM * PP_BK * 2;        // 32 KiB of token rows per stage
constexpr int PP_SCALES = 2048;   // MXFP8: per k-tile 256 + 256 rows x 4 E8M0 bytes (one per 32-element block)
typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128_o(uint32_t addr) {   // address + immediate byte offset (< 65536)
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ uint32_t lds_read_b32_o(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void glds4(const void *gsrc, void *lds_wave_base) {   // 4 bytes per lane
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}

// ---- MXFP8 element/scale helpers (bit-exact restatement: oracle/fp8_ref.mx_quantize)
__device__ __forceinline__ int mx_shared_exp(float amax) {   // floor(log2 amax) - 8, clamped to E8M0; amax = 0 -> 0
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
    return amax > 0.f ? (e < -127 ? -127 : e) : 0;
}
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d, int sexp) {
    const float y0 = fminf(fmaxf(ldexpf(a, -sexp), -448.f), 448.f), y1 = fminf(fmaxf(ldexpf(b, -sexp), -448.f), 448.f);
    const float y2 = fminf(fmaxf(ldexpf(c, -sexp), -448.f), 448.f), y3 = fminf(fmaxf(ldexpf(d, -sexp), -448.f), 448.f);
    int o = __builtin_amdgcn_cvt_pk_fp8_f32(y0, y1, 0, false);
    o = __builtin_amdgcn_cvt_pk_fp8_f32(y2, y3, o, true);
    return (uint32_t)o;
}
__device__ __forceinline__ float bf16_round_f32(float f) { return __uint_as_float((uint32_t)f32_to_bf16(f) << 16); }

// MX = true: operands are MXFP8 (e4m3 bytes + one E8M0 scale byte per 32 elements along K, scale arrays [rows, K/32]);
// a tile row is still 128 bytes (128 elements), a k-tile is two v_mfma_scale_f32_32x32x64_f8f6f4 steps.  Operand layout
// of that instruction, pinned on hardware by tools/microbench/mx_layout.hip: lane (r, h) holds row r, bytes 0-15 of its
// 8 VGPRs = k 16h..16h+15 (scale block 0), bytes 16-31 = k 32+16h.. (scale block 1); the scale byte a lane supplies
// (selected by opsel) belongs to block h of its row.
#ifdef TSIM_PP_STAMPS
// DIAGNOSTIC build only (python -m text_similarity_amd.build --stamps): wave 0 of every workgroup accumulates where a
// tile's cycles go.  [0] tiles, [1] wait for the first k-tile + top barriers, [2] k loop, [3] epilogue, [4] MFMA sections,
// [5] load sections, [6] barrier waits, [7] wait for the next k-tile's DMA (vmcnt).
__device__ unsigned long long g_pp_stamps[8];
#define PP_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PP_ACC(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_pp_stamps[i], (unsigned long long)(v)); } while (0)
#else
#define PP_STAMP(var) do { } while (0)
#define PP_ACC(i, v) do { } while (0)
#endif

template <int EPI, int KSEC, bool MX, int BN>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const void *__restrict__ X_, const void *__restrict__ W_,
                                                      const uint8_t *__restrict__ xs, const uint8_t *__restrict__ ws,
                                                      const float *__restrict__ bias, void *__restrict__ out_,
                                                      uint8_t *__restrict__ out_s, int M, int N, int K, int mtiles,
                                                      int ntiles, int wpk) {
    constexpr int ESZ = MX ? 1 : 2;                  // bytes per element
    constexpr int NSTEP = MX ? 2 : 4;                // MFMA k-steps per k-tile
    constexpr int NSEC = NSTEP / KSEC;
    constexpr int NI = BN / 128;                     // 32-feature fragments per wave (4 waves across the feature tile)
    constexpr int WFEAT = BN / 4;                    // features per wave
    constexpr int PP_STAGE = PP_XB + BN * 128;       // operand bytes per stage
    constexpr int PP_PPW = PP_STAGE / 1024 / 8;      // DMA pieces per wave and k-tile (first 4: token rows)
    constexpr int STAGE = PP_STAGE + (MX ? PP_SCALES : 0);
    constexpr bool F32 = EPI == PP_EPI_F32;
    static_assert(BN == 256 || BN == 128, "feature tile");
    static_assert(!MX || BN == 256, "the MXFP8 variant is built for 256-feature tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int r = lane & 31, h = lane >> 5;

    // Persistent workgroups: tile indices t = blockIdx.x, + gridDim.x, ... (gridDim.x % 8 == 0, so a workgroup's tiles
    // keep blockIdx % 8, i.e. their XCD).  Tile order: the ntiles feature tiles of one token tile share t % 8, so they
    // meet in one XCD's L2 (speed only).
    const int total = ((mtiles + 7) / 8) * 8 * ntiles;
    auto tile_at = [&](int t, int &m0, int &n0) __attribute__((always_inline)) {
        const int xcd = t & 7, jj = t >> 3;
        const int mt_id = (jj / ntiles) * 8 + xcd;
        m0 = mt_id * PP_BM;
        n0 = (jj % ntiles) * BN;
        return mt_id < mtiles;
    };
    auto next_tile = [&](int t) __attribute__((always_inline)) {   // next valid tile of this workgroup, or >= total
        int m, n;
        do t += (int)gridDim.x; while (t < total && !tile_at(t, m, n));
        return t;
    };

    // ---- LDS image of an operand region: 128-byte tile rows, two per 256-byte super-row, 16-byte chunk c of
    // super-row sr stored at chunk c ^ (sr & 15): the 32 rows x 2 k-halves of a ds_read_b128 fragment hit 16 distinct
    // chunks per 16 lanes (conflict-free).  Piece p = 1 KiB = 64 lanes x 16 B, LDS-linear.
    // DMA sources: wave-uniform operand base (SGPRs) + one 32-bit byte offset per piece, advanced by 128 per k-tile.
    // wpk: W was re-laid at load time as [feature tile][k-tile][LDS image of the tile] (pack_w_kernel), so a W piece is
    // 1 KiB of CONTIGUOUS memory instead of 8 row segments of 128 B: LDS-DMA moves contiguous pieces about twice as fast,
    // and the staging rate, not the matrix pipe, bounds this kernel.
    uint32_t src_off[PP_PPW];
    uint32_t advanced = 0;                                // k-tiles the offsets have moved since they were set
    const uint32_t wstep = wpk ? BN * 128 : 128;          // bytes per k-tile on the W side
#pragma unroll
    for (int i = 0; i < PP_PPW; ++i) {
        const int p = wave + i * 8;                       // < 32: X piece, else W piece
        const int sl = (p & 31) * 64 + lane;
        const int sr = sl >> 4, chp = sl & 15;
        const int ch = chp ^ (sr & 15);
        src_off[i] = (i >= 4 && wpk) ? (uint32_t)((p - 32) * 1024 + lane * 16)
                                     : (uint32_t)((sr * 2 + (ch >> 3)) * K * ESZ + (ch & 7) * 16);
    }
    const char *xbase = nullptr, *wbase = nullptr;
    // MXFP8 scales of a k-tile: one dword (4 blocks) per row; wave w < 4 brings token rows 64w.., wave w >= 4 feature rows
    const uint8_t *sc_base = nullptr;
    uint32_t sc_off = 0;
    auto set_sources = [&](int m0, int n0) __attribute__((always_inline)) {
        xbase = reinterpret_cast<const char *>(X_) + (int64_t)m0 * K * ESZ;
        wbase = reinterpret_cast<const char *>(W_) + (int64_t)n0 * K * ESZ;   // packed or not: a feature tile's data starts here
#pragma unroll
        for (int i = 0; i < PP_PPW; ++i) src_off[i] -= advanced * (i < 4 ? 128u : wstep);
        advanced = 0;
        if constexpr (MX) {
            sc_base = wave < 4 ? xs + (int64_t)(m0 + wave * 64) * (K / 32) : ws + (int64_t)(n0 + (wave - 4) * 64) * (K / 32);
            sc_off = (uint32_t)(lane * (K / 32));
        }
    };
    auto issue = [&](int slot) __attribute__((always_inline)) {   // stages the NEXT k-tile (offsets advance by themselves)
#pragma unroll
        for (int i = 0; i < PP_PPW; ++i) {
            const int p = wave + i * 8;
            glds16((i < 4 ? xbase : wbase) + (size_t)src_off[i], smem + slot * STAGE + p * 1024);
            src_off[i] += i < 4 ? 128u : wstep;
        }
        advanced += 1;
        if constexpr (MX) {
            glds4(sc_base + (size_t)sc_off, smem + slot * STAGE + PP_STAGE + wave * 256);
            sc_off += 4;
        }
    };
    auto frag_off = [&](int row) __attribute__((always_inline)) {   // k-step 0; bf16 k-step s: XOR (s << 5); MX: XOR (s << 6 | q << 5)
        const int sr = row >> 1;
        return (uint32_t)(sr * 256 + ((((row & 1) * 8 + h) ^ (sr & 15)) << 4));
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
    // fragment j of this lane sits 32 tile rows = 4096 bytes after fragment 0 (the swizzle term only sees row bits 1-4),
    // so one address VGPR per operand and immediate offsets address all of them
    const uint32_t xoff0 = frag_off(grp * 128 + r);
    const uint32_t woff0 = PP_XB + frag_off(wq * WFEAT + r);

    const uint32_t xsoff0 = PP_STAGE + (grp * 128 + r) * 4;            // MX: this lane's scale dwords, +128 per fragment
    const uint32_t wsoff0 = PP_STAGE + 1024 + (wq * WFEAT + r) * 4;

    const int nk = K * ESZ / 128;
    // The next tile's first k-tile is fetched DURING the epilogue, into the slot its parity selects; the epilogue stages its
    // output image (row-contiguous global stores) through the OTHER slot, 64 KiB at a time.  Without this a workgroup sits
    // idle while its stores drain (vmcnt is in order) and then again for the first DMA of the next tile: ~8 of 25 us per
    // tile on the K = 768 projections.
    constexpr bool PIPE = true;
    int t = blockIdx.x, m0, n0;
    if (!tile_at(t, m0, n0)) t = next_tile(t);
    if (t >= total) return;
    (void)tile_at(t, m0, n0);
    set_sources(m0, n0);
    uint32_t kidx = 0;                                    // k-tiles consumed so far: slot parity across tiles
    constexpr int EPI_STORES = F32 ? NI * 64 : (EPI == PP_EPI_GELU_MX ? 9 : PP_BM * (BN / 8) / 512);   // global stores per wave in one epilogue
    bool first = true;
    issue(0);
    for (;;) {
        f32x16 acc[NI][4];
    #pragma unroll
        for (int i = 0; i < NI; ++i)
    #pragma unroll
            for (int j = 0; j < 4; ++j)
    #pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
        // my pieces of this tile's first k-tile have landed.  They are OLDER than the previous epilogue's global stores
        // when the fetch was issued ahead of them (PIPE): leave those stores in flight (vmcnt is in order, 6 bits wide)
        PP_STAMP(ts0);
        if (first || !PIPE) wait_vmcnt<0>(); else wait_vmcnt<(EPI_STORES < 63 ? EPI_STORES : 63)>();
        first = false;
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one interval behind group 0
        PP_STAMP(ts1);
#ifdef TSIM_PP_STAMPS
        unsigned long long msum = 0, lsum = 0, b1sum = 0, b2sum = 0, dsum = 0, tprev = 0;
#endif

        for (int kt = 0; kt < nk; ++kt) {
            const uint32_t sbase = lds0 + ((kidx + kt) & 1) * STAGE;
            uint32_t xsv[4], wsv[NI];
#pragma unroll
            for (int sec = 0; sec < NSEC; ++sec) {
                // ---------------- L: fragments of KSEC k-steps (+ the next k-tile's DMA)
#ifdef TSIM_PP_STAMPS
                const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
                if (tprev) b2sum += tl0 - tprev;
#endif
                u32x4 xf[KSEC][4], wf[KSEC][NI], xg[KSEC][4], wg[KSEC][NI];   // xg/wg: second 16 bytes of an MXFP8 fragment
#pragma unroll
                for (int ks = 0; ks < KSEC; ++ks) {
                    const int st = sec * KSEC + ks;
                    const uint32_t sx = (uint32_t)(MX ? st << 6 : st << 5);
                    const uint32_t wa = sbase + (woff0 ^ sx), xa = sbase + (xoff0 ^ sx);
                    wf[ks][0] = lds_read_b128_o<0>(wa);
                    if constexpr (NI == 2) wf[ks][NI - 1] = lds_read_b128_o<4096>(wa);
                    xf[ks][0] = lds_read_b128_o<0>(xa);
                    xf[ks][1] = lds_read_b128_o<4096>(xa);
                    xf[ks][2] = lds_read_b128_o<8192>(xa);
                    xf[ks][3] = lds_read_b128_o<12288>(xa);
                    if constexpr (MX) {   // second 16 bytes of each 32-byte fragment: chunk + 2
                        const uint32_t wb = sbase + (woff0 ^ sx ^ 32u), xb = sbase + (xoff0 ^ sx ^ 32u);
                        wg[ks][0] = lds_read_b128_o<0>(wb);
                        if constexpr (NI == 2) wg[ks][NI - 1] = lds_read_b128_o<4096>(wb);
                        xg[ks][0] = lds_read_b128_o<0>(xb);
                        xg[ks][1] = lds_read_b128_o<4096>(xb);
                        xg[ks][2] = lds_read_b128_o<8192>(xb);
                        xg[ks][3] = lds_read_b128_o<12288>(xb);
                    }
                }
                if constexpr (MX) {
                    if (sec == 0) {
                        wsv[0] = lds_read_b32_o<0>(sbase + wsoff0);
                        if constexpr (NI == 2) wsv[NI - 1] = lds_read_b32_o<128>(sbase + wsoff0);
                        xsv[0] = lds_read_b32_o<0>(sbase + xsoff0);
                        xsv[1] = lds_read_b32_o<128>(sbase + xsoff0);
                        xsv[2] = lds_read_b32_o<256>(sbase + xsoff0);
                        xsv[3] = lds_read_b32_o<384>(sbase + xsoff0);
                    }
                }
                if (sec == 0 && kt + 1 < nk) issue((kidx + kt + 1) & 1);
                if (sec == NSEC - 1 && grp == 1) wait_vmcnt<0>();
#pragma unroll
                for (int ks = 0; ks < KSEC; ++ks) {
                    // the wait carries the fragments as operands so that no MFMA reading them is scheduled above it
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[ks][0]), "+v"(xf[ks][1]), "+v"(xf[ks][2]), "+v"(xf[ks][3]) :: "memory");
#pragma unroll
                    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(wf[ks][i]));
                    if constexpr (MX) {
                        asm volatile("" : "+v"(xg[ks][0]), "+v"(xg[ks][1]), "+v"(xg[ks][2]), "+v"(xg[ks][3]));
#pragma unroll
                        for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(wg[ks][i]));
                    }
                }
                if constexpr (MX) {
                    if (sec == 0) {
                        asm volatile("" : "+v"(xsv[0]), "+v"(xsv[1]), "+v"(xsv[2]), "+v"(xsv[3]));
#pragma unroll
                        for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(wsv[i]));
#pragma unroll
                        for (int i = 0; i < NI; ++i) wsv[i] >>= 8 * h;    // lane half h supplies block h of the step: byte 2s + h
#pragma unroll
                        for (int j = 0; j < 4; ++j) xsv[j] >>= 8 * h;
                    }
                }
#ifdef TSIM_PP_STAMPS
                const unsigned long long tl1 = __builtin_amdgcn_s_memtime();
                lsum += tl1 - tl0;
#endif
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- M
                PP_STAMP(tm0);
#ifdef TSIM_PP_STAMPS
                b1sum += tm0 - tl1;
#endif
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < KSEC; ++ks)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (MX) {
                                const i32x8 wv = __builtin_bit_cast(i32x8, __builtin_shufflevector(wf[ks][i], wg[ks][i], 0, 1, 2, 3, 4, 5, 6, 7));
                                const i32x8 xv = __builtin_bit_cast(i32x8, __builtin_shufflevector(xf[ks][j], xg[ks][j], 0, 1, 2, 3, 4, 5, 6, 7));
                                const int st = sec * KSEC + ks;      // opsel = byte 2*st of the (shifted) scale dword
                                if constexpr (F32) {
                                    if (st == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xv, wv, acc[i][j], 0, 0, 0, (int)xsv[j], 0, (int)wsv[i]);
                                    else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xv, wv, acc[i][j], 0, 0, 2, (int)xsv[j], 2, (int)wsv[i]);
                                } else {
                                    if (st == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[i][j], 0, 0, 0, (int)wsv[i], 0, (int)xsv[j]);
                                    else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[i][j], 0, 0, 2, (int)wsv[i], 2, (int)xsv[j]);
                                }
                            } else {
                                const bf16x8 wv = __builtin_bit_cast(bf16x8, wf[ks][i]);
                                const bf16x8 xv = __builtin_bit_cast(bf16x8, xf[ks][j]);
                                if constexpr (F32)   // rows (registers) = tokens, columns (lanes) = features
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv, wv, acc[i][j], 0, 0, 0);
                                else                 // rows (registers) = features, columns (lanes) = tokens
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc[i][j], 0, 0, 0);
                            }
                        }
                // keep the cluster inside its barrier interval: without a use here LLVM may sink the (pure) MFMAs into a later
                // block, merging two sections (seen on the MX variant: 96 fragment VGPRs live, scratch spills)
                // (in/out operands, no loop: with input-only operands of the template-sized array hipcc's host pass silently
                // drops the kernel's host stub — the library then fails to load with an undefined symbol)
                asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]));
                if constexpr (NI == 2) asm volatile("" : "+v"(acc[NI - 1][0]), "+v"(acc[NI - 1][1]), "+v"(acc[NI - 1][2]), "+v"(acc[NI - 1][3]));
                __builtin_amdgcn_s_setprio(0);
#ifdef TSIM_PP_STAMPS
                tprev = __builtin_amdgcn_s_memtime();
                msum += tprev - tm0;
#endif
                if (sec == NSEC - 1 && grp == 0) wait_vmcnt<0>();
#ifdef TSIM_PP_STAMPS
                if (sec == NSEC - 1) { const unsigned long long tw = __builtin_amdgcn_s_memtime(); dsum += tw - tprev; tprev = tw; }
#endif
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier: everyone is done with LDS
        PP_STAMP(ts2);
        kidx += nk;
        const int tn = next_tile(t);
        int m0n = 0, n0n = 0;
        if (tn < total) (void)tile_at(tn, m0n, n0n);

        if constexpr (F32) {
            // acc[i][j][g]: token m0 + grp*128 + j*32 + (g&3) + 8*(g>>2) + 4*h, feature n0 + wq*WFEAT + i*32 + r
            float *out = reinterpret_cast<float *>(out_);
            float bv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i)   // asm load: an ordinary load's use would make hipcc drain the DMA issued below
                asm volatile("global_load_dword %0, %1, off" : "=v"(bv[i]) : "v"(bias + n0 + wq * WFEAT + i * 32 + r) : "memory");
            if (tn < total) { set_sources(m0n, n0n); issue(kidx & 1); }
            if (tn < total) wait_vmcnt<PP_PPW + (MX ? 1 : 0)>(); else wait_vmcnt<0>();   // the bias loads are older than the DMA
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(bv[i]));
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = n0 + wq * WFEAT + i * 32 + r;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int64_t m = m0 + grp * 128 + j * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;   // < padded row count
                        out[m * N + n] = acc[i][j][g] + bv[i];
                    }
            }
        } else if constexpr (EPI == PP_EPI_GELU_MX) {
            // bias + GELU, rounded to bf16 (what the unfused path stores), then straight to MXFP8 — the next projection's
            // operand format: the 32 features i*32.. of a token are one scale block, 16 values in lane (r, 0) and 16 in
            // lane (r, 1).  Bytes (64 KiB) and scales (2 KiB) are staged in the slot the prefetch does not use.
            uint8_t *out = reinterpret_cast<uint8_t *>(out_);
            constexpr int SC0 = PP_BM * BN;               // scale image [256 tokens][8 blocks] behind the byte image
            u32x4 bvr[NI][4];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    asm volatile("global_load_dwordx4 %0, %1, off"
                                 : "=v"(bvr[i][gq]) : "v"(bias + n0 + wq * WFEAT + i * 32 + 8 * gq + 4 * h) : "memory");
            if (tn < total) { set_sources(m0n, n0n); issue(kidx & 1); }
            if (tn < total) wait_vmcnt<PP_PPW + 1>(); else wait_vmcnt<0>();   // the bias loads are older than the DMA
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(bvr[i][0]), "+v"(bvr[i][1]), "+v"(bvr[i][2]), "+v"(bvr[i][3]));
            const uint32_t stg = lds0 + ((kidx & 1) ^ 1) * STAGE;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y[16];
                    float amax = 0.f;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 bv = __builtin_bit_cast(f32x4, bvr[i][gq]);
                        float t0 = acc[i][j][4 * gq + 0] + bv[0], t1 = acc[i][j][4 * gq + 1] + bv[1];
                        float t2 = acc[i][j][4 * gq + 2] + bv[2], t3 = acc[i][j][4 * gq + 3] + bv[3];
                        gelu2(t0, t1);
                        gelu2(t2, t3);
                        y[4 * gq + 0] = bf16_round_f32(t0);
                        y[4 * gq + 1] = bf16_round_f32(t1);
                        y[4 * gq + 2] = bf16_round_f32(t2);
                        y[4 * gq + 3] = bf16_round_f32(t3);
                    }
#pragma unroll
                    for (int g = 0; g < 16; ++g) amax = fmaxf(amax, fabsf(y[g]));
                    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                    const int sexp = mx_shared_exp(amax);
                    const int row = grp * 128 + j * 32 + r;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int nloc = wq * WFEAT + i * 32 + 8 * gq + 4 * h;
                        const uint32_t ad = stg + row * BN + (((nloc >> 4) ^ (row & 15)) << 4) + (nloc & 15);
                        const uint32_t pk = mx_pack4(y[4 * gq], y[4 * gq + 1], y[4 * gq + 2], y[4 * gq + 3], sexp);
                        asm volatile("ds_write_b32 %0, %1" ::"v"(ad), "v"(pk) : "memory");
                    }
                    if (h == 0) {
                        const uint32_t ad = stg + SC0 + row * 8 + wq * NI + i;
                        const uint32_t sb = (uint32_t)(sexp + 127);
                        asm volatile("ds_write_b8 %0, %1" ::"v"(ad), "v"(sb) : "memory");
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            uint8_t *obase = out + (int64_t)m0 * N + n0;
#pragma unroll 2
            for (int it = 0; it < PP_BM * 16 / 512; ++it) {
                const int sl = it * 512 + threadIdx.x;
                const int row = sl >> 4, cp = sl & 15;       // rows past M land in the padded tail of the buffer
                u32x4 v = lds_read_b128_o<0>(stg + sl * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
                *reinterpret_cast<u32x4 *>(obase + (int64_t)row * N + ((cp ^ (row & 15)) << 4)) = v;
            }
            {
                const int row = threadIdx.x >> 1, half = threadIdx.x & 1;   // 256 rows x 8 scale bytes, 4 per thread
                uint32_t v = lds_read_b32_o<0>(stg + SC0 + row * 8 + half * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
                *reinterpret_cast<uint32_t *>(out_s + (int64_t)(m0 + row) * (N / 32) + n0 / 32 + half * 4) = v;
            }
        } else {
            // acc[i][j][g]: feature n0 + wq*WFEAT + i*32 + (g&3) + 8*(g>>2) + 4*h, token m0 + grp*128 + j*32 + r.
            // Output image in LDS (the slot the prefetch does not use): row = token (BN*2 bytes), 16-byte slot c at
            // c ^ (row & 15); one pass per wave group = token half (128 rows: 64 KiB of a 256-wide tile, 32 KiB of a 128-wide one).
            // LDS traffic by inline asm and raw barriers, bias by asm loads issued BEFORE the prefetch DMA: nothing here
            // drains that DMA.
            bf16_t *out = reinterpret_cast<bf16_t *>(out_);
            constexpr int NPASS = 2, ROWS = 128;
            u32x4 bvr[NI][4];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    asm volatile("global_load_dwordx4 %0, %1, off"
                                 : "=v"(bvr[i][gq]) : "v"(bias + n0 + wq * WFEAT + i * 32 + 8 * gq + 4 * h) : "memory");
            if (tn < total) { set_sources(m0n, n0n); issue(kidx & 1); }
            if (tn < total) wait_vmcnt<PP_PPW + (MX ? 1 : 0)>(); else wait_vmcnt<0>();   // the bias loads are older
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(bvr[i][0]), "+v"(bvr[i][1]), "+v"(bvr[i][2]), "+v"(bvr[i][3]));
            const uint32_t stg = lds0 + ((kidx & 1) ^ 1) * STAGE;
            uint64_t pk[NI][4][4];                            // this wave's results, packed bf16 x 4
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bv = __builtin_bit_cast(f32x4, bvr[i][gq]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float y0 = acc[i][j][4 * gq + 0] + bv[0], y1 = acc[i][j][4 * gq + 1] + bv[1];
                        float y2 = acc[i][j][4 * gq + 2] + bv[2], y3 = acc[i][j][4 * gq + 3] + bv[3];
                        if constexpr (EPI == PP_EPI_GELU) {
                            gelu2(y0, y1);
                            gelu2(y2, y3);
                        }
                        pk[i][gq][j] = (uint64_t)pack_bf16x2(y0, y1) | ((uint64_t)pack_bf16x2(y2, y3) << 32);
                    }
                }
            char *obase = reinterpret_cast<char *>(out + (int64_t)m0 * N + n0);
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                if (grp == pass) {
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int nloc = wq * WFEAT + i * 32 + 8 * gq + 4 * h;
                                const int row = j * 32 + r;                      // row inside this pass's image
                                const uint32_t ad = stg + row * (BN * 2) + (((nloc >> 3) ^ (row & 15)) << 4) + ((nloc & 4) << 1);
                                asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk[i][gq][j]) : "memory");
                            }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll 2
                for (int it = 0; it < ROWS * (BN / 8) / 512; ++it) {
                    const int sl = it * 512 + threadIdx.x;
                    const int row = sl / (BN / 8), cp = sl % (BN / 8);   // rows past M land in the padded tail of the buffer
                    u32x4 v = lds_read_b128_o<0>(stg + sl * 16);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
                    *reinterpret_cast<u32x4 *>(obase + (int64_t)(pass * ROWS + row) * N * 2 + ((cp ^ (row & 15)) << 4)) = v;
                }
                if (pass + 1 < NPASS) __builtin_amdgcn_s_barrier();   // the image is read: the other group may overwrite it
            }
        }
#ifdef TSIM_PP_STAMPS
        {
            const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
            PP_ACC(0, 1); PP_ACC(1, ts1 - ts0); PP_ACC(2, ts2 - ts1); PP_ACC(3, ts3 - ts2); PP_ACC(4, msum); PP_ACC(5, lsum); PP_ACC(6, b1sum + b2sum); PP_ACC(7, dsum);
        }
#endif
        if (tn >= total) break;
        t = tn;
        m0 = m0n;
        n0 = n0n;
    }
    wait_vmcnt<0>();
}

// out[m, :] = LayerNorm(y[m, :] + res[m, :]) * gamma + beta, one wave per token row, fp32 statistics (two-pass).
// HBM-bound: H * (4 + 2 + 2) bytes per row.
template <int VPL4, bool QUANT>   // float4 groups per lane = H / 256; QUANT: also emit the row as MXFP8 (q, sc)
__global__ __launch_bounds__(256) void res_ln_rows_kernel(const float *__restrict__ y, const bf16_t *__restrict__ res,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float eps, int M, int H, bf16_t *__restrict__ out,
                                                          uint8_t *__restrict__ q, uint8_t *__restrict__ sc) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[VPL4][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL4; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(y + row * H + c);
        const uint2 rr = *reinterpret_cast<const uint2 *>(res + row * H + c);
        v[i][0] = a.x + __uint_as_float(rr.x << 16);
        v[i][1] = a.y + __uint_as_float(rr.x & 0xffff0000u);
        v[i][2] = a.z + __uint_as_float(rr.y << 16);
        v[i][3] = a.w + __uint_as_float(rr.y & 0xffff0000u);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)H;
    float qs = 0.f;
#pragma unroll
    for (int i = 0; i < VPL4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) qs += (v[i][e] - mean) * (v[i][e] - mean);
    const float rstd = 1.0f / sqrtf(wave_sum(qs) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < VPL4; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
        const float4 be = *reinterpret_cast<const float4 *>(beta + c);
        uint2 o;
        o.x = pack_bf16x2((v[i][0] - mean) * rstd * g.x + be.x, (v[i][1] - mean) * rstd * g.y + be.y);
        o.y = pack_bf16x2((v[i][2] - mean) * rstd * g.z + be.z, (v[i][3] - mean) * rstd * g.w + be.w);
        *reinterpret_cast<uint2 *>(out + row * H + c) = o;
        if constexpr (QUANT) {   // the bf16 values just stored, 32-element block = 8 consecutive lanes
            const float f0 = __uint_as_float(o.x << 16), f1 = __uint_as_float(o.x & 0xffff0000u);
            const float f2 = __uint_as_float(o.y << 16), f3 = __uint_as_float(o.y & 0xffff0000u);
            float amax = fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(f3)));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
            const int sexp = mx_shared_exp(amax);
            *reinterpret_cast<uint32_t *>(q + row * H + c) = mx_pack4(f0, f1, f2, f3, sexp);
            if ((lane & 7) == 0) sc[row * (H / 32) + (c >> 5)] = (uint8_t)(sexp + 127);
        }
    }
}

// bf16 [n] -> MXFP8: e4m3 bytes [n] + E8M0 scale bytes [n / 32] (blocks of 32 consecutive elements; rows are multiples
// of 32 long, so the flat view is the row view).  8 elements per lane, 4 lanes per block.  Bit-exact restatement:
// oracle/fp8_ref.mx_quantize.  HBM-bound: 3.03 bytes per element.
__global__ __launch_bounds__(256) void quant_mx_kernel(const bf16_t *__restrict__ x, int64_t ngroups,
                                                       uint8_t *__restrict__ q, uint8_t *__restrict__ sc) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;      // group of 8 elements
    const bool live = t < ngroups;                                   // ngroups % 4 == 0: a block's 4 lanes agree
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (live) raw = *reinterpret_cast<const uint4 *>(x + t * 8);
    float v[8];
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const int sexp = mx_shared_exp(amax);
    const uint32_t pk[2] = {mx_pack4(v[0], v[1], v[2], v[3], sexp), mx_pack4(v[4], v[5], v[6], v[7], sexp)};
    if (live) {
        *reinterpret_cast<uint2 *>(q + t * 8) = make_uint2(pk[0], pk[1]);
        if ((t & 3) == 0) sc[t >> 2] = (uint8_t)(sexp + 127);
    }
}

// W [N, kbytes] row-major -> [N / BN][kbytes / 128][BN * 128 bytes in the kernel's LDS image order] (16 bytes per thread)
__global__ __launch_bounds__(256) void pack_w_kernel(const uint4 *__restrict__ W, uint4 *__restrict__ Wp, int N, int kbytes,
                                                     int BN) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= (int64_t)N * kbytes / 16) return;
    const int per = BN * 8, nk = kbytes / 128;
    const int blk = (int)(u / per), sl = (int)(u % per);
    const int nt = blk / nk, kt = blk % nk;
    const int sr = sl >> 4, ch = (sl & 15) ^ (sr & 15);
    const int row = sr * 2 + (ch >> 3), c = ch & 7;
    Wp[u] = W[((int64_t)(nt * BN + row) * kbytes + kt * 128 + c * 16) / 16];
}

int pack_w(const void *W, void *Wp, int N, int kbytes, int BN, hipStream_t st) {
    if (N % BN != 0 || kbytes % 128 != 0) return fail(TSIM_EINVAL, "pack_w: N=%d kbytes=%d BN=%d", N, kbytes, BN);
    const int64_t units = (int64_t)N * kbytes / 16;
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st,
                       static_cast<const uint4 *>(W), static_cast<uint4 *>(Wp), N, kbytes, BN);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

template <int EPI, int KSEC, bool MX, int BN = 256>
static int launch_pp(const void *X, const void *W, const uint8_t *xs, const uint8_t *ws, const float *bias, void *out,
                     uint8_t *out_s, int M, int N, int K, int wpk, hipStream_t st) {
    constexpr int lds = 2 * (PP_XB + BN * 128 + (MX ? PP_SCALES : 0));   // the epilogue stages through the idle slot
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = gemm_pp_kernel<EPI, KSEC, MX, BN>;
    static DevOnce lds_once;
    TSIM_MAX_LDS(lds_once, kern, lds);
    const int mtiles = (M + PP_BM - 1) / PP_BM, ntiles = N / BN;
    const int total = ((mtiles + 7) / 8) * 8 * ntiles;
    static int persist = -1;
    if (persist < 0) { const char *e = getenv("TSIM_PP_PERSIST"); persist = e ? atoi(e) : 1; }
    const int grid = persist && total > 256 ? 256 : total;     // one persistent workgroup per CU (a multiple of 8)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, X, W, xs, ws, bias, out, out_s, M, N, K, mtiles, ntiles, wpk);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

bool gemm_pp_supported(int N, int K) { return N % 128 == 0 && K % PP_BK == 0 && K >= 2 * PP_BK; }
bool gemm_pp_mx_supported(int N, int K) { return N % 256 == 0 && K % 128 == 0 && K >= 256; }

static int pp_ksec() {
    static int ksec = -1;
    if (ksec < 0) { const char *e = getenv("TSIM_PP_KSEC"); ksec = e ? atoi(e) : 2; }
    return ksec;
}

int gemm_pp_tile_width(int N) { return N % 256 != 0 ? 128 : 256; }

int gemm_pp(int epi, const bf16_t *X, const bf16_t *W, int w_packed, const float *bias, void *out, int M, int N, int K,
            hipStream_t st) {
    if (!gemm_pp_supported(N, K)) return fail(TSIM_EUNSUPPORTED, "gemm_pp: N=%d K=%d not tileable by 128x64", N, K);
    if (M <= 0) return TSIM_OK;
    const int ksec = pp_ksec();
    // feature tile: 256 wide unless N is not a multiple of 256 (the 128-wide tile stages 1.5x the bytes per FLOP and
    // measured slower on the N = 768 projections even though it fills the last round of workgroups better)
    static int bn_env = -1;
    if (bn_env < 0) { const char *e = getenv("TSIM_PP_BN"); bn_env = e ? atoi(e) : 0; }
    int bn = gemm_pp_tile_width(N);
    if (N % 256 == 0 && bn_env == 128 && !w_packed) bn = 128;   // (a packed W fixes the tile width it was packed for)
#define PP_GO(E)                                                                                              \
    do {                                                                                                      \
        if (bn == 128)                                                                                        \
            return ksec == 1 ? launch_pp<E, 1, false, 128>(X, W, nullptr, nullptr, bias, out, nullptr, M, N, K, w_packed, st) \
                             : launch_pp<E, 2, false, 128>(X, W, nullptr, nullptr, bias, out, nullptr, M, N, K, w_packed, st); \
        return ksec == 1 ? launch_pp<E, 1, false, 256>(X, W, nullptr, nullptr, bias, out, nullptr, M, N, K, w_packed, st)     \
                         : launch_pp<E, 2, false, 256>(X, W, nullptr, nullptr, bias, out, nullptr, M, N, K, w_packed, st);    \
    } while (0)
    switch (epi) {
        case PP_EPI_BIAS: PP_GO(PP_EPI_BIAS);
        case PP_EPI_GELU: PP_GO(PP_EPI_GELU);
        case PP_EPI_F32: PP_GO(PP_EPI_F32);
        default: return fail(TSIM_EINVAL, "gemm_pp: unknown epilogue %d", epi);
    }
#undef PP_GO
}

int gemm_pp_mx(int epi, const uint8_t *Xq, const uint8_t *Xs, const uint8_t *Wq, int w_packed, const uint8_t *Ws,
               const float *bias, void *out, uint8_t *out_scales, int M, int N, int K, hipStream_t st) {
    if (!gemm_pp_mx_supported(N, K)) return fail(TSIM_EUNSUPPORTED, "gemm_pp_mx: N=%d K=%d not tileable by 256x128", N, K);
    if (M <= 0) return TSIM_OK;
    switch (epi) {   // one v_mfma_scale_f32_32x32x64 step (8 MFMAs, 512 cycles) per section
        case PP_EPI_BIAS: return launch_pp<PP_EPI_BIAS, 1, true>(Xq, Wq, Xs, Ws, bias, out, nullptr, M, N, K, w_packed, st);
        case PP_EPI_GELU: return launch_pp<PP_EPI_GELU, 1, true>(Xq, Wq, Xs, Ws, bias, out, nullptr, M, N, K, w_packed, st);
        case PP_EPI_F32: return launch_pp<PP_EPI_F32, 1, true>(Xq, Wq, Xs, Ws, bias, out, nullptr, M, N, K, w_packed, st);
        case PP_EPI_GELU_MX:
            if (!out_scales) return fail(TSIM_EINVAL, "gemm_pp_mx: the MXFP8 epilogue needs a scale output");
            return launch_pp<PP_EPI_GELU_MX, 1, true>(Xq, Wq, Xs, Ws, bias, out, out_scales, M, N, K, w_packed, st);
        default: return fail(TSIM_EINVAL, "gemm_pp_mx: unknown epilogue %d", epi);
    }
}

int quant_mx(const bf16_t *x, int64_t rows, int K, uint8_t *q, uint8_t *scales, hipStream_t st) {
    if (K % 32 != 0) return fail(TSIM_EINVAL, "quant_mx: K=%d is not a multiple of the 32-element block", K);
    const int64_t ngroups = rows * K / 8;
    if (ngroups <= 0) return TSIM_OK;
    hipLaunchKernelGGL(quant_mx_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, st, x, ngroups, q, scales);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

int res_ln_rows(const float *y, const bf16_t *res, const float *gamma, const float *beta, float eps, bf16_t *out,
                uint8_t *q, uint8_t *q_scales, int M, int H, hipStream_t st) {
    if (M <= 0) return TSIM_OK;
    const unsigned g = (unsigned)((M + 3) / 4);
    const bool qt = q != nullptr && q_scales != nullptr;
#define RL_GO(V)                                                                                                          \
    do {                                                                                                                  \
        if (qt) hipLaunchKernelGGL((res_ln_rows_kernel<V, true>), dim3(g), dim3(256), 0, st, y, res, gamma, beta, eps, M, H, out, q, q_scales); \
        else hipLaunchKernelGGL((res_ln_rows_kernel<V, false>), dim3(g), dim3(256), 0, st, y, res, gamma, beta, eps, M, H, out, q, q_scales);   \
    } while (0)
    switch (H) {
        case 256: RL_GO(1); break;
        case 512: RL_GO(2); break;
        case 768: RL_GO(3); break;
        case 1024: RL_GO(4); break;
        default: return fail(TSIM_EUNSUPPORTED, "res_ln_rows: hidden size %d (256, 512, 768, 1024)", H);
    }
#undef RL_GO
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

#ifdef TSIM_PP_STAMPS
}  // namespace tsim
extern "C" int tsim_debug_pp_stamps(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsim::g_pp_stamps), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(tsim::g_pp_stamps), z, 64) != hipSuccess) return 1; }
    return 0;
}
namespace tsim {
#endif
}  // namespace tsim
