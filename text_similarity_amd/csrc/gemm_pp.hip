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
// Roofline: MFMA.  Per k-tile a workgroup stages 64 KiB for 4.2 MFLOP... = 128 FLOP/B from L2, 16 B/clk/CU at full
// MFMA rate (measured LDS-DMA ceiling: 40-56 B/clk/CU).
//
// Epilogues: bias | bias + GELU(erf) -> bf16, staged through LDS so that every global store is a 16-byte piece of a
// 512-byte row segment;  bias -> fp32 (pre-LayerNorm sums; the MFMA operands are swapped so that the feature runs
// along the lanes and each store instruction writes two full 128-byte lines).
#include "common.h"
#include "gemm_pp.h"

namespace tsim {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64;
constexpr int PP_XB = PP_BM * PP_BK * 2, PP_WB = PP_BN * PP_BK * 2, PP_STAGE = PP_XB + PP_WB;   // 32 + 32 KiB
constexpr int PP_LDS = 2 * PP_STAGE;                                                             // 128 KiB
constexpr int PP_PPW = PP_STAGE / 1024 / 8;                                                      // DMA pieces per wave and tile

__device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <int EPI, int KSEC>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W,
                                                      const float *__restrict__ bias, void *__restrict__ out_, int M,
                                                      int N, int K, int mtiles, int ntiles) {
    constexpr int NSEC = 4 / KSEC;
    constexpr bool F32 = EPI == PP_EPI_F32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int r = lane & 31, h = lane >> 5;

    // tile order: the ntiles feature tiles of one token tile share blockIdx % 8, i.e. one XCD's L2 (speed only)
    const int b = blockIdx.x, xcd = b & 7, jj = b >> 3;
    const int nt_id = jj % ntiles;
    const int mt_id = (jj / ntiles) * 8 + xcd;
    if (mt_id >= mtiles) return;
    const int m0 = mt_id * PP_BM, n0 = nt_id * PP_BN;

    // ---- LDS image of an operand region: 128-byte tile rows, two per 256-byte super-row, 16-byte chunk c of
    // super-row sr stored at chunk c ^ (sr & 15): the 32 rows x 2 k-halves of a ds_read_b128 fragment hit 16 distinct
    // chunks per 16 lanes (conflict-free).  Piece p = 1 KiB = 64 lanes x 16 B, LDS-linear.
    int src_off[PP_PPW];
#pragma unroll
    for (int i = 0; i < PP_PPW; ++i) {
        const int p = wave + i * 8;                       // < 32: X piece, else W piece
        const int sl = (p & 31) * 64 + lane;
        const int sr = sl >> 4, chp = sl & 15;
        const int ch = chp ^ (sr & 15);
        src_off[i] = (sr * 2 + (ch >> 3)) * K * 2 + (ch & 7) * 16;
    }
    const char *xbase = reinterpret_cast<const char *>(X + (int64_t)m0 * K);
    const char *wbase = reinterpret_cast<const char *>(W + (int64_t)n0 * K);
    auto issue = [&](int kt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PP_PPW; ++i) {
            const int p = wave + i * 8;
            glds16((i < PP_PPW / 2 ? xbase : wbase) + src_off[i] + kt * (PP_BK * 2), smem + slot * PP_STAGE + p * 1024);
        }
    };
    auto frag_off = [&](int row) __attribute__((always_inline)) {   // k-step 0; k-step s: XOR (s << 5)
        const int sr = row >> 1;
        return (uint32_t)(sr * 256 + ((((row & 1) * 8 + h) ^ (sr & 15)) << 4));
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
    uint32_t xoff[4], woff[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) xoff[j] = frag_off(grp * 128 + j * 32 + r);
#pragma unroll
    for (int i = 0; i < 2; ++i) woff[i] = PP_XB + frag_off(wq * 64 + i * 32 + r);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    const int nk = K / PP_BK;
    issue(0, 0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one interval behind group 0

    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t sbase = lds0 + (kt & 1) * PP_STAGE;
#pragma unroll
        for (int sec = 0; sec < NSEC; ++sec) {
            // ---------------- L: fragments of KSEC k-steps (+ the next k-tile's DMA)
            u32x4 xf[KSEC][4], wf[KSEC][2];
#pragma unroll
            for (int ks = 0; ks < KSEC; ++ks) {
                const uint32_t sx = (uint32_t)((sec * KSEC + ks) << 5);
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[ks][i] = lds_read_b128(sbase + (woff[i] ^ sx));
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[ks][j] = lds_read_b128(sbase + (xoff[j] ^ sx));
            }
            if (sec == 0 && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
            if (sec == NSEC - 1 && grp == 1) wait_vmcnt<0>();
            if constexpr (KSEC == 1)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[0][2]), "+v"(xf[0][3])
                             :: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[0][2]), "+v"(xf[0][3]),
                               "+v"(wf[KSEC - 1][0]), "+v"(wf[KSEC - 1][1]), "+v"(xf[KSEC - 1][0]), "+v"(xf[KSEC - 1][1]),
                               "+v"(xf[KSEC - 1][2]), "+v"(xf[KSEC - 1][3])
                             :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KSEC; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8 wv = __builtin_bit_cast(bf16x8, wf[ks][i]);
                        const bf16x8 xv = __builtin_bit_cast(bf16x8, xf[ks][j]);
                        if constexpr (F32)   // rows (registers) = tokens, columns (lanes) = features
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv, wv, acc[i][j], 0, 0, 0);
                        else                 // rows (registers) = features, columns (lanes) = tokens
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
            if (sec == NSEC - 1 && grp == 0) wait_vmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();           // pairs with group 1's last barrier: everyone is done with LDS

    if constexpr (F32) {
        // acc[i][j][g]: token m0 + grp*128 + j*32 + (g&3) + 8*(g>>2) + 4*h, feature n0 + wq*64 + i*32 + r
        float *out = reinterpret_cast<float *>(out_);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + wq * 64 + i * 32 + r;
            const float bv = bias[n];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int64_t m = m0 + grp * 128 + j * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;   // < padded row count
                    out[m * N + n] = acc[i][j][g] + bv;
                }
        }
    } else {
        // acc[i][j][g]: feature n0 + wq*64 + i*32 + (g&3) + 8*(g>>2) + 4*h, token m0 + grp*128 + j*32 + r.
        // Tile image in LDS: row = token (512 B), 16-byte slot c at c ^ (row & 15).
        bf16_t *out = reinterpret_cast<bf16_t *>(out_);
        auto tile_addr = [&](int row, int nloc) __attribute__((always_inline)) {
            return smem + row * (PP_BN * 2) + (((nloc >> 3) ^ (row & 15)) << 4) + ((nloc & 4) << 1);
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int nloc = wq * 64 + i * 32 + 8 * gq + 4 * h;
                const float4 bv = *reinterpret_cast<const float4 *>(bias + n0 + nloc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y0 = acc[i][j][4 * gq + 0] + bv.x, y1 = acc[i][j][4 * gq + 1] + bv.y;
                    float y2 = acc[i][j][4 * gq + 2] + bv.z, y3 = acc[i][j][4 * gq + 3] + bv.w;
                    if constexpr (EPI == PP_EPI_GELU) {
                        y0 = gelu_erf(y0);
                        y1 = gelu_erf(y1);
                        y2 = gelu_erf(y2);
                        y3 = gelu_erf(y3);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(y0, y1);
                    o.y = pack_bf16x2(y2, y3);
                    *reinterpret_cast<uint2 *>(tile_addr(grp * 128 + j * 32 + r, nloc)) = o;
                }
            }
        __syncthreads();
        char *obase = reinterpret_cast<char *>(out + (int64_t)m0 * N + n0);
#pragma unroll 4
        for (int sl = threadIdx.x; sl < PP_BM * 32; sl += 512) {
            const int row = sl >> 5, cp = sl & 31;       // rows past M land in the padded tail of the buffer
            *reinterpret_cast<uint4 *>(obase + (int64_t)row * N * 2 + ((cp ^ (row & 15)) << 4)) =
                *reinterpret_cast<const uint4 *>(smem + sl * 16);
        }
    }
}

// out[m, :] = LayerNorm(y[m, :] + res[m, :]) * gamma + beta, one wave per token row, fp32 statistics (two-pass).
// HBM-bound: H * (4 + 2 + 2) bytes per row.
template <int VPL4>   // float4 groups per lane = H / 256
__global__ __launch_bounds__(256) void res_ln_rows_kernel(const float *__restrict__ y, const bf16_t *__restrict__ res,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float eps, int M, int H, bf16_t *__restrict__ out) {
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
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) q += (v[i][e] - mean) * (v[i][e] - mean);
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < VPL4; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
        const float4 be = *reinterpret_cast<const float4 *>(beta + c);
        uint2 o;
        o.x = pack_bf16x2((v[i][0] - mean) * rstd * g.x + be.x, (v[i][1] - mean) * rstd * g.y + be.y);
        o.y = pack_bf16x2((v[i][2] - mean) * rstd * g.z + be.z, (v[i][3] - mean) * rstd * g.w + be.w);
        *reinterpret_cast<uint2 *>(out + row * H + c) = o;
    }
}

template <int EPI, int KSEC>
static int launch_pp(const bf16_t *X, const bf16_t *W, const float *bias, void *out, int M, int N, int K, hipStream_t st) {
    auto kern = gemm_pp_kernel<EPI, KSEC>;
    static bool attr_done = false;
    if (!attr_done) {
        TSIM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
        attr_done = true;
    }
    const int mtiles = (M + PP_BM - 1) / PP_BM, ntiles = N / PP_BN;
    const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), PP_LDS, st, X, W, bias, out, M, N, K, mtiles, ntiles);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

bool gemm_pp_supported(int N, int K) { return N % PP_BN == 0 && K % PP_BK == 0 && K >= 2 * PP_BK; }

int gemm_pp(int epi, const bf16_t *X, const bf16_t *W, const float *bias, void *out, int M, int N, int K,
            hipStream_t st) {
    if (!gemm_pp_supported(N, K)) return fail(TSIM_EUNSUPPORTED, "gemm_pp: N=%d K=%d not tileable by 256x64", N, K);
    if (M <= 0) return TSIM_OK;
    static int ksec = -1;
    if (ksec < 0) { const char *e = getenv("TSIM_PP_KSEC"); ksec = e ? atoi(e) : 2; }
#define PP_GO(E)                                                                      \
    return ksec == 1 ? launch_pp<E, 1>(X, W, bias, out, M, N, K, st) : launch_pp<E, 2>(X, W, bias, out, M, N, K, st)
    switch (epi) {
        case PP_EPI_BIAS: PP_GO(PP_EPI_BIAS);
        case PP_EPI_GELU: PP_GO(PP_EPI_GELU);
        case PP_EPI_F32: PP_GO(PP_EPI_F32);
        default: return fail(TSIM_EINVAL, "gemm_pp: unknown epilogue %d", epi);
    }
#undef PP_GO
}

int res_ln_rows(const float *y, const bf16_t *res, const float *gamma, const float *beta, float eps, bf16_t *out, int M,
                int H, hipStream_t st) {
    if (M <= 0) return TSIM_OK;
    const unsigned g = (unsigned)((M + 3) / 4);
    switch (H) {
        case 768: hipLaunchKernelGGL(res_ln_rows_kernel<3>, dim3(g), dim3(256), 0, st, y, res, gamma, beta, eps, M, H, out); break;
        case 256: hipLaunchKernelGGL(res_ln_rows_kernel<1>, dim3(g), dim3(256), 0, st, y, res, gamma, beta, eps, M, H, out); break;
        case 512: hipLaunchKernelGGL(res_ln_rows_kernel<2>, dim3(g), dim3(256), 0, st, y, res, gamma, beta, eps, M, H, out); break;
        case 1024: hipLaunchKernelGGL(res_ln_rows_kernel<4>, dim3(g), dim3(256), 0, st, y, res, gamma, beta, eps, M, H, out); break;
        default: return fail(TSIM_EUNSUPPORTED, "res_ln_rows: hidden size %d (256, 512, 768, 1024)", H);
    }
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

}  // namespace tsim
