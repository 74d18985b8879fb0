// Similarity-search kernels for gfx950 (MI355X): row L2-normalise, fused MFMA cosine + per-query top-k,
// candidate finalisation (fixed-order re-scoring), list merge, dense cos_sim, masked mean-pool.
//
// Reference call sites replaced (see include/tsim.h for the per-function citations):
//   /root/reference/src/pipeline/search_pipeline.py:73-78   expand_as + F.cosine_similarity + torch.topk
//   /root/reference/src/utils/metrics.py:81-101             cos_sim
//   /root/reference/src/modules/modules.py:158-171          AvgPoolingStrategy.forward
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "k1_topk.h"

namespace tsim {

// =====================================================================================================
// l2norm_rows: one wave per row, canonical float64 scale (common.h).  HBM-bound: reads rows*d*(4|2) B, writes
// rows*ld_out*2 B.
// =====================================================================================================
template <typename T>
__device__ __forceinline__ float load_as_f32(const T *p);
template <>
__device__ __forceinline__ float load_as_f32<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }

template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const T *__restrict__ x, int64_t rows, int d,
                                                          int64_t ld_in, bf16_t *__restrict__ out, int ld_out,
                                                          float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T *xr = x + row * ld_in;
    double ss = 0.0;
    for (int j = lane; j < d; j += 64) {
        const double v = (double)load_as_f32<T>(xr + j);
        ss = fma(v, v, ss);
    }
    const double inv = canonical_inv_norm(ss, eps);
    bf16_t *o = out + row * (int64_t)ld_out;
    for (int j = lane; j < ld_out; j += 64) o[j] = j < d ? canonical_unit_elem(load_as_f32<T>(xr + j), inv) : (bf16_t)0;
}

// =====================================================================================================
// K2: cos_topk_finalize — one wave per query.
//   1. select the KL best of the P2*KL partial entries by (MFMA score desc, index asc);
//   2. re-score them in the canonical order: float64 accumulation over j = 0..D-1, one rounding to f32;
//   3. order by (canonical score desc, index asc) and emit the first k.
// Step 2 is what makes results independent of MFMA summation order, tiling and shard count.
// =====================================================================================================
__device__ __forceinline__ bool key_before(float s1, int i1, float s2, int i2) {
    // true if (s1,i1) ranks strictly ahead of (s2,i2)
    return s1 > s2 || (s1 == s2 && i1 < i2);
}

template <int KL>
__global__ __launch_bounds__(256) void cos_topk_finalize_kernel(const float *__restrict__ part_s,
                                                                const int *__restrict__ part_i, int P2,
                                                                int64_t Q, const bf16_t *__restrict__ eq,
                                                                const bf16_t *__restrict__ ec, int D, int k,
                                                                float *__restrict__ out_s,
                                                                int64_t *__restrict__ out_i,
                                                                int64_t idx_offset) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int64_t E = (int64_t)P2 * KL;
    const float *ps = part_s + q * E;
    const int *pi = part_i + q * E;

    // 1a. one pass: every lane keeps the KL best of its E/64 entries in a sorted register list
    float ls[KL];
    int li[KL];
#pragma unroll
    for (int j = 0; j < KL; ++j) {
        ls[j] = -INFINITY;
        li[j] = 0x7fffffff;
    }
    auto consider = [&](float s, int i) {
        if (i >= 0 && key_before(s, i, ls[KL - 1], li[KL - 1])) {
#pragma unroll
            for (int j = 0; j < KL; ++j) {
                const bool ahead = key_before(s, i, ls[j], li[j]);
                const float ns = ahead ? ls[j] : s;
                const int ni = ahead ? li[j] : i;
                ls[j] = ahead ? s : ls[j];
                li[j] = ahead ? i : li[j];
                s = ns;
                i = ni;
            }
        }
    };
    // Lists are sorted and padded with (-inf, -1): a lane walks whole lists (list = lane, lane+64, ...), reads the
    // first four row ids (16 B) and stops at the first pad.  With pre-pass thresholds most lists are empty, so this
    // touches 16 B per list instead of its 8*KL bytes.
    for (int l = lane; l < P2; l += 64) {
        const float *lsrc = ps + (int64_t)l * KL;
        const int *isrc = pi + (int64_t)l * KL;
#pragma unroll 1
        for (int j = 0; j < KL; j += 4) {
            const int4 iv = *reinterpret_cast<const int4 *>(isrc + j);
            if (iv.x < 0) break;
            const float4 sv = *reinterpret_cast<const float4 *>(lsrc + j);
            consider(sv.x, iv.x);
            consider(sv.y, iv.y);
            consider(sv.z, iv.z);
            consider(sv.w, iv.w);
            if (iv.w < 0) break;
        }
    }

    // 1b. KL rounds of a 64-way merge of the list heads; the winning lane pops its head
    float my_s = -INFINITY;  // lane t < KL ends up holding the t-th selected candidate
    int my_i = -1;
    for (int t = 0; t < KL; ++t) {
        float bs = ls[0];
        int bi = li[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (key_before(os, oi, bs, bi)) {
                bs = os;
                bi = oi;
            }
        }
        if (bi == 0x7fffffff) break;  // fewer than KL valid entries (uniform)
        if (lane == t) {
            my_s = bs;
            my_i = bi;
        }
        if (li[0] == bi) {  // row ids are unique across partitions: exactly one lane owns the winner
#pragma unroll
            for (int j = 0; j + 1 < KL; ++j) {
                ls[j] = ls[j + 1];
                li[j] = li[j + 1];
            }
            ls[KL - 1] = -INFINITY;
            li[KL - 1] = 0x7fffffff;
        }
    }
    (void)my_s;

    // canonical re-score (lanes holding a candidate)
    float cs = -INFINITY;
    if (my_i >= 0) {
        const bf16_t *qp = eq + q * D;
        const bf16_t *cp = ec + (int64_t)my_i * D;
        double acc = 0.0;
        for (int j = 0; j < D; j += 8) {
            const uint4 qa = *reinterpret_cast<const uint4 *>(qp + j);
            const uint4 ca = *reinterpret_cast<const uint4 *>(cp + j);
            const uint32_t qw[4] = {qa.x, qa.y, qa.z, qa.w};
            const uint32_t cw[4] = {ca.x, ca.y, ca.z, ca.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = fma((double)__uint_as_float(qw[u] << 16), (double)__uint_as_float(cw[u] << 16), acc);
                acc = fma((double)__uint_as_float(qw[u] & 0xffff0000u),
                          (double)__uint_as_float(cw[u] & 0xffff0000u), acc);
            }
        }
        cs = (float)acc;
    }
    // final order among the <= KL candidates: rank by counting
    int rank = 0;
    for (int t = 0; t < KL; ++t) {
        const float os = __shfl(cs, t, 64);
        const int oi = __shfl(my_i, t, 64);
        if (oi >= 0 && key_before(os, oi, cs, my_i)) rank++;
    }
    if (lane < KL && my_i >= 0 && rank < k) {
        out_s[q * k + rank] = cs;
        out_i[q * k + rank] = (int64_t)my_i + idx_offset;
    }
    // fewer valid candidates than k: pad
    int nvalid = __popcll(__ballot(my_i >= 0));
    if (lane >= nvalid && lane < k) {
        out_s[q * k + lane] = -INFINITY;
        out_i[q * k + lane] = -1;
    }
}

// =====================================================================================================
// thr_select: gthr[q] = ordered-int form of the KL-th largest of the query's P2 block maxima (pre-pass output,
// [Q][P2] floats).  One wave per query; KL rounds of (lane-local max, wave max, owner removes one instance).
// =====================================================================================================
__global__ __launch_bounds__(256) void thr_select_kernel(const float *__restrict__ bmax, int P2, int64_t Q, int KL,
                                                         int *__restrict__ gthr) {
    constexpr int VPL = K1_PREPASS_MAX_P2 / 64;
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    float v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int e = lane + 64 * i;
        v[i] = e < P2 ? bmax[q * P2 + e] : -INFINITY;
    }
    float best = -INFINITY;
    for (int t = 0; t < KL; ++t) {
        float lm = v[0];
#pragma unroll
        for (int i = 1; i < VPL; ++i) lm = fmaxf(lm, v[i]);
        best = wave_max(lm);
        const unsigned long long owners = __ballot(lm == best);
        if (lane == __ffsll((long long)owners) - 1) {   // one owner removes one instance
            bool done = false;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const bool hit = !done && v[i] == best;
                v[i] = hit ? -INFINITY : v[i];
                done = done || hit;
            }
        }
    }
    if (lane == 0) gthr[q] = best > -INFINITY ? float_to_ordered(best) : K1_GTHR_INIT;
}

// =====================================================================================================
// merge of sorted per-shard / per-chunk lists: [nlists, Q, k_in] -> [Q, k_out]; one wave per query.
// =====================================================================================================
__device__ __forceinline__ bool key_before64(float s1, int64_t i1, float s2, int64_t i2) {
    return s1 > s2 || (s1 == s2 && i1 < i2);
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ scores,
                                                         const int64_t *__restrict__ idx, int nlists,
                                                         int64_t Q, int k_in, int k_out,
                                                         float *__restrict__ out_s,
                                                         int64_t *__restrict__ out_i) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int E = nlists * k_in;
    float last_s = INFINITY;
    int64_t last_i = -1;
    for (int t = 0; t < k_out; ++t) {
        float bs = -INFINITY;
        int64_t bi = INT64_MAX;
        for (int e = lane; e < E; e += 64) {
            const int l = e / k_in, j = e % k_in;
            const int64_t off = ((int64_t)l * Q + q) * k_in + j;
            const float s = scores[off];
            const int64_t i = idx[off];
            if (i >= 0 && key_before64(last_s, last_i, s, i) && key_before64(s, i, bs, bi)) {
                bs = s;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int64_t oi = __shfl_xor(bi, o, 64);
            if (key_before64(os, oi, bs, bi)) {
                bs = os;
                bi = oi;
            }
        }
        const bool none = bi == INT64_MAX;
        if (lane == 0) {
            out_s[q * k_out + t] = none ? -INFINITY : bs;
            out_i[q * k_out + t] = none ? -1 : bi;
        }
        if (!none) {
            last_s = bs;
            last_i = bi;
        } else {
            last_s = -INFINITY;
            last_i = INT64_MAX;  // nothing ranks after this: remaining slots pad
        }
    }
}

// =====================================================================================================
// dense cos_sim (A8): float32, rows normalised by division exactly like the reference, 64x64 tiles.
// Evaluation-sized inputs only; not on the search path.
// =====================================================================================================
__global__ __launch_bounds__(256) void cos_sim_kernel(const float *__restrict__ a, int64_t na,
                                                      const float *__restrict__ b, int64_t nb, int d,
                                                      float *__restrict__ out) {
    __shared__ float sa[16][65], sb[16][65];
    __shared__ float norm_a[64], norm_b[64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    // row norms of this tile's 64 + 64 rows: 4 threads per row
    {
        const int rr = threadIdx.x >> 2, part = threadIdx.x & 3;
        float s1 = 0.f, s2 = 0.f;
        if (i0 + rr < na)
            for (int kk = part; kk < d; kk += 4) { float v = a[(i0 + rr) * d + kk]; s1 = fmaf(v, v, s1); }
        if (j0 + rr < nb)
            for (int kk = part; kk < d; kk += 4) { float v = b[(j0 + rr) * d + kk]; s2 = fmaf(v, v, s2); }
        s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
        s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
        if (part == 0) { norm_a[rr] = sqrtf(s1); norm_b[rr] = sqrtf(s2); }
    }
    __syncthreads();
    float acc[4][4] = {};
    for (int k0 = 0; k0 < d; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int rr = e >> 4, kk = e & 15;
            float va = 0.f, vb = 0.f;
            if (k0 + kk < d) {
                if (i0 + rr < na) va = a[(i0 + rr) * d + k0 + kk] / norm_a[rr];
                if (j0 + rr < nb) vb = b[(j0 + rr) * d + k0 + kk] / norm_b[rr];
            }
            sa[kk][rr] = va;
            sb[kk][rr] = vb;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = sa[kk][ty * 4 + u]; bv[u] = sb[kk][tx * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(av[u], bv[v], acc[u][v]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t i = i0 + ty * 4 + u, j = j0 + tx * 4 + v;
            if (i < na && j < nb) out[i * nb + j] = acc[u][v];
        }
}

// =====================================================================================================
// masked mean-pool on a padded [B,S,H] tensor (A4).  One thread per (b, h); HBM-bound.
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void mean_pool_kernel(const T *__restrict__ hidden,
                                                        const int32_t *__restrict__ mask, int S, int H,
                                                        float *__restrict__ out) {
    const int64_t bidx = blockIdx.y;
    const int hh = blockIdx.x * 256 + threadIdx.x;
    if (hh >= H) return;
    const T *hp = hidden + bidx * S * (int64_t)H + hh;
    const int32_t *mp = mask + bidx * S;
    float sum = 0.f, msum = 0.f;
    for (int s = 0; s < S; ++s) {
        const float m = (float)mp[s];
        sum = fmaf(load_as_f32<T>(hp + (int64_t)s * H), m, sum);
        msum += m;
    }
    out[bidx * H + hh] = sum / fmaxf(msum, 1e-9f);
}

// =====================================================================================================
// host side
// =====================================================================================================
static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace tsim

using namespace tsim;

extern "C" int tsim_pad_dim(int d) {
    const int sizes[] = {128, 256, 384, 512, 768};
    for (int s : sizes)
        if (d <= s) return s;
    return 0;
}

extern "C" int tsim_l2norm_rows(const void *x, int x_dtype, int64_t rows, int d, int64_t ld_in, void *out_bf16,
                                int ld_out, float eps, void *stream) {
    TSIM_REQUIRE(x && out_bf16, "l2norm_rows: null pointer");
    TSIM_REQUIRE(rows >= 0 && d > 0 && ld_in >= d && ld_out >= d, "l2norm_rows: bad shape rows=%lld d=%d ld_in=%lld ld_out=%d",
                 (long long)rows, d, (long long)ld_in, ld_out);
    if (rows == 0) return TSIM_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (x_dtype == TSIM_F32)
        hipLaunchKernelGGL(l2norm_rows_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream),
                           (const float *)x, rows, d, ld_in, (bf16_t *)out_bf16, ld_out, eps);
    else if (x_dtype == TSIM_BF16)
        hipLaunchKernelGGL(l2norm_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream),
                           (const bf16_t *)x, rows, d, ld_in, (bf16_t *)out_bf16, ld_out, eps);
    else
        return fail(TSIM_EINVAL, "l2norm_rows: unknown dtype %d", x_dtype);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

extern "C" void tsim_time_next_topk(void *start_event, void *stop_event) {
    g_ev_start = reinterpret_cast<hipEvent_t>(start_event);
    g_ev_stop = reinterpret_cast<hipEvent_t>(stop_event);
}

extern "C" size_t tsim_cosine_topk_workspace_bytes(int64_t Q, int64_t N, int k) {
    if (Q <= 0 || N <= 0 || k <= 0 || k > 28) return 0;
    // the plan depends on the padded width only through the wave count; take the larger (4-wave) layout
    TopkPlan a, b;
    plan_topk(Q, N, 384, k, &a);
    plan_topk(Q, N, 768, k, &b);
    const size_t e = a.part_elems > b.part_elems ? a.part_elems : b.part_elems;
    return e * 8 + (size_t)Q * 4 + (size_t)Q * K1_PREPASS_MAX_P2 * 4 + 256;
}

extern "C" int tsim_cosine_topk(const void *eq, int64_t Q, const void *ec, int64_t N, int d, int ld, int k,
                                float *out_scores, int64_t *out_idx, int64_t idx_offset, void *workspace,
                                size_t workspace_bytes, void *stream) {
    TSIM_REQUIRE(eq && ec && out_scores && out_idx, "cosine_topk: null pointer");
    TSIM_REQUIRE(Q > 0 && N > 0, "cosine_topk: empty input Q=%lld N=%lld", (long long)Q, (long long)N);
    TSIM_REQUIRE(k >= 1 && k <= 28, "cosine_topk: k=%d outside 1..28", k);
    TSIM_REQUIRE(N < (1ll << 31) - 64 && Q < (1ll << 31) - 512, "cosine_topk: shard too large for 32-bit row ids");
    TSIM_REQUIRE(ld == tsim_pad_dim(d) && ld > 0, "cosine_topk: rows must be padded to tsim_pad_dim(d)=%d (got ld=%d)",
                 tsim_pad_dim(d), ld);
    TSIM_REQUIRE((((uintptr_t)eq | (uintptr_t)ec) & 15) == 0, "cosine_topk: embedding matrices must be 16-byte aligned");
    TopkPlan p;
    plan_topk(Q, N, ld, k, &p);
    const size_t need = p.part_elems * 8 + (size_t)Q * 4 + (size_t)Q * K1_PREPASS_MAX_P2 * 4;
    if (!workspace || workspace_bytes < need)
        return fail(TSIM_ENOMEM, "cosine_topk: workspace %zu B < %zu B", workspace_bytes, need);
    float *part_s = reinterpret_cast<float *>(workspace);
    int *part_i = reinterpret_cast<int *>(part_s + p.part_elems);
    int *gthr = part_i + p.part_elems;   // per-query shared threshold words, re-initialised every call
    float *bmax = reinterpret_cast<float *>(gthr + Q);
    hipStream_t st = as_stream(stream);
    TopkPlan pp;
    static int prepass_on = -1;
    if (prepass_on < 0) { const char *e = getenv("TSIM_K1_PREPASS"); prepass_on = e ? atoi(e) : 1; }
    if (prepass_on && plan_prepass(Q, N, p, &pp)) {
        // threshold pre-pass: block maxima over the first rows, KL-th largest per query -> initial shared bounds
        const int64_t S = (int64_t)pp.nchunks * pp.rows_per_chunk;
        int rc0 = k1_launch_blockmax(pp, ld, (const bf16_t *)eq, Q, (const bf16_t *)ec, S, bmax, st);
        if (rc0) return rc0;
        hipLaunchKernelGGL(thr_select_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, st, bmax, pp.P2, Q, p.KL, gthr);
        TSIM_HIP_CHECK(hipGetLastError());
    } else {
        TSIM_HIP_CHECK(hipMemsetAsync(gthr, 0x80, (size_t)Q * 4, st));
    }
#ifdef TSIM_PP_STAMPS
    if (getenv("TSIM_K1_DIAG_NOSEL"))   // DIAGNOSTIC: thresholds nothing can pass (results are wrong): the time without selection
        TSIM_HIP_CHECK(hipMemsetAsync(gthr, 0x7f, (size_t)Q * 4, st));
#endif
    hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    if (ev0) TSIM_HIP_CHECK(hipEventRecord(ev0, st));
    int rc = p.KL == 16 ? k1_launch_kl16(p, ld, (const bf16_t *)eq, Q, (const bf16_t *)ec, N, part_s, part_i, gthr, st)
                        : k1_launch_kl32(p, ld, (const bf16_t *)eq, Q, (const bf16_t *)ec, N, part_s, part_i, gthr, st);
    if (rc) return rc;
    if (ev1) TSIM_HIP_CHECK(hipEventRecord(ev1, st));
    const unsigned grid = (unsigned)((Q + 3) / 4);
    if (p.KL == 16)
        hipLaunchKernelGGL(cos_topk_finalize_kernel<16>, dim3(grid), dim3(256), 0, st, part_s, part_i, p.P2, Q,
                           (const bf16_t *)eq, (const bf16_t *)ec, ld, k, out_scores, out_idx, idx_offset);
    else
        hipLaunchKernelGGL(cos_topk_finalize_kernel<32>, dim3(grid), dim3(256), 0, st, part_s, part_i, p.P2, Q,
                           (const bf16_t *)eq, (const bf16_t *)ec, ld, k, out_scores, out_idx, idx_offset);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

extern "C" int tsim_topk_merge(const float *scores, const int64_t *idx, int nlists, int64_t Q, int k_in, int k_out,
                               float *out_scores, int64_t *out_idx, void *stream) {
    TSIM_REQUIRE(scores && idx && out_scores && out_idx, "topk_merge: null pointer");
    TSIM_REQUIRE(nlists >= 1 && Q >= 0 && k_in >= 1 && k_out >= 1, "topk_merge: bad shape");
    if (Q == 0) return TSIM_OK;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, as_stream(stream), scores,
                       idx, nlists, Q, k_in, k_out, out_scores, out_idx);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

extern "C" int tsim_cos_sim(const float *a, int64_t na, const float *b, int64_t nb, int d, float *out, void *stream) {
    TSIM_REQUIRE(a && b && out, "cos_sim: null pointer");
    TSIM_REQUIRE(na >= 0 && nb >= 0 && d > 0, "cos_sim: bad shape");
    if (na == 0 || nb == 0) return TSIM_OK;
    dim3 grid((unsigned)((nb + 63) / 64), (unsigned)((na + 63) / 64));
    hipLaunchKernelGGL(cos_sim_kernel, grid, dim3(256), 0, as_stream(stream), a, na, b, nb, d, out);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

extern "C" int tsim_mean_pool(const void *hidden, int hidden_dtype, const int32_t *mask, int64_t B, int S, int H,
                              float *out, void *stream) {
    TSIM_REQUIRE(hidden && mask && out, "mean_pool: null pointer");
    TSIM_REQUIRE(B >= 0 && S > 0 && H > 0 && B < 65536, "mean_pool: bad shape B=%lld S=%d H=%d", (long long)B, S, H);
    if (B == 0) return TSIM_OK;
    dim3 grid((unsigned)((H + 255) / 256), (unsigned)B);
    if (hidden_dtype == TSIM_F32)
        hipLaunchKernelGGL(mean_pool_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float *)hidden,
                           mask, S, H, out);
    else if (hidden_dtype == TSIM_BF16)
        hipLaunchKernelGGL(mean_pool_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t *)hidden,
                           mask, S, H, out);
    else
        return fail(TSIM_EINVAL, "mean_pool: unknown dtype %d", hidden_dtype);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}
