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
template <>
__device__ __forceinline__ float load_as_f32<unit_t>(const unit_t *p) { return (float)*p; }

// rho_max (optional): the largest rounding residual rho_r = || half(u_r) - u_r ||_2 of the rows written, u_r = the exact unit
// row x_r / max(|x_r|, eps) — the quantity the search's exactness guard is built on (guard_eps below).  Accumulated with an
// atomic max on the float's bit pattern (non-negative floats order like ints), so one word can span many calls (an index
// that grows): the caller zeroes it once.
template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const T *__restrict__ x, int64_t rows, int d,
                                                          int64_t ld_in, unit_t *__restrict__ out, int ld_out,
                                                          float eps, float *__restrict__ rho_max) {
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
    unit_t *o = out + row * (int64_t)ld_out;
    double r2 = 0.0;
    for (int j = lane; j < ld_out; j += 64) {
        unit_t hv = (unit_t)0;
        if (j < d) {
            const double u = (double)load_as_f32<T>(xr + j) * inv;
            hv = f64_to_f16(u);
            const double e = (double)(float)hv - u;
            r2 = fma(e, e, r2);
        }
        o[j] = hv;
    }
    if (rho_max) {
        const float rho = rho_round_up(sqrt(wave_sum_f64(r2)));
        if (lane == 0) rho_publish(rho_max, rho);
    }
}

// =====================================================================================================
// Exact scores.  Two definitions, both evaluated in float64 in ONE canonical order and rounded once to float32, so that
// GPU and oracle agree bit for bit (oracle/search_ref._lane_sum): lane l accumulates the products of elements j = l,
// l + 64, ... in that order (fma of an exact product == multiply + add), then an xor butterfly 32, 16, .., 1.
//   COS = true  (float32 rows): the reference's score, F.cosine_similarity of the float32 embeddings
//                /root/reference/src/pipeline/search_pipeline.py:76-77: x.y / (max(|x|, eps) * max(|y|, eps))
//                (oracle/search_ref.exact_cosine);
//   COS = false (unit rows as stored): the inner product of the stored unit rows (oracle/search_ref.canonical_scores).
// MFMA scores only SELECT candidates; every score that is returned or compared for the final order is one of these.
// =====================================================================================================
constexpr int XS_MAXI = 12;   // 64 * 12 = 768 elements per row at most
constexpr double XS_EPS = (double)1e-8f;

template <typename T>
struct ExactQuery {
    double v[XS_MAXI];   // this lane's elements j = lane + 64 i of the query row (0 beyond d)
    double norm;         // COS: max(|q|, eps)
};

// This lane's elements j = lane + 64 i, i < NI, of a row as float64 (0 beyond d).  BRANCH-FREE: the address of an element past
// the end is clamped to the row's last one and the value replaced after the load.  (`j < d ? load : 0` compiles to a branch
// around every load with s_waitcnt vmcnt(0) behind it: the six loads of a 384-wide row became six dependent round trips, the
// 16 candidates of a query 96 — 45 of the finalize kernel's 88 us at Q = 256.)  1 <= d <= 64 NI.
template <typename T, int NI>
__device__ __forceinline__ void row_elems_f64(double (&c)[NI], const T *row, int d, int lane) {
    float x[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = lane + 64 * i;
        x[i] = load_as_f32<T>(row + (j < d ? j : d - 1));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) c[i] = lane + 64 * i < d ? (double)x[i] : 0.0;
}

template <typename T, bool COS, int NI>
__device__ __forceinline__ void exact_load_query_ni(ExactQuery<T> &q, const T *row, int d, int lane) {
    double c[NI];
    row_elems_f64<T, NI>(c, row, d, lane);
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < XS_MAXI; ++i) {
        q.v[i] = i < NI ? c[i < NI ? i : 0] : 0.0;
        ss = fma(q.v[i], q.v[i], ss);   // (+0 beyond d: the sum is that of the elements below d, in their order)
    }
    q.norm = 1.0;
    if constexpr (COS) q.norm = fmax(sqrt(wave_sum_f64(ss)), XS_EPS);
}
template <typename T, bool COS>
__device__ __forceinline__ void exact_load_query(ExactQuery<T> &q, const T *row, int d, int lane) {
    if (d <= 64 * (XS_MAXI / 2)) exact_load_query_ni<T, COS, XS_MAXI / 2>(q, row, d, lane);   // wave-uniform
    else exact_load_query_ni<T, COS, XS_MAXI>(q, row, d, lane);
}

// per-lane partial sums of the query against one row: dot (and |row|^2 for COS)
template <typename T, bool COS, int NI>
__device__ __forceinline__ void exact_partials(const ExactQuery<T> &q, const T *row, int d, int lane, double &dot, double &cc) {
    double c[NI];
    row_elems_f64<T, NI>(c, row, d, lane);
    dot = 0.0;
    cc = 0.0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        dot = fma(q.v[i], c[i], dot);
        if constexpr (COS) cc = fma(c[i], c[i], cc);
    }
}

// exact score of the query against one row (all 64 lanes take part and get the same value)
template <typename T, bool COS>
__device__ __forceinline__ float exact_score(const ExactQuery<T> &q, const T *row, int d, int lane) {
    double dot, cc;
    if (d <= 64 * (XS_MAXI / 2)) exact_partials<T, COS, XS_MAXI / 2>(q, row, d, lane, dot, cc);   // wave-uniform
    else exact_partials<T, COS, XS_MAXI>(q, row, d, lane, dot, cc);
    dot = wave_sum_f64(dot);
    if constexpr (COS) {
        const double nc = fmax(sqrt(wave_sum_f64(cc)), XS_EPS);
        return (float)(dot / (q.norm * nc));
    }
    return (float)dot;
}

// NB rows against the query at once: lane t0 + u (u < NB) returns the exact score of the row held (as my_i) by lane
// t0 + u < nvalid; other lanes return garbage.  Same bits as exact_score: per lane the same fma chains, and the wave sums are
// the SAME xor-butterfly trees (32, 16, .., 1) — evaluated as a reduce-scatter: with NV values to sum, a lane sends the half of
// them its partner keeps and adds what it receives to the half it keeps itself (own + partner commutes, so both lanes of a
// pair would have computed the same bits), NV/2 + NV/4 + .. exchanges instead of 6 NV.  All NB rows' loads are in flight
// together: one memory round trip per batch — at small Q this kernel is a chain of round trips and butterflies and nothing else
// (one wave per query, 64 workgroups at Q = 256: 45 of its 88 us were four dependent gathers of four rows each).
template <int N, int O, int NV>   // N values still held per lane, next exchange with lane ^ O (compile-time indices throughout)
__device__ __forceinline__ void butterfly_reduce_scatter(double (&acc)[NV], int lane) {
    if constexpr (O > 0) {
        if constexpr (N > 1) {
            const bool up = (lane & O) != 0;   // this lane keeps the upper half
#pragma unroll
            for (int v = 0; v < N / 2; ++v) {
                const double send = up ? acc[v] : acc[v + N / 2];
                const double keep = up ? acc[v + N / 2] : acc[v];
                acc[v] = keep + __shfl_xor(send, O, 64);
            }
            butterfly_reduce_scatter<N / 2, O / 2>(acc, lane);
        } else {
            acc[0] += __shfl_xor(acc[0], O, 64);
            butterfly_reduce_scatter<1, O / 2>(acc, lane);
        }
    }
}

template <typename T, bool COS, int NB>
__device__ __forceinline__ float exact_score_batch(const ExactQuery<T> &q, const T *xc, int64_t ldc, int my_i, int t0, int nvalid,
                                                   int d, int lane) {
    constexpr int NV = (COS ? 2 : 1) * NB;
    static_assert(NV >= 2 && NV <= 32 && (NV & (NV - 1)) == 0, "batch size");
    constexpr int LOG = NV == 32 ? 5 : NV == 16 ? 4 : NV == 8 ? 3 : NV == 4 ? 2 : 1;
    double acc[NV];
    const T *rows[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int t = t0 + u < nvalid ? t0 + u : t0;   // past the end: repeat a valid one, result unused
        rows[u] = xc + (int64_t)__shfl(my_i, t, 64) * ldc;
    }
    if (d <= 64 * (XS_MAXI / 2)) {   // wave-uniform
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            double dot, cc;
            exact_partials<T, COS, XS_MAXI / 2>(q, rows[u], d, lane, dot, cc);
            if constexpr (COS) { acc[2 * u] = dot; acc[2 * u + 1] = cc; }
            else acc[u] = dot;
        }
    } else {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            double dot, cc;
            exact_partials<T, COS, XS_MAXI>(q, rows[u], d, lane, dot, cc);
            if constexpr (COS) { acc[2 * u] = dot; acc[2 * u + 1] = cc; }
            else acc[u] = dot;
        }
    }
    butterfly_reduce_scatter<NV, 32>(acc, lane);
    // value v ended up in the lanes with (lane >> (6 - LOG)) == v
    constexpr int SH = 6 - LOG;
    float score;
    if constexpr (COS) {
        const double other = __shfl_xor(acc[0], 1 << SH, 64);          // even v (dot) lanes receive cc
        const double nc = fmax(sqrt(other), XS_EPS);
        score = (float)(acc[0] / (q.norm * nc));
    } else {
        score = (float)acc[0];
    }
    const int u = lane - t0;
    const int src = ((COS ? 2 * u : u) << SH) & 63;
    return __shfl(score, src, 64);
}

__device__ __forceinline__ bool key_before(float s1, int i1, float s2, int i2) {
    // true if (s1,i1) ranks strictly ahead of (s2,i2)
    return s1 > s2 || (s1 == s2 && i1 < i2);
}
__device__ __forceinline__ float float_below(float f) {   // the next float towards -inf (finite f)
    if (f == 0.f) return -1.17549435e-38f;
    const int b = __float_as_int(f);
    return __int_as_float(f > 0.f ? b - 1 : b + 1);
}

// control words of one search call (zeroed on the stream before the first kernel)
enum { CTL_NFLAG = 0, CTL_NUNRES = 1, CTL_WORDS = 4 };
// per-query status written to out_status (all results are exact; the status says which pass produced them)
enum { ST_PASS1 = 0, ST_WIDENED = 1, ST_BRUTE = 2 };

struct GuardArgs {
    // COS (float32 rows given): eps = guard_eps(rho_q, rho_c, ld) — a BOUND on |MFMA score - exact score| for the query against
    // every row of the shard (common.h).  rho_q is computed from the query's two rows in the kernel; rho_c = *rho_c_max when
    // the caller has the measured residual maximum of the shard's unit rows (tsim_l2norm_rows), else the a-priori bound.
    // Unit rows only: the scores differ by float32 accumulation alone: eps = max(c1 * largest difference seen, floor),
    // floor = ld * 2^-23 (rigorous for unit rows; c1 covers callers whose rows are not quite unit).
    float c1;
    float floor;
    const float *rho_c_max;   // device, or null
    float rho_c_default;      // rho_apriori(ld)
    int ld;
    int *ctl;          // CTL_* words
    int *flag_q;       // [Q] flagged queries, compact
    int *flag_thr;     // [Q] per slot: collection threshold as an ordered int (k1_topk.h float_to_ordered)
    float *flag_eps;   // [Q] per slot: the query's eps (COS)
    int *unres_q;      // [Q] queries left to the brute-force pass, compact
    int *status;       // [Q] or null
};

// rho of a query row: || stored half row - exact unit row ||_2 from the float32 row already held in `q` (all 64 lanes take part)
__device__ __forceinline__ float query_rho(const ExactQuery<float> &q, const unit_t *urow, int d, int lane) {
    double r2 = 0.0;
    const double inv = 1.0 / q.norm;
    auto body = [&](auto nic) __attribute__((always_inline)) {
        constexpr int NI = decltype(nic)::value;
        double u[NI];
        row_elems_f64<unit_t, NI>(u, urow, d, lane);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const double e = u[i] - q.v[i] * inv;   // (0 - 0 beyond d)
            r2 = fma(e, e, r2);
        }
    };
    if (d <= 64 * (XS_MAXI / 2)) body(std::integral_constant<int, XS_MAXI / 2>{});   // wave-uniform
    else body(std::integral_constant<int, XS_MAXI>{});
    return rho_round_up(sqrt(wave_sum_f64(r2)));
}
__device__ __forceinline__ float guard_rho_c(const GuardArgs &g) { return g.rho_c_max ? *g.rho_c_max : g.rho_c_default; }
// collection threshold for "every row whose exact score could reach `target`": MFMA score >= target - eps.  Returned one float
// below (target - eps) so that the float rounding of the subtraction is on the safe side; eps = inf -> collect everything.
__device__ __forceinline__ float guard_tau(float target, float eps) {
    if (!(eps < 3.0e38f) || !(target > -3.0e38f)) return -3.4028234e38f;
    float tau = float_below((float)((double)target - (double)eps));
    if ((double)tau + (double)eps >= (double)target) tau = float_below(tau);
    return tau;
}

// The KL best entries of a query's partial lists by (MFMA score desc, index asc): lane t < KL returns the t-th
// (my_i = -1 when there are fewer).  ps / pi: the query's lists, nlists of KL entries each, sorted, padded with (-inf, -1).
// M = entries a LANE keeps (KL: all it could ever contribute).  The insertion network is the cost of this routine — the branch
// around it is taken whenever ANY lane has an entry to insert, i.e. nearly always: 8 VALU per position and entry, ~4 000 per
// query at KL = 16, and at Q = 4 096 the finalize kernel is VALU-throughput-bound on exactly that (45 us).  The KL winners are
// spread over 64 lanes, so a lane almost never contributes more than a few: with M = 4 the network is a quarter.  EXACTNESS is
// kept by a check, not by the odds: a lane's discarded entries all rank behind its M-th kept one, so they can only matter if the
// lane had all M of its kept entries popped by the merge; in that case (returned as false) the caller repeats with M = KL.
// (thr_update needs only a valid lower bound and never repeats.)
template <int KL, int LB = 4, int M = KL>
__device__ __forceinline__ bool select_kl_best(const float *__restrict__ ps, const int *__restrict__ pi, int P2, int lane,
                                               float &my_s_out, int &my_i_out) {
    // 1a. one pass: every lane keeps the M best of its E/64 entries in a sorted register list
    float ls[M];
    int li[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        ls[j] = -INFINITY;
        li[j] = 0x7fffffff;
    }
    bool dropped = false;   // this lane let go of an entry (or of the rest of a list) that it might have contributed
    auto consider = [&](float s, int i) {
        if (i >= 0) {
            if (key_before(s, i, ls[M - 1], li[M - 1])) {
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const bool ahead = key_before(s, i, ls[j], li[j]);
                    const float ns = ahead ? ls[j] : s;
                    const int ni = ahead ? li[j] : i;
                    ls[j] = ahead ? s : ls[j];
                    li[j] = ahead ? i : li[j];
                    s = ns;
                    i = ni;
                }
                if (M < KL && i != 0x7fffffff) dropped = true;   // a kept entry fell off the end
            } else if (M < KL) {
                dropped = true;
            }
        }
    };
    // Lists are sorted and padded with (-inf, -1): a lane walks whole lists (list = lane, lane+64, ...).  With pre-pass
    // thresholds most lists hold 0-3 entries, so the first four entries (16 + 16 B) of FOUR lists are requested together —
    // one memory round trip per 256 lists instead of two per list (the walk was a chain of dependent loads: 16 round trips
    // at Q = 256, where P2 = 512) — and only a list that is full that far is walked on.  (Eight lists at a time cost 32 more
    // registers: 160 instead of <= 128, one wave per SIMD less, and the kernel — pure latency — ran 152 us instead of 80 at
    // Q = 4096; LB = 8 is the small-batch form, where occupancy is idle anyway.)
    for (int l0 = lane; l0 < P2; l0 += 64 * LB) {
        int4 iv[LB];
        float4 sv[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int l = l0 + 64 * u;
            iv[u] = make_int4(-1, -1, -1, -1);
            sv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l < P2) {
                iv[u] = *reinterpret_cast<const int4 *>(pi + (int64_t)l * KL);
                sv[u] = *reinterpret_cast<const float4 *>(ps + (int64_t)l * KL);
            }
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            if (iv[u].x < 0) continue;
            consider(sv[u].x, iv[u].x);
            consider(sv[u].y, iv[u].y);
            consider(sv[u].z, iv[u].z);
            consider(sv[u].w, iv[u].w);
            if (iv[u].w < 0) continue;
            // the list is sorted: once an entry does not beat this lane's last kept one, nothing behind it does — and every step
            // of the walk is a dependent memory round trip (full lists, as after phase A: 3 per list, 8 lists per lane in a row)
            if (!key_before(sv[u].w, iv[u].w, ls[M - 1], li[M - 1])) {
                if (M < KL) dropped = true;   // (the list may go on)
                continue;
            }
            const float *lsrc = ps + (int64_t)(l0 + 64 * u) * KL;
            const int *isrc = pi + (int64_t)(l0 + 64 * u) * KL;
#pragma unroll 1
            for (int j = 4; j < KL; j += 4) {
                const int4 jv = *reinterpret_cast<const int4 *>(isrc + j);
                if (jv.x < 0) break;
                const float4 tv = *reinterpret_cast<const float4 *>(lsrc + j);
                consider(tv.x, jv.x);
                consider(tv.y, jv.y);
                consider(tv.z, jv.z);
                consider(tv.w, jv.w);
                if (jv.w < 0) break;
                if (!key_before(tv.w, jv.w, ls[M - 1], li[M - 1])) {
                    if (M < KL) dropped = true;
                    break;
                }
            }
        }
    }

    // 1b. KL rounds of a 64-way merge of the list heads; the winning lane pops its head
    float my_s = -INFINITY;
    int my_i = -1;
    int npop = 0;
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
            for (int j = 0; j + 1 < M; ++j) {
                ls[j] = ls[j + 1];
                li[j] = li[j + 1];
            }
            ls[M - 1] = -INFINITY;
            li[M - 1] = 0x7fffffff;
            ++npop;
        }
    }
    my_s_out = my_s;
    my_i_out = my_i;
    if constexpr (M < KL) return !__any(dropped && npop == M);
    return true;
}

// Two-phase main pass: after the list kernel has scored the first rows of the shard (phase A: lists p2_first .. +p2_count of
// every query), the KL-th best MFMA score found so far replaces the pre-pass bound in the query's shared threshold word.
// Phase B then streams the rest of the shard against a bound taken from 4x more rows than the pre-pass sample: its
// candidate events (the expensive, branchy side of the selection filter) fall accordingly.
template <int KL>
__global__ __launch_bounds__(256) void thr_update_kernel(const float *__restrict__ part_s, const int *__restrict__ part_i,
                                                         int P2_total, int p2_first, int p2_count, int64_t Q,
                                                         int *__restrict__ gthr) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    float my_s;
    int my_i;
    // (lanes keep four entries: should one lane hold more than four of the KL best, the KL-th of what was kept is still the score
    // of a real row with KL - 1 others at or above it: a valid, slightly weaker bound)
    (void)select_kl_best<KL, 4, 4>(part_s + (q * P2_total + p2_first) * KL, part_i + (q * P2_total + p2_first) * KL, p2_count, lane, my_s,
                                   my_i);
    const float kth = __shfl(my_s, KL - 1, 64);
    const int kth_i = __shfl(my_i, KL - 1, 64);
    if (lane == 0 && kth_i >= 0) atomicMax(gthr + q, float_to_ordered(kth));   // KL rows score at least kth
}

// =====================================================================================================
// K2: cos_topk_finalize — one wave per query.
//   1. select the KL best of the P2*KL partial entries by (MFMA score desc, index asc);
//   2. re-score them exactly (above);
//   3. order by (exact score desc, index asc) and emit the first k;
//   4. GUARD: every row outside the KL candidates has an MFMA score <= cut = the KL-th selected one, hence an exact
//      score <= cut + eps when eps bounds |MFMA - exact| for the query.  eps is estimated from the candidates themselves
//      (c1 x the largest difference observed, never below `floor`).  If cut + eps < the k-th exact score nothing outside
//      can reach the list and the result stands; otherwise the query is FLAGGED: the widening pass (K1 in COLLECT mode)
//      gathers every row whose MFMA score exceeds (k-th exact score - eps) and widen_finalize re-scores all of them.
// =====================================================================================================
// NB: rows per exact re-score batch, LB: lists per walk round trip.  <4, 4>: 120 registers, four waves per SIMD (large Q: the
// kernel is latency-bound and lives on occupancy); <16 or 8, 8>: everything in flight at once for Q <= 1024, where at most one
// workgroup per CU exists anyway.
template <int KL, typename T, bool COS, int NB, int LB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB > 4 ? 1 : KL == 16 ? 4 : 2))) void cos_topk_finalize_kernel(const float *__restrict__ part_s,
                                                                const int *__restrict__ part_i, int P2,
                                                                int64_t Q, int64_t N, const T *__restrict__ xq, int64_t ldq,
                                                                const T *__restrict__ xc, int64_t ldc, int d, int k,
                                                                const unit_t *__restrict__ uq, const int *__restrict__ gthr,
                                                                float *__restrict__ out_s,
                                                                int64_t *__restrict__ out_i,
                                                                int64_t idx_offset, GuardArgs g) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int64_t E = (int64_t)P2 * KL;
    const float *ps = part_s + q * E;
    const int *pi = part_i + q * E;

    float my_s;  // lane t < KL ends up holding the t-th selected candidate
    int my_i;
#if defined(TSIM_FIN_DIAG) && (TSIM_FIN_DIAG & 1)   // TIMING-ONLY (wrong results): no selection, the first list's entries
    my_s = lane < KL ? ps[lane] : -INFINITY;
    my_i = lane < KL ? pi[lane] : -1;
    if (my_i < 0 && lane < KL) { my_i = (int)((q * 977 + lane * 131) % N); my_s = 0.f; }
#else
    // (lanes keeping four entries + a repeat with all KL when the exactness check fails measured SLOWER here, 56-58 us against
    // 44-49: after the main pass the lists are short and the second instantiation costs more than the smaller network saves;
    // thr_update, whose lists are full, gains: 27 -> 16 us)
    (void)select_kl_best<KL, LB, KL>(ps, pi, P2, lane, my_s, my_i);
#endif
    const int nvalid = __popcll(__ballot(my_i >= 0));   // candidates sit in lanes 0 .. nvalid-1

    // 2. exact re-score: the wave works on one candidate at a time (coalesced row reads)
    ExactQuery<T> eqr;
    exact_load_query<T, COS>(eqr, xq + q * ldq, d, lane);
    float cs = -INFINITY;
#pragma unroll 1
    for (int t0 = 0; t0 < nvalid; t0 += NB) {   // NB candidates per step: their row reads overlap, their wave sums share shuffles
#if defined(TSIM_FIN_DIAG) && (TSIM_FIN_DIAG & 2)   // TIMING-ONLY (wrong results): no exact re-score
        const float sc = my_s;
#else
        const float sc = exact_score_batch<T, COS, NB>(eqr, xc, ldc, my_i, t0, nvalid, d, lane);
#endif
        if (lane >= t0 && lane < t0 + NB && lane < nvalid) cs = sc;
    }
    // 3. final order among the candidates: rank by counting
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
    if (lane >= nvalid && lane < k) {   // fewer valid candidates than k: pad
        out_s[q * k + lane] = -INFINITY;
        out_i[q * k + lane] = -1;
    }
    // 4. guard.  Rows that are not among the candidates: either they lost against the KL-th list entry (MFMA score <= it), or
    // the list kernel dropped them against the query's shared bound, whose final (largest) value is gthr[q] — also when the
    // bound came from a kernel that accumulates in another order (pre-pass in the 16x16x32 form, lists of 32 in the 32x32x16
    // form): cut = max of the two bounds whatever their origin.  Fewer than KL entries mean "every row was a candidate" only
    // if nothing was ever filtered (the bound word still holds its initial value).
    const int bkey = gthr ? gthr[q] : K1_GTHR_INIT;
    const bool filtered = bkey > K1_GTHR_INIT;
    bool safe = N <= KL || (nvalid < KL && !filtered);
    float tau = 0.f, eps = 0.f;
    if (!safe) {
        const float err = wave_max(lane < nvalid ? fabsf(my_s - cs) : 0.f);
        if constexpr (COS) {
            eps = guard_eps(query_rho(eqr, uq + q * g.ld, d, lane), guard_rho_c(g), g.ld);
            // the bound must hold on the candidates too; if it does not, the unit rows are not the canonical images of the
            // float32 rows (or rho_c_max is stale): trust nothing, score the whole shard exactly
            if (!(err <= eps)) eps = INFINITY;
        } else {
            eps = fmaxf(g.c1 * err, g.floor);
        }
        float cut = nvalid >= KL ? __shfl(my_s, KL - 1, 64) : -INFINITY;
        if (filtered) cut = fmaxf(cut, ordered_to_float(bkey));
        const unsigned long long kth = __ballot(lane < nvalid && rank == k - 1);   // at most one lane
        const float sk = kth ? __shfl(cs, __ffsll((long long)kth) - 1, 64) : -INFINITY;   // fewer than k candidates: no k-th score
        safe = kth != 0 && (double)cut + (double)eps < (double)sk;
        tau = guard_tau(sk, eps);   // rows at or below tau cannot reach sk
    }
    if (lane == 0) {
        if (!safe) {
            const int slot = atomicAdd(g.ctl + CTL_NFLAG, 1);
            g.flag_q[slot] = (int)q;
            g.flag_thr[slot] = float_to_ordered(tau);
            g.flag_eps[slot] = eps;
        }
        if (g.status) g.status[q] = safe ? ST_PASS1 : ST_WIDENED;
    }
}

// k rounds of "best entry strictly after the previous winner" over LDS arrays sc/ix[0..n): a 256-thread workgroup.
// red_s/red_i: 4-entry scratch.  Writes (-inf, -1) when the entries run out.  emit(t, score, row) is called by thread 0.
template <typename EMIT>
__device__ __forceinline__ void wg_select_topk(const float *sc, const int *ix, int n, int k, float *red_s, int *red_i,
                                               EMIT emit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float last_s = INFINITY;
    int last_i = -1;
    for (int t = 0; t < k; ++t) {
        float bs = -INFINITY;
        int bi = 0x7fffffff;
        for (int e = threadIdx.x; e < n; e += 256) {
            const float s = sc[e];
            const int i = ix[e];
            if (i >= 0 && key_before(last_s, last_i, s, i) && key_before(s, i, bs, bi)) {
                bs = s;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (key_before(os, oi, bs, bi)) {
                bs = os;
                bi = oi;
            }
        }
        if (lane == 0) {
            red_s[wave] = bs;
            red_i[wave] = bi;
        }
        __syncthreads();
        bs = red_s[0];
        bi = red_i[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (key_before(red_s[w], red_i[w], bs, bi)) {
                bs = red_s[w];
                bi = red_i[w];
            }
        __syncthreads();
        const bool none = bi == 0x7fffffff;
        if (threadIdx.x == 0) emit(t, none ? -INFINITY : bs, none ? -1 : bi);
        if (none) {
            last_s = -INFINITY;
            last_i = 0x7fffffff;   // nothing ranks after this: the remaining slots pad
        } else {
            last_s = bs;
            last_i = bi;
        }
    }
}

// =====================================================================================================
// widen_finalize: one workgroup per FLAGGED query (slot).  Re-scores every entry the widening pass collected for it
// (every row whose MFMA score exceeded the slot's threshold), selects the exact top-k and repeats the guard with the
// threshold that was actually used and the errors seen on this larger sample.  An overflowed buffer or a failed guard
// hands the query to the brute-force pass.
// =====================================================================================================
constexpr int COLL_CAP = 1024;   // entries per slot

template <typename T, bool COS>
__global__ __launch_bounds__(256) void widen_finalize_kernel(const unsigned long long *__restrict__ coll_buf,
                                                             const int *__restrict__ coll_cnt, int64_t Q, int64_t N,
                                                             const T *__restrict__ xq, int64_t ldq,
                                                             const T *__restrict__ xc, int64_t ldc, int d, int k,
                                                             float *__restrict__ out_s, int64_t *__restrict__ out_i,
                                                             int64_t idx_offset, GuardArgs g) {
    __shared__ float sc[COLL_CAP];
    __shared__ int ix[COLL_CAP];
    __shared__ float red_s[4];
    __shared__ int red_i[4];
    __shared__ float s_err[4];
    __shared__ float s_top[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int nflag = g.ctl[CTL_NFLAG];
    nflag = nflag < Q ? nflag : (int)Q;
    for (int slot = blockIdx.x; slot < nflag; slot += gridDim.x) {
        const int q = g.flag_q[slot];
        const int cnt = coll_cnt[slot];
        const int n = cnt < COLL_CAP ? cnt : COLL_CAP;
        bool resolved = cnt <= COLL_CAP;
        if (resolved) {   // workgroup-uniform
            ExactQuery<T> eqr;
            exact_load_query<T, COS>(eqr, xq + (int64_t)q * ldq, d, lane);
            float err = 0.f;
            for (int e = wave; e < n; e += 4) {
                const unsigned long long ent = coll_buf[(int64_t)slot * COLL_CAP + e];
                const int row = (int)(ent >> 32);
                const float s = exact_score<T, COS>(eqr, xc + (int64_t)row * ldc, d, lane);
                err = fmaxf(err, fabsf(__uint_as_float((uint32_t)ent) - s));
                if (lane == 0) {
                    sc[e] = s;
                    ix[e] = row;
                }
            }
            if (lane == 0) s_err[wave] = err;
            __syncthreads();
            wg_select_topk(sc, ix, n, k, red_s, red_i, [&](int t, float s, int row) {
                out_s[(int64_t)q * k + t] = s;
                out_i[(int64_t)q * k + t] = row < 0 ? -1 : (int64_t)row + idx_offset;
                s_top[t] = s;
            });
            __syncthreads();
            // guard again with the threshold the collection used: every row that was NOT collected has an MFMA score below
            // it, hence an exact score below thr + eps.  COS: eps is the query's bound from the first pass (and it must hold on
            // everything that was re-scored here, else the inputs are inconsistent -> brute force); unit rows only: the largest
            // difference seen on this larger sample with half the safety factor of the first pass (not below 1).
            const float errmax = fmaxf(fmaxf(s_err[0], s_err[1]), fmaxf(s_err[2], s_err[3]));
            float eps;
            if constexpr (COS) {
                eps = g.flag_eps[slot];
                if (!(errmax <= eps)) eps = INFINITY;
            } else {
                eps = fmaxf(fmaxf(0.5f * g.c1, 1.f) * errmax, g.floor);
            }
            const float thr = ordered_to_float(g.flag_thr[slot]);
            resolved = n >= k && (double)thr + (double)eps < (double)s_top[k - 1];
        }
        if (threadIdx.x == 0) {
            if (!resolved) g.unres_q[atomicAdd(g.ctl + CTL_NUNRES, 1)] = q;
            if (g.status) g.status[q] = resolved ? ST_WIDENED : ST_BRUTE;
        }
        __syncthreads();
    }
}

// =====================================================================================================
// Brute-force exact pass for the queries nothing else resolved: every row of the shard is scored exactly.
//   bf_partial: workgroup (chunk c, slot u): rows of chunk c in blocks of BF_SB; a running exact top-k per chunk.
//   bf_merge:   one workgroup per slot merges the NCH chunk lists and writes the query's final list.
// HBM-bound (N*d*4 bytes per query); last resort, and the serving path for k > 28 on small shards.
// =====================================================================================================
constexpr int BF_SB = 2048;
constexpr int BF_MAXK = 64;

template <typename T, bool COS>
__global__ __launch_bounds__(256) void bf_partial_kernel(int64_t Q, int64_t N, int rows_per_chunk,
                                                         const T *__restrict__ xq, int64_t ldq,
                                                         const T *__restrict__ xc, int64_t ldc, int d, int k,
                                                         float *__restrict__ bf_s, int *__restrict__ bf_i, GuardArgs g) {
    __shared__ float sc[BF_SB + BF_MAXK];
    __shared__ int ix[BF_SB + BF_MAXK];
    __shared__ float top_s[BF_MAXK];
    __shared__ int top_i[BF_MAXK];
    __shared__ float red_s[4];
    __shared__ int red_i[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = gridDim.x, chunk = blockIdx.x;
    int nu = g.ctl[CTL_NUNRES];
    nu = nu < Q ? nu : (int)Q;
    const int64_t r0 = (int64_t)chunk * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < N ? r0 + rows_per_chunk : N;
    for (int u = blockIdx.y; u < nu; u += gridDim.y) {
        const int q = g.unres_q[u];
        ExactQuery<T> eqr;
        exact_load_query<T, COS>(eqr, xq + (int64_t)q * ldq, d, lane);
        if (threadIdx.x < BF_MAXK) {
            top_s[threadIdx.x] = -INFINITY;
            top_i[threadIdx.x] = -1;
        }
        __syncthreads();
        for (int64_t b = r0; b < r1; b += BF_SB) {
            const int nb = (int)(r1 - b < BF_SB ? r1 - b : BF_SB);
            for (int e = wave; e < nb; e += 4) {
                const float s = exact_score<T, COS>(eqr, xc + (b + e) * ldc, d, lane);
                if (lane == 0) {
                    sc[e] = s;
                    ix[e] = (int)(b + e);
                }
            }
            if (threadIdx.x < k) {   // the running list competes with the new block
                sc[nb + threadIdx.x] = top_s[threadIdx.x];
                ix[nb + threadIdx.x] = top_i[threadIdx.x];
            }
            __syncthreads();
            wg_select_topk(sc, ix, nb + k, k, red_s, red_i, [&](int t, float s, int row) {
                top_s[t] = s;
                top_i[t] = row;
            });
            __syncthreads();
        }
        if (threadIdx.x < k) {
            bf_s[((int64_t)u * nch + chunk) * k + threadIdx.x] = top_s[threadIdx.x];
            bf_i[((int64_t)u * nch + chunk) * k + threadIdx.x] = top_i[threadIdx.x];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void bf_merge_kernel(int64_t Q, int nch, int k, const float *__restrict__ bf_s,
                                                       const int *__restrict__ bf_i, float *__restrict__ out_s,
                                                       int64_t *__restrict__ out_i, int64_t idx_offset, GuardArgs g) {
    __shared__ float red_s[4];
    __shared__ int red_i[4];
    int nu = g.ctl[CTL_NUNRES];
    nu = nu < Q ? nu : (int)Q;
    for (int u = blockIdx.x; u < nu; u += gridDim.x) {
        const int q = g.unres_q[u];
        wg_select_topk(bf_s + (int64_t)u * nch * k, bf_i + (int64_t)u * nch * k, nch * k, k, red_s, red_i,
                       [&](int t, float s, int row) {
                           out_s[(int64_t)q * k + t] = s;
                           out_i[(int64_t)q * k + t] = row < 0 ? -1 : (int64_t)row + idx_offset;
                       });
        if (threadIdx.x == 0 && g.status) g.status[q] = ST_BRUTE;
    }
}

// k > 28: every query goes to the widening pass (or straight to the brute-force pass: all_brute).  gthr[q] = B, the k-th
// largest block maximum = a lower bound of the k-th best MFMA score: k rows score >= B on the MFMA, hence >= B - eps exactly,
// so the k-th best EXACT score is >= B - eps and every row of the exact top-k has an MFMA score >= B - 2 eps: that is the
// collection threshold.  One wave per query (COS: the query's rho comes from its two rows).
template <bool COS>
__global__ __launch_bounds__(256) void flag_all_kernel(int64_t Q, const int *__restrict__ gthr, bool all_brute,
                                                       const float *__restrict__ xq, int64_t ldq, const unit_t *__restrict__ uq,
                                                       int d, GuardArgs g) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q == 0 && lane == 0) g.ctl[all_brute ? CTL_NUNRES : CTL_NFLAG] = (int)Q;
    if (q >= Q) return;
    float eps = g.floor;
    if constexpr (COS) {
        if (!all_brute) {
            ExactQuery<float> eqr;
            exact_load_query<float, true>(eqr, xq + q * ldq, d, lane);
            eps = guard_eps(query_rho(eqr, uq + q * g.ld, d, lane), guard_rho_c(g), g.ld);
        }
    }
    if (lane != 0) return;
    if (all_brute) {
        g.unres_q[q] = (int)q;
    } else {
        g.flag_q[q] = (int)q;
        const int key = gthr[q];
        g.flag_thr[q] = key <= K1_GTHR_INIT ? key : float_to_ordered(guard_tau(ordered_to_float(key), 2.f * eps * 1.000001f));
        g.flag_eps[q] = eps;
    }
    if (g.status) g.status[q] = all_brute ? ST_BRUTE : ST_WIDENED;
}

// =====================================================================================================
// thr_select: gthr[q] = ordered-int form of the KL-th largest of the query's P2 block maxima (pre-pass output,
// [Q][P2] floats).  One wave per query; KL rounds of (lane-local max, wave max, owner removes one instance).
// =====================================================================================================
// VPL = values per lane >= P2 / 64.  (With the one 32-value form every round cost ~140 VALU instructions whatever P2 was — 20 us at
// any Q, and four waves per SIMD share the pipe; typical P2 is 512.)
// zero_base / zero_words: the call's control words and per-slot counters, cleared here instead of by a memset of their own when
// this kernel is the one in front of the main pass (one launch and its gap less per search).
template <int VPL>
__global__ __launch_bounds__(256) void thr_select_kernel(const float *__restrict__ bmax, int P2, int64_t Q, int KL,
                                                         int *__restrict__ gthr, int *__restrict__ zero_base, int zero_words) {
    for (int w = blockIdx.x * 256 + threadIdx.x; w < zero_words; w += gridDim.x * 256) zero_base[w] = 0;
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

static void launch_thr_select(const float *bmax, int P2, int64_t Q, int KL, int *gthr, hipStream_t st, int *zero_base = nullptr,
                              int zero_words = 0) {
    const dim3 grid((unsigned)((Q + 3) / 4));
    static_assert(K1_PREPASS_MAX_P2 == 2048, "dispatch below covers P2 <= 2048");
    if (P2 <= 512) hipLaunchKernelGGL(thr_select_kernel<8>, grid, dim3(256), 0, st, bmax, P2, Q, KL, gthr, zero_base, zero_words);
    else if (P2 <= 1024) hipLaunchKernelGGL(thr_select_kernel<16>, grid, dim3(256), 0, st, bmax, P2, Q, KL, gthr, zero_base, zero_words);
    else hipLaunchKernelGGL(thr_select_kernel<32>, grid, dim3(256), 0, st, bmax, P2, Q, KL, gthr, zero_base, zero_words);
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
                                                         int64_t lstride_s, int64_t lstride_i,
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
            const int64_t off = q * k_in + j;
            const float s = scores[(int64_t)l * lstride_s + off];
            const int64_t i = idx[(int64_t)l * lstride_i + off];
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

extern "C" int tsim_l2norm_rows(const void *x, int x_dtype, int64_t rows, int d, int64_t ld_in, void *out_f16,
                                int ld_out, float eps, float *rho_max, void *stream) {
    TSIM_REQUIRE(x && out_f16, "l2norm_rows: null pointer");
    TSIM_REQUIRE(rows >= 0 && d > 0 && ld_in >= d && ld_out >= d, "l2norm_rows: bad shape rows=%lld d=%d ld_in=%lld ld_out=%d",
                 (long long)rows, d, (long long)ld_in, ld_out);
    if (rows == 0) return TSIM_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (x_dtype == TSIM_F32)
        hipLaunchKernelGGL(l2norm_rows_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream),
                           (const float *)x, rows, d, ld_in, (unit_t *)out_f16, ld_out, eps, rho_max);
    else if (x_dtype == TSIM_BF16)
        hipLaunchKernelGGL(l2norm_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream),
                           (const bf16_t *)x, rows, d, ld_in, (unit_t *)out_f16, ld_out, eps, rho_max);
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

namespace tsim {
constexpr int TOPK_MAX_LISTS = 28;   // largest k the list kernels (KL = 32) serve
constexpr int TOPK_MAX_K = BF_MAXK;  // largest k at all (widening / brute-force passes)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Workspace layout of one search call (byte offsets).
struct SearchWs {
    size_t part_s, part_i, gthr, bmax, ctl, flag_q, flag_thr, flag_eps, unres_q, coll_cnt, coll_buf, bf_s, bf_i, total;
    int bf_nch, bf_rows;
};

static void plan_workspace(int64_t Q, int64_t N, int k, SearchWs *w) {
    size_t part = 0;
    if (k <= TOPK_MAX_LISTS) {   // the plan depends on the padded width only through the wave count: take the larger layout
        for (int D : {384, 768}) {
            TopkPlan a, pa, pb;
            plan_topk(Q, N, D, k, &a);
            size_t e = a.part_elems;
            if (N >= 4 * 131072) {   // two-phase main pass: the lists of both launches side by side
                plan_topk(Q, 131072, D, k, &pa);
                plan_topk(Q, N - 131072, D, k, &pb);
                if (pa.part_elems + pb.part_elems > e) e = pa.part_elems + pb.part_elems;
            }
            if (e > part) part = e;
        }
    }
    int nch = (int)((N + 255) / 256);
    w->bf_nch = nch < 64 ? (nch < 1 ? 1 : nch) : 64;
    w->bf_rows = (int)((N + w->bf_nch - 1) / w->bf_nch);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align256(bytes); return at; };
    w->part_s = take(part * 4);
    w->part_i = take(part * 4);
    w->gthr = take((size_t)Q * 4);
    w->bmax = take((size_t)Q * K1_PREPASS_MAX_P2 * 4);
    w->ctl = take(CTL_WORDS * 4);
    w->flag_q = take((size_t)Q * 4);
    w->flag_thr = take((size_t)Q * 4);
    w->flag_eps = take((size_t)Q * 4);
    w->unres_q = take((size_t)Q * 4);
    w->coll_cnt = take((size_t)Q * 4);
    w->coll_buf = take((size_t)Q * COLL_CAP * 8);
    w->bf_s = take((size_t)Q * w->bf_nch * k * 4);
    w->bf_i = take((size_t)Q * w->bf_nch * k * 4);
    w->total = o;
}

// plan of the widening pass: enough chunks that ONE flagged query block still spreads over the chip
static void plan_collect(int64_t Q, int64_t N, int D, TopkPlan *p) {
    plan_topk(Q, N, D, 1, p);
    int64_t nch = 256, max_ch = (N + 255) / 256;
    if (nch > max_ch) nch = max_ch;
    if (nch < 8) {   // the kernel's block map wants 1, 2, 4 or >= 8 chunks
        int p2 = 1;
        while (p2 * 2 <= nch) p2 *= 2;
        nch = p2;
    }
    int64_t rpc = (N + nch - 1) / nch;
    rpc = (rpc + K1_TILE_ROWS - 1) / K1_TILE_ROWS * K1_TILE_ROWS;
    p->rows_per_chunk = (int)rpc;
    p->nchunks = (int)((N + rpc - 1) / rpc);
    if (p->nchunks < 8 && (p->nchunks & (p->nchunks - 1))) {   // rounding produced 3, 5, 6 or 7 chunks: use fewer, longer ones
        int p2 = 1;
        while (p2 * 2 <= p->nchunks) p2 *= 2;
        rpc = (N + p2 - 1) / p2;
        rpc = (rpc + K1_TILE_ROWS - 1) / K1_TILE_ROWS * K1_TILE_ROWS;
        p->rows_per_chunk = (int)rpc;
        p->nchunks = (int)((N + rpc - 1) / rpc);
    }
    p->P2 = p->nchunks * 2;
}

// block maxima over the WHOLE shard in >= 2*kneed partitions (k > 28: the kneed-th largest block maximum is a lower bound
// of the kneed-th best MFMA score); false when the shard is too small for that many partitions
static bool plan_fullmax(int64_t Q, int64_t N, int D, int kneed, TopkPlan *p) {
    plan_topk(Q, N, D, 1, p);
    int64_t nch = 512 / p->nqb;
    if (nch < kneed) nch = kneed;            // P2 = 2 nch >= 2 kneed
    if (nch > K1_PREPASS_MAX_P2 / 2) nch = K1_PREPASS_MAX_P2 / 2;
    int64_t rpc = (N + nch - 1) / nch;
    rpc = (rpc + K1_TILE_ROWS - 1) / K1_TILE_ROWS * K1_TILE_ROWS;
    p->rows_per_chunk = (int)rpc;
    p->nchunks = (int)((N + rpc - 1) / rpc);
    p->P2 = p->nchunks * 2;
    p->part_elems = (size_t)Q * p->P2;
    // every partition (a lane half of a chunk: rows 4h..4h+3 of each group of 8) must hold at least one row
    return p->nchunks >= 8 && p->P2 >= kneed && p->P2 <= K1_PREPASS_MAX_P2 && N - (int64_t)(p->nchunks - 1) * rpc >= 8;
}

// Two-phase main pass (large shards): phase A = the first K1_PHASE_A_ROWS rows, phase B = the rest.
constexpr int64_t K1_PHASE_A_ROWS = 131072;
static bool plan_two_phase(int64_t Q, int64_t N, int D, int k, TopkPlan *pa, TopkPlan *pb) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("TSIM_K1_PHASES"); on = e ? atoi(e) : 1; }
    // one or two query blocks: the chip is filled by corpus chunks alone and two extra launches cost more than the tighter
    // bound saves (Q = 256: 0.37 -> 0.44 ms); from four query blocks on the second phase wins (Q = 4096: -3 %)
    if (!on || N < 4 * K1_PHASE_A_ROWS || Q <= 768) return false;
    plan_topk(Q, K1_PHASE_A_ROWS, D, k, pa);
    plan_topk(Q, N - K1_PHASE_A_ROWS, D, k, pb);
    return true;
}

static float guard_c1() {
    static float c1 = -1.f;
    if (c1 < 0.f) { const char *e = getenv("TSIM_GUARD_C1"); c1 = e ? (float)atof(e) : 4.0f; if (!(c1 >= 1.f)) c1 = 1.f; }
    return c1;
}

template <typename T, bool COS>
static int search_tail(const SearchWs &w, char *ws, int64_t Q, int64_t N, const unit_t *eq, const unit_t *ec, int ld,
                       const T *xq, int64_t ldq, const T *xc, int64_t ldc, int d, int k, float *out_s, int64_t *out_i,
                       int64_t idx_offset, const GuardArgs &g, bool run_collect, hipStream_t st) {
    // widening pass + its finalisation (workgroups leave at once when nothing was flagged) ...
    if (run_collect) {
        TopkPlan cp;
        plan_collect(Q, N, ld, &cp);
        K1Collect coll{};
        coll.qcount = g.ctl + CTL_NFLAG;
        coll.qmap = g.flag_q;
        coll.buf = reinterpret_cast<unsigned long long *>(ws + w.coll_buf);
        coll.cnt = reinterpret_cast<int *>(ws + w.coll_cnt);
        coll.cap = COLL_CAP;
        int rc = k1_launch_collect(cp, ld, eq, Q, ec, N, g.flag_thr, coll, st);
        if (rc) return rc;
        const unsigned wg = (unsigned)(Q < 2048 ? Q : 2048);
        hipLaunchKernelGGL((widen_finalize_kernel<T, COS>), dim3(wg), dim3(256), 0, st, coll.buf, coll.cnt, Q, N, xq, ldq, xc,
                           ldc, d, k, out_s, out_i, idx_offset, g);
        TSIM_HIP_CHECK(hipGetLastError());
    }
    // ... and the brute-force pass for whatever is still unresolved
    float *bf_s = reinterpret_cast<float *>(ws + w.bf_s);
    int *bf_i = reinterpret_cast<int *>(ws + w.bf_i);
    const unsigned us = (unsigned)(Q < 64 ? Q : 64);
    hipLaunchKernelGGL((bf_partial_kernel<T, COS>), dim3(w.bf_nch, us), dim3(256), 0, st, Q, N, w.bf_rows, xq, ldq, xc, ldc, d, k,
                       bf_s, bf_i, g);
    TSIM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(bf_merge_kernel, dim3((unsigned)(Q < 1024 ? Q : 1024)), dim3(256), 0, st, Q, w.bf_nch, k, bf_s, bf_i,
                       out_s, out_i, idx_offset, g);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

template <int KL, typename T, bool COS>
static void launch_finalize(const TopkPlan &p, const float *part_s, const int *part_i, int64_t Q, int64_t N, const T *xq,
                            int64_t ldq, const T *xc, int64_t ldc, int d, int k, const unit_t *uq, const int *gthr, float *out_s,
                            int64_t *out_i, int64_t idx_offset, const GuardArgs &g, hipStream_t st) {
    const dim3 grid((unsigned)((Q + 3) / 4));
    constexpr int LBW = KL == 16 ? 8 : 4;   // (eight lists of 32 at a time spill)
    static int wide_on = -1;
    if (wide_on < 0) { const char *e = getenv("TSIM_FIN_WIDE"); wide_on = e ? atoi(e) : 1; }
    if (wide_on && Q <= 1024 && d <= 384)
        hipLaunchKernelGGL((cos_topk_finalize_kernel<KL, T, COS, 16, LBW>), grid, dim3(256), 0, st, part_s, part_i, p.P2, Q, N, xq, ldq,
                           xc, ldc, d, k, uq, gthr, out_s, out_i, idx_offset, g);
    else if (wide_on && Q <= 1024)
        hipLaunchKernelGGL((cos_topk_finalize_kernel<KL, T, COS, 8, LBW>), grid, dim3(256), 0, st, part_s, part_i, p.P2, Q, N, xq, ldq,
                           xc, ldc, d, k, uq, gthr, out_s, out_i, idx_offset, g);
    else
        hipLaunchKernelGGL((cos_topk_finalize_kernel<KL, T, COS, 4, 4>), grid, dim3(256), 0, st, part_s, part_i, p.P2, Q, N, xq, ldq,
                           xc, ldc, d, k, uq, gthr, out_s, out_i, idx_offset, g);
}
}  // namespace tsim

extern "C" int tsim_cosine_topk_plan(int64_t Q, int64_t N, int ld, int k, int32_t plan[4]) {
    if (Q <= 0 || N <= 0 || k <= 0 || k > TOPK_MAX_K || !plan || tsim_pad_dim(ld) != ld)
        return fail(TSIM_EINVAL, "tsim_cosine_topk_plan: bad arguments (Q=%lld N=%lld ld=%d k=%d)", (long long)Q, (long long)N, ld, k);
    TopkPlan p;
    plan_topk(Q, N, ld, k <= TOPK_MAX_LISTS ? k : 10, &p);
    plan[0] = p.nqb;
    plan[1] = p.nchunks;
    plan[2] = p.rows_per_chunk;
    // block mapping of cos_topk_partial_kernel: >= 8 chunks: XCD x owns chunks x, x+8, ... with all their query blocks;
    // fewer: 8 / nchunks XCDs share a chunk and split its query blocks
    plan[3] = p.nchunks >= 8 ? ((p.nchunks + 7) / 8) * p.nqb : (p.nqb + (8 / p.nchunks) - 1) / (8 / p.nchunks);
    return TSIM_OK;
}

extern "C" size_t tsim_cosine_topk_workspace_bytes(int64_t Q, int64_t N, int k) {
    if (Q <= 0 || N <= 0 || k <= 0 || k > TOPK_MAX_K) return 0;
    SearchWs w;
    plan_workspace(Q, N, k, &w);
    return w.total;
}

extern "C" int tsim_cosine_topk_ex(const void *eq, const float *eq_f32, int64_t ldq_f32, int64_t Q, const void *ec,
                                   const float *ec_f32, int64_t ldc_f32, const float *ec_rho_max, int64_t N, int d, int ld, int k,
                                   float *out_scores, int64_t *out_idx, int32_t *out_status, int64_t idx_offset,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    TSIM_REQUIRE(eq && ec && out_scores && out_idx, "cosine_topk: null pointer");
    TSIM_REQUIRE(Q > 0 && N > 0, "cosine_topk: empty input Q=%lld N=%lld", (long long)Q, (long long)N);
    TSIM_REQUIRE(k >= 1 && k <= TOPK_MAX_K, "cosine_topk: k=%d outside 1..%d", k, TOPK_MAX_K);
    TSIM_REQUIRE(N < (1ll << 31) - 64 && Q < (1ll << 31) - 512, "cosine_topk: shard too large for 32-bit row ids");
    TSIM_REQUIRE(ld == tsim_pad_dim(d) && ld > 0, "cosine_topk: rows must be padded to tsim_pad_dim(d)=%d (got ld=%d)",
                 tsim_pad_dim(d), ld);
    TSIM_REQUIRE((((uintptr_t)eq | (uintptr_t)ec) & 15) == 0, "cosine_topk: embedding matrices must be 16-byte aligned");
    TSIM_REQUIRE((eq_f32 == nullptr) == (ec_f32 == nullptr), "cosine_topk: pass both float32 matrices or neither");
    const bool cosf = eq_f32 != nullptr;
    if (cosf) TSIM_REQUIRE(ldq_f32 >= d && ldc_f32 >= d, "cosine_topk: float32 row strides %lld/%lld < d=%d", (long long)ldq_f32,
                           (long long)ldc_f32, d);
    SearchWs w;
    plan_workspace(Q, N, k, &w);
    if (!workspace || workspace_bytes < w.total)
        return fail(TSIM_ENOMEM, "cosine_topk: workspace %zu B < %zu B", workspace_bytes, w.total);
    char *ws = reinterpret_cast<char *>(workspace);
    float *part_s = reinterpret_cast<float *>(ws + w.part_s);
    int *part_i = reinterpret_cast<int *>(ws + w.part_i);
    int *gthr = reinterpret_cast<int *>(ws + w.gthr);   // per-query shared threshold words, re-initialised every call
    float *bmax = reinterpret_cast<float *>(ws + w.bmax);
    hipStream_t st = as_stream(stream);
    const unit_t *uq = (const unit_t *)eq, *uc = (const unit_t *)ec;

    GuardArgs g;
    g.c1 = guard_c1();
    // unit rows only: float32 accumulation of ld exact products of unit rows, any order, rounding or truncation per step
    // (ld * 2^-23 |a||b|, |a|,|b| <= 1 + 2^-10) + the final rounding of the exact score
    g.floor = (float)ld * 1.1920929e-7f * 1.003f + 2.4e-7f;
    g.rho_c_max = ec_rho_max;
    g.rho_c_default = rho_apriori(ld);
    g.ld = ld;
    g.flag_eps = reinterpret_cast<float *>(ws + w.flag_eps);
    g.ctl = reinterpret_cast<int *>(ws + w.ctl);
    g.flag_q = reinterpret_cast<int *>(ws + w.flag_q);
    g.flag_thr = reinterpret_cast<int *>(ws + w.flag_thr);
    g.unres_q = reinterpret_cast<int *>(ws + w.unres_q);
    g.status = out_status;
    // ctl .. coll_cnt are contiguous: one memset clears the control words and the per-slot counters — or the threshold kernel of
    // the pre-pass does (nothing in front of it touches them)
    const size_t ctl_bytes = w.coll_cnt + align256((size_t)Q * 4) - w.ctl;
    bool ctl_cleared = false;

    bool run_collect = true;
    if (k <= TOPK_MAX_LISTS) {
        TopkPlan p, pp;
        plan_topk(Q, N, ld, k, &p);
        static int prepass_on = -1;
        if (prepass_on < 0) { const char *e = getenv("TSIM_K1_PREPASS"); prepass_on = e ? atoi(e) : 1; }
        if (prepass_on && plan_prepass(Q, N, p, &pp)) {
            // threshold pre-pass: block maxima over the first rows, KL-th largest per query -> initial shared bounds
            const int64_t S = (int64_t)pp.nchunks * pp.rows_per_chunk;
            int rc0 = k1_launch_blockmax(pp, ld, uq, Q, uc, S, bmax, st);
            if (rc0) return rc0;
            launch_thr_select(bmax, pp.P2, Q, p.KL, gthr, st, reinterpret_cast<int *>(ws + w.ctl), (int)(ctl_bytes / 4));
            TSIM_HIP_CHECK(hipGetLastError());
            ctl_cleared = true;
        } else {
            TSIM_HIP_CHECK(hipMemsetAsync(gthr, 0x80, (size_t)Q * 4, st));
        }
        if (!ctl_cleared) TSIM_HIP_CHECK(hipMemsetAsync(ws + w.ctl, 0, ctl_bytes, st));
#ifdef TSIM_PP_STAMPS
        if (getenv("TSIM_K1_DIAG_NOSEL"))   // DIAGNOSTIC: thresholds nothing can pass (results are wrong): the time without selection
            TSIM_HIP_CHECK(hipMemsetAsync(gthr, 0x7f, (size_t)Q * 4, st));
#endif
        hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
        g_ev_start = g_ev_stop = nullptr;
        if (ev0) TSIM_HIP_CHECK(hipEventRecord(ev0, st));
        TopkPlan pa, pb;
        if (plan_two_phase(Q, N, ld, k, &pa, &pb)) {
            // main pass in two launches: rows [0, NA) with the pre-pass bounds, then the rest with the KL-th best score of
            // phase A as the bound (thr_update_kernel).  Lists of both phases sit side by side: P2 = pa.P2 + pb.P2.
            const int64_t NA = K1_PHASE_A_ROWS;
            p.P2 = pa.P2 + pb.P2;
            K1Collect ra{}, rb{};
            ra.p2_base = 0; ra.p2_total = p.P2; ra.row_base = 0;
            rb.p2_base = pa.P2; rb.p2_total = p.P2; rb.row_base = (int)NA;
            int rc = p.KL == 16 ? k1_launch_kl16(pa, ld, uq, Q, uc, NA, part_s, part_i, gthr, st, ra)
                                : k1_launch_kl32(pa, ld, uq, Q, uc, NA, part_s, part_i, gthr, st, ra);
            if (rc) return rc;
            const unsigned ug = (unsigned)((Q + 3) / 4);
            if (p.KL == 16) hipLaunchKernelGGL(thr_update_kernel<16>, dim3(ug), dim3(256), 0, st, part_s, part_i, p.P2, 0, pa.P2, Q, gthr);
            else hipLaunchKernelGGL(thr_update_kernel<32>, dim3(ug), dim3(256), 0, st, part_s, part_i, p.P2, 0, pa.P2, Q, gthr);
            TSIM_HIP_CHECK(hipGetLastError());
            rc = p.KL == 16 ? k1_launch_kl16(pb, ld, uq, Q, uc + NA * ld, N - NA, part_s, part_i, gthr, st, rb)
                            : k1_launch_kl32(pb, ld, uq, Q, uc + NA * ld, N - NA, part_s, part_i, gthr, st, rb);
            if (rc) return rc;
        } else {
            int rc = p.KL == 16 ? k1_launch_kl16(p, ld, uq, Q, uc, N, part_s, part_i, gthr, st)
                                : k1_launch_kl32(p, ld, uq, Q, uc, N, part_s, part_i, gthr, st);
            if (rc) return rc;
        }
        if (ev1) TSIM_HIP_CHECK(hipEventRecord(ev1, st));
        if (cosf) {
            if (p.KL == 16) launch_finalize<16, float, true>(p, part_s, part_i, Q, N, eq_f32, ldq_f32, ec_f32, ldc_f32, d, k,
                                                             uq, gthr, out_scores, out_idx, idx_offset, g, st);
            else launch_finalize<32, float, true>(p, part_s, part_i, Q, N, eq_f32, ldq_f32, ec_f32, ldc_f32, d, k, uq, gthr,
                                                  out_scores, out_idx, idx_offset, g, st);
        } else {
            if (p.KL == 16) launch_finalize<16, unit_t, false>(p, part_s, part_i, Q, N, uq, ld, uc, ld, ld, k, uq, gthr,
                                                               out_scores, out_idx, idx_offset, g, st);
            else launch_finalize<32, unit_t, false>(p, part_s, part_i, Q, N, uq, ld, uc, ld, ld, k, uq, gthr, out_scores,
                                                    out_idx, idx_offset, g, st);
        }
        TSIM_HIP_CHECK(hipGetLastError());
    } else {
        // k > 28: no list kernel.  Block maxima over the whole shard give a lower bound of the k-th best MFMA score; every row
        // above (bound - margin) is collected and re-scored; widen_finalize's guard decides whether that was enough.
        TSIM_HIP_CHECK(hipMemsetAsync(ws + w.ctl, 0, ctl_bytes, st));
        TopkPlan fp;
        const bool ok = plan_fullmax(Q, N, ld, k, &fp);
        if (ok) {
            int rc0 = k1_launch_blockmax(fp, ld, uq, Q, uc, N, bmax, st);
            if (rc0) return rc0;
            launch_thr_select(bmax, fp.P2, Q, k, gthr, st);
            TSIM_HIP_CHECK(hipGetLastError());
        }
        if (cosf)
            hipLaunchKernelGGL(flag_all_kernel<true>, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, st, Q, gthr, !ok, eq_f32, ldq_f32,
                               uq, d, g);
        else
            hipLaunchKernelGGL(flag_all_kernel<false>, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, st, Q, gthr, !ok,
                               (const float *)nullptr, (int64_t)0, uq, d, g);
        TSIM_HIP_CHECK(hipGetLastError());
        run_collect = ok;
    }
    if (cosf)
        return search_tail<float, true>(w, ws, Q, N, uq, uc, ld, eq_f32, ldq_f32, ec_f32, ldc_f32, d, k, out_scores, out_idx,
                                        idx_offset, g, run_collect, st);
    return search_tail<unit_t, false>(w, ws, Q, N, uq, uc, ld, uq, ld, uc, ld, ld, k, out_scores, out_idx, idx_offset, g,
                                      run_collect, st);
}

extern "C" int tsim_cosine_topk(const void *eq, int64_t Q, const void *ec, int64_t N, int d, int ld, int k,
                                float *out_scores, int64_t *out_idx, int64_t idx_offset, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return tsim_cosine_topk_ex(eq, nullptr, 0, Q, ec, nullptr, 0, nullptr, N, d, ld, k, out_scores, out_idx, nullptr, idx_offset,
                               workspace, workspace_bytes, stream);
}

extern "C" int tsim_topk_merge_strided(const float *scores, const int64_t *idx, int nlists, int64_t Q, int k_in, int k_out,
                                       int64_t list_stride_scores, int64_t list_stride_idx, float *out_scores,
                                       int64_t *out_idx, void *stream) {
    TSIM_REQUIRE(scores && idx && out_scores && out_idx, "topk_merge: null pointer");
    TSIM_REQUIRE(nlists >= 1 && Q >= 0 && k_in >= 1 && k_out >= 1, "topk_merge: bad shape");
    TSIM_REQUIRE(nlists == 1 || (list_stride_scores >= Q * k_in && list_stride_idx >= Q * k_in),
                 "topk_merge: list strides %lld / %lld < Q * k_in = %lld", (long long)list_stride_scores,
                 (long long)list_stride_idx, (long long)(Q * k_in));
    if (Q == 0) return TSIM_OK;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, as_stream(stream), scores,
                       idx, nlists, Q, k_in, k_out, list_stride_scores, list_stride_idx, out_scores, out_idx);
    TSIM_HIP_CHECK(hipGetLastError());
    return TSIM_OK;
}

extern "C" int tsim_topk_merge(const float *scores, const int64_t *idx, int nlists, int64_t Q, int k_in, int k_out,
                               float *out_scores, int64_t *out_idx, void *stream) {
    return tsim_topk_merge_strided(scores, idx, nlists, Q, k_in, k_out, Q * k_in, Q * k_in, out_scores, out_idx, stream);
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
