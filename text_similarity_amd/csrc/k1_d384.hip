// K1 main pass at the headline shape: D = 384, candidate lists of 16 (k <= 12).  Schedules (env, read once):
//   TSIM_K1_PP   (0) two-group ping-pong schedule
//   TSIM_K1_PAIR (1) two tiles per barrier; 0 = one tile per barrier
//   TSIM_K1_DEEP (0) query blocks up to which the five-slot ring (one tile per barrier, 96 KiB in flight) is used (measured
//                    slower at Q = 256: 0.201 vs 0.188 ms)
//   TSIM_K1_QW2  (0) four waves x 64 queries (one wave per SIMD, 16x16x32 form) instead of eight x 32
//   TSIM_K1_M16  (1) score tiles from v_mfma_f32_16x16x32_f16 (k1_topk.h, M16); 0 = v_mfma_f32_32x32x16_f16
#include "k1_topk.h"

namespace tsim {
int k1_launch_d384_kl16(const TopkPlan &p, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N, float *part_s,
                        int *part_i, int *gthr, hipStream_t st, K1Collect coll) {
    static int pair = -1, pp = -1, m16 = -1, deep = -1, qw2 = -1;
    if (deep < 0) { const char *e = getenv("TSIM_K1_DEEP"); deep = e ? atoi(e) : 0; }
    if (qw2 < 0) { const char *e = getenv("TSIM_K1_QW2"); qw2 = e ? atoi(e) : 0; }
    if (qw2) return launch_k1<384, 4, 2, 16, false, true, false, false, true>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
    if (m16 < 0) { const char *e = getenv("TSIM_K1_M16"); m16 = e ? atoi(e) : 1; }
    if (pair < 0) { const char *e = getenv("TSIM_K1_PAIR"); pair = e ? atoi(e) : 1; }
    if (pp < 0) { const char *e = getenv("TSIM_K1_PP"); pp = e ? atoi(e) : 0; }
    if (pp) return launch_k1<384, 8, 1, 16, false, false, false, true>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
    if (!pp && m16 && p.nqb <= deep) return launch_k1<384, 8, 1, 16, false, false, false, false, true, 5>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
    if (pair && m16) return launch_k1<384, 8, 1, 16, false, true, false, false, true>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
    if (pair) return launch_k1<384, 8, 1, 16, false, true>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
    return launch_k1<384, 8, 1, 16>(p, eq, Q, ec, N, part_s, part_i, gthr, st, coll);
}
}  // namespace tsim

#ifdef TSIM_PP_STAMPS
extern "C" int tsim_debug_k1_stamps(unsigned long long *out, int n, int reset) {
    unsigned long long tmp[K1_NSTAMPS];
    if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(tsim::g_k1_stamps), sizeof(tmp)) != hipSuccess) return 1;
    for (int i = 0; i < n && i < K1_NSTAMPS; ++i) out[i] = tmp[i];
    if (reset) { unsigned long long z[K1_NSTAMPS] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(tsim::g_k1_stamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
