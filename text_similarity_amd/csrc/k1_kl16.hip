// K1 instantiations for candidate lists of 16 entries (k <= 12).
#include "k1_topk.h"

namespace tsim {
int k1_launch_kl16(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                   float *part_s, int *part_i, int *gthr, hipStream_t st) {
    return launch_k1_kl<16>(p, D, eq, Q, ec, N, part_s, part_i, gthr, st);
}

int k1_launch_blockmax(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                       float *bmax, hipStream_t st) {
    return launch_k1_kl<16, true>(p, D, eq, Q, ec, N, bmax, nullptr, nullptr, st);
}
}  // namespace tsim

#ifdef TSIM_PP_STAMPS
extern "C" int tsim_debug_k1_stamps(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tsim::g_k1_stamps), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(tsim::g_k1_stamps), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif
