// K1 instantiations for candidate lists of 16 entries (k <= 12).
#include "k1_topk.h"

namespace tsim {
int k1_launch_kl16(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                   float *part_s, int *part_i, int *gthr, hipStream_t st, K1Collect range) {
    return launch_k1_kl<16>(p, D, eq, Q, ec, N, part_s, part_i, gthr, st, range);
}

int k1_launch_blockmax(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N,
                       float *bmax, hipStream_t st) {
    return launch_k1_kl<16, true>(p, D, eq, Q, ec, N, bmax, nullptr, nullptr, st);
}
}  // namespace tsim

