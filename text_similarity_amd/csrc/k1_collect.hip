// K1 instantiations of the widening pass behind the exactness guard (COLLECT mode: fixed thresholds, global append buffers).
#include "k1_topk.h"

namespace tsim {
int k1_launch_collect(const TopkPlan &p, int D, const unit_t *eq, int64_t Q, const unit_t *ec, int64_t N, int *gthr_slots,
                      K1Collect coll, hipStream_t st) {
    return launch_k1_kl<16, false, true>(p, D, eq, Q, ec, N, nullptr, nullptr, gthr_slots, st, coll);
}
}  // namespace tsim
