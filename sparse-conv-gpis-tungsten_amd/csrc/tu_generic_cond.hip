// tu_generic_cond.hip — single-query conditioning (unit-test surface) (all-features path instance; gpis_lane.hpp, gpis_launch.hpp).
// One kernel per translation unit: the all-features instance inlines the evaluator at every call site and each of these
// kernels takes 1-2 minutes to compile.
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void conditioning(const DevModel *d_model, size_t n, const gpis_query *q, const float *tv, const float *tg, gpis_cond_coeff *co, Counters *cnt, hipStream_t s)
{
    k_conditioning<0><<<grid_of(n, kBlock), kBlock, 0, s>>>(d_model, n, q, tv, tg, co, cnt);
}

}}   // namespace gpis::launch
