// tu_generic_nee.hip — neePDF / neeGrad queries (BSDF NEE hooks, the NEE driver) (all-features path instance; gpis_lane.hpp, gpis_launch.hpp).
// One kernel per translation unit: the all-features instance inlines the evaluator at every call site and each of these
// kernels takes 1-2 minutes to compile.
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void nee(const DevModel *d_model, size_t n, const gpis_nee_query *q, float *pdf, float *grad3, Counters *cnt, const uint8_t *mask, hipStream_t s)
{
    k_nee<0><<<grid_of(n, kBlock), kBlock, 0, s>>>(d_model, n, q, pdf, grad3, cnt, mask);
}

}}   // namespace gpis::launch
