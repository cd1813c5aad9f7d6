// tu_generic_nee.hip — neePDF / neeGrad queries (BSDF NEE hooks, the NEE driver) (all-features path instance; gpis_lane.hpp, gpis_launch.hpp).
// One kernel per translation unit: the all-features instance inlines the evaluator at every call site and each of these
// kernels takes 1-2 minutes to compile.
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void nee(int inst, const DevModel *d_model, size_t n, const gpis_nee_query *q, float *pdf, float *grad3, Counters *cnt, const uint8_t *mask, hipStream_t s)
{
    if (inst == INST_1D) k_nee<spec_1d::Path><<<grid_of(n, kBlock), kBlock, 0, s>>>(d_model, n, q, pdf, grad3, cnt, mask);
    else k_nee<generic::Path><<<grid_of(n, kBlock), kBlock, 0, s>>>(d_model, n, q, pdf, grad3, cnt, mask);
}

}}   // namespace gpis::launch
