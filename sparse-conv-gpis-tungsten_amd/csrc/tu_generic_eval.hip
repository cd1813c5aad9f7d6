// tu_generic_eval.hip — single-query evaluateValue / evaluateGradient (unit-test surface) (all-features path instance; gpis_lane.hpp, gpis_launch.hpp).
// One kernel per translation unit: the all-features instance inlines the evaluator at every call site and each of these
// kernels takes 1-2 minutes to compile.
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void eval_value(const DevModel *d_model, size_t n, const gpis_query *q, float *value, int32_t *gp_id, Counters *cnt, hipStream_t s)
{
    k_eval_value<0><<<grid_of(n, kBlock), kBlock, 0, s>>>(d_model, n, q, value, gp_id, cnt);
}
void eval_gradient(const DevModel *d_model, size_t n, const gpis_query *q, float *grad3, Counters *cnt, hipStream_t s)
{
    k_eval_gradient<0><<<grid_of(n, kBlock), kBlock, 0, s>>>(d_model, n, q, grad3, cnt);
}

}}   // namespace gpis::launch
