// tu_generic_sd.hip — lane-per-ray sampleDistance (all-features path instance; gpis_lane.hpp, gpis_launch.hpp).
// One kernel per translation unit: the all-features instance inlines the evaluator at every call site and each of these
// kernels takes 1-2 minutes to compile.
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void lane_generic_sample_distance(const DevModel *d_model, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff,
                                  const uint8_t *mask, Counters *cnt, hipStream_t s)
{
    k_sample_distance<generic::Path><<<grid_of(n, kBlock), kBlock, 0, s>>>(d_model, n, rays, out, coeff, mask, cnt);
}

}}   // namespace gpis::launch
