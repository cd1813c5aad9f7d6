// tu_guided_tr.hip — resident guided march, transmittance (gpis_guide.hpp, gpis_launch.hpp).
#include "gpis_guide.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void guided_transmittance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField *d_guide, size_t n, const gpis_ray_in *rays,
                          uint8_t *visible, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s)
{
    if (small_arg) k_guided_transmittance<true><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, d_guide, n, rays, visible, mask, cnt, guide_cnt);
    else k_guided_transmittance<false><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, d_guide, n, rays, visible, mask, cnt, guide_cnt);
}

}}   // namespace gpis::launch
