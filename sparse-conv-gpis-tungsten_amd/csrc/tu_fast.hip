// tu_fast.hip — wave-cooperative march for single-realization media and its cell table (gpis_fast.hpp, gpis_launch.hpp).
#include "gpis_fast.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

int fast_table_build(const DevModel &M, FastTable *t) { return gpis::fast_table_build(M, nullptr, t); }
void fast_sample_distance(const DevModel *d_model, const FastTable &T, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff,
                          const uint8_t *mask, Counters *cnt, hipStream_t s)
{
    (void)gpis::fast_sample_distance(d_model, &T, n, rays, out, coeff, mask, cnt, s);
}
void fast_transmittance(const DevModel *d_model, const FastTable &T, size_t n, const gpis_ray_in *rays, uint8_t *visible, const uint8_t *mask,
                        Counters *cnt, hipStream_t s)
{
    (void)gpis::fast_transmittance(d_model, &T, n, rays, visible, mask, cnt, s);
}

}}   // namespace gpis::launch
