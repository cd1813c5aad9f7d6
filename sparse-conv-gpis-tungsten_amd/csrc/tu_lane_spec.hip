// tu_lane_spec.hip — lane-per-ray march, the three specialised path instances (gpis_launch.hpp).
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void lane_generic_sample_distance(const DevModel *, size_t, const gpis_ray_in *, gpis_seg_out *, gpis_cond_coeff *, const uint8_t *, Counters *, hipStream_t);   // tu_generic_sd.hip / tu_generic_tr.hip
void lane_generic_transmittance(const DevModel *, size_t, const gpis_ray_in *, uint8_t *, const uint8_t *, Counters *, hipStream_t);

void lane_sample_distance(int inst, const DevModel *d_model, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff,
                          const uint8_t *mask, Counters *cnt, hipStream_t s)
{
    const unsigned grid = grid_of(n, kBlock);
    switch (inst) {
    case INST_1D: k_sample_distance<spec_1d::Path><<<grid, kBlock, 0, s>>>(d_model, n, rays, out, coeff, mask, cnt); break;
    case INST_3D: k_sample_distance<spec_3d::Path><<<grid, kBlock, 0, s>>>(d_model, n, rays, out, coeff, mask, cnt); break;
    case INST_3D_MULTIRES: k_sample_distance<spec_3d_multires::Path><<<grid, kBlock, 0, s>>>(d_model, n, rays, out, coeff, mask, cnt); break;
    default: lane_generic_sample_distance(d_model, n, rays, out, coeff, mask, cnt, s); break;
    }
}
void lane_transmittance(int inst, const DevModel *d_model, size_t n, const gpis_ray_in *rays, uint8_t *visible, const uint8_t *mask,
                        Counters *cnt, hipStream_t s)
{
    const unsigned grid = grid_of(n, kBlock);
    switch (inst) {
    case INST_1D: k_transmittance<spec_1d::Path><<<grid, kBlock, 0, s>>>(d_model, n, rays, visible, mask, cnt); break;
    case INST_3D: k_transmittance<spec_3d::Path><<<grid, kBlock, 0, s>>>(d_model, n, rays, visible, mask, cnt); break;
    case INST_3D_MULTIRES: k_transmittance<spec_3d_multires::Path><<<grid, kBlock, 0, s>>>(d_model, n, rays, visible, mask, cnt); break;
    default: lane_generic_transmittance(d_model, n, rays, visible, mask, cnt, s); break;
    }
}

}}   // namespace gpis::launch
