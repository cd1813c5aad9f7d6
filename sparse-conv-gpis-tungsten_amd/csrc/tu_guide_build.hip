// tu_guide_build.hip — tabulation of the certified guide field and its two self-checks (gpis_guide.hpp, gpis_launch.hpp).
#include "gpis_guide.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

int guide_build(const DevModel &M, const DevModel *d_model, const FastTable &T, int half, int ppc, GuideField *F, bool sparse)
{
    return gpis::guide_build(M, d_model, T, half, ppc, F, sparse);
}
void guide_selfcheck(const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const float *points3, unsigned long long *stats,
                     float *max_ratio, float *sum_bound, hipStream_t s)
{
    k_guide_selfcheck<0><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n, points3, stats, max_ratio, sum_bound);
}
void guide_raycheck(const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays, uint32_t steps,
                    unsigned long long *stats, hipStream_t s)
{
    k_guide_raycheck<0><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n, rays, steps, stats);
}

}}   // namespace gpis::launch
