// tu_guided_sd.hip — resident guided march, sampleDistance (the benchmark's dominant kernel; gpis_guide.hpp, gpis_launch.hpp).
#include "gpis_guide.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void guided_sample_distance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField *d_guide, size_t n, const gpis_ray_in *rays,
                            gpis_seg_out *out, gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s)
{
    if (small_arg) k_guided_sample_distance<true><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, d_guide, n, rays, out, coeff, mask, cnt, guide_cnt);
    else k_guided_sample_distance<false><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, d_guide, n, rays, out, coeff, mask, cnt, guide_cnt);
}
void guided_sample_distance_nograd(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField *d_guide, size_t n, const gpis_ray_in *rays,
                                   gpis_seg_out *out, gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s)
{
    if (small_arg) k_guided_sample_distance_nograd<true><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, d_guide, n, rays, out, coeff, mask, cnt, guide_cnt);
    else k_guided_sample_distance_nograd<false><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, d_guide, n, rays, out, coeff, mask, cnt, guide_cnt);
}
int fast_stats_read(unsigned long long *out32)
{
#ifdef GPIS_FAST_STATS
    // diagnostic build only: read and clear the cooperative loop's work counters (this TU's copy: the guided sampleDistance kernel)
    unsigned long long h[32];
    if (hipDeviceSynchronize() != hipSuccess) return GPIS_ERR_DEVICE;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gpis::g_fast_stats), sizeof h) != hipSuccess) return GPIS_ERR_DEVICE;
    for (int i = 0; i < 32; ++i) out32[i] = h[i];
    for (int i = 0; i < 32; ++i) h[i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(gpis::g_fast_stats), h, sizeof h) != hipSuccess) return GPIS_ERR_DEVICE;
    return GPIS_OK;
#else
    (void)out32;
    return GPIS_ERR_UNSUPPORTED;
#endif
}

}}   // namespace gpis::launch
