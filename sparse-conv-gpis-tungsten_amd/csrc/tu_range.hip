// tu_range.hip — guided march with in-wave refill from a range of the batch (gpis_guide_range.hpp; GPIS_OPT_RANGE_LEN, off by default).
#include "gpis_guide_range.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

void range_sample_distance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays,
                           gpis_seg_out *out, gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt,
                           uint32_t range_len, hipStream_t s)
{
    RangeArgs a{};
    a.n = n; a.rays = rays; a.out = out; a.coeff = coeff; a.visible = nullptr; a.mask = mask; a.cnt = cnt; a.guide_cnt = guide_cnt;
    a.range_len = range_len;
    const unsigned grid = (unsigned)((n + range_len - 1) / range_len);
    if (small_arg) k_guided_range_sd<true><<<grid, kFastBlock, 0, s>>>(d_model, T, F, a);
    else k_guided_range_sd<false><<<grid, kFastBlock, 0, s>>>(d_model, T, F, a);
    if (hipPeekAtLastError() != hipSuccess) return;
    if (small_arg) k_guided_range_grad<true><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n, rays, out, coeff, mask, cnt);
    else k_guided_range_grad<false><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n, rays, out, coeff, mask, cnt);
}
void range_grad(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays, gpis_seg_out *out,
                gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, hipStream_t s)
{
    if (small_arg) k_guided_range_grad<true><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n, rays, out, coeff, mask, cnt);
    else k_guided_range_grad<false><<<grid_of(n, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n, rays, out, coeff, mask, cnt);
}
void range_transmittance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays,
                         uint8_t *visible, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, uint32_t range_len, hipStream_t s)
{
    RangeArgs a{};
    a.n = n; a.rays = rays; a.out = nullptr; a.coeff = nullptr; a.visible = visible; a.mask = mask; a.cnt = cnt; a.guide_cnt = guide_cnt;
    a.range_len = range_len;
    const unsigned grid = (unsigned)((n + range_len - 1) / range_len);
    if (small_arg) k_guided_range_tr<true><<<grid, kFastBlock, 0, s>>>(d_model, T, F, a);
    else k_guided_range_tr<false><<<grid, kFastBlock, 0, s>>>(d_model, T, F, a);
}

}}   // namespace gpis::launch
