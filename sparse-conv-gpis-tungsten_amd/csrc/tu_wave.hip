// tu_wave.hip — wavefront form of the guided march (gpis_wave.hpp, gpis_launch.hpp).
#include "gpis_wave.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

size_t wave_state_bytes() { return sizeof(WaveState); }
void wave_step(bool want_sample, const DevModel *d_model, const GuideField &F, size_t n_active, const uint32_t *active, int init, const gpis_ray_in *rays,
               const uint8_t *mask, const WaveBufs &b, unsigned long long *guide_cnt, hipStream_t s)
{
    if (want_sample) k_wave_step<true><<<grid_of(n_active, 256), 256, 0, s>>>(d_model, F, n_active, active, init, rays, mask, (WaveState *)b.state, b.k0, b.v0, b.d_req, guide_cnt);
    else k_wave_step<false><<<grid_of(n_active, 256), 256, 0, s>>>(d_model, F, n_active, active, init, rays, mask, (WaveState *)b.state, b.k0, b.v0, b.d_req, guide_cnt);
}
void wave_eval(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n_req, const uint32_t *sorted, const gpis_ray_in *rays,
               const WaveBufs &b, Counters *cnt, hipStream_t s)
{
    if (small_arg) k_wave_eval<true><<<grid_of(n_req, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n_req, sorted, rays, (WaveState *)b.state, cnt);
    else k_wave_eval<false><<<grid_of(n_req, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n_req, sorted, rays, (WaveState *)b.state, cnt);
}
void wave_tail(bool want_sample, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n_active, const uint32_t *active,
               const gpis_ray_in *rays, const WaveBufs &b, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s)
{
    if (want_sample) k_wave_tail<true><<<(unsigned)n_active, kFastBlock, 0, s>>>(d_model, T, F, n_active, active, rays, (WaveState *)b.state, cnt, guide_cnt);
    else k_wave_tail<false><<<(unsigned)n_active, kFastBlock, 0, s>>>(d_model, T, F, n_active, active, rays, (WaveState *)b.state, cnt, guide_cnt);
}
void wave_grad_keys(const DevModel *d_model, const GuideField &F, size_t n, const gpis_ray_in *rays, const WaveBufs &b, hipStream_t s)
{
    k_wave_grad_keys<<<grid_of(n, 256), 256, 0, s>>>(d_model, F, n, rays, (const WaveState *)b.state, b.k0, b.v0, b.d_req);
}
void wave_grad(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n_req, const uint32_t *sorted, const gpis_ray_in *rays,
               const WaveBufs &b, Counters *cnt, hipStream_t s)
{
    if (small_arg) k_wave_grad<true><<<grid_of(n_req, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n_req, sorted, rays, (WaveState *)b.state, cnt);
    else k_wave_grad<false><<<grid_of(n_req, kFastBlock), kFastBlock, 0, s>>>(d_model, T, F, n_req, sorted, rays, (WaveState *)b.state, cnt);
}
void wave_finish_sd(const DevModel *d_model, size_t n, const gpis_ray_in *rays, const uint8_t *mask, const WaveBufs &b, gpis_seg_out *out,
                    gpis_cond_coeff *coeff, Counters *cnt, hipStream_t s)
{
    k_wave_finish_sd<<<grid_of(n, 256), 256, 0, s>>>(d_model, n, rays, mask, (const WaveState *)b.state, out, coeff, cnt);
}
void wave_finish_tr(size_t n, const uint8_t *mask, const WaveBufs &b, uint8_t *visible, Counters *cnt, hipStream_t s)
{
    k_wave_finish_tr<<<grid_of(n, 256), 256, 0, s>>>(n, mask, (const WaveState *)b.state, visible, cnt);
}

}}   // namespace gpis::launch
