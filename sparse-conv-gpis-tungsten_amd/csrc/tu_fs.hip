// tu_fs.hip — function-space comparison path (config C4; gpis_fs.hpp, gpis_launch.hpp).
#include "gpis_fs.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

size_t fs_workspace_bytes_per_block() { return sizeof(FsGlob); }
void fs_march(bool want_sample, unsigned grid, const DevModel *d_model, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, gpis_seg_out *out,
              uint8_t *visible, void *workspace, hipStream_t s)
{
    if (want_sample) k_fs_march<true><<<grid, 64, 0, s>>>(d_model, n, rays, states, out, visible, (FsGlob *)workspace);
    else k_fs_march<false><<<grid, 64, 0, s>>>(d_model, n, rays, states, out, visible, (FsGlob *)workspace);
}
void fs_linalg(unsigned grid, int op, int n, size_t count, const double *in, double *out, double *evals, void *workspace, hipStream_t s)
{
    k_fs_linalg<0><<<grid, 64, 0, s>>>(op, n, count, in, out, evals, (FsGlob *)workspace);
}
int fs_prof_read(unsigned long long *out16, int reset)
{
#ifdef GPIS_FS_PROF
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(gpis::g_fs_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return GPIS_ERR_DEVICE;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(gpis::g_fs_prof), z, sizeof z) != hipSuccess) return GPIS_ERR_DEVICE; }
    return GPIS_OK;
#else
    (void)out16; (void)reset;
    return GPIS_ERR_UNSUPPORTED;
#endif
}

}}   // namespace gpis::launch
