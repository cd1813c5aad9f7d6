// tu_persist_b.hip — persistent refilling march (gpis_persist.inc), instances spec_3d_multires and generic (gpis_launch.hpp).
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

template <class P, bool WANT_SAMPLE>
static int occupancy_of()
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_persist_march<P, WANT_SAMPLE>, kBlock, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb;
}
int persist_b_blocks_per_cu(int inst, bool want_sample)
{
    if (inst == INST_3D_MULTIRES) return want_sample ? occupancy_of<spec_3d_multires::Persist, true>() : occupancy_of<spec_3d_multires::Persist, false>();
    return want_sample ? occupancy_of<generic::Persist, true>() : occupancy_of<generic::Persist, false>();
}
void persist_b_march(int inst, bool want_sample, unsigned grid, const DevModel *d_model, const PersistArgs &a, hipStream_t s)
{
    if (inst == INST_3D_MULTIRES) {
        if (want_sample) k_persist_march<spec_3d_multires::Persist, true><<<grid, kBlock, 0, s>>>(d_model, a);
        else k_persist_march<spec_3d_multires::Persist, false><<<grid, kBlock, 0, s>>>(d_model, a);
    } else {
        if (want_sample) k_persist_march<generic::Persist, true><<<grid, kBlock, 0, s>>>(d_model, a);
        else k_persist_march<generic::Persist, false><<<grid, kBlock, 0, s>>>(d_model, a);
    }
}

}}   // namespace gpis::launch
