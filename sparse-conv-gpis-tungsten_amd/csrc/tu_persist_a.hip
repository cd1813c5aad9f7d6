// tu_persist_a.hip — persistent refilling march (gpis_persist.inc), instances spec_1d and spec_3d (gpis_launch.hpp).
#include "gpis_lane.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis { namespace launch {

int persist_b_blocks_per_cu(int inst, bool want_sample);                                                                        // tu_persist_b.hip
void persist_b_march(int inst, bool want_sample, unsigned grid, const DevModel *d_model, const PersistArgs &a, hipStream_t s);

template <class P, bool WANT_SAMPLE>
static int occupancy_of()
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_persist_march<P, WANT_SAMPLE>, kBlock, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb;
}
int persist_blocks_per_cu(int inst, bool want_sample)
{
    if (inst == INST_1D) return want_sample ? occupancy_of<spec_1d::Persist, true>() : occupancy_of<spec_1d::Persist, false>();
    if (inst == INST_3D) return want_sample ? occupancy_of<spec_3d::Persist, true>() : occupancy_of<spec_3d::Persist, false>();
    return persist_b_blocks_per_cu(inst, want_sample);
}
void persist_march(int inst, bool want_sample, unsigned grid, const DevModel *d_model, const PersistArgs &a, hipStream_t s)
{
    if (inst == INST_1D) {
        if (want_sample) k_persist_march<spec_1d::Persist, true><<<grid, kBlock, 0, s>>>(d_model, a);
        else k_persist_march<spec_1d::Persist, false><<<grid, kBlock, 0, s>>>(d_model, a);
    } else if (inst == INST_3D) {
        if (want_sample) k_persist_march<spec_3d::Persist, true><<<grid, kBlock, 0, s>>>(d_model, a);
        else k_persist_march<spec_3d::Persist, false><<<grid, kBlock, 0, s>>>(d_model, a);
    } else {
        persist_b_march(inst, want_sample, grid, d_model, a, s);
    }
}

}}   // namespace gpis::launch
