// gpis_fast.hpp — single-realization fast path (cell impulse table + wave-coherent cell sweep).
// Placeholder until the kernels land: the table is never built, so the generic path runs.
#pragma once
#include "gpis_device.hpp"

namespace gpis {

struct FastTable {
    float4 *cells;   // nullptr → fast path unused
};

inline bool fast_supported(const DevModel &) { return false; }
inline int fast_table_build(const DevModel &, const DevModel *, FastTable *t) { t->cells = nullptr; return GPIS_OK; }
inline void fast_table_free(FastTable *t) { t->cells = nullptr; }
inline int fast_sample_distance(const DevModel *, const FastTable *, size_t, const gpis_ray_in *, gpis_seg_out *, gpis_cond_coeff *,
                                const uint8_t *, Counters *, hipStream_t) { return GPIS_ERR_UNSUPPORTED; }
inline int fast_transmittance(const DevModel *, const FastTable *, size_t, const gpis_ray_in *, uint8_t *, const uint8_t *, Counters *,
                              hipStream_t) { return GPIS_ERR_UNSUPPORTED; }

}   // namespace gpis
