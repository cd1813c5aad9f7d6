// gpis_sort.hip — key/value radix sort of a ray batch (rocPRIM through hipCUB), used by the wavefront
// drivers to regroup the segments of a bounce so that the 64 lanes of a wave are neighbours in lattice
// space.  Kept in its own translation unit: the sort templates are heavy and independent of the
// path code.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stddef.h>

namespace gpis {

// Two-call convention of the library: temp == nullptr only reports the scratch size.
hipError_t sort_pairs_u32(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *vals_in, uint32_t *vals_out, size_t n, hipStream_t stream)
{
    if (n > (size_t)0x7FFFFFFF)
        return hipErrorInvalidValue;      // the library's count is an int
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 32, stream);
}

}   // namespace gpis
