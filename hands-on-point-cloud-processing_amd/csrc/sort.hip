// sort.hip — the device-wide sorts of the index builds, instantiated ONCE: rocPRIM's LSD radix sorts (AMD's own primitives
// library, wave64-aware; no CUB-compatibility layer in between).  Stable, deterministic.
//   * (cell << 32 | x bits, point index) pairs of the grid build (grid.hip) and (voxel key, point index) of the voxel filter
//   * u32 keys of the seed selection (ground.hip)
//   * segmented u32 keys: the neighbour rows of a radius search, each into ascending index order (radius_grid.hip)
// Call with temp == nullptr to get the scratch size (the rocPRIM two-call convention).
#include "sort.hpp"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace pcr {

hipError_t sort_pairs_u64_u32(void* temp, size_t& temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const uint32_t* vals_in,
                              uint32_t* vals_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream)
{
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}

hipError_t sort_keys_u32(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, size_t n, unsigned begin_bit, unsigned end_bit,
                         hipStream_t stream)
{
    return rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, n, begin_bit, end_bit, stream);
}

hipError_t segmented_sort_keys_u32(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, size_t n, size_t segments,
                                   const uint32_t* begin_offsets, const uint32_t* end_offsets, unsigned begin_bit, unsigned end_bit, hipStream_t stream)
{
    return rocprim::segmented_radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (unsigned int)n, (unsigned int)segments, begin_offsets, end_offsets, begin_bit,
                                              end_bit, stream);
}

}  // namespace pcr
