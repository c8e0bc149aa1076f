// sort.hpp — device-wide radix sorts (sort.hip: rocPRIM, instantiated once).  temp == nullptr: only temp_bytes is written.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace pcr {

hipError_t sort_pairs_u64_u32(void* temp, size_t& temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const uint32_t* vals_in,
                              uint32_t* vals_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
hipError_t sort_keys_u32(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, size_t n, unsigned begin_bit, unsigned end_bit,
                         hipStream_t stream);
hipError_t segmented_sort_keys_u32(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, size_t n, size_t segments,
                                   const uint32_t* begin_offsets, const uint32_t* end_offsets, unsigned begin_bit, unsigned end_bit, hipStream_t stream);

}  // namespace pcr
