// re_sort.hip -- the one place that instantiates rocPRIM (device radix sort of (u64 key, u32 value) pairs), kept out of the other
// translation units' compile time.  Used by the device-side re-bucket bookkeeping (re_api.hip: rebucket_on_device).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdint>

namespace re {
// stable LSD radix sort on bits [begin_bit, end_bit) of the keys; tmp == nullptr: only *tmp_bytes is computed
hipError_t sort_pairs_u64_u32(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                              uint32_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}
}  // namespace re
