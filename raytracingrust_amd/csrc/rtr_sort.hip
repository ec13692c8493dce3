// rtr_sort.hip -- the device-wide sort of the library, on rocPRIM's LSD radix sort (this translation unit alone pays for
// the template instantiation): 64-bit Morton keys ascending, for the device-side BVH builder (rtr_lbvh.hip).  rocPRIM's
// two-call convention: temp == nullptr returns the temporary storage size in temp_bytes.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

namespace rtr {

hipError_t sort_keys_u64(void *temp, size_t &temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, uint32_t n,
                         uint32_t key_bits, hipStream_t stream) {
    return rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t)n, 0u, key_bits, stream);
}

} // namespace rtr
