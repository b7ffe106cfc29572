// lifecycle_sort.hip -- orders one step's free-slot-queue operations by their 64-bit
// key with rocPRIM's device radix sort (only the key bits in use are sorted).  This is
// bookkeeping beside the hot path, so a library sort is used as-is; everything on
// the hot path is hand-written in kernels.hip.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.h"

namespace psamd {

hipError_t sort_ops_tmp_bytes(size_t n, int key_bits, size_t *bytes)
{
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int *)nullptr,
                                     (int *)nullptr, n, 0u, (unsigned)key_bits, (hipStream_t)0);
}

hipError_t sort_ops(hipStream_t st, const DeviceState &d, int n, int key_bits)
{
    size_t need = 0;
    hipError_t e = sort_ops_tmp_bytes((size_t)n, key_bits, &need);
    if (e != hipSuccess) return e;
    if (need > d.sort_tmp_bytes) return hipErrorOutOfMemory;
    size_t bytes = d.sort_tmp_bytes;
    return rocprim::radix_sort_pairs(d.sort_tmp, bytes, d.op_keys, d.op_keys_sorted, d.op_args, d.op_args_sorted,
                                     (size_t)n, 0u, (unsigned)key_bits, st);
}

}  // namespace psamd
