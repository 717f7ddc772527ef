// sx_sort.hip — order the device run records by start offset on the device (rocPRIM radix
// sort), so that the host only has to walk them once.  Records are 16 bytes: the key is
// `start` (u64), the payload the other 8 bytes.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "sx_device.hpp"

namespace sx {

__global__ __launch_bounds__(256) void split_records_kernel(const DevRun* recs, uint32_t n, uint64_t* keys, uint64_t* vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const DevRun r = recs[i];
    const bool invalid = r.len == kRecInvalidLen && r.chars_flags == kRecInvalidFlags;
    keys[i] = invalid ? ~0ull : r.start;  // unused slots sort to the end
    vals[i] = ((uint64_t)r.chars_flags << 32) | r.len;
}
__global__ __launch_bounds__(256) void join_records_kernel(const uint64_t* keys, const uint64_t* vals, uint32_t n, DevRun* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    DevRun r;
    r.start = keys[i]; r.len = (uint32_t)vals[i]; r.chars_flags = (uint32_t)(vals[i] >> 32);
    out[i] = r;
}

// scratch must hold sort_scratch_bytes(n) bytes; sorted records are written back to `recs`
size_t sort_scratch_bytes(uint32_t n) {
    size_t tmp = 0;
    uint64_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, nul, nul, nul, nul, (size_t)n, 0, 64, (hipStream_t)0);
    return (size_t)n * 8 * 4 + tmp + 1024;
}

hipError_t sort_records(DevRun* recs, uint32_t n, void* scratch, size_t scratch_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t* k0 = (uint64_t*)scratch;
    uint64_t* v0 = k0 + n;
    uint64_t* k1 = v0 + n;
    uint64_t* v1 = k1 + n;
    void* tmp = (void*)(v1 + n);
    size_t tmp_bytes = scratch_bytes - (size_t)n * 32;
    const unsigned blocks = (n + 255) / 256;
    hipLaunchKernelGGL(split_records_kernel, dim3(blocks), dim3(256), 0, stream, recs, n, k0, v0);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, 64, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(join_records_kernel, dim3(blocks), dim3(256), 0, stream, k1, v1, n, recs);
    return hipGetLastError();
}

}  // namespace sx
