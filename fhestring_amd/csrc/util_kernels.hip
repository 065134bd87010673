// Small data-movement kernels that are not part of any bootstrap (kept out of the profiled kernel sources:
// fhestring_amd/kernel_sources.py ties the roofline counters to those files' hashes).
#include <hip/hip_runtime.h>

#include "pbs_kernels.h"

namespace fhs {

// n pool blocks -> n consecutive rows (the inverse of scatter_blocks_kernel): one launch + ONE device-to-host copy bring
// a whole FheString back (fhs_download_string) instead of one synchronous 16 KB copy per block.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint64_t *const *__restrict__ src, uint64_t *__restrict__ out) {
    const uint64_t *s = src[blockIdx.x];
    uint64_t *o = out + (size_t)blockIdx.x * BIG_CT;
    for (int e = threadIdx.x; e < BIG_CT; e += 256) o[e] = s[e];
}
hipError_t launch_gather_rows(const uint64_t *const *d_src, uint64_t *d_out, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, s, d_src, d_out);
    return hipGetLastError();
}

}  // namespace fhs
