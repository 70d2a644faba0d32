// Calibration of rocprofv3's WRITE_SIZE / FETCH_SIZE on gfx950 in THIS code's access pattern (MI355X_MICROARCH.md: "calibrate on a known
// byte count in your own access pattern"): a 4 GiB non-temporal 16-byte-per-lane fill (the record / cells kernels' stores), the same
// fill with plain stores, and a 4 GiB 16-byte-per-lane read.  Run under  rocprofv3 --kernel-trace --pmc WRITE_SIZE  and
// --pmc FETCH_SIZE  (separate passes): tools/pmc_calibration.sh divides the counters by the known byte counts.
//   hipcc -O3 --offload-arch=gfx950 tools/pmc_calibration.hip -o tools/_bin/pmc_calibration
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef u64 v2u64 __attribute__((ext_vector_type(2)));
__global__ void fill_nt(v2u64 *p, u64 n) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) { v2u64 v; v.x = i; v.y = ~i; __builtin_nontemporal_store(v, p + i); }
}
__global__ void fill_plain(v2u64 *p, u64 n) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) { v2u64 v; v.x = i; v.y = ~i; p[i] = v; }
}
__global__ void read_all(const v2u64 *p, u64 n, u64 *sink) {
    u64 acc = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) { const v2u64 v = p[i]; acc += v.x ^ v.y; }
    if (acc == 0x123456789abcdefull) *sink = acc;
}
int main() {
    const u64 bytes = 4ull << 30, n = bytes / 16;
    v2u64 *p; u64 *sink;
    if (hipMalloc(reinterpret_cast<void **>(&p), bytes) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&sink), 8) != hipSuccess) return 1;
    hipLaunchKernelGGL(fill_nt, dim3(4096), dim3(256), 0, 0, p, n);
    hipLaunchKernelGGL(fill_plain, dim3(4096), dim3(256), 0, 0, p, n);
    hipLaunchKernelGGL(read_all, dim3(4096), dim3(256), 0, 0, p, n, sink);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    std::printf("bytes per kernel: %llu\n", bytes);
    return 0;
}
