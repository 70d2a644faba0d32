// Does the WIDTH of a lane's non-temporal store matter?  The record kernel's product rows leave as 16 bytes per lane (1 KB per wave instruction), its
// per-column planes as 8 bytes per lane (512 B per instruction: limb_t entries) -- profiles/r06_record_kernel_phases.txt.  The same 1 GB written with
// 16 / 8 / 4 bytes per lane per store instruction (contiguous across the wave in every case), one workgroup per 64 KB piece (the record kernel's
// granularity) and per 4 KB (a dense window), and in "planes": a workgroup writes 32 runs of 2 KB (16 B per lane) or 64 runs of 1 KB (8 B per lane)
// that lie 2 KB apart inside its 64 KB -- every run whole 64-byte lines, as the record planes are.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/store_width_probe tools/store_width_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned u32;
typedef u64 v2 __attribute__((ext_vector_type(2)));
template <int W> __device__ __forceinline__ void stw(char *p, u64 a, u64 b) {
    if (W == 16) { v2 v = {a, b}; __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p)); }
    else if (W == 8) __builtin_nontemporal_store(a, reinterpret_cast<u64 *>(p));
    else __builtin_nontemporal_store((u32)a, reinterpret_cast<u32 *>(p));
}
// PIECE bytes per workgroup, written front to back: step s of the workgroup covers 256 * W contiguous bytes
template <int W, int PIECE>
__global__ __launch_bounds__(256) void fill(char *A, u64 bytes) {
    const u64 t = threadIdx.x;
    char *p = A + (u64)blockIdx.x * PIECE;
    for (u64 k = 0; k < PIECE / (256 * W); ++k) stw<W>(p + k * 256 * W + t * W, blockIdx.x, k);
}
template <int W, int PIECE> static float tbs(char *A, u64 bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(bytes / PIECE);
    hipLaunchKernelGGL((fill<W, PIECE>), dim3(grid), dim3(256), 0, 0, A, bytes);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((fill<W, PIECE>), dim3(grid), dim3(256), 0, 0, A, bytes);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (float)(bytes / (ms / 3) / 1e9);
}
int main(int argc, char **argv) {
    const u64 GB = 1ull << 30; const int n = argc > 1 ? std::atoi(argv[1]) : 6;
    std::printf("1 GB hipMalloc buffers, non-temporal stores, TB/s            64 KB per workgroup              4 KB per workgroup\n");
    std::printf("buffer                                              16 B/lane  8 B/lane  4 B/lane     16 B/lane  8 B/lane  4 B/lane\n");
    for (int i = 0; i < n; ++i) {
        char *c; CK(hipMalloc((void **)&c, GB));
        std::printf("%-50d  %8.2f  %8.2f  %8.2f      %8.2f  %8.2f  %8.2f\n", i, tbs<16, 65536>(c, GB), tbs<8, 65536>(c, GB), tbs<4, 65536>(c, GB),
                    tbs<16, 4096>(c, GB), tbs<8, 4096>(c, GB), tbs<4, 4096>(c, GB));
    }
    return 0;
}
