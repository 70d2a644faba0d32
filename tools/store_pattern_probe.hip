// The store PATTERN against the placement classes: torch's fill_ on two streams shows 6.7 / 7.1 TB/s for every pair of 1.275 GB buffers where the
// probes' persistent 64 KB-per-workgroup fill shows 5.0 / 6.9.  One 4 GB buffer alone and pairs (chunk 0 + j), non-temporal 16-byte stores:
//   P0  persistent grid (2,048 workgroups), 64 KB pieces, 4 KB per workgroup step        (the probes' pattern; the product kernels' is close to it)
//   P1  one workgroup per 64 KB piece (no persistence)
//   P2  one workgroup per 16 KB piece, 4 KB per step
//   P3  one workgroup per 16 KB piece, every lane writes 64 contiguous bytes (4 x 16)   (an elementwise kernel's vectorised pattern)
//   P4  one workgroup per 4 KB
//   P5  persistent grid, but a workgroup's consecutive pieces lie 1/8 of the buffer apart (XCD-contiguous eighths)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/store_pattern_probe tools/store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef unsigned long long u64;
typedef u64 v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(void *p, u64 a, u64 b) { v2 v = {a, b}; __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p)); }
template <int P>
__global__ __launch_bounds__(256) void fill(char *A, char *B, u64 bytes) {   // B != nullptr: even work items go to A, odd to B
    const u64 t = threadIdx.x;
    const u64 piece = P == 0 || P == 1 || P == 5 ? 65536 : (P == 4 ? 4096 : 16384);
    const u64 items = (B ? 2 : 1) * (bytes / piece);
    for (u64 w = blockIdx.x; w < items; w += gridDim.x) {
        u64 c = B ? w >> 1 : w;
        if (P == 5) { const u64 per = (bytes / piece + 7) / 8; c = (c & 7) * per + (c >> 3); if (c >= bytes / piece) continue; }
        char *p = ((B && (w & 1)) ? B : A) + c * piece;
        if (P == 3) { for (int k = 0; k < 4; ++k) st16(p + t * 64 + k * 16, c, k); }
        else for (u64 k = 0; k < piece / 4096; ++k) st16(p + k * 4096 + t * 16, c, k);
    }
}
template <int P> static float tbs(char *A, char *B, u64 bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u64 piece = P == 0 || P == 1 || P == 5 ? 65536 : (P == 4 ? 4096 : 16384);
    const u64 items = (B ? 2 : 1) * (bytes / piece);
    const unsigned grid = (P == 0 || P == 5) ? 2048u : (unsigned)items;
    hipLaunchKernelGGL(fill<P>, dim3(grid), dim3(256), 0, 0, A, B, bytes);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fill<P>, dim3(grid), dim3(256), 0, 0, A, B, bytes);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (float)((B ? 2.0 : 1.0) * bytes / (ms / 3) / 1e9);
}
int main(int argc, char **argv) {
    const u64 GB = 1ull << 30, n = argc > 1 ? std::atoi(argv[1]) : 24;
    std::vector<char *> c(n);
    for (u64 i = 0; i < n; ++i) CK(hipMalloc((void **)&c[i], GB));
    std::printf("            P0     P1     P2     P3     P4     P5   (TB/s)\n");
    std::printf("alone     %5.2f  %5.2f  %5.2f  %5.2f  %5.2f  %5.2f\n", tbs<0>(c[0], nullptr, GB), tbs<1>(c[0], nullptr, GB), tbs<2>(c[0], nullptr, GB), tbs<3>(c[0], nullptr, GB), tbs<4>(c[0], nullptr, GB), tbs<5>(c[0], nullptr, GB));
    for (u64 j = 1; j < n; ++j)
        std::printf("0 + %-3llu  %5.2f  %5.2f  %5.2f  %5.2f  %5.2f  %5.2f\n", j, tbs<0>(c[0], c[j], GB), tbs<1>(c[0], c[j], GB), tbs<2>(c[0], c[j], GB), tbs<3>(c[0], c[j], GB), tbs<4>(c[0], c[j], GB), tbs<5>(c[0], c[j], GB));
    char *big; CK(hipMalloc((void **)&big, 4 * GB));
    std::printf("4 GB alone %5.2f  %5.2f  %5.2f  %5.2f  %5.2f  %5.2f\n", tbs<0>(big, nullptr, 4 * GB), tbs<1>(big, nullptr, 4 * GB), tbs<2>(big, nullptr, 4 * GB), tbs<3>(big, nullptr, 4 * GB), tbs<4>(big, nullptr, 4 * GB), tbs<5>(big, nullptr, 4 * GB));
    return 0;
}
