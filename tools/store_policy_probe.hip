// Is the placement-class effect a property of NON-TEMPORAL stores?  1 GB chunks, the concurrent pair fill (alternate workgroups), every partner of
// chunk 0, with four store forms: non-temporal (what the product kernels use), plain, plain + sc1 (write-through hint via __builtin_amdgcn... n/a:
// use glc through inline asm), and a single stream for reference.  build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/store_policy_probe tools/store_policy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef unsigned long long u64;
constexpr u64 PIECE = 65536, BLK = 4096;
typedef u64 v2 __attribute__((ext_vector_type(2)));
template <int POLICY> __device__ __forceinline__ void st16(void *p, u64 a, u64 b) {
    v2 v = {a, b};
    if (POLICY == 0) __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p));
    else if (POLICY == 1) *reinterpret_cast<v2 *>(p) = v;
    else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
template <int POLICY>
__global__ __launch_bounds__(256) void fill(char *A, char *B, u64 pieces) {
    const u64 t = threadIdx.x, items = (B ? 2 : 1) * pieces;
    for (u64 w = blockIdx.x; w < items; w += gridDim.x) {
        const u64 c = B ? w >> 1 : w;
        char *p = ((B && (w & 1)) ? B : A) + c * PIECE;
        for (u64 k = 0; k < PIECE / BLK; ++k) st16<POLICY>(p + k * BLK + t * 16, c, k);
    }
}
template <int POLICY> static float tbs(char *A, char *B, u64 bytes_each) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill<POLICY>, dim3(2048), dim3(256), 0, 0, A, B, bytes_each / PIECE);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fill<POLICY>, dim3(2048), dim3(256), 0, 0, A, B, bytes_each / PIECE);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (float)((B ? 2.0 : 1.0) * bytes_each / (ms / 3) / 1e9);
}
int main(int argc, char **argv) {
    const u64 GB = 1ull << 30, n = argc > 1 ? std::atoi(argv[1]) : 40;
    std::vector<char *> c(n);
    for (u64 i = 0; i < n; ++i) CK(hipMalloc((void **)&c[i], GB));
    std::printf("partner   nt    plain   sc0sc1   sc1    (pair fill with chunk 0, TB/s; hipMalloc 1 GB each)\n");
    for (u64 j = 1; j < n; ++j)
        std::printf("0 + %-3llu  %5.2f  %5.2f  %5.2f  %5.2f\n", j, tbs<0>(c[0], c[j], GB), tbs<1>(c[0], c[j], GB), tbs<2>(c[0], c[j], GB), tbs<3>(c[0], c[j], GB));
    std::printf("alone     %5.2f  %5.2f  %5.2f  %5.2f\n", tbs<0>(c[0], nullptr, GB), tbs<1>(c[0], nullptr, GB), tbs<2>(c[0], nullptr, GB), tbs<3>(c[0], nullptr, GB));
    return 0;
}
