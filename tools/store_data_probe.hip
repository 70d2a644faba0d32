// Does the store rate depend on WHAT is stored?  One 5.37 GB hipMalloc buffer, one non-temporal store stream (64 KB per workgroup, 16 bytes per lane),
// the same addresses every time; the data: zeros, small integers (canonical cells: a limb in the low 8 bytes, 24 zero bytes), pseudo-random 16-byte
// values (Montgomery cells: x R mod p is uniform in [0, p)).  build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/store_data_probe tools/store_data_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned long long u64;
constexpr u64 PIECE = 65536, BLK = 4096;
__device__ __forceinline__ void st16(void *p, u64 a, u64 b) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    v2 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p));
}
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
template <int MODE>
__global__ __launch_bounds__(256) void fill(char *A, u64 pieces) {
    const u64 t = threadIdx.x;
    for (u64 c = blockIdx.x; c < pieces; c += gridDim.x) {
        char *p = A + c * PIECE;
        for (u64 k = 0; k < PIECE / BLK; ++k) {
            const u64 id = (c * (PIECE / BLK) + k) * 256 + t;
            u64 a, b;
            if (MODE == 0) { a = 0; b = 0; }
            else if (MODE == 1) { a = (id & 1) ? 0 : mix(id); b = 0; }          // canonical cells of 64-bit values: 8 value bytes, 24 zero bytes per 32-byte cell
            else { a = mix(id); b = mix(id ^ 0x9e3779b97f4a7c15ull); }         // dense cells
            st16(p + k * BLK + t * 16, a, b);
        }
    }
}
template <int MODE> float run(char *A, u64 bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill<MODE>, dim3(2048), dim3(256), 0, 0, A, bytes / PIECE);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(fill<MODE>, dim3(2048), dim3(256), 0, 0, A, bytes / PIECE);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (float)(bytes / (ms / 4) / 1e9);
}
int main() {
    const u64 bytes = 256ull * 5 * 131066 * 32 / PIECE * PIECE;
    for (int b = 0; b < 4; ++b) {
        char *A; CK(hipMalloc(&A, bytes));
        std::printf("buffer %d: zeros %.2f TB/s   canonical-like (8 of 32 bytes non-zero) %.2f   dense random %.2f   zeros again %.2f\n", b, run<0>(A, bytes), run<1>(A, bytes), run<2>(A, bytes), run<0>(A, bytes));
    }
    return 0;
}
