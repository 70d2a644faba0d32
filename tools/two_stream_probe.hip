// Two store streams into two buffers: at which level do same-class buffers collide?  (profiles/r06_placement_counters.txt)
// N plain hipMalloc buffers of the lookup column size (5.37 GB).  For every partner i of buffer 0 the same 2 x 5.37 GB are written as
//   V1 interleaved   every workgroup writes chunk c of A and chunk c of B, 4 KB of A then 4 KB of B (what lookup_fill_kernel does)
//   V2 wave-split    waves 0-1 of the workgroup write A's chunk, waves 2-3 B's chunk, concurrently
//   V3 wg-split      workgroup 2k writes chunk k of A, workgroup 2k+1 chunk k of B
//   V4 serial        one launch writes A, the next writes B (two single-stream kernels)
//   V5 skewed        as V1 but B's 4 KB blocks are visited half a chunk ahead of A's (rotated inside the chunk)
//   V6 xcd-split     workgroups of even XCDs (blockIdx % 8 even) write A, of odd XCDs write B
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/two_stream_probe tools/two_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned long long u64;
constexpr u64 CHUNK = 65536, BLK = 4096;   // bytes per workgroup and stream; bytes per 256-thread store (16 B per lane)

__device__ __forceinline__ void st16(void *p, u64 a, u64 b) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    v2 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p));
}
template <int V>
__global__ __launch_bounds__(256) void fill2(char *A, char *B, u64 chunks) {
    const u64 t = threadIdx.x;
    if (V == 1 || V == 5) {
        const u64 c = blockIdx.x;
        char *a = A + c * CHUNK, *b = B + c * CHUNK;
        for (u64 k = 0; k < CHUNK / BLK; ++k) {
            const u64 kb = V == 5 ? (k + CHUNK / BLK / 2) % (CHUNK / BLK) : k;
            st16(a + k * BLK + t * 16, c, k);
            st16(b + kb * BLK + t * 16, c, k);
        }
    } else if (V == 2) {
        const u64 c = blockIdx.x;
        char *p = (t < 128 ? A : B) + c * CHUNK;
        const u64 tt = t & 127;
        for (u64 k = 0; k < CHUNK / (BLK / 2); ++k) st16(p + k * (BLK / 2) + tt * 16, c, k);
    } else if (V == 3) {
        const u64 c = blockIdx.x >> 1;
        char *p = ((blockIdx.x & 1) ? B : A) + c * CHUNK;
        for (u64 k = 0; k < CHUNK / BLK; ++k) st16(p + k * BLK + t * 16, c, k);
    } else if (V == 4) {
        const u64 c = blockIdx.x;
        char *p = A + c * CHUNK;
        for (u64 k = 0; k < CHUNK / BLK; ++k) st16(p + k * BLK + t * 16, c, k);
    } else if (V == 6) {   // grid = 2 * chunks rounded to 16: XCD x = blockIdx % 8; pair (x / 2) of XCDs shares the chunks c = 4 * (blockIdx / 16) + x / 2 ... simply:
        const u64 x = blockIdx.x & 7, g = blockIdx.x >> 3;      // g-th workgroup of XCD x
        const u64 c = g * 4 + (x >> 1);                         // chunk: four XCD pairs interleave
        if (c >= chunks) return;
        char *p = ((x & 1) ? B : A) + c * CHUNK;
        for (u64 k = 0; k < CHUNK / BLK; ++k) st16(p + k * BLK + t * 16, c, k);
    }
}
template <int V>
float run(char *A, char *B, u64 bytes, int reps) {
    const u64 chunks = bytes / CHUNK;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto once = [&]() {
        if (V == 3) hipLaunchKernelGGL(fill2<3>, dim3((unsigned)(2 * chunks)), dim3(256), 0, 0, A, B, chunks);
        else if (V == 4) { hipLaunchKernelGGL(fill2<4>, dim3((unsigned)chunks), dim3(256), 0, 0, A, B, chunks); hipLaunchKernelGGL(fill2<4>, dim3((unsigned)chunks), dim3(256), 0, 0, B, A, chunks); }
        else if (V == 6) hipLaunchKernelGGL(fill2<6>, dim3((unsigned)(2 * ((chunks + 3) / 4) * 4)), dim3(256), 0, 0, A, B, chunks);
        else hipLaunchKernelGGL(fill2<V>, dim3((unsigned)chunks), dim3(256), 0, 0, A, B, chunks);
    };
    once();
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) once();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 10;
    const u64 bytes = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 256ull * 5 * 131066 * 32 / CHUNK * CHUNK;
    std::vector<char *> pool(n);
    for (int i = 0; i < n; ++i) { CK(hipMalloc(&pool[i], bytes)); std::printf("buffer %2d at %p\n", i, (void *)pool[i]); }
    std::printf("bytes per buffer %llu; times in ms for 2 buffers (TB/s)\n", bytes);
    std::printf("%-8s %14s %14s %14s %14s %14s %14s\n", "partner", "V1 interleaved", "V2 wave-split", "V3 wg-split", "V4 serial", "V5 skewed", "V6 xcd-split");
    for (int i = 1; i < n; ++i) {
        float v[6] = {run<1>(pool[0], pool[i], bytes, 3), run<2>(pool[0], pool[i], bytes, 3), run<3>(pool[0], pool[i], bytes, 3),
                      run<4>(pool[0], pool[i], bytes, 3), run<5>(pool[0], pool[i], bytes, 3), run<6>(pool[0], pool[i], bytes, 3)};
        std::printf("0 + %-4d", i);
        for (float x : v) std::printf(" %7.3f (%4.2f)", x, 2.0 * bytes / x / 1e9);
        std::printf("\n");
    }
    // and every buffer alone (single stream), to see whether a class shows with one stream
    std::printf("alone:  ");
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(fill2<4>, dim3((unsigned)(bytes / CHUNK)), dim3(256), 0, 0, pool[i], pool[i], bytes / CHUNK);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fill2<4>, dim3((unsigned)(bytes / CHUNK)), dim3(256), 0, 0, pool[i], pool[i], bytes / CHUNK);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); std::printf(" %6.3f", ms / 3); }
    std::printf("\n");
    return 0;
}
