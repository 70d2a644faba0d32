// Probe 3: which store-stream shapes reach the fill rate?  All variants write the same 1.25 GB with 16-byte stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
// V0: grid-stride, one 16-B store per thread per iteration (classic fill)
__global__ __launch_bounds__(256) void v0(uint8_t *b, u64 n16) {
    const ulonglong2 v = make_ulonglong2(1, 2);
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) reinterpret_cast<ulonglong2 *>(b)[i] = v;
}
// V1: each block owns a contiguous chunk and streams through it 4 KB per iteration
__global__ __launch_bounds__(256) void v1(uint8_t *b, u64 n16, u64 chunk16) {
    const ulonglong2 v = make_ulonglong2(1, 2);
    const u64 beg = (u64)blockIdx.x * chunk16, end = beg + chunk16 < n16 ? beg + chunk16 : n16;
    for (u64 i = beg + threadIdx.x; i < end; i += 256) reinterpret_cast<ulonglong2 *>(b)[i] = v;
}
// V2: each WAVE owns a contiguous chunk (1 KB per iteration per wave)
__global__ __launch_bounds__(256) void v2(uint8_t *b, u64 n16, u64 chunk16) {
    const ulonglong2 v = make_ulonglong2(1, 2);
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u64 beg = wave * chunk16, end = beg + chunk16 < n16 ? beg + chunk16 : n16;
    for (u64 i = beg + (threadIdx.x & 63); i < end; i += 64) reinterpret_cast<ulonglong2 *>(b)[i] = v;
}
// V3: like V2 but unrolled x4 (4 stores in flight back to back)
__global__ __launch_bounds__(256) void v3(uint8_t *b, u64 n16, u64 chunk16) {
    const ulonglong2 v = make_ulonglong2(1, 2);
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u64 beg = wave * chunk16, end = beg + chunk16 < n16 ? beg + chunk16 : n16;
    for (u64 i = beg + (threadIdx.x & 63); i + 192 < end; i += 256) {
        ulonglong2 *p = reinterpret_cast<ulonglong2 *>(b) + i;
        p[0] = v; p[64] = v; p[128] = v; p[192] = v;
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const u64 bytes = 19456ull * 64512; const u64 n16 = bytes / 16;
    uint8_t *buf; hipMalloc(&buf, bytes + (1 << 20));
    for (int grid : {1024, 2048, 4864, 8192, 19456}) {
        float t = timeit([&] { hipLaunchKernelGGL(v0, dim3(grid), dim3(256), 0, 0, buf, n16); });
        printf("V0 grid-stride fill        grid %6d  %.3f ms %.0f GB/s\n", grid, t, bytes / t / 1e6);
    }
    for (int grid : {1024, 2048, 4864, 19456}) {
        u64 chunk = (n16 + grid - 1) / grid;
        float t = timeit([&] { hipLaunchKernelGGL(v1, dim3(grid), dim3(256), 0, 0, buf, n16, chunk); });
        printf("V1 block-private chunk     grid %6d  %.3f ms %.0f GB/s\n", grid, t, bytes / t / 1e6);
    }
    for (int grid : {1024, 2048, 4864, 19456}) {
        u64 chunk = (n16 + (u64)grid * 4 - 1) / ((u64)grid * 4);
        float t = timeit([&] { hipLaunchKernelGGL(v2, dim3(grid), dim3(256), 0, 0, buf, n16, chunk); });
        printf("V2 wave-private chunk      grid %6d  %.3f ms %.0f GB/s\n", grid, t, bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(v3, dim3(grid), dim3(256), 0, 0, buf, n16, chunk); });
        printf("V3 wave-private, unroll 4  grid %6d  %.3f ms %.0f GB/s\n", grid, t, bytes / t / 1e6);
    }
    return 0;
}
