// Does hipExtAnyOrderLaunch (AQL barrier bit cleared) let consecutive launches of ONE stream overlap on gfx950, and what does it
// buy a store-bound launch sequence?  (1) two one-workgroup spinners of ~100 us each: in order 200 us, overlapped 100 us.
// (2) twenty fills of 1.25 GB (one 256-thread workgroup per 256 KB, non-temporal 16-byte stores), in order / any order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin_kernel(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && threadIdx.x == 0) atomicAdd(sink, 1u);
}
__global__ __launch_bounds__(256) void fill_kernel(uint4 *p, unsigned per_wg16) {
    uint4 *q = p + (size_t)blockIdx.x * per_wg16;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = {blockIdx.x, 1u, 2u, 3u};
    for (unsigned i = threadIdx.x; i < per_wg16; i += 256) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(q + i));
}
// (3) does a launch START before its predecessor has ENDED?  A spins 100 us then sets *flag; B reads *flag when it starts.
__global__ void late_flag_kernel(unsigned long long ticks, unsigned *flag) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void early_look_kernel(const unsigned *flag, unsigned *seen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) seen[0] = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename F> static float timed(hipStream_t st, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st)); f(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned *sink; CK(hipMalloc(&sink, 4)); CK(hipMemset(sink, 0, 4));
    const unsigned long long ticks = 10000;   // wall_clock64: 100 MHz
    void *sargs[] = {(void *)&ticks, (void *)&sink};
    auto spin = [&](int flags) { CK(hipExtLaunchKernel((const void *)spin_kernel, dim3(1), dim3(64), sargs, 0, st, nullptr, nullptr, flags)); };
    spin(0); CK(hipStreamSynchronize(st));
    for (int rep = 0; rep < 2; ++rep) {
        printf("two spinners of 100 us: in order %.3f ms, any order %.3f ms\n", timed(st, [&] { spin(0); spin(0); }), timed(st, [&] { spin(0); spin(hipExtAnyOrderLaunch); }));
    }
    for (int n : {1, 2, 4, 8}) {
        const float ms = timed(st, [&] { for (int k = 0; k < n; ++k) spin(k ? hipExtAnyOrderLaunch : 0); });
        printf("%d spinners, all but the first any order: %.3f ms\n", n, ms);
    }
    {
        unsigned *flag, *seen; CK(hipMalloc(&flag, 4)); CK(hipMalloc(&seen, 4));
        for (int flags = 0; flags < 2; ++flags)
            for (int wgs : {1, 4096}) {
                CK(hipMemset(flag, 0, 4)); CK(hipMemset(seen, 0xff, 4)); CK(hipStreamSynchronize(st)); CK(hipDeviceSynchronize());
                void *fa[] = {(void *)&ticks, (void *)&flag}; void *la[] = {(void *)&flag, (void *)&seen};
                CK(hipExtLaunchKernel((const void *)late_flag_kernel, dim3(1), dim3(64), fa, 0, st, nullptr, nullptr, 0));
                CK(hipExtLaunchKernel((const void *)early_look_kernel, dim3(wgs), dim3(64), la, 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0));
                CK(hipStreamSynchronize(st));
                unsigned h; CK(hipMemcpy(&h, seen, 4, hipMemcpyDeviceToHost));
                printf("successor (%d workgroups, %s) saw the predecessor's end-of-kernel flag = %u  (0: it started before the predecessor ended)\n", wgs, flags ? "any order" : "in order", h);
            }
    }
    const size_t bytes = 1250ull << 20; const unsigned per_wg16 = (256u << 10) / 16, n_wg = bytes / (256u << 10);
    uint4 *bufs[3]; for (auto &b : bufs) CK(hipMalloc(&b, bytes));
    for (int rep = 0; rep < 3; ++rep)
        for (int flags = 0; flags < 2; ++flags) {
            auto run = [&] { for (int k = 0; k < 20; ++k) { uint4 *p = bufs[k % 3]; void *fa[] = {(void *)&p, (void *)&per_wg16};
                CK(hipExtLaunchKernel((const void *)fill_kernel, dim3(n_wg), dim3(256), fa, 0, st, nullptr, nullptr, (flags && k) ? hipExtAnyOrderLaunch : 0)); } };
            run(); CK(hipStreamSynchronize(st));
            const float ms = timed(st, run);
            printf("20 fills of 1.25 GB, %s: %.4f ms per fill (%.2f TB/s)\n", flags ? "any order" : "in order ", ms / 20, bytes / (ms / 20) / 1e9);
        }
    return 0;
}
