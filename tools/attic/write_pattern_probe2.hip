// Probe 2: what slows a store-bound kernel down?  Same bytes as one trace_kernel launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
struct Cfg { int valu_per_step; int tail_stores; int barrier_every; int nt; int wide_only; };
template <int NT>
__device__ __forceinline__ void st16(uint8_t *p, ulonglong2 v) {
    if (NT) { __builtin_nontemporal_store(v.x, (u64 *)p); __builtin_nontemporal_store(v.y, (u64 *)p + 1); }
    else *reinterpret_cast<ulonglong2 *>(p) = v;
}
template <int NT>
__global__ __launch_bounds__(256) void k(uint8_t *base, u64 nwaves, int steps, Cfg c, u64 *sink) {
    extern __shared__ u64 sh[];
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    u64 x = wave * 64 + lane, y = 0x9e3779b97f4a7c15ull;
    uint8_t *rec = base + wave * (u64)65536;
    const int main_steps = steps - (c.tail_stores ? 6 : 0);
    for (int s = 0; s < main_steps; ++s) {
        for (int i = 0; i < c.valu_per_step; ++i) { x = x * y + (x >> 7); }   // ~4-5 VALU ops each (64-bit mul)
        if (c.barrier_every && (s % c.barrier_every) == 0) __syncthreads();
        if (c.wide_only) { st16<NT>(rec + (u64)s * 1536 + lane * 16, make_ulonglong2(x, y)); if (lane < 32) st16<NT>(rec + (u64)s * 1536 + 1024 + lane * 16, make_ulonglong2(y, x)); }
        else {
            st16<NT>(rec + (u64)s * 1024 + lane * 16, make_ulonglong2(x, y));
            if (NT) __builtin_nontemporal_store(x, (u64 *)(rec + 43008 + (u64)s * 512 + lane * 8)); else *(u64 *)(rec + 43008 + (u64)s * 512 + lane * 8) = x;
        }
    }
    if (c.tail_stores) {  // 6 steps' worth of bytes (9 KB) as 18 narrow 8-byte-per-lane stores of 63 lanes
        for (int t = 0; t < 18; ++t) if (lane < 63) *(u64 *)(rec + 55296 + t * 512 + lane * 8) = x + t;
    }
    if (x == 42) sink[0] = x;
}
float run(int nt, uint8_t *buf, u64 nwaves, int steps, Cfg c, u64 *sink, int lds) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto launch = [&]() { if (nt) hipLaunchKernelGGL(k<1>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), lds, 0, buf, nwaves, steps, c, sink);
                          else hipLaunchKernelGGL(k<0>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), lds, 0, buf, nwaves, steps, c, sink); };
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const u64 nwaves = 19456; const int steps = 42; const u64 bytes = nwaves * steps * 1536ull;
    uint8_t *buf; hipMalloc(&buf, nwaves * 65536ull + (1 << 20)); u64 *sink; hipMalloc(&sink, 64);
    struct { const char *name; Cfg c; int nt; int lds; } tests[] = {
        {"base (16B+8B stores, 8 blocks/CU)", {0, 0, 0, 0, 0}, 0, 0},
        {"nontemporal stores", {0, 0, 0, 1, 0}, 1, 0},
        {"16B-only stores", {0, 0, 0, 0, 1}, 0, 0},
        {"+8 mul-adds per step", {8, 0, 0, 0, 0}, 0, 0},
        {"+32 mul-adds per step", {32, 0, 0, 0, 0}, 0, 0},
        {"+64 mul-adds per step", {64, 0, 0, 0, 0}, 0, 0},
        {"narrow 63-lane tail stores", {0, 1, 0, 0, 0}, 0, 0},
        {"barrier every 8 steps", {0, 0, 8, 0, 0}, 0, 0},
        {"4 blocks/CU (LDS 40KB)", {0, 0, 0, 0, 0}, 0, 40960},
        {"2 blocks/CU (LDS 80KB)", {0, 0, 0, 0, 0}, 0, 81920},
        {"1 block/CU (LDS 160KB)", {0, 0, 0, 0, 0}, 0, 163840},
        {"32 mul-adds + 4 blocks/CU", {32, 0, 0, 0, 0}, 0, 40960},
        {"32 mul-adds + tail + barrier", {32, 1, 8, 0, 0}, 0, 0},
    };
    for (auto &t : tests) {
        hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        float ms = run(t.nt, buf, nwaves, steps, t.c, sink, t.lds);
        printf("%-36s %.3f ms  %.0f GB/s\n", t.name, ms, bytes / ms / 1e6);
    }
    return 0;
}
