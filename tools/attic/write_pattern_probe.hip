// Probe: HBM write bandwidth vs. store pattern (no compute).  Build: hipcc --offload-arch=gfx950 -O3
// Each wave writes STEPS "steps"; per step one 16-byte store + one 8-byte store per lane, like trace_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
// pattern 0: grid-coalesced fill (every wave writes consecutive 1 KB chunks interleaved across the grid)
// pattern 1: per-wave private 96 KB region, 1 KB (lo) + 512 B (hi) contiguous per step
// pattern 2: per-wave private region, per step 2 x 512 B (lo planes) + 2 x 256 B (hi planes)  [current trace layout]
// pattern 3: step-major across waves: lo[(step*NW + wave)*1KB], hi[(step*NW+wave)*512B]
// pattern 4: like 2 but the 4 waves of a block interleaved: rows of the 4 items adjacent (4 KB per block-step)
template <int P>
__global__ __launch_bounds__(256) void k(uint8_t *base, u64 nwaves, int steps, u64 hi_base) {
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    const ulonglong2 v = make_ulonglong2(wave, lane);
    for (int s = 0; s < steps; ++s) {
        uint8_t *lo, *hi;
        if (P == 0) { lo = base + ((u64)s * nwaves + wave) * 1024 + lane * 16; hi = base + hi_base + ((u64)s * nwaves + wave) * 512 + lane * 8; }
        if (P == 1) { lo = base + wave * (u64)steps * 1536 + (u64)s * 1024 + lane * 16; hi = base + wave * (u64)steps * 1536 + (u64)steps * 1024 + (u64)s * 512 + lane * 8; }
        if (P == 2) { const int h = lane >> 5, i = lane & 31;
            uint8_t *rec = base + wave * (u64)steps * 1536;
            lo = rec + (u64)h * steps * 512 + (u64)s * 512 + i * 16; hi = rec + (u64)steps * 1024 + (u64)h * steps * 256 + (u64)s * 256 + i * 8; }
        if (P == 3) { lo = base + ((u64)s * nwaves + wave) * 1024 + lane * 16; hi = base + hi_base + ((u64)s * nwaves + wave) * 512 + lane * 8; }
        if (P == 4) { const u64 blk = wave >> 2, w4 = wave & 3;
            uint8_t *rec = base + blk * (u64)steps * 1536 * 4;
            lo = rec + ((u64)s * 4 + w4) * 1024 + lane * 16; hi = rec + (u64)steps * 4096 + ((u64)s * 4 + w4) * 512 + lane * 8; }
        *reinterpret_cast<ulonglong2 *>(lo) = v;
        *reinterpret_cast<u64 *>(hi) = v.x;
    }
}
template <int P> float run(uint8_t *buf, u64 nwaves, int steps, u64 hi_base) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<P>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, 0, buf, nwaves, steps, hi_base);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<P>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, 0, buf, nwaves, steps, hi_base);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const u64 nwaves = 19456; const int steps = 42;  // 42 * 1.5 KB = 63 KB per wave ~ one mul_mod record
    const u64 bytes = nwaves * steps * 1536ull;
    uint8_t *buf; hipMalloc(&buf, bytes + (1 << 20));
    const u64 hi_base = nwaves * steps * 1024ull;
    float t[5] = {run<0>(buf, nwaves, steps, hi_base), run<1>(buf, nwaves, steps, hi_base), run<2>(buf, nwaves, steps, hi_base),
                  run<3>(buf, nwaves, steps, hi_base), run<4>(buf, nwaves, steps, hi_base)};
    const char *names[5] = {"0 grid-coalesced", "1 private 1KB+512B", "2 private 2x512B+2x256B (current)", "3 step-major across waves", "4 block-interleaved 4KB"};
    for (int p = 0; p < 5; ++p) printf("pattern %-36s %.3f ms  %.0f GB/s\n", names[p], t[p], bytes / t[p] / 1e6);
    return 0;
}
