// Which SIMD does wave w of a workgroup land on?  (HW_REG_HW_ID bits 5:4 = SIMD id on gfx9.)  If wave 0 of every 4-wave workgroup sits on
// the same SIMD of its CU, the chain kernel's serial owner waves (wave 0 of every element) share one SIMD while three idle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out, unsigned long long spin) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);   // keep the workgroups resident together
}
int main() {
    for (int nw : {4, 8, 1}) {
        const int n_wg = 1024 * 4 / nw;
        unsigned *d; hipMalloc(&d, n_wg * nw * 4);
        hipLaunchKernelGGL(probe, dim3(n_wg), dim3(64 * nw), 0, 0, d, 2000ull);
        std::vector<unsigned> h(n_wg * nw); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        unsigned hist[8][4] = {};
        for (int b = 0; b < n_wg; ++b) for (int w = 0; w < nw; ++w) hist[w][(h[b * nw + w] >> 4) & 3]++;
        printf("%d-wave workgroups (%d of them): SIMD of wave w, counts over SIMD 0..3\n", nw, n_wg);
        for (int w = 0; w < nw; ++w) printf("  wave %d: %u %u %u %u\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        // per CU: how many wave-0s share the busiest SIMD (se, sh, cu = bits 15:13, 12, 11:8; xcc from the block index % 8)
        hipFree(d);
    }
    return 0;
}
