// Where do the waves of the chain kernel's workgroups land?  (developer probe)
// 1,024 workgroups of NW waves with the chain kernel's LDS footprint, all resident at once (each spins ~30 us); every wave
// records HW_ID (SIMD, CU, SH, SE) and XCC_ID.  Output: per CU, how many workgroups it holds and on how many distinct SIMDs
// their wave 0 sits -- the chain kernel's serial part runs in wave 0 of every workgroup.
//   hipcc --offload-arch=gfx950 -O2 -o tools/wave_simd_probe tools/wave_simd_probe.hip && tools/wave_simd_probe [NW] [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
template <int NW>
__global__ __launch_bounds__(64 * NW) void probe(unsigned *out, unsigned spin) {
    __shared__ unsigned lds[2600];   // ~10.4 KB like ChainLds<64, 4>
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
    if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * NW + (threadIdx.x >> 6))] = hw; out[2 * (blockIdx.x * NW + (threadIdx.x >> 6)) + 1] = xcc + lds[1] - 1; }
}
int main(int argc, char **argv) {
    const int NW = argc > 1 ? atoi(argv[1]) : 4, blocks = argc > 2 ? atoi(argv[2]) : 1024;
    unsigned *d; hipMalloc(&d, sizeof(unsigned) * 2 * blocks * NW);
    hipStream_t st = nullptr;
    if (argc > 3) {   // [mask word hex] [words carrying it]: launch on a stream with that CU mask and list the CUs that got work
        uint32_t mask[8]; const uint32_t m = (uint32_t)strtoul(argv[3], nullptr, 16); const int words = argc > 4 ? atoi(argv[4]) : 8;
        for (int w = 0; w < 8; ++w) mask[w] = w < words ? m : 0u;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
    }
    for (int rep = 0; rep < 2; ++rep) {
        if (NW == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, st, d, 3000u);
        else if (NW == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(128), 0, st, d, 3000u);
        else hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(512), 0, st, d, 3000u);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * blocks * NW);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;   // key: xcc|se|sh|cu -> blocks
    int w_eq_simd = 0;
    std::map<unsigned, std::multiset<int>> w0_simd;
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < NW; ++w) {
            const unsigned hw = h[2 * (b * NW + w)], xcc = h[2 * (b * NW + w) + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            if (w == 0) { per_cu[key].push_back(b); w0_simd[key].insert((int)simd); }
            if ((int)simd == (w & 3)) ++w_eq_simd;
        }
    }
    printf("NW %d blocks %d: CUs seen %zu, waves with simd == wave %% 4: %d of %d\n", NW, blocks, per_cu.size(), w_eq_simd, blocks * NW);
    std::map<int, int> hist_blocks, hist_distinct, hist_maxshare;
    for (auto &kv : per_cu) {
        hist_blocks[(int)kv.second.size()]++;
        std::set<int> ds(w0_simd[kv.first].begin(), w0_simd[kv.first].end());
        hist_distinct[(int)ds.size()]++;
        int mx = 0; for (int sd : ds) mx = std::max<int>(mx, (int)w0_simd[kv.first].count(sd));
        hist_maxshare[mx]++;
    }
    printf("workgroups per CU:"); for (auto &kv : hist_blocks) printf("  %d -> %d CUs", kv.first, kv.second); printf("\n");
    printf("distinct SIMDs holding a wave 0, per CU:"); for (auto &kv : hist_distinct) printf("  %d -> %d CUs", kv.first, kv.second); printf("\n");
    printf("most wave 0s on one SIMD, per CU:"); for (auto &kv : hist_maxshare) printf("  %d -> %d CUs", kv.first, kv.second); printf("\n");
    if (st) {   // CUs per XCC that got work under the mask
        std::map<unsigned, int> per_xcc; for (auto &kv : per_cu) per_xcc[kv.first >> 12]++;
        printf("CUs with work per XCC:"); for (auto &kv : per_xcc) printf("  xcc%u -> %d", kv.first, kv.second); printf("\n");
    }
    int shown = 0;
    for (auto &kv : per_cu) { if (shown++ >= 4) break; printf("cu %05x blocks:", kv.first); for (int b : kv.second) printf(" %d", b); printf("  wave-0 simds:"); for (int sd : w0_simd[kv.first]) printf(" %d", sd); printf("\n"); }
    return 0;
}
