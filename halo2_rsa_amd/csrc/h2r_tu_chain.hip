// libh2r.so, translation unit "chain": the dependent mul_mod chain per element -- recip_kernel, chain_kernel, chain_dual_kernel
// (h2r_kernels.hpp) -- and the launcher that picks the build for a shape and a batch.
#include "h2r_internal.hpp"

namespace h2r {
namespace {

// grid_cap: upper bound on the chain kernel's workgroups (0 = one per element); a smaller grid walks the batch
template <int K, int NW, bool DEEP>
hipError_t launch_chain_t(const ChainArgs &ca, u64 grid_cap, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    if (ca.batch == 0) return hipSuccess;
    if (ca.pre) {   // the shared modulus' Barrett constants, once, ahead of the elements' chains
        hipLaunchKernelGGL((recip_kernel<K, NW>), dim3(1), dim3(64 * NW), 0, st, ca.n, ca.kreal, const_cast<u32 *>(ca.pre));
        if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
    }
    const u64 grid = grid_cap && grid_cap < ca.batch ? grid_cap : ca.batch;
    if (ca.state) hipExtLaunchKernelGGL((chain_kernel<K, NW, DEEP, true>), dim3((unsigned)grid), dim3(64 * NW), 0, st, ea, eb, 0, ca);   // a segment of a long exponent
    else hipExtLaunchKernelGGL((chain_kernel<K, NW, DEEP, false>), dim3((unsigned)grid), dim3(64 * NW), 0, st, ea, eb, 0, ca);
    return hipGetLastError();
}
// co_running: the call's record kernel of the PREVIOUS batch runs next to this chain kernel (pipeline mode)
}  // namespace

hipError_t launch_chain_shape(u32 num_cus, const ChainArgs &ca, bool co_running, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    // Footprint next to a record kernel: at most four 4-wave (two 8-wave) workgroups per CU, the residency the
    // batch-1024 RSA-2048 call has; a larger batch is walked by that grid instead of queueing more workgroups (a chain
    // kernel with 8,192 workgroups kept every CU full of its waves and cost the record kernel 15 % of its store rate).
    const u64 cap4 = co_running ? 4ull * num_cus : 0, cap2 = co_running ? 2ull * num_cus : 0;
    // The chain kernel is compiled for K = 8, 16, 32, 64, 96, 128 digits; any other size runs as the next larger one with
    // zero high digits (ca.kreal digits in memory).  NW = waves per element (a multiple of the 64-column groups).
    // K = 96 (RSA-3072) is its own build: run as K = 128 it did 1.8x the multiply-accumulates and made the chain kernel
    // the longer leg of the pipeline (0.57-0.65 ms against a 0.49 ms record kernel per 1,024 signatures).
    const u32 K = ca.kreal <= 8 ? 8 : ca.kreal <= 16 ? 16 : ca.kreal <= 32 ? 32 : ca.kreal <= 64 ? 64 : ca.kreal <= 96 ? 96 : 128;
    switch (K) {
        case 8: return launch_chain_t<8, 1, false>(ca, 4 * cap4, st, ea, eb);
        case 16: return launch_chain_t<16, 1, false>(ca, 4 * cap4, st, ea, eb);
        case 32: {
            // one wavefront per element (h2r_chain_wave.hpp): workgroups of four independent chains
            if (ca.batch == 0) return hipSuccess;
            if (knobs().chain_wave != 0 && knobs().chain_nw == 0) {
                if (ca.pre) {
                    hipLaunchKernelGGL((recip_kernel<32, 4>), dim3(1), dim3(256), 0, st, ca.n, ca.kreal, const_cast<u32 *>(ca.pre));
                    if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
                }
                const u64 wgs = (ca.batch + CHAIN_WAVE_WPB - 1) / CHAIN_WAVE_WPB;
                const u64 grid = cap4 && cap4 < wgs ? cap4 : wgs;
                if (ca.state) hipExtLaunchKernelGGL((chain_wave_kernel<32, true>), dim3((unsigned)grid), dim3(64 * CHAIN_WAVE_WPB), 0, st, ea, eb, 0, ca);
                else hipExtLaunchKernelGGL((chain_wave_kernel<32, false>), dim3((unsigned)grid), dim3(64 * CHAIN_WAVE_WPB), 0, st, ea, eb, 0, ca);
                return hipGetLastError();
            }
            return launch_chain_t<32, 4, false>(ca, cap4, st, ea, eb);
        }
        case 64: {
            // Two chains per element side by side (chain_dual_kernel: squarings and multiplies of one exponent bit in lockstep, eight
            // waves): for latency-bound batches -- at most two elements per CU -- of variable exponents (two independent mul_mods per
            // bit) and of DENSE fixed exponents (a zero bit costs the multiply group a dropped mul_mod).  BASELINE config 5: 3,072
            // dependent mul_mods become 2,048 steps.
            if (knobs().chain_nw == 0 && knobs().chain_deep < 0 && ca.mode != CHAIN_MULMOD && !ca.pre && ca.batch <= 2ull * num_cus) {
                u32 pop = 0;
                for (u32 wi = 0; wi < (ca.e.nbits + 31) / 32; ++wi) pop += (u32)__builtin_popcount(ca.e.words[wi]);
                const bool dense = ca.mode == CHAIN_POW_VAR || (ca.e.nbits >= 64 && 4 * pop >= ca.e.nbits);
                if (dense) {
                    const u64 grid = cap2 && cap2 < ca.batch ? cap2 : ca.batch;
                    hipExtLaunchKernelGGL((chain_dual_kernel<64, true>), dim3((unsigned)grid), dim3(512), 0, st, ea, eb, 0, ca);
                    return hipGetLastError();
                }
            }
            // Throughput build (6 blocks per CU) when the batch fills the chip; latency build (deep operand prefetch,
            // 139 VGPRs) when there are at most two elements per CU and each chain's own latency is what the call
            // waits for (BASELINE config 5: 256 elements x 3,072 dependent mul_mods: 9.4 -> 7.8 ms, tools/c5_sweep.sh).
            const int nw_env = knobs().chain_nw, deep_env = knobs().chain_deep;
            const bool small = ca.batch <= 512;
            const int nw = nw_env ? nw_env : 4;
            const bool deep = deep_env >= 0 ? deep_env != 0 : small;
            if (nw == 8) return deep ? launch_chain_t<64, 8, true>(ca, cap2, st, ea, eb) : launch_chain_t<64, 8, false>(ca, cap2, st, ea, eb);
            if (nw == 2) return launch_chain_t<64, 2, false>(ca, 2 * cap4, st, ea, eb);
            return deep ? launch_chain_t<64, 4, true>(ca, cap4, st, ea, eb) : launch_chain_t<64, 4, false>(ca, cap4, st, ea, eb);
        }
        case 96: return launch_chain_t<96, 6, false>(ca, cap4 * 2 / 3, st, ea, eb);
        default: return launch_chain_t<128, 8, false>(ca, cap2, st, ea, eb);
    }
}

}  // namespace h2r
