// libh2r.so, translation unit "step": every instantiation of step_kernel (h2r_kernels.hpp) -- the records of call k and the
// chains of call k+1 in one launch -- and its launcher.
#include <algorithm>

#include "h2r_internal.hpp"
#include "h2r_sha256.hpp"

namespace h2r {
namespace {

template <int K, int NW, int LW, int L, bool WAVE = false>
hipError_t launch_step_t(u32 num_cus, const ChainArgs &ca, const TraceArgs &ta, const AuxArgs *aa, const AuxArgs *va, const Sha256Args *sha,
                         hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    constexpr int IPB = (64 * NW) / TraceGeo<L>::TPI;                   // record items per workgroup of this launch
    const u64 rec_blocks = (ta.n_items + IPB - 1) / IPB;
    // chain workgroups per CU: four 4-wave ones (what runs next to a record kernel on two queues), two 6- or 8-wave ones
    // (measured, profiles/r03_step_shapes.txt: RSA-3072 1.5 / 2 / 3 / 4 per CU -> 1.91 / 2.12 / 1.79 / 1.77 M
    //  assigns/s; RSA-4096 1 / 1.5 / 2 / 3 -> 1.04 / 1.22 / 1.44 / 1.18 M; 128 x 32-bit limbs 1 / 1.5 / 2 / 3 -> 0.599 / 0.606 / 0.606 / 0.544 M)
    u64 per_cu4 = NW == 4 ? 4ull * num_cus : 2ull * num_cus;
    if (knobs().step_chain_x2_per_cu > 0) per_cu4 = (u64)knobs().step_chain_x2_per_cu * num_cus / 2;
    // (WAVE: a chain workgroup is NW independent one-wave chains, so the role needs a quarter of the workgroups)
    u32 n_chain = (u32)std::min<u64>(WAVE ? (ca.batch + NW - 1) / NW : ca.batch, per_cu4);
    n_chain = (n_chain + 7) & ~7u;                                     // keeps blockIdx % 8 (the XCD) of the record role's workgroups
    AuxArgs none;
    std::memset(&none, 0, sizeof none);
    const u64 n_aux = aa ? aa->batch : 0;
    Sha256Args no_sha;
    std::memset(&no_sha, 0, sizeof no_sha);
    const u64 n_sha = sha ? ((sha->batch + 64 * NW - 1) / (64 * NW) + 7) & ~7ull : 0;   // one thread per message; a multiple of 8 (the XCD of what follows)
    const dim3 grid((unsigned)(n_sha + n_chain + rec_blocks + n_aux));
    if (va || sha) hipExtLaunchKernelGGL((step_kernel<K, NW, LW, L, true, WAVE>), grid, dim3(64 * NW), 0, st, ea, eb, 0,
                                         ca, ta, aa ? *aa : none, va ? *va : none, sha ? *sha : no_sha, (u32)n_sha, n_chain, (u32)rec_blocks);
    else hipExtLaunchKernelGGL((step_kernel<K, NW, LW, L, false, WAVE>), grid, dim3(64 * NW), 0, st, ea, eb, 0,
                               ca, ta, aa ? *aa : none, none, sha ? *sha : no_sha, (u32)n_sha, n_chain, (u32)rec_blocks);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_step_shape(const StepShape &s, u32 num_cus, const ChainArgs &ca, const TraceArgs &ta, const AuxArgs *aa, const AuxArgs *va,
                             const Sha256Args *sha, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    if (s.L == 32) return launch_step_t<64, 4, 64, 32>(num_cus, ca, ta, aa, va, sha, st, ea, eb);
    if (s.L == 16) {
        // RSA-1024: the chain role as one-wave chains (h2r_chain_wave.hpp) unless the verifier's witness is folded into the role (four-wave form only)
        const bool wave = knobs().chain_wave != 0 && !va;
        return wave ? launch_step_t<32, 4, 64, 16, true>(num_cus, ca, ta, aa, va, sha, st, ea, eb) : launch_step_t<32, 4, 64, 16>(num_cus, ca, ta, aa, va, sha, st, ea, eb);
    }
    if (s.L == 128) return launch_step_t<128, 8, 32, 128>(num_cus, ca, ta, aa, va, sha, st, ea, eb);
    if (s.L == 64) return launch_step_t<128, 8, 64, 64>(num_cus, ca, ta, aa, va, sha, st, ea, eb);
    return launch_step_t<96, 6, 64, 48>(num_cus, ca, ta, aa, va, sha, st, ea, eb);
}
u32 step_shared_bytes_shape(const StepShape &s) {
    if (s.L == 32) return (u32)sizeof(StepShared<64, 4, 64, 32>);
    if (s.L == 16) return (u32)std::max(sizeof(StepShared<32, 4, 64, 16>), sizeof(StepShared<32, 4, 64, 16, true>));
    if (s.L == 128) return (u32)sizeof(StepShared<128, 8, 32, 128>);
    if (s.L == 64) return (u32)sizeof(StepShared<128, 8, 64, 64>);
    return (u32)sizeof(StepShared<96, 6, 64, 48>);
}

}  // namespace h2r
