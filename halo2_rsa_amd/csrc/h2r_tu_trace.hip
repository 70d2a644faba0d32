// libh2r.so, translation unit "trace": every instantiation of trace_kernel (h2r_kernels.hpp) and its launcher.
#include "h2r_internal.hpp"

namespace h2r {
namespace {

template <int LW, int L>
hipError_t launch_trace_t(const TraceArgs &ta, u32 lds_per_cu, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    // the kernel hard-codes the accumulator row strides layout_compute derives for (LW, L)
    if (ta.acc_lo_group != (LW == 64 ? 3ull * (2 * L * 16) : (u64)L * 16) || ta.acc_hi_group != ta.acc_lo_group ||
        ta.acc_lo_row != (LW == 64 ? 2u * L * 16 : 0u) || ta.acc_spg != (LW == 64 ? 2u : 1u)) return hipErrorInvalidValue;
    // (measured for the RSA-2048 shape: 128- and 64-thread workgroups are no better at any residency)
    constexpr int BT = TraceGeo<L>::BT, IPB = TraceGeo<L>::IPB;
    const u64 blocks = (ta.n_items + IPB - 1) / IPB;
    if (blocks == 0) return hipSuccess;
    // Residency cap: `residency` workgroups per CU (0 = whatever fits).  The cap is enforced the way occupancy is
    // enforced on this hardware -- by the workgroup's LDS allocation: the launch requests as much (untouched) dynamic
    // LDS as makes exactly `residency` workgroups fill a CU's LDS.
    u32 dyn = ta.dyn_lds;
    if (ta.residency) {
        static const u32 static_lds = [] {   // the kernel's own (static) LDS, once per instantiation
            hipFuncAttributes fa;
            return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&trace_kernel<LW, L, BT>)) == hipSuccess ? (u32)fa.sharedSizeBytes : 0u;
        }();
        const u32 per = lds_per_cu / ta.residency;
        dyn = per > static_lds + 1024 ? ((per - static_lds - 512) & ~15u) : 0u;   // `residency` fit, `residency + 1` do not
    }
    if (dyn > 48 * 1024) {   // large requests must be announced (once per device; harmless to repeat)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&trace_kernel<LW, L, BT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        (void)hipGetLastError();
    }
    hipExtLaunchKernelGGL((trace_kernel<LW, L, BT>), dim3((unsigned)blocks), dim3(BT), dyn, st, ea, eb, 0, ta);
    return hipGetLastError();
}
// one instantiation per supported num_limbs: L = STEP, 2 STEP, ..., MAXL
template <int LW, int STEP, int I>
hipError_t launch_trace_w(u32 L, const TraceArgs &ta, u32 lds_per_cu, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    if constexpr (I == 0) return hipErrorInvalidValue;
    else {
        if (L == (u32)(I * STEP)) return launch_trace_t<LW, I * STEP>(ta, lds_per_cu, st, ea, eb);
        return launch_trace_w<LW, STEP, I - 1>(L, ta, lds_per_cu, st, ea, eb);
    }
}

}  // namespace

hipError_t launch_trace_shape(u32 w, u32 L, u32 lds_per_cu, const TraceArgs &ta, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    if (!shape_supported(w, L)) return hipErrorInvalidValue;
    if (w == 64) return launch_trace_w<64, kLStep64, kLMax64 / kLStep64>(L, ta, lds_per_cu, st, ea, eb);
    return launch_trace_w<32, kLStep32, kLMax32 / kLStep32>(L, ta, lds_per_cu, st, ea, eb);
}

}  // namespace h2r
