// Internal to libh2r.so: what the library's translation units share.
//
// The library is compiled as several translation units so that a forced build stays under a minute (the device
// compile of the template kernel families is what a build costs; each family is instantiated in exactly one unit):
//   h2r_tu_trace.hip   trace_kernel<LW, L>         (one record per mul_mod; every supported shape)
//   h2r_tu_chain.hip   recip / chain / chain_dual / chain_wave kernels (the dependent mul_mod chain per element)
//   h2r_tu_step.hip    step_kernel<K, NW, LW, L>   (records of call k + chains of call k+1 in one launch)
//   h2r_tu_cells.hip   cells_kernel<LW, ABL, MONT, NWV> (the advice image directly from the operands)
//   h2r_api.hip        the C ABI, the ctx, the pipelines, and every small kernel
// The launchers below take plain values, never the ctx: `struct h2r_ctx` stays private to h2r_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>
#include <cstring>

#include "h2r.h"
#include "h2r_kernels.hpp"
#include "h2r_cells.hpp"

namespace h2r {

// Developer knobs: compiled in ONLY by the -DH2R_DEV_KNOBS build (python -m halo2_rsa_amd._build <name> -DH2R_DEV_KNOBS,
// selected with H2R_LIB by the sweeps under tools/).  The product library never reads the environment: every knob
// keeps the measured default below.
struct Knobs {
    int chain_nw = 0, chain_deep = -1;          // H2R_CHAIN_NW, H2R_CHAIN_DEEP
    long chain_wave = -1;                        // H2R_CHAIN_WAVE=0|1: the one-wavefront chain for 32-digit elements (-1 = the measured default)
    long trace_dyn_lds = -1, trace_prio = -1;    // H2R_TRACE_DYN_LDS, H2R_TRACE_PRIO
    long chain_prio = -1, ablate = 0;            // H2R_CHAIN_PRIO, H2R_ABLATE (needs the -DH2R_ABLATION build)
    int pipe_stream_prio = -1;                   // H2R_PIPE_STREAM_PRIO = low (default) | normal | high  -> -1 | 0 | +1
    bool chain_timing = false;                   // H2R_CHAIN_TIMING (needs the -DH2R_CHAIN_TIMING build)
    bool pipe_serialize = false;                 // H2R_PIPE_SERIALIZE=1: with two record streams, a record kernel also waits for the previous one
    long plain_overlap = -1;                     // H2R_PLAIN_OVERLAP=0: the plain pow exports never overlap their sub-batches internally
    long arena_chunk_mb = 0;                     // H2R_ARENA_CHUNK_MB: physical chunk size of the arena's regions
    long pipe_sub_batch = 0;                     // H2R_PIPE_SUB_BATCH: elements per chain + record kernel pair inside a pipelined call (multiple of 256)
    long pipe_pace = -1;                         // H2R_PIPE_PACE=0|1: sub-batch i+1's chain kernel waits for sub-batch i-1's record kernel
    long pipe_step = -1;                         // H2R_PIPE_STEP=0: never issue a pipeline step as one launch (the two-queue form for every shape)
    long pipe_twoq_l16 = 0;                      // H2R_PIPE_TWOQ_L16=n: RSA-1024 takes the two-queue form for every call of up to n (-1: never; 0 = the measured rule)
    long pipe_form = -1;                         // H2R_PIPE_FORM=0|1: skip the queue probe; 1 = the two-queue form, 0 = the one-launch step
    long step_chain_x2_per_cu = 0;               // H2R_STEP_CHAIN_X2_PER_CU=n: n/2 chain workgroups per CU in a step launch (0 = the measured default)
    unsigned long pipe_cu_mask = 0;              // H2R_PIPE_CU_MASK=<hex word>: the record stream is created with this 32-bit CU mask repeated over the device (experiment)
    long pipe_cu_mask_words = 0;                 // H2R_PIPE_CU_MASK_WORDS=n: only the first n 32-bit words carry the mask, the rest are zero
    long verify_fold = -1;                       // H2R_VERIFY_FOLD=0|1: the verifier's witness inside the step launch's chain role (-1 = the measured default per shape)
    long exp_segments = -1;                      // H2R_EXP_SEGMENTS=n: segments a long exponent is walked in (0 / 1 = never; -1 = the default rule, exp_segment_count)
    long single_call_segments = -1;              // H2R_SINGLE_CALL_SEGMENTS=n: segments of a SHORT exponent in a single stream-ordered call of 513..1,536 RSA-2048 elements
    long rowprog_stage_rows = 0;                 // H2R_ROWPROG_STAGE_ROWS=64|128|256: rows (= threads) of a row-program workgroup (0 = the rule in launch_row_prog)
    long cells_nwv = 0;                          // H2R_CELLS_NWV=1|8: waves per cells_kernel workgroup of a Montgomery ctx (0 = the rule at ctx creation)
    Knobs() {
#ifdef H2R_DEV_KNOBS
        auto num = [](const char *name, long dflt) { const char *v = std::getenv(name); return v ? std::atol(v) : dflt; };
        chain_nw = (int)num("H2R_CHAIN_NW", 0); chain_deep = (int)num("H2R_CHAIN_DEEP", -1); chain_wave = num("H2R_CHAIN_WAVE", -1);
        trace_dyn_lds = num("H2R_TRACE_DYN_LDS", -1); trace_prio = num("H2R_TRACE_PRIO", -1);
        chain_prio = num("H2R_CHAIN_PRIO", -1); ablate = num("H2R_ABLATE", 0);
        const char *pe = std::getenv("H2R_PIPE_STREAM_PRIO");
        pipe_stream_prio = !pe ? -1 : (!std::strcmp(pe, "high") ? 1 : (!std::strcmp(pe, "low") ? -1 : 0));
        chain_timing = std::getenv("H2R_CHAIN_TIMING") != nullptr;
        { const char *g = std::getenv("H2R_PIPE_SERIALIZE"); pipe_serialize = g && g[0] == '1'; }
        { const char *m = std::getenv("H2R_PIPE_CU_MASK"); pipe_cu_mask = m ? std::strtoul(m, nullptr, 16) : 0; pipe_cu_mask_words = num("H2R_PIPE_CU_MASK_WORDS", 0); }
        verify_fold = num("H2R_VERIFY_FOLD", -1); exp_segments = num("H2R_EXP_SEGMENTS", -1); single_call_segments = num("H2R_SINGLE_CALL_SEGMENTS", -1);
        pipe_step = num("H2R_PIPE_STEP", -1); pipe_form = num("H2R_PIPE_FORM", -1); pipe_twoq_l16 = num("H2R_PIPE_TWOQ_L16", 0); step_chain_x2_per_cu = num("H2R_STEP_CHAIN_X2_PER_CU", 0);
        rowprog_stage_rows = num("H2R_ROWPROG_STAGE_ROWS", 0); cells_nwv = num("H2R_CELLS_NWV", 0);
        pipe_sub_batch = num("H2R_PIPE_SUB_BATCH", 0); arena_chunk_mb = num("H2R_ARENA_CHUNK_MB", 0); plain_overlap = num("H2R_PLAIN_OVERLAP", -1); pipe_pace = num("H2R_PIPE_PACE", -1);
#endif
    }
};
inline const Knobs &knobs() { static const Knobs k; return k; }   // one instance for the whole library (inline function, static local)

// Shapes with a compiled record kernel.  BigIntChip::new only asserts bits_len % limb_width == 0 (chip.rs:1175); here
// num_limbs must also be a multiple of 4 (64-bit limbs, up to 4096 bits: RSA-1024/1536/2048/3072/4096 ...) or of 8
// (32-bit limbs, up to 4096 bits), so that every accumulator row is a whole number of 64-byte store segments.
constexpr u32 kLStep64 = 4, kLMax64 = 64, kLStep32 = 8, kLMax32 = 128;
inline bool shape_supported(u32 w, u32 L) {
    if (w == 64) return L >= kLStep64 && L <= kLMax64 && L % kLStep64 == 0;
    if (w == 32) return L >= kLStep32 && L <= kLMax32 && L % kLStep32 == 0;
    return false;
}

// The shapes with a step build: (limb width, limbs) -> (chain digits K, waves NW).  A step's workgroup is the chain role's.
struct StepShape { u32 w, L, K, NW; };
constexpr StepShape kStepShapes[] = {{64, 32, 64, 4}, {64, 16, 32, 4}, {32, 128, 128, 8}, {64, 64, 128, 8}, {64, 48, 96, 6}};
inline const StepShape *step_shape_of(u32 w, u32 L, u32 K) {
    for (const StepShape &s : kStepShapes) if (w == s.w && L == s.L && K == s.K) return &s;
    return nullptr;
}

// h2r_tu_trace.hip.  ea/eb (nullable): start/stop events stamped by the dispatch itself.
hipError_t launch_trace_shape(u32 w, u32 L, u32 lds_per_cu, const TraceArgs &ta, hipStream_t st, hipEvent_t ea, hipEvent_t eb);
// h2r_tu_chain.hip.  co_running: the call's record kernel of the PREVIOUS batch runs next to this chain kernel (pipeline mode)
hipError_t launch_chain_shape(u32 num_cus, const ChainArgs &ca, bool co_running, hipStream_t st, hipEvent_t ea, hipEvent_t eb);
// h2r_tu_step.hip.  One step: the records described by `ta` (an earlier sub-batch) and the chains described by `ca`, one launch on `st`.
hipError_t launch_step_shape(const StepShape &s, u32 num_cus, const ChainArgs &ca, const TraceArgs &ta, const AuxArgs *aa, const AuxArgs *va,
                             const Sha256Args *sha, hipStream_t st, hipEvent_t ea, hipEvent_t eb);
u32 step_shared_bytes_shape(const StepShape &s);
// h2r_tu_cells.hip.  `lds` = the dynamic LDS request (residency rule applied by the caller); nwv = waves per workgroup (Montgomery, long shapes).
hipError_t launch_cells_shape(u32 w, bool mont, u32 nwv, u32 lds, const CellsArgs &ca, hipStream_t st, hipEvent_t ea, hipEvent_t eb);

}  // namespace h2r
