// C ABI of libh2r (see include/h2r.h).  Host side: context, layouts, kernel launches, flatten.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the functions are resolved with dlsym (no link-time dependency on librccl)

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#define H2R_TU_API 1   // this unit defines the plain kernels of the shared headers
#include "h2r.h"
#include "h2r_internal.hpp"
#include "h2r_varrows.hpp"
#include "h2r_copymap.hpp"
#include "h2r_layout.hpp"
#include "h2r_lookup.hpp"
#include "h2r_muled.hpp"
#include "h2r_rowprog.hpp"
#include "h2r_sha256.hpp"
#include "h2r_check.hpp"

using namespace h2r;

namespace {

thread_local char g_hip_err[256] = "";

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    std::snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
    return false;
}
// No C++ exception crosses the C ABI ("never abort/throw across the boundary"): every export is a function-try-block.
// A failed host allocation (std::vector, std::map, new) becomes H2R_E_NOMEM, anything else H2R_E_INTERNAL; the size queries return 0.
#define H2R_CATCH_STATUS catch (const std::bad_alloc &) { return H2R_E_NOMEM; } catch (...) { return H2R_E_INTERNAL; }
#define H2R_CATCH_ZERO catch (...) { return 0; }
#define H2R_CATCH_VOID catch (...) { }
#define H2R_CATCH_STR catch (...) { return ""; }
#define HIP_TRY(expr)                                   \
    do {                                                \
        if (!hip_ok((expr), #expr)) return H2R_E_HIP;   \
    } while (0)

// Every export that touches the device selects the ctx's device for the duration of the call and restores the caller
// thread's current device on return (the library has no thread-global side effects).
struct DeviceGuard {
    int prev = -1; bool switched = false; hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define H2R_ON_DEVICE(dev_)                                     \
    DeviceGuard h2r_device_guard_(dev_);                        \
    if (!hip_ok(h2r_device_guard_.err, "hipSetDevice")) return H2R_E_HIP

// ---- optional per-kernel event timing --------------------------------------------------------------
struct ProfRec { u32 kernel; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
u32 g_prof_cap = 0;
u32 g_prof_gen = 0;   // bumped whenever the recorded events are destroyed (h2r_profile_enable)
struct ProfScope {  // start/stop events of one launch when profiling is armed
    // ext == false: the events are recorded around the launch (two marker packets on the stream).
    // ext == true:  the caller hands a/b to hipExtLaunchKernelGGL, which stamps them from the dispatch packet's own
    //               completion signal -- no extra packets (each marker costs ~5 us of queue time between kernels).
    hipStream_t st; hipEvent_t a = nullptr, b = nullptr; u32 kernel; bool on = false, ext;
    ProfScope(u32 k, hipStream_t s, bool ext_ = false) : st(s), kernel(k), ext(ext_) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_cap && g_prof.size() < g_prof_cap && hipEventCreate(&a) == hipSuccess) {
            if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); a = nullptr; return; }
            on = true;
            if (!ext) (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope() {
        if (!on) return;
        if (!ext) (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(ProfRec{kernel, a, b});
    }
};

// (developer knobs: h2r_internal.hpp)

// Element strides, in units of 256 bytes.  The record kernel's store rate depends on how consecutive elements' records
// fall on the HBM channel interleave (sweep of the stride at batch 1024, profiles/r01_elem_stride_sweep.txt): the
// compact pow element (19 * 255 + 20 = 4865 units) is the best, strides whose residue modulo 256 units lies around
// 40..50 lose up to 10 % (that is where the verify element, 4904 units, landed), most others lose ~2 %.  Rule: odd
// multiple of 256 bytes, and residues 24..62 are bumped to 65.
inline u64 odd_stride_256(u64 bytes) {
    u64 u = round_up(bytes, 256) / 256;
    u |= 1;
    const u64 r = u % 256;
    if (r >= 24 && r <= 62) u += 65 - r;
    return u * 256;
}

// The event a pipeline waits on for one call's record kernel.  With the profiler armed it is the profiler's own
// stop event (borrowed; alive while g_prof_gen == gen), so that no extra marker packet sits between kernels.
struct DoneRef { hipEvent_t ev = nullptr; u32 gen = 0; bool borrowed = false; };

struct Workspace {  // carve-up of the scratch of one batch call (relative to the 256-byte aligned base)
    u64 off_pre;   // the shared modulus' Barrett constants (recip_kernel), behind the operands
    u64 off_n;     // [batch][L] limbs: every element's modulus, copied by the chain kernel -- what the record writer reads, so
                   // that the caller's n buffer is needed only while the call's own launches run (h2r.h, pipelined form)
    u64 off_state; // [batch][2][L] limbs: the (squared, acc) pair of every element between two segments of a long exponent
    u64 total;     // one [batch * T][4][L] limb array: a, b, q, r of every mul_mod, then off_pre, off_n, off_state, plus alignment slack
};
Workspace workspace_plan(u32 limb_bytes, u32 L, u64 batch, u32 T) {
    Workspace w;
    const u64 arr = round_up(batch * T * (u64)L * limb_bytes, 256);
    w.off_pre = 4 * arr;
    w.off_n = w.off_pre + round_up(4ull * chain_pre_words(128), 256);
    w.off_state = w.off_n + round_up(batch * (u64)L * limb_bytes, 256);
    w.total = w.off_state + round_up(2 * batch * (u64)L * limb_bytes, 256) + 256;
    return w;
}

}  // namespace

struct h2r_ctx {
    h2r_params params;
    h2r_layout layout;
    u32 L, K;           // limbs, 32-bit digits
    U256 word_max;
    u8 *const_rec_dev;  // device copy of the constant record
    u32 *advice_desc_dev = nullptr;   // [h2r_advice_rows] packed row descriptors of the advice image (advice_pack)
    u32 cells_nwv = 1;                // cells_kernel: waves per workgroup (Montgomery cells of the long shapes share one set of planes among several)
    u64 *cells_ktab_dev = nullptr;    // cells_kernel: columns 0, 1, >= 2 of the accumulated_extra constants (CELLS_KT_WORDS words)
    std::vector<u8> const_rec_host;
    // lookup-table row offsets for the multiplicity histogram
    u32 tab0_len, tab1_off, tab1_len, tab2_off, tab2_len, hist_len;
    // RefreshAux::new(w, L, L).increased_limbs_vec (host copy and device copy)
    u8 refresh_inc[2 * 128 + 8]; u32 refresh_nf;
    u64 field_p[4];   // the field modulus (a_b encoding, chip.rs:859)
    FieldConsts fc;   // its Montgomery constants (lookup compression, is_zero's inverse witness)
    h2r_advice_repr repr;      // representation of every field element that crosses the boundary (h2r_ctx_create_ex; default: row-major, canonical)
    MontK mk;                  // the short Montgomery multipliers of the field (H2R_ADVICE_MONTGOMERY)
    MontK *mk_dev = nullptr;
    u32 num_cus, lds_per_cu;   // of the ctx's device
    // the plain (stream-ordered) pow exports overlap chain and record kernels INSIDE a large call through this pipeline
    // (created on first use; calls on one ctx from several threads take turns queueing)
    mutable std::mutex pipe_mu;
    mutable h2r_pipeline *pipe = nullptr;
    // a side stream of the ctx for the short row-program kernels of a whole-element image: they write other rows than the pow rows'
    // kernel and are latency-bound (is_zero's inverse witnesses), so they run NEXT to it instead of in front of it (created on first use)
    mutable std::mutex side_mu;
    mutable hipStream_t side_stream = nullptr;
    mutable hipEvent_t side_fork = nullptr, side_join = nullptr;
    // row programs of the Fresh-op advice images (h2r_rowprog.hpp), built on first use; key = op | assert_one << 8
    struct RowProg { std::vector<RpRow> host; RpRow *dev = nullptr; std::vector<u32> inv_rows; u32 *inv_dev = nullptr; };
    mutable std::mutex prog_mu;
    mutable std::map<u32, RowProg> progs;
    // h2r_advice_check: the per-kind table (selectors, lookup bit lengths) of a (lookup configuration, layout) pair, on the device
    mutable std::mutex check_mu;
    mutable std::map<std::vector<u8>, CheckKind *> check_tabs;
};

namespace {

void put_le(u8 *dst, const U256 &v, u32 nbytes) { std::memcpy(dst, v.v, nbytes); }

// Fill the input-independent planes of is_equal_muled (accumulated_extra chain, chip.rs:869-875).
void build_const_record(h2r_ctx *c) {
    const h2r_layout &lo = c->layout;
    c->const_rec_host.assign(lo.record_stride, 0);
    u8 *r = c->const_rec_host.data();
    const u32 w = lo.limb_width, C = lo.num_cols;
    U256 acc_extra;
    for (u32 i = 0; i < C; ++i) {
        acc_extra = acc_extra + c->word_max;                 // :869-870
        U256 q_acc = acc_extra.shr(w);                       // :871
        u64 mod_acc = acc_extra.low(w);
        U256 nq = q_acc.shl(w);
        put_le(r + lo.plane_off[H2R_PL_ACCX_LO] + (u64)i * 16, acc_extra, 16);
        put_le(r + lo.plane_off[H2R_PL_NQ2_LO] + (u64)i * 16, nq, 16);
        if (lo.plane_elem[H2R_PL_ACCX_HI]) {
            std::memcpy(r + lo.plane_off[H2R_PL_ACCX_HI] + (u64)i * 8, &acc_extra.v[2], 8);
            std::memcpy(r + lo.plane_off[H2R_PL_NQ2_HI] + (u64)i * 8, &nq.v[2], 8);
        }
        put_le(r + lo.plane_off[H2R_PL_QACC] + (u64)i * lo.carry_bytes, q_acc, lo.carry_bytes);
        std::memcpy(r + lo.plane_off[H2R_PL_MODACC] + (u64)i * lo.limb_bytes, &mod_acc, lo.limb_bytes);
        u64 amnq = (acc_extra - nq).low(w);
        std::memcpy(r + lo.plane_off[H2R_PL_AMNQ2] + (u64)i * lo.limb_bytes, &amnq, lo.limb_bytes);
        acc_extra = q_acc;                                   // :875
    }
}

// The template kernel families are instantiated in their own translation units (h2r_internal.hpp): thin ctx-taking wrappers here.
hipError_t launch_trace(const h2r_ctx *c, const TraceArgs &ta, hipStream_t st, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr) {
    return launch_trace_shape(c->layout.limb_width, c->L, c->lds_per_cu, ta, st, ea, eb);
}
// co_running: the call's record kernel of the PREVIOUS batch runs next to this chain kernel (pipeline mode)
hipError_t launch_chain(const h2r_ctx *c, const ChainArgs &ca, bool co_running, hipStream_t st, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr) {
    return launch_chain_shape(c->num_cus, ca, co_running, st, ea, eb);
}

struct ScratchGuard {  // stream-ordered scratch when the caller passes workspace == NULL
    void *p = nullptr; hipStream_t st = nullptr; bool owned = false;
    ~ScratchGuard() { if (owned && p) (void)hipFreeAsync(p, st); }
};

void fill_trace_args(const h2r_ctx *c, TraceArgs &ta) {
    std::memset(&ta, 0, sizeof ta);
    const h2r_layout &lo = c->layout;
    for (int p = 0; p < H2R_PL_COUNT; ++p) ta.off[p] = lo.plane_off[p];
    ta.wm[0] = c->word_max.v[0]; ta.wm[1] = c->word_max.v[1]; ta.wm[2] = c->word_max.v[2];
    ta.carry_bits = lo.carry_bits; ta.carry_sub_bits = lo.carry_sub_bits;
    ta.carry_nsub = lo.carry_nsub; ta.carry_sub_stride = lo.carry_sub_stride;
    ta.record_stride = lo.record_stride;
    ta.acc_spg = lo.acc_steps_per_group; ta.acc_lo_row = lo.acc_lo_row_bytes; ta.acc_lo_group = lo.acc_lo_group_bytes; ta.acc_hi_group = lo.acc_hi_group_bytes;
    ta.const_rec = c->const_rec_dev;
    ta.ablate = (u32)knobs().ablate;
    if (knobs().trace_dyn_lds >= 0) ta.dyn_lds = (u32)knobs().trace_dyn_lds;
    if (knobs().trace_prio >= 0) ta.prio = (u32)knobs().trace_prio;
}

// run_path(..., args_only): fill the two kernels' arguments in and launch nothing (the pipeline issues them itself)
struct PathArgs { ChainArgs ca; TraceArgs ta; bool has_trace = false; };
// A long exponent walked as segments of its bits (ChainArgs::state): bits [bit_lo, bit_hi) = the mul_mods [t_lo, t_lo + t_cnt) of every element
struct ExpSegment { u32 bit_lo, bit_hi, t_lo, t_cnt; };
// Common driver: chain kernel (q, r of every mul_mod) then trace kernel (the witness records).
int32_t run_path(const h2r_ctx *c, u32 mode, const void *a, const void *b, const void *n, const void *e_limbs,
                 u32 e_num_limbs, u32 exp_limb_bits, const ExpBits *eb, u32 check_in_field, u64 batch, u32 flags,
                 u32 T, void *trace, u64 elem_stride, u64 off_records, const h2r_pow_layout *pl, void *out,
                 uint8_t *status, void *workspace, hipStream_t st, hipStream_t trace_st = nullptr,
                 hipEvent_t chain_done = nullptr, hipEvent_t trace_done = nullptr, DoneRef *done_ref = nullptr,
                 void *shared_pre = nullptr, PathArgs *args_only = nullptr, void *n_copy_at = nullptr, const ExpSegment *seg = nullptr) {
    // shared_pre / n_copy_at: where the shared modulus' Barrett constants / the elements' moduli go when `workspace` is a
    // slice of a larger call's plan
    // trace_st != nullptr (pipeline mode): the record-writing kernel runs on trace_st after `chain_done`
    if (!c || !n || !a || !status) return H2R_E_NULL;
    if (trace_st && !workspace) return H2R_E_NULL;
    if (c->params.device < 0) return H2R_E_UNSUPPORTED;  // host-only context
    if (batch == 0) return H2R_OK;
    if (batch * (u64)(T ? T : 1) >= (1ull << 32)) return H2R_E_UNSUPPORTED;  // item index is 32-bit in the kernels
    if (T == 0) {  // e == 0: no mul_mod at all; result is the constant 1 (chip.rs:729)
        // handled by the chain kernel (loop of zero bits); still need a dummy ops buffer
    }
    H2R_ON_DEVICE(c->params.device);
    const h2r_layout &lo = c->layout;
    const Workspace wp = workspace_plan(lo.limb_bytes, c->L, batch, T ? T : 1);
    ScratchGuard sg; sg.st = st;
    u8 *ws = static_cast<u8 *>(workspace);
    if (!ws) {
        HIP_TRY(hipMallocAsync(&sg.p, wp.total, st));
        sg.owned = true; ws = static_cast<u8 *>(sg.p);
    }
    ws = reinterpret_cast<u8 *>(round_up(reinterpret_cast<u64>(ws), 256));
    // off_records = UINT64_MAX: a witness-only element (h2r_pow_layout_compact) -- the chain kernel leaves what it writes into a trace
    // (a Var element's exponent bits and selected operands, the result), no record kernel follows
    const bool records = trace && T && off_records != UINT64_MAX;
    ChainArgs ca;
    std::memset(&ca, 0, sizeof ca);
    ca.a = static_cast<const u32 *>(a); ca.b = static_cast<const u32 *>(b); ca.n = static_cast<const u32 *>(n);
    ca.e_limbs = static_cast<const u32 *>(e_limbs);
    ca.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : c->K;
    ca.batch = batch; ca.kreal = c->K; ca.mode = mode; ca.T = T ? T : 1;
    ca.e_num_limbs = e_num_limbs; ca.exp_limb_bits = exp_limb_bits; ca.digits_per_limb = lo.limb_width / 32;
    ca.check_in_field = check_in_field;
    ca.ops = reinterpret_cast<u32 *>(ws);
    ca.out = static_cast<u32 *>(out); ca.status = status;
    if (pl && trace) {
        ca.trace = static_cast<u8 *>(trace); ca.elem_stride = elem_stride;
        ca.off_e_bits = pl->off_e_bits; ca.off_selected = pl->off_selected; ca.selected_stride = pl->selected_stride;
        ca.off_result = pl->off_result; ca.write_result_to_trace = 1;
        if (mode != CHAIN_POW_VAR) { ca.off_e_bits = 0; ca.off_selected = 0; }
    }
    if (eb) ca.e = *eb;
    if (seg) { ca.state = reinterpret_cast<u32 *>(ws + wp.off_state); ca.bit_lo = seg->bit_lo; ca.bit_hi = seg->bit_hi; ca.t_base = seg->t_lo; }
    u8 *n_copy = records ? (n_copy_at ? static_cast<u8 *>(n_copy_at) : ws + wp.off_n) : nullptr;
    ca.n_copy = reinterpret_cast<u32 *>(n_copy);
    // 128-digit chains (RSA-4096 at 64-bit limbs) are the longer leg next to their record kernel: their waves get issue
    // priority there (1.00 -> 1.05 M assigns/s; no effect measured for the shorter chains)
    if (trace_st && c->K > 96 && lo.limb_width == 64) ca.prio = 1;   // (the 32-bit-limb 4096-bit shape is record-bound)
    // [r6] ... and so are RSA-1024's one-wave chains in the two-queue form's smallest calls (1,280 .. 1,535 per call: 14.4-14.5 -> 14.7-14.8 M assigns/s;
    // neutral from 1,536: profiles/r06_two_queue_rsa1024.txt, 6.)
    if (trace_st && c->L == 16 && c->K == 32 && lo.limb_width == 64 && batch >= 1280 && batch < 1536) ca.prio = 1;
    if (knobs().chain_prio >= 0) ca.prio = (u32)knobs().chain_prio;
    // one key, many elements: the Barrett constants of the shared modulus are computed once (recip_kernel) instead of by
    // every element's workgroup (big_integer/chip.rs:562-567 divides by the same n every time)
    // ... except in pipeline mode for the shapes whose chain kernel hides behind the record kernel anyway: the one-workgroup
    // recip_kernel in front of the chain kernel makes the chain kernel start 17 us AFTER the record kernel it shares the CUs
    // with instead of together with it, and the record kernel -- the longer leg -- then runs 0.195 -> 0.209 ms (same-box A/B,
    // bench.py --shared-modulus: 4.49 -> 4.8 M assigns/s without the precomputation).
    const bool chain_hidden = trace_st && records && (lo.limb_width == 32 || c->L <= 32);
    if ((flags & H2R_F_SHARED_MODULUS) && batch > 1 && !chain_hidden)
        ca.pre = reinterpret_cast<const u32 *>(shared_pre ? static_cast<u8 *>(shared_pre) : ws + wp.off_pre);
    // Pipeline mode: the record stream must wait for this chain kernel.  The event it waits on is the dispatch's own
    // stop event (the profiler's when armed, else chain_done) -- no separate marker packet.
    hipEvent_t chain_wait = nullptr;
#ifdef H2R_CHAIN_TIMING   // developer build (tools/chain_timing.py): dump block 0's s_memtime stamps
    static u64 *dbg_buf = nullptr;
    if (knobs().chain_timing) { if (!dbg_buf) (void)hipMalloc(&dbg_buf, 4096 * 8); (void)hipMemsetAsync(dbg_buf, 0, 4096 * 8, st); ca.dbg_time = dbg_buf; }
#endif
    if (args_only) args_only->ca = ca;
    else {
        ProfScope ps(H2R_KERNEL_CHAIN, st, true);
        const bool piped = trace_st && records;
        chain_wait = ps.on ? ps.b : (piped ? chain_done : nullptr);
        if (c->K > 128) return H2R_E_UNSUPPORTED;
        HIP_TRY(launch_chain(c, ca, trace_st != nullptr, st, ps.a, chain_wait));
    }
#ifdef H2R_CHAIN_TIMING
    if (ca.dbg_time) {
        static u64 host[4096];
        (void)hipStreamSynchronize(st); (void)hipMemcpy(host, ca.dbg_time, sizeof host, hipMemcpyDeviceToHost);
        FILE *f = std::fopen("/tmp/h2r_chain_timing.txt", "w");
        if (f) { for (int i = 0; i < 4000 && host[i]; ++i) std::fprintf(f, "%llu\n", (unsigned long long)host[i]); std::fclose(f); }
    }
#endif
    if (records) {
        TraceArgs ta;
        fill_trace_args(c, ta);
        const u64 lb = lo.limb_width / 8;   // the four values of an item are L limbs apart
        ta.opA = ws; ta.opB = ws + c->L * lb; ta.opQ = ws + 2 * c->L * lb; ta.opR = ws + 3 * c->L * lb; ta.op_stride = 4ull * c->L;
        ta.n = n_copy; ta.n_stride = c->L;   // the chain kernel's copy: the caller's n is read inside the call only
        ta.status = status; ta.n_items = batch * T; ta.T = T;
        if (seg) { ta.n_items = batch * seg->t_cnt; ta.T = seg->t_cnt; ta.t_lo = seg->t_lo; ta.T_ops = T; }   // this segment's mul_mods of every element
        ta.trace = static_cast<u8 *>(trace); ta.elem_stride = elem_stride; ta.off_records = off_records;
        if (args_only) { args_only->ta = ta; args_only->has_trace = true; return H2R_OK; }
        hipStream_t ts = st;
        // LDS share of the record kernel's workgroups (the occupancy lever on this hardware: an LDS request the kernel never
        // touches).  ALONE the 64-bit-limb shapes up to RSA-2048 write fastest with FEW concurrent store streams: one
        // workgroup (4 records in flight) per CU -- 0.227 -> 0.213 ms, 5.87 TB/s for RSA-2048 (ta.residency = 1: the request
        // is derived from the device's LDS size and the kernel's static LDS).  NEXT TO the following batch's chain kernel the
        // request decides two things at once -- how many record workgroups share a CU and how much LDS is left for chain
        // workgroups -- so it is a measured per-shape share of the CU's LDS, not a workgroup count
        // (profiles/r02_residency_sweep.txt, profiles/history/r01_pipeline_sweep.txt; MI355X, 160 KB per CU).
        const bool tune = knobs().trace_dyn_lds < 0;
        if (tune && lo.limb_width == 64 && c->L <= 32) ta.residency = 1;
        if (trace_st) {
            if (tune) {
                ta.residency = 0;
                u32 share_256;   // request = share_256 / 256 of the CU's LDS
                if (lo.limb_width == 32) share_256 = 50;                               // 32,000 B of 160 KB: three workgroups per CU
                else if (c->L == 32 && batch > 512) share_256 = 50;                    // RSA-2048 at throughput batches: three
                else if (c->L > 48) share_256 = 31;                                    // 4096-bit: 20,000 B (1.15 -> 1.01 ms per 1,024)
                else if (c->L > 32) share_256 = 62;                                    // RSA-3072: 40,000 B (0.82 -> 0.71 ms)
                else share_256 = 70;                                                   // 45,000 B: two workgroups, room for the chain's
                ta.dyn_lds = (u32)(((u64)c->lds_per_cu * share_256 / 256) & ~15ull);
            }
            // This cross-queue wait puts a barrier packet between two record kernels: 5 us of the 10-11 us between them
            // (tools/boundary_probe2.hip).  Measured alternatives, none shipped (profiles/r02_gap_experiments.txt): a gate
            // kernel polling a count the chain blocks publish (their agent-scope releases disturb the record kernel's store
            // stream, and a one-wave kernel costs as much as the barrier packet); the record kernel's own workgroups polling
            // a number the chain stream publishes (5-7 us per step faster, but a record kernel that starts early HOLDS the
            // LDS that the chain kernel -- or any kernel the caller queued in front of it -- needs to get onto the CUs).
            HIP_TRY(hipStreamWaitEvent(trace_st, chain_wait, 0));
            ts = trace_st;
        }
        ProfScope ps(H2R_KERNEL_TRACE, ts, true);
        HIP_TRY(launch_trace(c, ta, ts, ps.a, ps.on ? ps.b : trace_done));
        if (done_ref) {
            if (ps.on) { std::lock_guard<std::mutex> lk(g_prof_mu); *done_ref = DoneRef{ps.b, g_prof_gen, true}; }
            else *done_ref = DoneRef{trace_done, 0, false};
        }
    } else if (done_ref) {
        *done_ref = DoneRef{};
    }
    return H2R_OK;
}

int32_t exp_to_bits(const uint8_t *e_le, size_t e_len, ExpBits *eb, u32 *T) {
    if (!e_le && e_len) return H2R_E_NULL;
    std::memset(eb, 0, sizeof *eb);
    eb->nbits = exp_num_bits(e_le, e_len);
    if (eb->nbits > 8 * sizeof eb->bytes) return H2R_E_UNSUPPORTED;
    std::memcpy(eb->bytes, e_le, (eb->nbits + 7) / 8);
    u32 t = 0;
    for (u32 i = 0; i < eb->nbits; ++i) t += 1 + exp_bit(e_le, i);  // one square per bit, one mul per set bit
    *T = t;
    return H2R_OK;
}

// Walk the sections of a Fresh-op region (AuxGeom) in stream order: emit(device_offset, length).
// Sections start on 16-byte boundaries in the device buffer; the flat stream is their concatenation.
template <typename F>
void fresh_sections(const AuxGeom &g, u32 op, F &&emit) {
    u64 off = 0;
    const u32 L = g.L;
    auto add = [&](u32 n) { emit(off, (u64)n * g.STEP); off += g.add_sz(n); };
    auto eq = [&](u32 n) { emit(off, 2ull * n); off += g.eq_sz(n); };
    auto subu = [&](u32 n1) { emit(off, (u64)n1 * g.RA); off += g.cl_sz(n1); add(n1); eq(n1 + 1); };
    auto sub = [&](u32 nA, u32 nB) {
        const u32 m = nA > nB ? nA : nB, n1 = m + 1;
        add(m); subu(n1);
        emit(off, 2); off += 16;
        emit(off, (u64)n1 * g.LB); off += AuxGeom::a16((u64)n1 * g.LB);
        emit(off, (u64)m * g.LB); off += AuxGeom::a16((u64)m * g.LB);
        subu(n1);
    };
    auto lt = [&]() { sub(L, L); eq(L); emit(off, 2); off += 16; };
    switch (op) {
        case FRESH_ADD: add(L); break;
        case FRESH_SUB: case FRESH_IS_LESS_THAN_OR_EQUAL: sub(L, L); break;
        case FRESH_ADD_MOD: add(L); sub(L + 1, L); emit(off, (u64)(L + 2) * g.LB); break;
        case FRESH_SUB_MOD: sub(L, L); sub(L, L + 1); emit(off, (u64)(L + 2) * g.LB); break;
        case FRESH_IS_ZERO: case FRESH_IS_EQUAL_FRESH: eq(L); break;
        case FRESH_IS_LESS_THAN: case FRESH_IS_IN_FIELD: lt(); break;
        case FRESH_IS_GREATER_THAN: sub(L, L); emit(off, 1); break;
        case FRESH_IS_GREATER_THAN_OR_EQUAL: lt(); emit(off, 1); break;
        default: break;
    }
}
template <typename F>
void in_field_sections(const AuxGeom &g, F &&emit) { fresh_sections(g, FRESH_IS_IN_FIELD, emit); }

}  // namespace

extern "C" {

uint32_t h2r_abi_version(void) { return H2R_VERSION; }

int32_t h2r_ctx_create(const h2r_params *params, h2r_ctx **out) try { return h2r_ctx_create_ex(params, nullptr, out); } H2R_CATCH_STATUS

int32_t h2r_ctx_advice_repr(const h2r_ctx *ctx, h2r_advice_repr *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    *out = ctx->repr;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_ctx_create_ex(const h2r_params *params, const h2r_advice_repr *repr, h2r_ctx **out) try {
    if (!params || !out) return H2R_E_NULL;
    *out = nullptr;
    if (repr) {
        if (repr->struct_size != sizeof(h2r_advice_repr)) return H2R_E_UNSUPPORTED;   // a caller built against another header
        if ((repr->flags & ~(H2R_ADVICE_COLUMNS | H2R_ADVICE_MONTGOMERY)) || (repr->col_stride & 15)) return H2R_E_SHAPE;
        if (repr->col_stride && !(repr->flags & H2R_ADVICE_COLUMNS)) return H2R_E_SHAPE;
    }
    const u32 w = params->limb_width;
    if (w == 0 || params->bits_len == 0 || params->bits_len % w != 0) return H2R_E_SHAPE;  // chip.rs:1175
    const u32 L = params->bits_len / w;
    if (w != 32 && w != 64) return H2R_E_UNSUPPORTED;
    const u32 fbits = field_num_bits(params->field);
    if (!fbits) return H2R_E_SHAPE;
    if (L > 4096) return H2R_E_UNSUPPORTED;
    U256 wm = compute_mul_word_max(w, L);
    if (wm.bits() > fbits) return H2R_E_FIELD_TOO_SMALL;  // chip.rs:1178
    if (!shape_supported(w, L)) return H2R_E_UNSUPPORTED;
    h2r_ctx *c = new (std::nothrow) h2r_ctx();
    if (!c) return H2R_E_NOMEM;
    std::unique_ptr<h2r_ctx, void (*)(h2r_ctx *)> guard(c, h2r_ctx_destroy);   // (released on success: an exception below must not leak the ctx)
    c->params = *params; c->L = L; c->K = params->bits_len / 32; c->word_max = wm; c->const_rec_dev = nullptr;
    c->num_cus = 256; c->lds_per_cu = 160 * 1024;
    std::memset(c->refresh_inc, 0, sizeof c->refresh_inc);
    c->refresh_nf = (L <= 128) ? refresh_aux_increased_limbs(w, L, L, c->refresh_inc) : 0;
    layout_compute(w, L, &c->layout);
    field_modulus(params->field, c->field_p);
    field_consts_init(c->field_p, &c->fc);
    montk_init(c->field_p, &c->mk);
    c->repr.struct_size = sizeof(h2r_advice_repr); c->repr.flags = repr ? repr->flags : 0u; c->repr.col_stride = repr ? repr->col_stride : 0;
    build_const_record(c);
    // histogram rows: composition table of the limb sub-limbs, then of the carry sub-limbs when its width
    // differs, then the carry overflow table
    const h2r_layout &lo = c->layout;
    c->tab0_len = 1u << lo.limb_sub_bits;
    if (lo.carry_sub_bits == lo.limb_sub_bits) { c->tab1_off = 0; c->tab1_len = 0; }
    else { c->tab1_off = c->tab0_len; c->tab1_len = 1u << lo.carry_sub_bits; }
    const u32 ovb = lo.carry_bits % lo.carry_sub_bits;
    c->tab2_off = c->tab0_len + c->tab1_len; c->tab2_len = ovb ? (1u << ovb) : 0;
    c->hist_len = c->tab2_off + c->tab2_len;
    if (params->device < 0) {  // host-only context: layouts, flatten and parameter queries; no device work
        *out = guard.release();
        return H2R_OK;
    }
    DeviceGuard dg(params->device);
    {
        int v = 0;
        c->num_cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, params->device) == hipSuccess && v > 0) ? (u32)v : 256u;
        c->lds_per_cu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, params->device) == hipSuccess && v > 0) ? (u32)v : 160u * 1024;
        (void)hipGetLastError();
    }
    if (!hip_ok(dg.err, "hipSetDevice") ||
        !hip_ok(hipMalloc(reinterpret_cast<void **>(&c->const_rec_dev), lo.record_stride), "hipMalloc(const record)") ||
        !hip_ok(hipMemcpy(c->const_rec_dev, c->const_rec_host.data(), lo.record_stride, hipMemcpyHostToDevice), "hipMemcpy(const record)")) {
        return H2R_E_HIP;
    }
    {   // the advice image's row table
        const u32 rows = advice_rows_per_record(L, lo.carry_nsub), nrc = (lo.carry_nsub + 3) / 4;
        std::vector<u32> desc(rows);
        for (u32 r = 0; r < rows; ++r) desc[r] = advice_pack(advice_decode(r, L, nrc));
        if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&c->advice_desc_dev), rows * sizeof(u32)), "hipMalloc(advice rows)") ||
            !hip_ok(hipMemcpy(c->advice_desc_dev, desc.data(), rows * sizeof(u32), hipMemcpyHostToDevice), "hipMemcpy(advice rows)")) {
            return H2R_E_HIP;
        }
    }
    {   // cells_kernel's table of the input-independent accumulated_extra chain (chip.rs:869-875): X[0] = 0, X[i+1] = (X[i] + W) >> w
        // is at its fixed point L 2^w - L from i = 2 on (W = (2^w - 1)(L 2^w - L + 1)), so three columns describe every column
        const u8 *r = c->const_rec_host.data();
        auto col = [&](u32 i, u64 (&e)[10]) {
            std::memset(e, 0, sizeof e);
            std::memcpy(&e[0], r + lo.plane_off[H2R_PL_ACCX_LO] + (u64)i * 16, 16);
            std::memcpy(&e[6], r + lo.plane_off[H2R_PL_NQ2_LO] + (u64)i * 16, 16);
            if (lo.plane_elem[H2R_PL_ACCX_HI]) {
                std::memcpy(&e[2], r + lo.plane_off[H2R_PL_ACCX_HI] + (u64)i * 8, 8);
                std::memcpy(&e[8], r + lo.plane_off[H2R_PL_NQ2_HI] + (u64)i * 8, 8);
            }
            std::memcpy(&e[3], r + lo.plane_off[H2R_PL_QACC] + (u64)i * lo.carry_bytes, lo.carry_bytes);
            std::memcpy(&e[5], r + lo.plane_off[H2R_PL_MODACC] + (u64)i * lo.limb_bytes, lo.limb_bytes);
            std::memcpy(&e[9], r + lo.plane_off[H2R_PL_AMNQ2] + (u64)i * lo.limb_bytes, lo.limb_bytes);
        };
        u64 kt[CELLS_TAB_WORDS] = {0}, e[10], e2[10];
        for (u32 i = 0; i < 3; ++i) { col(i, e); std::memcpy(&kt[10 * i], e, sizeof e); }
        col(2, e2);
        bool fixed_point = true;
        for (u32 i = 3; i < lo.num_cols; ++i) { col(i, e); fixed_point = fixed_point && std::memcmp(e, e2, sizeof e) == 0; }
        if (!fixed_point) return H2R_E_UNSUPPORTED;
        for (int k = 0; k < 3; ++k) kt[CELLS_KT_WM + k] = c->word_max.v[k];
        for (int k = 0; k < 4; ++k) kt[CELLS_KT_P + k] = c->fc.p[k];
        std::memcpy(&kt[CELLS_KT_FC], &c->fc, sizeof c->fc);
        {   // the column rows' fast-path sources, packed against this shape's LDS plan
            const bool mont = (c->repr.flags & H2R_ADVICE_MONTGOMERY) != 0;
            // Montgomery cells of the long shapes (64 limbs and more): EIGHT waves share one set of planes (h2r_cells.hpp); measured per shape,
            // TB/s with 1 / 4 / 8 waves: 128 x 32-bit 3.4 / 5.3 / 6.4, 96 x 32-bit 3.2 / 4.8 / 5.4, 64 x 32-bit 4.2 / 4.3 / 5.2, 64 x 64-bit 3.8 / 3.6 / 4.4,
            // 48 x 64-bit 3.9 / 3.1 / 3.5, 32 x 64-bit 5.2 / 2.6 / 2.9 (tools/cells_nwv_probe.py, profiles/r05_cells_representations.txt)
            const u32 nwv_env = (knobs().cells_nwv == 1 || knobs().cells_nwv == 8) ? (u32)knobs().cells_nwv : 0u;   // (developer A/B)
            c->cells_nwv = !mont ? 1u : (nwv_env ? nwv_env : (L >= 64 ? 8u : 1u));
            // (a device whose CU has less LDS than the eight-wave plan asks for runs the one-wave plan instead of failing at the first launch)
            if (c->cells_nwv > 1 && cells_lds_plan(w, L, mont, c->cells_nwv).total > c->lds_per_cu) c->cells_nwv = 1;
            const CellsLds lp = cells_lds_plan(w, L, mont, c->cells_nwv);
            u32 *fs = reinterpret_cast<u32 *>(&kt[CELLS_KT_FSRC]);
            for (u32 k = 0; k < ADVICE_COL_ROWS * 3; ++k) {
                fs[k] = cells_pack_fast_src(lp, w, cells_fast_src(k / 3, k % 3, false), mont);
                fs[CELLS_SRC_WORDS + k] = cells_pack_fast_src(lp, w, cells_fast_src(k / 3, k % 3, true), mont);
            }
        }
        if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&c->mk_dev), sizeof(MontK)), "hipMalloc(Montgomery table)") ||
            !hip_ok(hipMemcpy(c->mk_dev, &c->mk, sizeof(MontK), hipMemcpyHostToDevice), "hipMemcpy(Montgomery table)")) {
            return H2R_E_HIP;
        }
        if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&c->cells_ktab_dev), sizeof kt), "hipMalloc(cells table)") ||
            !hip_ok(hipMemcpy(c->cells_ktab_dev, kt, sizeof kt, hipMemcpyHostToDevice), "hipMemcpy(cells table)")) {
            return H2R_E_HIP;
        }
    }
    *out = guard.release();
    return H2R_OK;
} H2R_CATCH_STATUS

void h2r_ctx_destroy(h2r_ctx *ctx) try {
    if (!ctx) return;
    if (ctx->params.device >= 0) {
        DeviceGuard dg(ctx->params.device);
        if (ctx->const_rec_dev) (void)hipFree(ctx->const_rec_dev);
        if (ctx->advice_desc_dev) (void)hipFree(ctx->advice_desc_dev);
        if (ctx->cells_ktab_dev) (void)hipFree(ctx->cells_ktab_dev);
        if (ctx->mk_dev) (void)hipFree(ctx->mk_dev);
        for (auto &kv : ctx->check_tabs) if (kv.second) (void)hipFree(kv.second);
        for (auto &kv : ctx->progs) { if (kv.second.dev) (void)hipFree(kv.second.dev); if (kv.second.inv_dev) (void)hipFree(kv.second.inv_dev); }
        if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
        if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
        if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
        if (ctx->pipe) h2r_pipeline_destroy(ctx->pipe);
    }
    delete ctx;
} H2R_CATCH_VOID

int32_t h2r_compute_range_lens(uint32_t limb_width, uint32_t num_limbs, uint32_t comp[3], uint32_t over[3]) try {
    if (!comp || !over) return H2R_E_NULL;
    if (limb_width < kNumLookupLimbs || limb_width > 64 || num_limbs == 0) return H2R_E_SHAPE;
    compute_range_lens(limb_width, num_limbs, comp, over);
    return H2R_OK;
} H2R_CATCH_STATUS
int32_t h2r_rsa_compute_range_lens(uint32_t num_limbs, uint32_t comp[4], uint32_t over[3]) try {
    if (!comp || !over) return H2R_E_NULL;
    if (num_limbs == 0) return H2R_E_SHAPE;
    compute_range_lens(64, num_limbs, comp, over);  // src/chip.rs:250-251, LIMB_WIDTH = 64
    comp[3] = 32 / kNumLookupLimbs;                 // src/chip.rs:252
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_trace_layout(const h2r_ctx *ctx, h2r_layout *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    *out = ctx->layout;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_pow_fixed_layout(const h2r_ctx *ctx, const uint8_t *e_le, size_t e_len, h2r_pow_layout *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    ExpBits eb; u32 T;
    int32_t rc = exp_to_bits(e_le, e_len, &eb, &T);
    if (rc) return rc;
    const h2r_layout &lo = ctx->layout;
    std::memset(out, 0, sizeof *out);
    out->num_mul_mods = T; out->num_exp_bits = eb.nbits;
    out->off_records = 0;
    out->off_result = (u64)T * lo.record_stride;
    out->off_e_bits = UINT64_MAX; out->off_selected = UINT64_MAX; out->selected_stride = 0;
    out->elem_stride = odd_stride_256(out->off_result + (u64)lo.num_limbs * lo.limb_bytes);
    out->stream_bytes = (u64)T * lo.stream_bytes + (u64)lo.num_limbs * lo.limb_bytes;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_pow_var_layout(const h2r_ctx *ctx, uint32_t e_num_limbs, uint32_t exp_limb_bits, h2r_pow_layout *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    const h2r_layout &lo = ctx->layout;
    if (e_num_limbs == 0 || exp_limb_bits == 0 || exp_limb_bits > lo.limb_width) return H2R_E_SHAPE;
    const u64 nbits = (u64)e_num_limbs * exp_limb_bits;
    if (nbits > (1u << 20)) return H2R_E_UNSUPPORTED;
    std::memset(out, 0, sizeof *out);
    out->num_mul_mods = (u32)(2 * nbits); out->num_exp_bits = (u32)nbits;
    out->exp_limb_bits = exp_limb_bits; out->e_num_limbs = e_num_limbs;
    out->off_records = 0;
    const u64 limbs_bytes = round_up((u64)lo.num_limbs * lo.limb_bytes, 256);
    out->off_selected = 2 * nbits * lo.record_stride;
    out->selected_stride = limbs_bytes;
    out->off_result = out->off_selected + nbits * limbs_bytes;
    out->off_e_bits = out->off_result + limbs_bytes;
    out->elem_stride = odd_stride_256(out->off_e_bits + nbits);
    out->stream_bytes = nbits + nbits * (2 * lo.stream_bytes + (u64)lo.num_limbs * lo.limb_bytes) + (u64)lo.num_limbs * lo.limb_bytes;
    return H2R_OK;
} H2R_CATCH_STATUS

uint64_t h2r_workspace_bytes(const h2r_ctx *ctx, uint64_t batch, uint32_t num_mul_mods) try {
    if (!ctx) return 0;
    return workspace_plan(ctx->layout.limb_bytes, ctx->L, batch, num_mul_mods ? num_mul_mods : 1).total;
} H2R_CATCH_ZERO

int32_t h2r_mul_mod_batch(const h2r_ctx *ctx, const void *a, const void *b, const void *n, uint64_t batch,
                          uint32_t flags, void *trace, void *r_out, uint8_t *status, void *workspace,
                          h2r_stream_t stream) try {
    if (!ctx || !b) return H2R_E_NULL;
    return run_path(ctx, CHAIN_MULMOD, a, b, n, nullptr, 0, 0, nullptr, 0, batch, flags, 1, trace,
                    ctx->layout.record_stride, 0, nullptr, r_out, status, workspace, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

int32_t h2r_square_mod_batch(const h2r_ctx *ctx, const void *a, const void *n, uint64_t batch, uint32_t flags,
                             void *trace, void *r_out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    return h2r_mul_mod_batch(ctx, a, a, n, batch, flags, trace, r_out, status, workspace, stream);  // chip.rs:648
} H2R_CATCH_STATUS

namespace {
int32_t launch_verify_aux(const h2r_ctx *ctx, const void *sig, const void *n, const uint64_t *hashed, uint64_t batch, uint32_t flags,
                          void *trace, const h2r_verify_layout &vl, void *powed_out, uint8_t *is_valid_out, uint8_t *status, hipStream_t st);
AuxArgs verify_aux_args(const h2r_ctx *ctx, const void *sig, const void *n, const uint64_t *hashed, uint64_t batch, uint32_t flags,
                        void *trace, const h2r_verify_layout &vl, void *powed_out, uint8_t *is_valid_out, uint8_t *status);
int32_t launch_in_field(const h2r_ctx *ctx, const void *x, const void *n, uint64_t batch, uint32_t flags, void *in_field_trace, hipStream_t st);
int32_t in_field_args(const h2r_ctx *ctx, const void *x, const void *n, uint64_t batch, uint32_t flags, void *in_field_trace, AuxArgs *aa, u32 *lds);
}

namespace {
bool plain_call_overlaps(const h2r_ctx *c, u64 batch);
u32 exp_segment_count(const h2r_ctx *c, u64 batch, u32 nbits, bool has_trace, bool single_call = false);
int32_t overlapped_pow_fixed(const h2r_ctx *ctx, const void *x, const void *n, const uint8_t *e_le, size_t e_len, uint64_t batch,
                             uint32_t flags, void *trace, const h2r_pow_layout &pl, uint64_t elem_stride, void *out, uint8_t *status,
                             void *workspace, hipStream_t st, u32 check_in_field, u32 T,
                             const void *e_limbs = nullptr, u32 e_num_limbs = 0, u32 exp_limb_bits = 0);
}

static int32_t pow_fixed_impl(const h2r_ctx *ctx, const void *x, const void *n, const uint8_t *e_le, size_t e_len,
                              uint64_t batch, uint32_t flags, void *trace, void *out, uint8_t *status,
                              void *workspace, h2r_stream_t stream, u32 check_in_field) {
    if (!ctx) return H2R_E_NULL;
    ExpBits eb; u32 T;
    int32_t rc = exp_to_bits(e_le, e_len, &eb, &T);
    if (rc) return rc;
    h2r_pow_layout pl;
    rc = h2r_pow_fixed_layout(ctx, e_le, e_len, &pl);
    if (rc) return rc;
    // a large call with a trace: sub-batches whose chain kernels run next to the previous sub-batch's record kernel (a side
    // stream of the ctx), joined back onto the caller's stream before returning -- stream-ordered as ever for the caller
    if (trace && T && x && n && status && ctx->params.device >= 0 && (plain_call_overlaps(ctx, batch) || exp_segment_count(ctx, batch, eb.nbits, true, true) > 1))
        return overlapped_pow_fixed(ctx, x, n, e_le, e_len, batch, flags, trace, pl, pl.elem_stride, out, status, workspace,
                                    static_cast<hipStream_t>(stream), check_in_field, T);
    return run_path(ctx, CHAIN_POW_FIXED, x, nullptr, n, nullptr, 0, 0, &eb, check_in_field, batch, flags, T, trace,
                    pl.elem_stride, pl.off_records, &pl, out, status, workspace, static_cast<hipStream_t>(stream));
}

static int32_t pow_var_impl(const h2r_ctx *ctx, const void *x, const void *e_limbs, uint32_t e_num_limbs, uint32_t exp_limb_bits,
                            const void *n, uint64_t batch, uint32_t flags, void *trace, void *out, uint8_t *status,
                            void *workspace, h2r_stream_t stream, u32 check_in_field) {
    if (!ctx || !e_limbs) return H2R_E_NULL;
    h2r_pow_layout pl;
    int32_t rc = h2r_pow_var_layout(ctx, e_num_limbs, exp_limb_bits, &pl);
    if (rc) return rc;
    // a long exponent on a latency-bound batch: walked as segments of its bits, each segment's records next to the next one's chains
    if (trace && pl.num_mul_mods && x && n && status && ctx->params.device >= 0 && exp_segment_count(ctx, batch, e_num_limbs * exp_limb_bits, true) > 1)
        return overlapped_pow_fixed(ctx, x, n, nullptr, 0, batch, flags, trace, pl, pl.elem_stride, out, status, workspace,
                                    static_cast<hipStream_t>(stream), check_in_field, pl.num_mul_mods, e_limbs, e_num_limbs, exp_limb_bits);
    return run_path(ctx, CHAIN_POW_VAR, x, nullptr, n, e_limbs, e_num_limbs, exp_limb_bits, nullptr, check_in_field, batch, flags,
                    pl.num_mul_mods, trace, pl.elem_stride, pl.off_records, &pl, out, status, workspace,
                    static_cast<hipStream_t>(stream));
}

int32_t h2r_pow_mod_fixed_exp_batch(const h2r_ctx *ctx, const void *x, const void *n, const uint8_t *e_le,
                                    size_t e_len, uint64_t batch, uint32_t flags, void *trace, void *out,
                                    uint8_t *status, void *workspace, h2r_stream_t stream) try {
    return pow_fixed_impl(ctx, x, n, e_le, e_len, batch, flags, trace, out, status, workspace, stream, 0);
} H2R_CATCH_STATUS

int32_t h2r_pow_mod_batch(const h2r_ctx *ctx, const void *x, const void *e_limbs, uint32_t e_num_limbs,
                          uint32_t exp_limb_bits, const void *n, uint64_t batch, uint32_t flags, void *trace,
                          void *out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    return pow_var_impl(ctx, x, e_limbs, e_num_limbs, exp_limb_bits, n, batch, flags, trace, out, status, workspace, stream, 0);
} H2R_CATCH_STATUS

// RSAChip::modpow_public_key (src/chip.rs:99-114): assert_in_field witness (:106), then the pow path with the in-field
// predicate folded into the chain kernel's status.
int32_t h2r_modpow_public_key_batch(const h2r_ctx *ctx, const void *x, const void *n, const uint8_t *e_le,
                                    size_t e_len, uint64_t batch, uint32_t flags, void *trace, void *in_field_trace,
                                    void *out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    const int32_t rc = pow_fixed_impl(ctx, x, n, e_le, e_len, batch, flags, trace, out, status, workspace, stream, 1);
    if (rc || !in_field_trace) return rc;
    return launch_in_field(ctx, x, n, batch, flags, in_field_trace, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

int32_t h2r_modpow_public_key_var_batch(const h2r_ctx *ctx, const void *x, const void *e_limbs, uint32_t e_num_limbs,
                                        uint32_t exp_limb_bits, const void *n, uint64_t batch, uint32_t flags, void *trace,
                                        void *in_field_trace, void *out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    const int32_t rc = pow_var_impl(ctx, x, e_limbs, e_num_limbs, exp_limb_bits, n, batch, flags, trace, out, status, workspace, stream, 1);
    if (rc || !in_field_trace) return rc;
    return launch_in_field(ctx, x, n, batch, flags, in_field_trace, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

int32_t h2r_verify_layout_fixed(const h2r_ctx *ctx, const uint8_t *e_le, size_t e_len, h2r_verify_layout *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    if (ctx->layout.limb_width != 64 || ctx->L < 9) return H2R_E_SHAPE;  // RSAChip::LIMB_WIDTH, src/chip.rs:203
    std::memset(out, 0, sizeof *out);
    int32_t rc = h2r_pow_fixed_layout(ctx, e_le, e_len, &out->pow);
    if (rc) return rc;
    const AuxGeom g(ctx->L, 64);
    out->off_in_field = out->pow.elem_stride;
    u64 sb = 0;
    in_field_sections(g, [&](u64, u64 len) { sb += len; });
    out->in_field_stream_bytes = sb;
    out->off_em = out->off_in_field + round_up(g.in_field_sz(), 256);
    out->em_stream_bytes = 2ull * ctx->L + 34;
    out->elem_stride = odd_stride_256(out->off_em + g.em_sz());
    out->stream_bytes = out->in_field_stream_bytes + out->pow.stream_bytes + out->em_stream_bytes;
    return H2R_OK;
} H2R_CATCH_STATUS


int32_t h2r_verify_pkcs1v15_batch(const h2r_ctx *ctx, const void *sig, const void *n, const uint8_t *e_le, size_t e_len,
                                  const uint64_t *hashed, uint64_t batch, uint32_t flags, void *trace, void *powed_out,
                                  uint8_t *is_valid_out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!ctx || !sig || !n || !hashed || !trace || !powed_out || !status) return H2R_E_NULL;
    h2r_verify_layout vl;
    int32_t rc = h2r_verify_layout_fixed(ctx, e_le, e_len, &vl);
    if (rc) return rc;
    ExpBits eb; u32 T;
    rc = exp_to_bits(e_le, e_len, &eb, &T);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (T && ctx->params.device >= 0 && plain_call_overlaps(ctx, batch))   // large call: overlapping sub-batches, as pow_fixed_impl
        rc = overlapped_pow_fixed(ctx, sig, n, e_le, e_len, batch, flags, trace, vl.pow, vl.elem_stride, powed_out, status, workspace, st, 1, T);
    else
        rc = run_path(ctx, CHAIN_POW_FIXED, sig, nullptr, n, nullptr, 0, 0, &eb, 1, batch, flags, T, trace, vl.elem_stride,
                      vl.pow.off_records, &vl.pow, powed_out, status, workspace, st);
    if (rc || batch == 0) return rc;
    return launch_verify_aux(ctx, sig, n, hashed, batch, flags, trace, vl, powed_out, is_valid_out, status, st);
} H2R_CATCH_STATUS

// RSAPubE::Var arm of the same call (src/chip.rs:108-110: pow_mod with the chip's exp_limb_bits)
int32_t h2r_verify_layout_var(const h2r_ctx *ctx, uint32_t e_num_limbs, uint32_t exp_limb_bits, h2r_verify_layout *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    if (ctx->layout.limb_width != 64 || ctx->L < 9) return H2R_E_SHAPE;
    std::memset(out, 0, sizeof *out);
    int32_t rc = h2r_pow_var_layout(ctx, e_num_limbs, exp_limb_bits, &out->pow);
    if (rc) return rc;
    const AuxGeom g(ctx->L, 64);
    out->off_in_field = out->pow.elem_stride;
    u64 sb = 0;
    in_field_sections(g, [&](u64, u64 len) { sb += len; });
    out->in_field_stream_bytes = sb;
    out->off_em = out->off_in_field + round_up(g.in_field_sz(), 256);
    out->em_stream_bytes = 2ull * ctx->L + 34;
    out->elem_stride = odd_stride_256(out->off_em + g.em_sz());
    out->stream_bytes = out->in_field_stream_bytes + out->pow.stream_bytes + out->em_stream_bytes;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_verify_pkcs1v15_var_batch(const h2r_ctx *ctx, const void *sig, const void *n, const void *e_limbs, uint32_t e_num_limbs,
                                      uint32_t exp_limb_bits, const uint64_t *hashed, uint64_t batch, uint32_t flags, void *trace,
                                      void *powed_out, uint8_t *is_valid_out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!ctx || !sig || !n || !e_limbs || !hashed || !trace || !powed_out || !status) return H2R_E_NULL;
    h2r_verify_layout vl;
    int32_t rc = h2r_verify_layout_var(ctx, e_num_limbs, exp_limb_bits, &vl);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = run_path(ctx, CHAIN_POW_VAR, sig, nullptr, n, e_limbs, e_num_limbs, exp_limb_bits, nullptr, 1, batch, flags, vl.pow.num_mul_mods, trace,
                  vl.elem_stride, vl.pow.off_records, &vl.pow, powed_out, status, workspace, st);
    if (rc || batch == 0) return rc;
    return launch_verify_aux(ctx, sig, n, hashed, batch, flags, trace, vl, powed_out, is_valid_out, status, st);
} H2R_CATCH_STATUS

int32_t h2r_verify_trace_flatten(const h2r_ctx *ctx, const h2r_verify_layout *vl, const void *elem_host, void *stream_out) try {
    if (!ctx || !vl || !elem_host || !stream_out) return H2R_E_NULL;
    if (vl->pow.off_records == UINT64_MAX) return H2R_E_SHAPE;   // h2r_verify_layout_compact: no records to flatten
    const u8 *e = static_cast<const u8 *>(elem_host);
    u8 *o = static_cast<u8 *>(stream_out);
    const AuxGeom g(ctx->L, ctx->layout.limb_width);
    in_field_sections(g, [&](u64 off, u64 len) { std::memcpy(o, e + vl->off_in_field + off, len); o += len; });
    int32_t rc = h2r_pow_trace_flatten(ctx, &vl->pow, e, o);
    if (rc) return rc;
    o += vl->pow.stream_bytes;
    std::memcpy(o, e + vl->off_em, vl->em_stream_bytes); o += vl->em_stream_bytes;
    if ((u64)(o - static_cast<u8 *>(stream_out)) != vl->stream_bytes) return H2R_E_SHAPE;
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- the caller of the path: RSASignatureVerifier::verify_pkcs1v15_signature (src/lib.rs:183-246) ---------------------
// SHA-256 of every element's message, the reversed digest packed into the four hashed-message limbs (:213-239), and -- in
// h2r_signature_verifier_batch -- RSAChip::verify_pkcs1v15_signature on them, all in stream order on the caller's stream.
int32_t h2r_sha256_hashed_msg_batch(const h2r_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, uint64_t fixed_len, uint64_t batch,
                                    uint8_t *digest_out, uint64_t *hashed_out, void *hm_trace, uint64_t hm_stride, h2r_stream_t stream) try {
    if (!ctx || (!msgs && (msg_off || fixed_len))) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (hm_trace && hm_stride == 0) hm_stride = HM_REGION;
    if ((reinterpret_cast<u64>(digest_out) | reinterpret_cast<u64>(hashed_out) | reinterpret_cast<u64>(hm_trace) | hm_stride) & 15) return H2R_E_SHAPE;
    if (hm_trace && hm_stride < HM_REGION) return H2R_E_SHAPE;
    if (batch == 0) return H2R_OK;
    if (batch >= (1ull << 37)) return H2R_E_UNSUPPORTED;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Sha256Args sa;
    sa.msgs = msgs; sa.off = msg_off; sa.fixed_len = fixed_len; sa.batch = batch;
    sa.digest = digest_out; sa.hashed = hashed_out; sa.region = static_cast<u8 *>(hm_trace); sa.region_stride = hm_stride;
    sa.done = nullptr; sa.target = 0;
    ProfScope ps(H2R_KERNEL_SHA256, st, true);
    hipExtLaunchKernelGGL(sha256_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, ps.a, ps.on ? ps.b : nullptr, 0, sa);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_signature_verifier_batch(const h2r_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, uint64_t fixed_len, const void *sig,
                                     const void *n, const uint8_t *e_le, size_t e_len, uint64_t batch, uint32_t flags, void *trace,
                                     void *hm_trace, uint64_t hm_stride, uint8_t *digest_out, uint64_t *hashed_out, void *powed_out,
                                     uint8_t *is_valid_out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!hashed_out) return H2R_E_NULL;
    if (ctx && (ctx->layout.limb_width != 64 || ctx->L < 9)) return H2R_E_SHAPE;   // before any launch: RSAChip::LIMB_WIDTH
    const int32_t rc = h2r_sha256_hashed_msg_batch(ctx, msgs, msg_off, fixed_len, batch, digest_out, hashed_out, hm_trace, hm_stride, stream);
    if (rc) return rc;
    return h2r_verify_pkcs1v15_batch(ctx, sig, n, e_le, e_len, hashed_out, batch, flags, trace, powed_out, is_valid_out, status, workspace, stream);
} H2R_CATCH_STATUS

struct h2r_pipeline {
    const h2r_ctx *ctx;
    // Record kernels alternate between two side streams: call k+1's may start as soon as its own chain kernel is
    // done, so it fills the CUs while call k's drains (a single side stream leaves a ~20 us barrier + dispatch gap
    // between consecutive record kernels, measured with rocprofv3 --kernel-trace).
    hipStream_t aux[2];
    enum { MAX_DEPTH = 4 };
    u32 depth;        // buffer sets the caller rotates through: call k may reuse call k-depth's buffers
    hipEvent_t chain_done[MAX_DEPTH], trace_done[MAX_DEPTH];
    hipEvent_t sub_done[2];    // record kernels of the sub-batches inside one call (pipeline_issue)
    DoneRef done[MAX_DEPTH];   // what marks the end of the record kernel of the call in each slot
    hipStream_t done_stream[MAX_DEPTH];
    u32 k;            // calls issued
    u32 joined;       // calls whose record kernel the user stream has been ordered after
    // One-launch steps (step_kernel): the records of the last sub-batch issued are written by the NEXT launch, together
    // with that launch's chains, or alone at the join.
    u32 *sha_done_dev = nullptr;   // messages hashed by SHA roles of this pipeline's step launches (device word, monotonic)
    u32 sha_issued = 0;            // ... as issued by the host
    bool pending = false;
    TraceArgs pending_ta;
    hipStream_t pending_st = nullptr;
    hipEvent_t flush_done = nullptr;   // orders another stream behind a flush
    // Do the caller's stream and the two side streams sit on THREE hardware queues?  HIP multiplexes a process's streams onto
    // GPU_MAX_HW_QUEUES queues (4 by default) in creation order; two streams of one queue never overlap, and the two-queue form then runs
    // BEHIND the one-launch step it replaces (4.8 against 5.5 M assigns/s under torchrun with RCCL's streams in the process).  The library
    // cannot read the assignment, so it measures it once per caller stream: three 150 us one-wave spinners (pipeline_three_queues).
    std::map<hipStream_t, int> queue_probe;   // 1: three queues, 0: some pair shares one
    float probe_ms = 0.f, probe_span_ms = 0.f; // the last probe: host wall time, device-clock span
};

// ---- multi-GPU: RCCL behind the C ABI --------------------------------------------------------------------------------------
extern "C++" {
namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    bool ok = false;
};
const Rccl &rccl() {
    static const Rccl r = [] {
        Rccl x;
        // the copy the process already has (torch's librccl.so, an application's) wins; else the system's
        for (const char *name : {"librccl.so.1", "librccl.so"}) if (!x.handle) x.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (!x.handle) x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) return x;
        auto sym = [&](const char *n) { return dlsym(x.handle, n); };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(sym("ncclGetVersion"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.Broadcast && x.AllGather && x.AllReduce && x.GroupStart && x.GroupEnd && x.GetErrorString;
        return x;
    }();
    return r;
}
bool rccl_ok(ncclResult_t e, const char *what) {
    if (e == ncclSuccess) return true;
    std::snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
    return false;
}
#define RCCL_TRY(expr) do { if (!rccl_ok((expr), #expr)) return H2R_E_HIP; } while (0)
int32_t rccl_ready() {
    if (rccl().ok) return H2R_OK;
    std::snprintf(g_hip_err, sizeof g_hip_err, "librccl.so.1 could not be loaded: %s", rccl().handle ? "symbols missing" : (dlerror() ? dlerror() : "not found"));
    return H2R_E_HIP;
}
}  // namespace
}  // extern "C++"

struct h2r_dist {
    const h2r_ctx *ctx; ncclComm_t comm; u32 rank, world;
    double *scratch;   // one word of plain device memory for the barrier (RCCL looks its buffers up in the runtime's allocation
                       // map: a stream-ordered pool allocation is not in it -- "Memobj map does not have ptr", seen once in five runs)
};

int32_t h2r_dist_unique_id(uint8_t id_out[H2R_DIST_ID_BYTES]) try {
    if (!id_out) return H2R_E_NULL;
    const int32_t rc = rccl_ready();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == H2R_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RCCL_TRY(rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_dist_init(const h2r_ctx *ctx, const uint8_t id[H2R_DIST_ID_BYTES], uint32_t rank, uint32_t world, h2r_dist **out) try {
    if (!ctx || !id || !out) return H2R_E_NULL;
    *out = nullptr;
    if (world == 0 || rank >= world) return H2R_E_SHAPE;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    const int32_t rc = rccl_ready();
    if (rc) return rc;
    H2R_ON_DEVICE(ctx->params.device);
    ncclUniqueId nid;
    std::memcpy(&nid, id, sizeof nid);
    ncclComm_t comm = nullptr;
    RCCL_TRY(rccl().CommInitRank(&comm, (int)world, nid, (int)rank));
    double *scratch = nullptr;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&scratch), sizeof(double)), "hipMalloc(barrier word)")) { (void)rccl().CommDestroy(comm); return H2R_E_HIP; }
    h2r_dist *d = new (std::nothrow) h2r_dist{ctx, comm, rank, world, scratch};
    if (!d) { (void)hipFree(scratch); (void)rccl().CommDestroy(comm); return H2R_E_HIP; }
    *out = d;
    return H2R_OK;
} H2R_CATCH_STATUS

void h2r_dist_destroy(h2r_dist *d) try {
    if (!d) return;
    { DeviceGuard dg(d->ctx->params.device); (void)hipDeviceSynchronize(); (void)rccl().CommDestroy(d->comm); (void)hipFree(d->scratch); }
    delete d;
} H2R_CATCH_VOID
// the version of the RCCL the exports above run on (ncclGetVersion: major * 10000 + minor * 100 + patch), 0 when it cannot be loaded
int32_t h2r_dist_version(void) try { int v = 0; return (rccl().ok && rccl().GetVersion && rccl().GetVersion(&v) == ncclSuccess) ? v : 0; } catch (...) { return 0; }
uint32_t h2r_dist_rank(const h2r_dist *d) try { return d ? d->rank : 0; } H2R_CATCH_ZERO
uint32_t h2r_dist_world(const h2r_dist *d) try { return d ? d->world : 0; } H2R_CATCH_ZERO

int32_t h2r_dist_shard_range(uint64_t total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi) try {
    if (!lo || !hi) return H2R_E_NULL;
    if (world == 0 || rank >= world) return H2R_E_SHAPE;
    const u64 base = total / world, rem = total % world;
    *lo = rank * base + std::min<u64>(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_dist_bcast(h2r_dist *d, void *buf, uint64_t bytes, uint32_t root, h2r_stream_t stream) try {
    if (!d || !buf) return H2R_E_NULL;
    if (root >= d->world) return H2R_E_SHAPE;
    if (bytes == 0) return H2R_OK;
    H2R_ON_DEVICE(d->ctx->params.device);
    RCCL_TRY(rccl().Broadcast(buf, buf, bytes, ncclUint8, (int)root, d->comm, static_cast<hipStream_t>(stream)));
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_dist_gather_results(h2r_dist *d, const void *results_shard, const uint8_t *status_shard, uint64_t shard_elems,
                                void *results_all, uint8_t *status_all, h2r_stream_t stream) try {
    if (!d || !results_shard || !results_all || (!status_shard != !status_all)) return H2R_E_NULL;
    if (shard_elems == 0) return H2R_OK;
    H2R_ON_DEVICE(d->ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const u64 bytes = shard_elems * (u64)d->ctx->L * d->ctx->layout.limb_bytes;
    RCCL_TRY(rccl().GroupStart());
    const ncclResult_t r1 = rccl().AllGather(results_shard, results_all, bytes, ncclUint8, d->comm, st);
    const ncclResult_t r2 = status_shard ? rccl().AllGather(status_shard, status_all, shard_elems, ncclUint8, d->comm, st) : ncclSuccess;
    const ncclResult_t r3 = rccl().GroupEnd();
    RCCL_TRY(r1); RCCL_TRY(r2); RCCL_TRY(r3);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_dist_allreduce_max_f64(h2r_dist *d, double *values, uint64_t count, h2r_stream_t stream) try {
    if (!d) return H2R_E_NULL;
    H2R_ON_DEVICE(d->ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (count == 0 || !values) {   // barrier: a one-element reduction on the communicator's scratch word, ordered on `stream`
        HIP_TRY(hipMemsetAsync(d->scratch, 0, sizeof(double), st));
        RCCL_TRY(rccl().AllReduce(d->scratch, d->scratch, 1, ncclDouble, ncclMax, d->comm, st));
        return H2R_OK;
    }
    RCCL_TRY(rccl().AllReduce(values, values, count, ncclDouble, ncclMax, d->comm, st));
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- placement-aware trace arena ---------------------------------------------------------------------------------------
// Where a trace buffer lies physically decides how fast the record kernel writes it: per 1.25 GB region of config 2 one of
// 0.183 / 0.197 / 0.216 / 0.222 ms per launch (6.8 / 6.35 / 5.8 / 5.65 TB/s), stable for the life of the allocation, a few
// regions in ten fast (tools/buffer_speed_probe.py; physically contiguous memory is always the 5.8 TB/s kind).  The cause
// was not found, so the arena LOOKS: it maps `candidates` regions (HIP virtual-memory API, 256 MB physical chunks), runs the
// record kernel in the production geometry on each, keeps the `regions` fastest and gives the others back.
struct h2r_arena {
    struct Region { void *va = nullptr; u64 mapped = 0, chunk = 0, n_mapped = 0; std::vector<hipMemGenericAllocationHandle_t> handles; float ms = 0.f;
                    bool plain = false; /* a hipMalloc allocation (image arena) instead of stitched physical chunks */ };
    int device = 0;
    u64 region_bytes = 0;
    std::vector<Region> kept;          // fastest first
    std::vector<float> measured;       // every candidate, in allocation order
};

namespace {
void arena_free_region(h2r_arena::Region &r) {
    if (r.plain) { if (r.va) (void)hipFree(r.va); r.va = nullptr; return; }
    // chunk k is mapped at va + k * chunk for k < n_mapped (a candidate that failed half-way has fewer than handles.size())
    for (u64 k = 0; r.va && k < r.n_mapped; ++k) (void)hipMemUnmap(static_cast<u8 *>(r.va) + k * r.chunk, r.chunk);
    for (auto h : r.handles) (void)hipMemRelease(h);
    if (r.va) (void)hipMemAddressFree(r.va, r.mapped);
    r.va = nullptr; r.handles.clear(); r.n_mapped = 0;
}
}  // namespace

int32_t h2r_arena_create(const h2r_ctx *ctx, uint64_t elem_stride, uint64_t first_record_off, uint32_t records_per_elem,
                         uint64_t batch, uint32_t regions, uint32_t candidates, h2r_stream_t stream, h2r_arena **out) try {
    return h2r_arena_create_ex(ctx, elem_stride, first_record_off, records_per_elem, batch, regions, candidates, 0, stream, out);
} H2R_CATCH_STATUS

// max_look_bytes != 0: an upper bound on the device memory the look may hold at any time BESIDES the kept regions (rejected
// candidates kept so that the next one lands elsewhere; the placeholder rounds are skipped) -- a service that shares the device
namespace {
// a streaming fill in the product kernels' store pattern (16 bytes per lane, non-temporal, whole 4 KB runs per workgroup step,
// XCD-contiguous blocks): what h2r_image_arena_create times on a candidate region
__global__ __launch_bounds__(256) void arena_fill_kernel(u8 *p, u64 bytes) {
    const u64 per_block = 64ull << 10;
    const u64 b = xcd_contiguous_block(blockIdx.x, gridDim.x);
    u8 *q = p + b * per_block;
    const u64 n = bytes - b * per_block < per_block ? bytes - b * per_block : per_block;
    for (u64 o = (u64)threadIdx.x * 16; o + 16 <= n; o += 4096) st16(q + o, 0x0123456789abcdefull ^ o, b);
}
using ArenaMeasure = std::function<int32_t(void *va, hipStream_t st, hipEvent_t ea, hipEvent_t eb, float *ms)>;
int32_t arena_build(const h2r_ctx *ctx, u64 region_bytes, uint32_t regions, uint32_t candidates, uint64_t max_look_bytes, hipStream_t st,
                    const ArenaMeasure &measure, bool plain, h2r_arena **out);
}  // namespace

int32_t h2r_arena_create_ex(const h2r_ctx *ctx, uint64_t elem_stride, uint64_t first_record_off, uint32_t records_per_elem,
                            uint64_t batch, uint32_t regions, uint32_t candidates, uint64_t max_look_bytes, h2r_stream_t stream,
                            h2r_arena **out) try {
    if (!ctx || !out) return H2R_E_NULL;
    *out = nullptr;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    const h2r_layout &lo = ctx->layout;
    if (!regions || candidates < regions || !batch || !records_per_elem ||
        elem_stride < first_record_off + (u64)records_per_elem * lo.record_stride) return H2R_E_SHAPE;
    if (batch * (u64)records_per_elem >= (1ull << 32)) return H2R_E_UNSUPPORTED;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // operands of the measurement launches: any values do (the record kernel's store pattern does not depend on them)
    const u64 n_items = batch * records_per_elem;
    const u64 ops_bytes = n_items * 4ull * ctx->L * lo.limb_bytes;
    u8 *scratch = nullptr;
    HIP_TRY(hipMalloc(&scratch, ops_bytes + batch + 4096));
    struct ScratchFree { u8 *p; ~ScratchFree() { if (p) (void)hipFree(p); } } scratch_free{scratch};
    HIP_TRY(hipMemsetAsync(scratch, 0x5a, ops_bytes, st));
    HIP_TRY(hipMemsetAsync(scratch + ops_bytes, 0, batch + 4096, st));
    // three launches of the record kernel in the production geometry
    const ArenaMeasure measure = [&](void *va, hipStream_t s2, hipEvent_t ea, hipEvent_t eb, float *ms_out) -> int32_t {
        TraceArgs ta;
        fill_trace_args(ctx, ta);
        const u64 lb = lo.limb_width / 8;
        ta.opA = scratch; ta.opB = scratch + ctx->L * lb; ta.opQ = scratch + 2 * ctx->L * lb; ta.opR = scratch + 3 * ctx->L * lb;
        ta.op_stride = 4ull * ctx->L;
        ta.n = scratch; ta.n_stride = 0;
        ta.status = scratch + ops_bytes; ta.n_items = n_items; ta.T = records_per_elem;
        ta.trace = static_cast<u8 *>(va); ta.elem_stride = elem_stride; ta.off_records = first_record_off;
        if (knobs().trace_dyn_lds < 0 && lo.limb_width == 64 && ctx->L <= 32) ta.residency = 1;   // the kernel's stand-alone launch shape
        float sum = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            if (!hip_ok(launch_trace(ctx, ta, s2, ea, eb), "launch_trace")) return H2R_E_HIP;
            if (!hip_ok(hipStreamSynchronize(s2), "hipStreamSynchronize")) return H2R_E_HIP;
            float ms = 0.f;
            if (!hip_ok(hipEventElapsedTime(&ms, ea, eb), "hipEventElapsedTime")) return H2R_E_HIP;
            if (rep) sum += ms;   // the first launch touches the pages
        }
        *ms_out = sum / 2.f;
        return H2R_OK;
    };
    return arena_build(ctx, batch * elem_stride, regions, candidates, max_look_bytes, st, measure, false, out);
} H2R_CATCH_STATUS

// The same look for ANY large output the kernels stream into -- advice images, the lookup argument's A' / S' columns: where such a
// buffer lies physically decides its store rate exactly as for the trace (cells_kernel 1.82-2.39 ms, lookup_fill_kernel 5.05-6.73 TB/s
// by buffer).  The candidates are timed with a streaming fill in the product kernels' store pattern; h2r_arena_region etc. apply.
int32_t h2r_image_arena_create(const h2r_ctx *ctx, uint64_t region_bytes, uint32_t regions, uint32_t candidates, uint64_t max_look_bytes,
                               h2r_stream_t stream, h2r_arena **out) try {
    if (!ctx || !out) return H2R_E_NULL;
    *out = nullptr;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (!regions || candidates < regions || region_bytes < (1ull << 20) || (region_bytes >> 16) >= (1ull << 31)) return H2R_E_SHAPE;
    H2R_ON_DEVICE(ctx->params.device);
    const ArenaMeasure measure = [&](void *va, hipStream_t s2, hipEvent_t ea, hipEvent_t eb, float *ms_out) -> int32_t {
        const unsigned blocks = (unsigned)((region_bytes + (64ull << 10) - 1) >> 16);
        float sum = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            hipExtLaunchKernelGGL(arena_fill_kernel, dim3(blocks), dim3(256), 0, s2, ea, eb, 0, static_cast<u8 *>(va), region_bytes);
            if (!hip_ok(hipGetLastError(), "arena_fill_kernel") || !hip_ok(hipStreamSynchronize(s2), "hipStreamSynchronize")) return H2R_E_HIP;
            float ms = 0.f;
            if (!hip_ok(hipEventElapsedTime(&ms, ea, eb), "hipEventElapsedTime")) return H2R_E_HIP;
            if (rep) sum += ms;
        }
        *ms_out = sum / 2.f;
        return H2R_OK;
    };
    // (plain hipMalloc candidates: regions stitched from physical chunks with the virtual-memory API aborted inside the runtime now and then
    //  under this look -- "Memobj map does not have ptr", tools/image_arena_probe.py -- and a streaming buffer has no use for the stitching)
    return arena_build(ctx, region_bytes, regions, candidates, max_look_bytes, static_cast<hipStream_t>(stream), measure, true, out);
} H2R_CATCH_STATUS

namespace {
int32_t arena_build(const h2r_ctx *ctx, u64 region_bytes, uint32_t regions, uint32_t candidates, uint64_t max_look_bytes, hipStream_t st,
                    const ArenaMeasure &measure, bool plain, h2r_arena **out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->params.device;
    size_t gran = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (!gran) gran = 2u << 20;
    const u64 chunk_target = (knobs().arena_chunk_mb > 0 ? (u64)knobs().arena_chunk_mb : 256ull) << 20;
    const u64 n_chunks = (region_bytes + chunk_target - 1) / chunk_target;
    const u64 chunk = round_up((region_bytes + n_chunks - 1) / n_chunks, gran);
    std::unique_ptr<h2r_arena> a(new (std::nothrow) h2r_arena());
    if (!a) return H2R_E_HIP;
    a->device = ctx->params.device; a->region_bytes = region_bytes;
    hipEvent_t ea = nullptr, eb = nullptr;
    HIP_TRY(hipEventCreate(&ea));
    if (!hip_ok(hipEventCreate(&eb), "hipEventCreate")) { (void)hipEventDestroy(ea); return H2R_E_HIP; }
    // one candidate: reserve, create, map, touch, measure
    auto make_candidate = [&](h2r_arena::Region &r) -> int32_t {
        r.mapped = n_chunks * chunk; r.chunk = chunk; r.n_mapped = 0;
        if (plain) {
            r.plain = true; r.mapped = region_bytes;
            if (!hip_ok(hipMalloc(&r.va, region_bytes), "hipMalloc")) { r.va = nullptr; return H2R_E_HIP; }
            return measure(r.va, st, ea, eb, &r.ms);   // (its first launch touches the pages)
        }
        if (!hip_ok(hipMemAddressReserve(&r.va, r.mapped, 0, nullptr, 0), "hipMemAddressReserve")) { r.va = nullptr; return H2R_E_HIP; }
        for (u64 k = 0; k < n_chunks; ++k) {
            hipMemGenericAllocationHandle_t h;
            if (!hip_ok(hipMemCreate(&h, chunk, &prop, 0), "hipMemCreate")) return H2R_E_HIP;
            r.handles.push_back(h);
            if (!hip_ok(hipMemMap(static_cast<u8 *>(r.va) + k * chunk, chunk, 0, h, 0), "hipMemMap")) return H2R_E_HIP;
            r.n_mapped = k + 1;
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        if (!hip_ok(hipMemSetAccess(r.va, r.mapped, &acc, 1), "hipMemSetAccess")) return H2R_E_HIP;
        if (!hip_ok(hipMemsetAsync(r.va, 0, region_bytes, st), "hipMemsetAsync")) return H2R_E_HIP;
        return measure(r.va, st, ea, eb, &r.ms);
    };
    std::vector<h2r_arena::Region> cands;
    int32_t rc = H2R_OK;
    // Rejected candidates keep their memory while the look goes on -- given back at once, the next candidate would be
    // handed the very same physical blocks -- up to 64 GB of them; beyond that the oldest are released (so the look also
    // fits traces of tens of GB, with less variety among the candidates).
    std::vector<h2r_arena::Region> rejected;
    u64 held_budget = 64ull << 30;
    size_t free_at_start = 0, total_at_start = 0;
    if (hipMemGetInfo(&free_at_start, &total_at_start) == hipSuccess) held_budget = std::min<u64>(held_budget, free_at_start / 2);
    else (void)hipGetLastError();
    if (max_look_bytes) held_budget = std::min<u64>(held_budget, max_look_bytes);
    auto keep_best = [&](size_t keep, bool release_all) {   // sorts; everything behind the first `keep` moves to `rejected`
        std::stable_sort(cands.begin(), cands.end(), [](const h2r_arena::Region &x, const h2r_arena::Region &y) { return x.ms < y.ms; });
        for (size_t i = keep; i < cands.size(); ++i) rejected.push_back(std::move(cands[i]));
        if (cands.size() > keep) cands.resize(keep);
        size_t first = 0;
        if (!release_all) { u64 held = 0; first = rejected.size(); while (first > 0 && held + rejected[first - 1].mapped <= held_budget) held += rejected[--first].mapped; }
        else first = rejected.size();
        for (size_t i = 0; i < first; ++i) arena_free_region(rejected[i]);
        rejected.erase(rejected.begin(), rejected.begin() + (long)first);
    };
    auto run_round = [&]() {
        for (u32 ci = 0; ci < candidates && rc == H2R_OK; ++ci) {
            cands.emplace_back();
            rc = make_candidate(cands.back());
            if (rc == H2R_OK) { a->measured.push_back(cands.back().ms); keep_best(regions, false); }
            else if (cands.size() > regions) {   // (out of memory, most likely) with enough good regions in hand: stop looking
                arena_free_region(cands.back()); cands.pop_back();
                (void)hipGetLastError();
                rc = H2R_OK;
                return;
            }
        }
    };
    run_round();
    // The kept regions should be of ONE class: a pipeline rotates through all of them, and a step into a second-class region
    // is 8 % longer (driver box of round 2: 1 fast candidate of 24, kept 0.183 / 0.199 ms).  While the slowest kept region is
    // more than 4 % behind the fastest, keep looking -- up to three more rounds' worth of candidates, one at a time.
    if (rc == H2R_OK && regions > 1 && region_bytes <= (12ull << 30)) {
        for (u32 extra = 0; extra < 3 * candidates && rc == H2R_OK && cands.size() == regions && cands.back().ms > 1.04f * cands.front().ms; ++extra) {
            cands.emplace_back();
            const int32_t r1 = make_candidate(cands.back());
            if (r1 != H2R_OK) { arena_free_region(cands.back()); cands.pop_back(); (void)hipGetLastError(); break; }   // out of memory: what we have
            a->measured.push_back(cands.back().ms);
            keep_best(regions, false);
        }
    }
    if (rc == H2R_OK && region_bytes <= (12ull << 30) && candidates >= 4 && !max_look_bytes && !plain) {
        // No fast class among the candidates?  On a box whose memory has not been churned yet (about one in five) the first
        // ~60 GB handed out are ALL of the slow class -- as physically contiguous memory always is -- while regions mapped after
        // some allocate / free traffic, or behind a large allocation, do contain fast ones (tools/no_fast_box_probe.py,
        // profiles/r03_placement.txt: 0 of 24, then 2 of 24 behind 64 GB, 2 of 24 behind 128 GB, 1 of 24 behind 192 GB).
        // Up to four more rounds, each after giving everything back and behind a placeholder of another size; a region counts as
        // fast when it is 7 % faster than the median of everything measured (the slow class is the majority on every box seen).
        auto have_fast = [&]() {
            std::vector<float> t(a->measured);
            std::sort(t.begin(), t.end());
            const float med = t[t.size() / 2];
            return cands.size() == regions && cands.back().ms <= 0.93f * med && cands.back().ms <= 1.04f * cands.front().ms;   // fast, and of one class
        };
        for (u32 round = 1; round <= 4 && rc == H2R_OK && !have_fast(); ++round) {
            keep_best(regions, true);
            size_t free_b = 0, total_b = 0;
            void *placeholder = nullptr;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const u64 want = (u64)candidates * n_chunks * chunk + (8ull << 30);
                const u64 size = std::min<u64>((u64)round * (48ull << 30), free_b / 2);
                const u64 ph = free_b > want + (16ull << 30) ? std::min<u64>(size, free_b - want - (16ull << 30)) : 0;
                if (ph && hipMalloc(&placeholder, ph) != hipSuccess) { placeholder = nullptr; (void)hipGetLastError(); }
            }
            for (u32 ci = 0; ci < candidates && rc == H2R_OK && !have_fast(); ++ci) {
                cands.emplace_back();
                const int32_t r1 = make_candidate(cands.back());
                if (r1 != H2R_OK) { arena_free_region(cands.back()); cands.pop_back(); (void)hipGetLastError(); break; }
                a->measured.push_back(cands.back().ms);
                keep_best(regions, false);
            }
            if (placeholder) (void)hipFree(placeholder);
        }
    }
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    if (rc) { for (auto &r : cands) arena_free_region(r); for (auto &r : rejected) arena_free_region(r); return rc; }
    keep_best(regions, true);
    for (auto &r : cands) a->kept.push_back(std::move(r));
    *out = a.release();
    return H2R_OK;
}
}  // namespace

void *h2r_arena_region(const h2r_arena *a, uint32_t i) try { return (a && i < a->kept.size()) ? a->kept[i].va : nullptr; } catch (...) { return nullptr; }
uint64_t h2r_arena_region_bytes(const h2r_arena *a) try { return a ? a->region_bytes : 0; } H2R_CATCH_ZERO
double h2r_arena_region_ms(const h2r_arena *a, uint32_t i) try { return (a && i < a->kept.size()) ? (double)a->kept[i].ms : 0.0; } catch (...) { return 0.0; }
uint32_t h2r_arena_measurements(const h2r_arena *a, double *ms_out, uint32_t cap) try {
    if (!a) return 0;
    for (u32 i = 0; i < a->measured.size() && i < cap && ms_out; ++i) ms_out[i] = a->measured[i];
    return (uint32_t)a->measured.size();
} H2R_CATCH_ZERO
void h2r_arena_destroy(h2r_arena *a) try {
    if (!a) return;
    {
        DeviceGuard dg(a->device);
        (void)hipDeviceSynchronize();
        for (auto &r : a->kept) arena_free_region(r);
    }
    delete a;
} H2R_CATCH_VOID

namespace {
// Which calls are issued as one-launch steps: the shape both roles of step_kernel are built for (RSA-2048: 64-bit limbs,
// 32 limbs -- 64-digit chains on four waves, record workgroups of 256 threads), at batches the throughput chain build serves.
// Measured against the two-queue form on the same boxes (bench.py, H2R_PIPE_STEP=0|1): 1,024 per call +1..5 %, 2,048 per call
// +0..3 %, 8,192 as ONE step launch -4..+1 % -- so a call above 4,096 is walked as several launches of at most 4,096.
const StepShape *step_shape(const h2r_ctx *c) { return step_shape_of(c->layout.limb_width, c->L, c->K); }
bool step_eligible(const h2r_ctx *c, u64 batch, const void *trace, u32 T) {
    return knobs().pipe_step != 0 && knobs().chain_nw == 0 && step_shape(c) && batch > 512 && trace && T;
}
// The shapes and call sizes with a two-queue form (chain kernels on the caller's stream, record kernels alternating between two side streams):
// RSA-2048 up to 2,048 per call (profiles/r04_two_queue.txt), and [r6] RSA-1024 from 1,280 per call now that its chain is the one-wave kernel
// (profiles/r06_two_queue_rsa1024.txt: 16.2-16.7 M assigns/s against the step's 14.5-15.9 M at 1,536-2,048 per call, 15.1-15.9 against 13.7-14.5 M at
// 8,192, 13.9-14.2 -> 14.7-14.8 M at 1,280 with chain priority; at 1,024 per call the step stays ahead or level in every variant).
bool two_queue_shape(const h2r_ctx *c, u64 batch) {
    if (c->layout.limb_width != 64) return false;
    if (c->L == 32) return batch <= 2048;
    if (c->L == 16 && c->K == 32 && knobs().chain_wave != 0 && knobs().chain_nw == 0) {
        if (knobs().pipe_twoq_l16 > 0) return (long)batch <= knobs().pipe_twoq_l16;     // (developer build: the upper bound swept)
        return knobs().pipe_twoq_l16 == 0 && batch >= 1280;                             // (H2R_PIPE_TWOQ_L16=-1: never)
    }
    return false;
}
constexpr u64 kStepMax = 4096;   // elements per step launch: a larger call is walked as equal parts of at most this size
// One step: the records described by `ta` (an earlier sub-batch) and the chains described by `ca`, one launch on `st` (h2r_tu_step.hip).
hipError_t launch_step(const h2r_ctx *c, const ChainArgs &ca, const TraceArgs &ta, const AuxArgs *aa, const AuxArgs *va, const Sha256Args *sha, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    const StepShape *s = step_shape(c);
    if (!s) return hipErrorInvalidValue;
    return launch_step_shape(*s, c->num_cus, ca, ta, aa, va, sha, st, ea, eb);
}
u32 step_shared_bytes(const h2r_ctx *c) {
    const StepShape *s = step_shape(c);
    return s ? step_shared_bytes_shape(*s) : 0;
}
// The pending records alone (the end of a train of steps, or a call that cannot be issued as a step); `st` is ordered behind them.
int32_t pipeline_flush(h2r_pipeline *p, hipStream_t st) {
    if (!p->pending) return H2R_OK;
    TraceArgs ta = p->pending_ta;
    ta.residency = (p->ctx->layout.limb_width == 64 && p->ctx->L <= 32) ? 1 : 0; ta.dyn_lds = 0;   // the record kernel's stand-alone launch shape
    {
        ProfScope ps(H2R_KERNEL_TRACE, p->pending_st, true);
        HIP_TRY(launch_trace(p->ctx, ta, p->pending_st, ps.a, ps.b));
    }
    p->pending = false;
    if (st != p->pending_st) {
        HIP_TRY(hipEventRecord(p->flush_done, p->pending_st));
        HIP_TRY(hipStreamWaitEvent(st, p->flush_done, 0));
    }
    return H2R_OK;
}
}  // namespace

int32_t h2r_pipeline_create(const h2r_ctx *ctx, h2r_pipeline **out) try { return h2r_pipeline_create_ex(ctx, 2, 1, out); } H2R_CATCH_STATUS

namespace {
__global__ void queue_probe_kernel(unsigned long long ticks, unsigned long long *stamp) {   // one wave that holds its queue for `ticks` of the 100 MHz wall clock
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (stamp && threadIdx.x == 0) { stamp[0] = t0; stamp[1] = wall_clock64(); }   // when it ran, on the device's own clock
}
}  // namespace
namespace {
// force: measure again although `st` has a cached verdict (h2r_pipeline_info: the caller's way to refresh it after streams were created or
// destroyed -- HIP may have re-assigned queues, and a destroyed stream's handle can come back for a new stream)
bool pipeline_three_queues(h2r_pipeline *p, hipStream_t st, bool force = false) {
    if (p->aux[0] == p->aux[1]) return false;
    if (knobs().pipe_form >= 0) return knobs().pipe_form == 1;    // (developer build: H2R_PIPE_FORM forces the form, e.g. under a profiler)
    auto it = p->queue_probe.find(st);
    if (!force && it != p->queue_probe.end()) return it->second != 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (cs != hipStreamCaptureStatusNone) return false;           // (a capture cannot be timed: the form that needs no particular queues; not cached)
    // everything queued so far out of the way, then the three spinners back to back: one spin of wall time when they overlap, three when they share
    hipStream_t ss[3] = {st, p->aux[0], p->aux[1]};
    for (hipStream_t s : ss) if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return false; }
    const unsigned long long spin_ticks = 15000;                  // 150 us
    float best = 1e9f, span = 1e9f;                               // host wall time / device-clock span (first start to last end), best round of each
    bool overlapped = false;
    unsigned long long *stamps = nullptr;                         // [3][2]: every spinner's start and end on the DEVICE's wall clock
    if (hipHostMalloc(reinterpret_cast<void **>(&stamps), 6 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); stamps = nullptr; }
    for (int rep = 0; rep < 4; ++rep) {                           // (the first round also loads the kernel; the best of the other three counts)
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(queue_probe_kernel, dim3(1), dim3(64), 0, ss[k], spin_ticks, stamps ? stamps + 2 * k : nullptr);
        bool ok = hipGetLastError() == hipSuccess;
        for (hipStream_t s : ss) ok = (hipStreamSynchronize(s) == hipSuccess) && ok;
        if (!ok) { (void)hipGetLastError(); if (stamps) (void)hipHostFree(stamps); return false; }
        const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (!rep) continue;
        best = std::min(best, ms);
        if (stamps) {   // the device's own account of the round: all three ran at one instant iff the latest start precedes the earliest end
            const bool ov = std::max({stamps[0], stamps[2], stamps[4]}) < std::min({stamps[1], stamps[3], stamps[5]});
            const float sp = (float)(std::max({stamps[1], stamps[3], stamps[5]}) - std::min({stamps[0], stamps[2], stamps[4]})) * 1e-5f;   // 100 MHz ticks
            if (ov && sp < span) { span = sp; overlapped = true; }
            else if (!overlapped) span = std::min(span, sp);
        }
    }
    if (stamps) (void)hipHostFree(stamps);
    p->probe_ms = best; p->probe_span_ms = stamps ? span : 0.f;
    // Measured (profiles/r05_queue_probe.txt, 16 runs: plain / torchrun + RCCL, 4 / 8 hardware queues, low / normal side-stream priority):
    // host wall time 0.179-0.181 ms whenever the two-queue form then ran at 5.5-5.6 M assigns/s, 0.199-0.203 ms whenever it ran at 4.2-5.0 M --
    // two of the streams share a queue and their packets overlap only partly.  A false "shared" costs 2 % (the step runs at 5.5 M), a false
    // "three queues" 10-25 %.
    // [r6] The host clock carries launch latency and its jitter (clean rounds of 0.178-0.186 ms in this round's runs: one at 0.185 sat on the
    // old 0.186 threshold).  The spinners' own device-clock stamps do not: first start to last end is 0.1539-0.1545 ms on three queues and
    // 0.1595-0.1597 ms when two streams share one (profiles/r06_queue_probe.txt, 32 runs on two boxes, the same eight conditions), a gap of
    // twenty times the spread.  The verdict is the device's: all three running at one instant AND span <= 0.157 ms; the host time only has
    // to be sane (<= 0.195 ms).  Without a stamp buffer the host clock decides alone, at the old threshold.
#ifdef H2R_DEV_KNOBS
    if (std::getenv("H2R_PROBE_DEBUG")) std::fprintf(stderr, "h2r queue probe: host %.4f ms, device span %.4f ms, overlapped %d\n", best, span, (int)overlapped);
#endif
    const int three = stamps ? ((overlapped && span <= 0.157f && best <= 0.195f) ? 1 : 0) : (best <= 0.186f ? 1 : 0);
    p->queue_probe[st] = three;
    return three != 0;
}
}  // namespace

int32_t h2r_pipeline_info(h2r_pipeline *p, h2r_stream_t stream, uint64_t batch, h2r_pipeline_info_t *out) try {
    if (!p || !out) return H2R_E_NULL;
    const bool v1 = out->struct_size == offsetof(h2r_pipeline_info_t, probe_span_ms);   // (callers built before probe_span_ms existed)
    if (!v1 && out->struct_size != sizeof(h2r_pipeline_info_t)) return H2R_E_UNSUPPORTED;
    const h2r_ctx *ctx = p->ctx;
    H2R_ON_DEVICE(ctx->params.device);
    out->depth = p->depth; out->side_streams = p->aux[0] != p->aux[1] ? 2u : 1u;
    const bool shape = batch && two_queue_shape(ctx, batch) && p->aux[0] != p->aux[1] && p->depth >= 3;
    out->three_queues = shape ? (pipeline_three_queues(p, static_cast<hipStream_t>(stream), true) ? 1u : 0u) : 2u;   // 2: not asked (the shape has no two-queue form)
    out->probe_ms = p->probe_ms;
    if (!v1) out->probe_span_ms = p->probe_span_ms;
    out->record_form = (shape && out->three_queues == 1) ? H2R_PIPE_TWO_QUEUE : (step_eligible(ctx, batch, reinterpret_cast<void *>(1), 19) ? H2R_PIPE_ONE_LAUNCH_STEP : H2R_PIPE_SIDE_STREAM);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_pipeline_create_ex(const h2r_ctx *ctx, uint32_t depth, uint32_t side_streams, h2r_pipeline **out) try {
    if (!ctx || !out) return H2R_E_NULL;
    *out = nullptr;
    if (depth < 2 || depth > h2r_pipeline::MAX_DEPTH || side_streams < 1 || side_streams > 2) return H2R_E_SHAPE;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    H2R_ON_DEVICE(ctx->params.device);
    h2r_pipeline *p = new (std::nothrow) h2r_pipeline();
    if (!p) return H2R_E_HIP;
    p->ctx = ctx; p->k = 0; p->joined = 0; p->depth = depth;
    const int n_aux = (int)side_streams;
    for (int i = 0; i < 2; ++i) p->aux[i] = nullptr;
    for (int i = 0; i < h2r_pipeline::MAX_DEPTH; ++i) { p->chain_done[i] = nullptr; p->trace_done[i] = nullptr; }
    p->sub_done[0] = p->sub_done[1] = nullptr;
    bool ok = true;
    // The side streams are created at the LOWEST stream priority: HIP multiplexes streams of one priority onto a
    // few hardware queues (GPU_MAX_HW_QUEUES, default 4), and two streams that share a queue never overlap -- under
    // torchrun, RCCL's own streams pushed the side stream onto the caller's queue and the pipeline ran serially
    // (0.362 vs 0.257 ms/step).  Another priority level means another queue; low rather than high because the
    // record kernel should fill what the chain kernel leaves, not the other way round (measured: tools/dist_ab.sh).
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio = knobs().pipe_stream_prio < 0 ? prio_least : (knobs().pipe_stream_prio > 0 ? prio_greatest : 0);
    for (int i = 0; ok && i < n_aux; ++i) {
        if (knobs().pipe_cu_mask) {   // developer experiment: the record kernels on a subset of the CUs
            uint32_t mask[8];
            for (int w = 0; w < 8; ++w) mask[w] = (knobs().pipe_cu_mask_words == 0 || w < knobs().pipe_cu_mask_words) ? (uint32_t)knobs().pipe_cu_mask : 0u;
            ok = hip_ok(hipExtStreamCreateWithCUMask(&p->aux[i], 8, mask), "hipExtStreamCreateWithCUMask");
            continue;
        }
        ok = hip_ok(hipStreamCreateWithPriority(&p->aux[i], hipStreamNonBlocking, prio), "hipStreamCreate");
    }
    if (ok && n_aux == 1) p->aux[1] = p->aux[0];
    for (u32 i = 0; ok && i < p->depth; ++i)
        ok = hip_ok(hipEventCreate(&p->chain_done[i]), "hipEventCreate") &&
             hip_ok(hipEventCreate(&p->trace_done[i]), "hipEventCreate");
    for (int i = 0; ok && i < 2; ++i) ok = hip_ok(hipEventCreate(&p->sub_done[i]), "hipEventCreate");
    if (ok) ok = hip_ok(hipEventCreateWithFlags(&p->flush_done, hipEventDisableTiming), "hipEventCreate");
    if (ok) ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&p->sha_done_dev), 256), "hipMalloc(sha count)") &&
                 hip_ok(hipMemset(p->sha_done_dev, 0, 256), "hipMemset(sha count)");
    if (!ok) { h2r_pipeline_destroy(p); return H2R_E_HIP; }
    *out = p;
    return H2R_OK;
} H2R_CATCH_STATUS

void h2r_pipeline_destroy(h2r_pipeline *p) try {
    if (!p) return;
    DeviceGuard dg(p->ctx->params.device);
    // a caller that did not join: the records still owed go out on the stream of the last call (which must still exist:
    // h2r.h asks for h2r_pipeline_join() before a stream with pipelined calls on it is destroyed)
    if (p->pending) { (void)pipeline_flush(p, p->pending_st); (void)hipStreamSynchronize(p->pending_st); }
    if (p->flush_done) (void)hipEventDestroy(p->flush_done);
    if (p->sha_done_dev) (void)hipFree(p->sha_done_dev);
    for (int i = 0; i < 2; ++i) if (p->aux[i]) (void)hipStreamSynchronize(p->aux[i]);
    for (int i = 0; i < h2r_pipeline::MAX_DEPTH; ++i) {
        if (p->chain_done[i]) (void)hipEventDestroy(p->chain_done[i]);
        if (p->trace_done[i]) (void)hipEventDestroy(p->trace_done[i]);
    }
    for (int i = 0; i < 2; ++i) if (p->sub_done[i]) (void)hipEventDestroy(p->sub_done[i]);
    if (p->aux[1] && p->aux[1] != p->aux[0]) (void)hipStreamDestroy(p->aux[1]);
    if (p->aux[0]) (void)hipStreamDestroy(p->aux[0]);
    delete p;
} H2R_CATCH_VOID

namespace {
// Order `st` after the record kernel of the call in `slot`.
int32_t pipeline_wait_slot(h2r_pipeline *p, u32 slot, hipStream_t st) {
    DoneRef &d = p->done[slot];
    if (!d.ev) return H2R_OK;   // that call launched no record kernel
    bool alive = true;
    if (d.borrowed) { std::lock_guard<std::mutex> lk(g_prof_mu); alive = d.gen == g_prof_gen; }
    if (!alive) {
        // the profiler's events were released in between: fall back to the tail of that record stream, which is
        // behind the kernel in question (over-synchronises, never under-synchronises)
        HIP_TRY(hipEventRecord(p->trace_done[slot], p->done_stream[slot]));
        d = DoneRef{p->trace_done[slot], 0, false};
    }
    HIP_TRY(hipStreamWaitEvent(st, d.ev, 0));
    return H2R_OK;
}
}  // namespace

namespace { void call_plan(const h2r_ctx *c, u64 batch, bool busy, std::vector<u64> &sizes, bool &pace); }

int32_t h2r_pipeline_call_plan(const h2r_ctx *ctx, uint64_t batch, uint32_t pipeline_busy_, uint64_t *sizes_out, uint32_t cap,
                               uint32_t *n_out, uint32_t *paced_out) try {
    if (!ctx || !n_out) return H2R_E_NULL;
    std::vector<u64> sizes; bool pace = false;
    call_plan(ctx, batch, pipeline_busy_ != 0, sizes, pace);
    *n_out = (uint32_t)sizes.size();
    if (paced_out) *paced_out = pace ? 1u : 0u;
    if (sizes_out) for (size_t i = 0; i < sizes.size() && i < cap; ++i) sizes_out[i] = sizes[i];
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
u32 exp_segment_count(const h2r_ctx *c, u64 batch, u32 nbits, bool has_trace, bool single_call);
void exp_segment_plan(u32 n_seg, u32 nbits, const ExpBits *eb, std::vector<ExpSegment> &out);
}
int32_t h2r_exp_segment_plan(const h2r_ctx *ctx, uint64_t batch, const uint8_t *e_le_bytes, size_t e_len, uint32_t var_exp_bits,
                             uint32_t *bit_bounds_out, uint32_t *mul_mod_bounds_out, uint32_t cap, uint32_t *n_out) try {
    if (!ctx || !n_out) return H2R_E_NULL;
    ExpBits eb; u32 T = 0;
    u32 nbits = var_exp_bits;
    if (!var_exp_bits) {
        const int32_t rc = exp_to_bits(e_le_bytes, e_len, &eb, &T);
        if (rc) return rc;
        nbits = eb.nbits;
    }
    const u32 n_seg = exp_segment_count(ctx, batch, nbits, true, false);
    std::vector<ExpSegment> segs;
    exp_segment_plan(n_seg, nbits, var_exp_bits ? nullptr : &eb, segs);
    *n_out = n_seg;
    for (u32 i = 0; i <= n_seg && i < cap; ++i) {
        if (bit_bounds_out) bit_bounds_out[i] = i < n_seg ? segs[i].bit_lo : nbits;
        if (mul_mod_bounds_out) mul_mod_bounds_out[i] = i < n_seg ? segs[i].t_lo : segs[n_seg - 1].t_lo + segs[n_seg - 1].t_cnt;
    }
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_pipeline_join(h2r_pipeline *p, h2r_stream_t stream) try {
    if (!p) return H2R_E_NULL;
    if (p->pending) {
        H2R_ON_DEVICE(p->ctx->params.device);
        const int32_t rf = pipeline_flush(p, static_cast<hipStream_t>(stream));
        if (rf) return rf;
    }
    for (; p->joined < p->k; ++p->joined) {
        const int32_t rc = pipeline_wait_slot(p, p->joined % p->depth, static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
// Is the record kernel of the previous pipelined call still queued or running?
bool pipeline_busy(h2r_pipeline *p) {
    if (p->pending) return true;
    if (p->k == 0) return false;
    const DoneRef &d = p->done[(p->k - 1) % p->depth];
    if (!d.ev) return false;
    if (d.borrowed) { std::lock_guard<std::mutex> lk(g_prof_mu); if (d.gen != g_prof_gen) return false; }
    return hipEventQuery(d.ev) == hipErrorNotReady;
}

// How one pipelined call is walked: the sizes of its sub-batches (each gets its own chain and record kernel) and whether
// sub-batch i+1's chain kernel is held back until sub-batch i's record kernel starts.  What is at stake is the start and
// the end of a call, not its steady state: an unsplit call exposes its whole chain kernel when no record kernel is in flight
// (0.94 ms for 8,192 RSA-2048 signatures against 1.77 ms of record kernel), while at steady state one large launch per call
// is the fastest form (fewer kernel boundaries: 1.77-1.81 ms per 8,192 against 1.93 as eight launches).  Measured per shape
// (tools/sub_batch_ab.sh, profiles/r02_sub_batches.txt):
//  * record-bound shapes (64-bit limbs, RSA-1536/2048): a call that finds the pipeline EMPTY is walked as sub-batches growing
//    by 3/2 from one chain-kernel grid (1024, 1536, 2304, ...): each chain kernel then fits next to the record kernel of the
//    sub-batch before it; paced (an unpaced train of chain kernels slows the record kernels it overlaps by 5 %).  A call
//    that finds a record kernel in flight is not split.
//  * chain-bound shapes (RSA-3072: chain kernel 0.65 ms, record kernel 0.49 ms per 1,024): uniform sub-batches, unpaced --
//    the chain kernels run back to back either way, the record kernels hide behind them, and the last record kernel of a
//    call is a quarter as long (4,096 signatures: 1.39-1.44 -> 1.60-1.62 M assigns/s over six calls).
//  * RSA-1024 and the 32-bit-limb shapes: no gain measured from any split; one launch.
// (busy: a record kernel of the previous call is still queued or running)
void call_plan(const h2r_ctx *c, u64 batch, bool busy, std::vector<u64> &sizes, bool &pace) {
    sizes.clear(); pace = false;
    const u64 unit = (u64)c->num_cus * (c->K > 64 ? 2 : 4);   // one chain-kernel grid: four 4-wave (two 8-wave) workgroups per CU
    if (knobs().pipe_sub_batch > 0) {
        pace = knobs().pipe_pace != 0;
        for (u64 o = 0; o < batch; o += (u64)knobs().pipe_sub_batch) sizes.push_back(std::min<u64>((u64)knobs().pipe_sub_batch, batch - o));
        if (sizes.empty()) sizes.push_back(batch);
        return;
    }
    const bool w64 = c->layout.limb_width == 64;
    if (w64 && c->L > 32 && batch > 3 * unit) {                         // chain-bound (RSA-3072, RSA-4096 at 64-bit limbs)
        for (u64 o = 0; o < batch; o += 2 * unit) sizes.push_back(std::min<u64>(2 * unit, batch - o));
        return;
    }
    if (w64 && c->L > 16 && c->L <= 32 && batch > unit + unit / 2 && !busy) {   // record-bound, pipeline empty
        pace = true;
        u64 cur = unit, left = batch;
        while (left) {
            u64 take = std::min(cur, left);
            if (left - take < cur / 2) take = left;
            sizes.push_back(take); left -= take;
            cur = (cur * 3 / 2) & ~255ull;
        }
        return;
    }
    sizes.push_back(batch);
}
// A LONG exponent on a latency-bound batch (at most two elements per CU: every chain's own length is what the call waits for --
// BASELINE config 5: 256 elements x 3,072 dependent mul_mods, 7-8.5 ms of chain and 7.9 ms of record kernel) is walked as SEGMENTS of
// its bits: the chain kernel of bits [lo, hi) of every element, then -- on the record stream -- the record kernel of those bits'
// mul_mods, next to the chain kernel of the following segment.  A call's records then trail its own chains by one segment instead of
// by the whole chain: a single call drops from chain + records to chain + 1/S records, and a train of pipelined calls loses the
// exposed first chain / last record kernel.  Same values, same buffers; the (squared, acc) pair crosses launches in the workspace.
u32 exp_segment_count(const h2r_ctx *c, u64 batch, u32 nbits, bool has_trace, bool single_call) {
    if (!has_trace || batch == 0) return 1;
    // single_call: a stream-ordered export that neither follows nor is followed by another call's kernels.  EXPERIMENT, off in the
    // product (the knob's default): the same cut for a SHORT exponent (e = 65537: 19 mul_mods) on a batch that fills the chip but is
    // too small to be walked as sub-batches of elements -- measured +2 % with two segments at 1,024 RSA-2048 elements and a loss
    // everywhere else (a 9-mul_mod chain kernel is latency-bound, every segment adds a cross-queue wait): profiles/r03_exp_segments.txt
    if (single_call && nbits >= 8 && nbits < 512 && c->layout.limb_width == 64 && c->L == 32 && batch > 512 && !plain_call_overlaps(c, batch)) {
        const long k = knobs().single_call_segments;
        return k >= 0 ? (u32)std::max<long>(1, std::min<long>(k, nbits / 2)) : 1;
    }
    if (batch > 2ull * c->num_cus || nbits < 512) return 1;
    if (knobs().exp_segments >= 0) return knobs().exp_segments > 1 ? (u32)std::min<long>(knobs().exp_segments, nbits / 32) : 1;
    // measured (tools/exp_segments_ab.sh, config 5, same box, 1 / 2 / 4 / 8 / 16 / 32 segments): pipelined 26.2 / 28.1 / 29.3 / 30.5 / 30.3 / 29.3 k
    // assigns/s, single calls 17.4 / 21.4 / 25.2 / 27.4 / 28.7 / 27.6 k
    return std::min<u32>(16, nbits / 128);   // >= 128 bits (128-256 mul_mods per element, ~0.5 ms of chain) per segment
}
// The segments themselves: equal parts of the exponent's bits (long exponents: boundaries on 32-bit words of e) with the mul_mods each covers
// -- one squaring per bit plus one multiply per set bit of a fixed exponent `eb`, two per bit of a variable one (eb == nullptr).
void exp_segment_plan(u32 n_seg, u32 nbits, const ExpBits *eb, std::vector<ExpSegment> &out) {
    out.clear();
    const u32 word = nbits >= 512 ? ~31u : ~0u;
    u32 t_lo = 0;
    for (u32 sgi = 0; sgi < n_seg; ++sgi) {
        ExpSegment sg;
        sg.bit_lo = (u32)((u64)nbits * sgi / n_seg) & word;
        sg.bit_hi = sgi + 1 == n_seg ? nbits : (u32)((u64)nbits * (sgi + 1) / n_seg) & word;
        sg.t_lo = t_lo; sg.t_cnt = 0;
        for (u32 bi = sg.bit_lo; bi < sg.bit_hi; ++bi) sg.t_cnt += eb ? 1u + ((eb->words[bi >> 5] >> (bi & 31)) & 1u) : 2u;
        t_lo += sg.t_cnt;
        out.push_back(sg);
    }
}
void pipeline_plan(h2r_pipeline *p, u64 batch, bool assume_empty, std::vector<u64> &sizes, bool &pace) {
    // (the busy query is only made where the answer matters)
    const h2r_ctx *c = p->ctx;
    const bool may_grow = c->layout.limb_width == 64 && c->L > 16 && c->L <= 32;
    call_plan(c, batch, may_grow && !assume_empty && pipeline_busy(p), sizes, pace);
}

// One pipelined call: chain kernel (+ `after_chain`, e.g. the verifier's aux kernel) on the caller's stream, the
// record kernel on a side stream, then the lazy join of the call whose buffers the next call may reuse.
int32_t pipeline_issue(h2r_pipeline *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len, uint64_t batch,
                       uint32_t flags, void *trace, const h2r_pow_layout &pl, uint64_t elem_stride, void *out,
                       uint8_t *status, void *workspace, hipStream_t st, const std::function<int32_t()> &after_chain,
                       u32 check_in_field = 1, bool assume_empty = false, const AuxArgs *witness_aux = nullptr, u32 witness_aux_lds = 0,
                       const void *e_limbs = nullptr, u32 e_num_limbs = 0, u32 exp_limb_bits = 0, const Sha256Args *sha = nullptr,
                       const AuxArgs *verify_aux = nullptr) {
    // verify_aux (nullable): the verifier's in-field + encoded-message witness of the whole call (what `after_chain` launches as a
    // kernel): when every sub-batch of the call goes out as a step launch, the chain role writes it element by element instead
    // sha (nullable): the SHA-256 / hashed-message step of RSASignatureVerifier for THIS call's messages; `after_chain` consumes
    // its output.  It rides on the call's first step launch when there is one, and is a kernel of its own on `st` otherwise.
    // witness_aux (nullable): what `after_chain` would launch, when that is a kernel whose output belongs to the call's
    // TRACE (the assert_in_field witness): a call issued as one-launch steps writes it together with its records
    // e_limbs (nullable): per-element variable exponents (BigIntChip::pow_mod, chip.rs:664-696) instead of the fixed e_le
    const h2r_ctx *ctx = p->ctx;
    ExpBits eb; u32 T;
    int32_t rc = H2R_OK;
    const u32 mode = e_limbs ? CHAIN_POW_VAR : CHAIN_POW_FIXED;
    if (e_limbs) { std::memset(&eb, 0, sizeof eb); T = pl.num_mul_mods; }
    else rc = exp_to_bits(e_le, e_len, &eb, &T);
    if (rc) return rc;
    const u64 e_bytes = (u64)e_num_limbs * ctx->layout.limb_bytes;   // per element
    const u32 slot = p->k % p->depth;
    p->done[slot] = DoneRef{};
    if (knobs().pipe_serialize && p->aux[0] != p->aux[1] && p->k > 0) {   // order this record stream behind the previous record kernel
        rc = pipeline_wait_slot(p, (p->k - 1) % p->depth, p->aux[p->k & 1]);
        if (rc) return rc;
    }
    // A large call may be walked as sub-batches, each with its own chain and record kernel (pipeline_plan): sub-batch
    // i+1's chain kernel then runs next to sub-batch i's record kernel INSIDE the call, exactly as consecutive calls do.
    // The sub-batches are slices of the caller's buffers and of the whole call's workspace plan ([batch*T][4][L],
    // element-major), so audits and emitters see one call.
    std::vector<u64> sizes; bool pace = false;
    const u32 nbits_all = e_limbs ? e_num_limbs * exp_limb_bits : eb.nbits;
    const u32 n_seg_single = assume_empty ? exp_segment_count(ctx, batch, nbits_all, trace && T, true) : 1;   // (a single stream-ordered call)
    // RSA-2048 on a pipeline created with TWO side streams and three or more buffer sets: the two-queue form -- chain kernels on the caller's
    // stream, record kernels alternating between the side streams, so that call k + 1's record kernel starts while call k's tail drains --
    // beats the one-launch step (same box, alternating: 5.41 / 5.41 M assigns/s against 5.18 / 5.32 M at 1,024 per call, 5.55-5.58 against
    // 5.50-5.51 M at 2,048).  The other step shapes were chain-bound enough to lose that way in round 4 (RSA-1024 with its four-wave chain 9.4
    // against 12.4 M, RSA-3072 2.1 against 2.5 M, RSA-4096 1.2 against 1.5 M; 128 x 32-bit limbs: the same): tools/two_queue_ab.sh,
    // profiles/r04_two_queue.txt.  [r6] RSA-1024 with the one-wave chain wins from 1,280 per call (two_queue_shape).
    // (calls of up to 2,048: at 4,096 per call one launch has little boundary left to hide and the step is ahead again, 5.37-5.42 against 5.27-5.33 M)
    // ... provided the three streams sit on three hardware queues, which the pipeline measures once per caller stream (pipeline_three_queues);
    // with a shared queue the call falls back to the one-launch step
    const bool overlap_records = two_queue_shape(ctx, batch) && p->aux[0] != p->aux[1] && p->depth >= 3 && knobs().pipe_step < 1 &&
                                 pipeline_three_queues(p, st);
    const bool as_steps = step_eligible(ctx, batch, trace, T) && n_seg_single <= 1 && !overlap_records;
    if (p->pending && (!as_steps || p->pending_st != st)) {   // the records still owed go out alone, `st` behind them
        rc = pipeline_flush(p, st);
        if (rc) return rc;
    }
    if (as_steps) {
        // every sub-batch is one launch: its chains together with the records of the sub-batch before it (of this call or of
        // the previous one).  With records pending the call is not split; a call that starts a train is, so that its first,
        // exposed chain kernel is short (the sizes an empty pipeline gets)
        if (p->pending) sizes.push_back(batch); else call_plan(ctx, batch, false, sizes, pace);
        std::vector<u64> capped;
        for (u64 sz : sizes) {
            const u64 parts = (sz + kStepMax - 1) / kStepMax;
            const u64 each = round_up((sz + parts - 1) / parts, 256);
            for (u64 o2 = 0; o2 < sz; o2 += each) capped.push_back(std::min(each, sz - o2));
        }
        sizes.swap(capped);
    } else {
        pipeline_plan(p, batch, assume_empty, sizes, pace);
        if (overlap_records && ctx->L == 16 && batch > 4096 && knobs().pipe_sub_batch <= 0) {
            // RSA-1024 in the two-queue form: a call above 4,096 as uniform sub-batches of 2,048, unpaced (8,192 per call 15.1 -> 15.6-15.9 M assigns/s,
            // 16,384 13.9-14.1 -> 14.4-14.5 M; paced or as sub-batches of 4,096 it loses: profiles/r06_two_queue_rsa1024.txt)
            sizes.clear(); pace = false;
            for (u64 o2 = 0; o2 < batch; o2 += 2048) sizes.push_back(std::min<u64>(2048, batch - o2));
        }
    }
    const bool split = sizes.size() > 1;
    const h2r_layout &lo = ctx->layout;
    const Workspace wp = workspace_plan(lo.limb_bytes, ctx->L, batch, T ? T : 1);
    u8 *ws = reinterpret_cast<u8 *>(round_up(reinterpret_cast<u64>(workspace), 256));
    const u64 in_bytes = (u64)ctx->K * 4, ws_elem = (u64)(T ? T : 1) * 4 * ctx->L * lo.limb_bytes;
    auto launch_sha_alone = [&]() -> int32_t {
        if (!sha || sha->batch == 0) return H2R_OK;
        ProfScope ps(H2R_KERNEL_SHA256, st, true);
        hipExtLaunchKernelGGL(sha256_kernel, dim3((unsigned)((sha->batch + 63) / 64)), dim3(64), 0, st, ps.a, ps.on ? ps.b : nullptr, 0, *sha);
        HIP_TRY(hipGetLastError());
        return H2R_OK;
    };
    if (!as_steps || !p->pending) {   // no step launch to ride on (two-queue form, or a call that starts a train with a chain kernel)
        rc = launch_sha_alone();
        if (rc) return rc;
    }
    if (as_steps) {
        const bool aux_as_role = witness_aux && witness_aux->batch && witness_aux_lds <= step_shared_bytes(ctx);
        bool aux_done = false;
        bool sha_done = !sha || !p->pending;
        // the verifier's witness inside the chain role: only when the whole call is step launches (a call that starts a train keeps
        // the kernel behind it) and the roles' LDS holds its staging.  When this very launch's SHA role produces the hashed limbs
        // the chain role waits for that role's message count (Sha256Args::done / target)
        const AuxGeom vg(ctx->L, lo.limb_width);
        // measured per shape (tools/verify_fold_ab.sh, ms per 1,024-signature step, digests / messages): RSA-2048 0.2040 / 0.2079 -> 0.1973 /
        // 0.1989 with the fold; the chain-bound shapes do not gain -- RSA-1024 0.0975 / 0.1097 -> 0.0988 / 0.1080, RSA-4096 0.6745 / 0.6854 ->
        // 0.6773 / 0.6795 -- or lose: RSA-3072 0.4195 / 0.4258 -> 0.4381 / 0.4601 (the witness lengthens the role the launch waits for)
        const bool fold_shape = knobs().verify_fold >= 0 ? knobs().verify_fold != 0 : ctx->L == 32;
        const bool fold_verify = fold_shape && verify_aux && verify_aux->batch && p->pending && (!sha || p->sha_done_dev) && vg.in_field_sz() + vg.em_sz() <= step_shared_bytes(ctx);
        u64 off = 0;
        for (size_t i = 0; i < sizes.size(); off += sizes[i], ++i) {
            const u64 nb = sizes[i];
            const u8 *xs = static_cast<const u8 *>(x) + off * in_bytes;
            const u8 *ns = static_cast<const u8 *>(n) + ((flags & H2R_F_SHARED_MODULUS) ? 0 : off * in_bytes);
            PathArgs pa;
            rc = run_path(ctx, mode, xs, nullptr, ns, e_limbs ? static_cast<const u8 *>(e_limbs) + off * e_bytes : nullptr, e_num_limbs, exp_limb_bits,
                          e_limbs ? nullptr : &eb, check_in_field, nb, flags, T,
                          static_cast<u8 *>(trace) + off * elem_stride, elem_stride, pl.off_records, &pl,
                          out ? static_cast<u8 *>(out) + off * in_bytes : nullptr, status + off, split ? ws + off * ws_elem : workspace,
                          st, p->aux[0], nullptr, nullptr, nullptr, split ? ws + wp.off_pre : nullptr, &pa,
                          split ? ws + wp.off_n + off * in_bytes : nullptr);
            if (rc) return rc;
            if (!pa.has_trace) return H2R_E_SHAPE;
            if (p->pending) {
                // THIS call's assert_in_field witness rides on the first step launch the call issues: it needs only x and n,
                // which are therefore read in `stream` order inside the call, like every other input
                const bool with_aux = aux_as_role && !aux_done;
                ProfScope ps(H2R_KERNEL_STEP, st, true);
                AuxArgs va;
                if (fold_verify) {   // the slice of the call this launch's chains cover
                    va = *verify_aux;
                    va.x = xs; va.n = ns;
                    va.hashed = verify_aux->hashed + off * 4;
                    va.powed = static_cast<const u8 *>(verify_aux->powed) + off * in_bytes;
                    va.batch = nb;
                    va.trace = verify_aux->trace + off * elem_stride;
                    va.is_valid = verify_aux->is_valid ? verify_aux->is_valid + off : nullptr;
                    va.status = verify_aux->status + off;
                }
                Sha256Args sr;
                u32 sha_target = p->sha_issued;
                if (!sha_done) {
                    sr = *sha;
                    if (fold_verify) { sha_target += (u32)sha->batch; sr.done = p->sha_done_dev; sr.target = sha_target; }
                }
                HIP_TRY(launch_step(ctx, pa.ca, p->pending_ta, with_aux ? witness_aux : nullptr, fold_verify ? &va : nullptr, sha_done ? nullptr : &sr, st, ps.a, ps.b));
                p->sha_issued = sha_target;   // (only a launch that went out counts: the device word never runs behind the host's target)
                aux_done = aux_done || with_aux;
                sha_done = true;
            } else {
                ProfScope ps(H2R_KERNEL_CHAIN, st, true);
                HIP_TRY(launch_chain(ctx, pa.ca, false, st, ps.a, ps.b));
            }
            p->pending = true; p->pending_ta = pa.ta; p->pending_st = st;
        }
        p->done_stream[slot] = st;
        p->k += 1;
        if (!aux_done && !fold_verify) rc = after_chain();   // (a call that starts a train as one chain kernel: the in-field kernel behind it)
        if (rc) return rc;
        for (; p->joined + p->depth <= p->k; ++p->joined) {   // calls issued the two-queue way earlier on
            rc = pipeline_wait_slot(p, p->joined % p->depth, st);
            if (rc) return rc;
        }
        return H2R_OK;
    }
    const u32 n_seg = n_seg_single > 1 ? n_seg_single : (sizes.size() == 1 ? exp_segment_count(ctx, batch, nbits_all, trace && T) : 1);
    if (n_seg > 1) {
        std::vector<ExpSegment> segs;
        exp_segment_plan(n_seg, nbits_all, e_limbs ? nullptr : &eb, segs);
        for (u32 sgi = 0; sgi < n_seg; ++sgi) {
            const ExpSegment &sg = segs[sgi];
            const bool last = sgi + 1 == n_seg;
            DoneRef cur{};
            rc = run_path(ctx, mode, x, nullptr, n, e_limbs, e_num_limbs, exp_limb_bits, e_limbs ? nullptr : &eb, check_in_field, batch, flags, T,
                          trace, elem_stride, pl.off_records, &pl, out, status, workspace, st, p->aux[p->k & 1], p->chain_done[slot],
                          last ? p->trace_done[slot] : p->sub_done[sgi & 1], last ? &p->done[slot] : &cur, nullptr, nullptr, nullptr, &sg);
            if (rc) return rc;
        }
        p->done_stream[slot] = p->aux[p->k & 1];
        p->k += 1;
        rc = after_chain();
        if (rc) return rc;
        for (; p->joined + p->depth <= p->k; ++p->joined) {
            rc = pipeline_wait_slot(p, p->joined % p->depth, st);
            if (rc) return rc;
        }
        return H2R_OK;
    }
    DoneRef prev{};   // the record kernel of the sub-batch before the current one
    u64 o = 0;
    for (size_t i = 0; i < sizes.size(); o += sizes[i], ++i) {
        const u64 nb = sizes[i];
        const bool last = i + 1 == sizes.size();
        DoneRef cur{};
        const u8 *xs = static_cast<const u8 *>(x) + o * in_bytes;
        const u8 *ns = static_cast<const u8 *>(n) + ((flags & H2R_F_SHARED_MODULUS) ? 0 : o * in_bytes);
        rc = run_path(ctx, mode, xs, nullptr, ns, e_limbs ? static_cast<const u8 *>(e_limbs) + o * e_bytes : nullptr, e_num_limbs, exp_limb_bits,
                      e_limbs ? nullptr : &eb, check_in_field, nb, flags, T,
                      static_cast<u8 *>(trace) + o * elem_stride, elem_stride, pl.off_records, &pl,
                      out ? static_cast<u8 *>(out) + o * in_bytes : nullptr, status + o, split ? ws + o * ws_elem : workspace,
                      st, p->aux[p->k & 1], p->chain_done[slot], last ? p->trace_done[slot] : p->sub_done[i & 1],
                      last ? &p->done[slot] : &cur, split ? ws + wp.off_pre : nullptr, nullptr,
                      split ? ws + wp.off_n + o * in_bytes : nullptr);
        if (rc) return rc;
        // paced: as consecutive calls are paced by the lazy join below, sub-batch i+1's chain kernel starts with sub-batch
        // i's record kernel, not earlier
        if (prev.ev && pace) {
            bool alive = true;
            if (prev.borrowed) { std::lock_guard<std::mutex> lk(g_prof_mu); alive = prev.gen == g_prof_gen; }
            if (alive) HIP_TRY(hipStreamWaitEvent(st, prev.ev, 0));
        }
        prev = cur;
    }
    p->done_stream[slot] = p->aux[p->k & 1];
    p->k += 1;
    rc = after_chain();
    if (rc) return rc;
    // lazy join: the NEXT call reuses the buffers of call k - depth, so order the user stream after that call's
    // record kernel now -- behind this call's chain kernel, which therefore overlaps the record kernels in flight
    for (; p->joined + p->depth <= p->k; ++p->joined) {
        rc = pipeline_wait_slot(p, p->joined % p->depth, st);
        if (rc) return rc;
    }
    return H2R_OK;
}

// Does a plain (non-pipelined) pow call of this size gain from being walked as overlapping sub-batches?  The shapes and
// sizes pipeline_plan splits for an empty pipeline.
bool plain_call_overlaps(const h2r_ctx *c, u64 batch) {
    if (knobs().plain_overlap == 0) return false;
    const u64 unit = (u64)c->num_cus * (c->K > 64 ? 2 : 4);
    if (c->layout.limb_width != 64) return false;
    if (c->L > 32) return batch > 3 * unit;
    return c->L > 16 && batch > unit + unit / 2;
}

int32_t overlapped_pow_fixed(const h2r_ctx *ctx, const void *x, const void *n, const uint8_t *e_le, size_t e_len, uint64_t batch,
                             uint32_t flags, void *trace, const h2r_pow_layout &pl, uint64_t elem_stride, void *out, uint8_t *status,
                             void *workspace, hipStream_t st, u32 check_in_field, u32 T,
                             const void *e_limbs, u32 e_num_limbs, u32 exp_limb_bits) {
    // (e_limbs: per-element variable exponents -- BigIntChip::pow_mod -- instead of e_le; only its long-exponent walk comes this way)
    H2R_ON_DEVICE(ctx->params.device);
    std::lock_guard<std::mutex> lk(ctx->pipe_mu);
    if (!ctx->pipe) {
        const int32_t rc = h2r_pipeline_create_ex(ctx, 2, 1, &ctx->pipe);
        if (rc) return rc;
    }
    ScratchGuard sg; sg.st = st;
    void *ws = workspace;
    if (!ws) {   // freed in stream order behind the join below, i.e. after the record kernels that read it
        HIP_TRY(hipMallocAsync(&sg.p, workspace_plan(ctx->layout.limb_bytes, ctx->L, batch, T).total, st));
        sg.owned = true; ws = sg.p;
    }
    const int32_t rc = pipeline_issue(ctx->pipe, x, n, e_le, e_len, batch, flags, trace, pl, elem_stride, out, status, ws, st,
                                      []() -> int32_t { return H2R_OK; }, check_in_field, true, nullptr, 0, e_limbs, e_num_limbs, exp_limb_bits);
    const int32_t rj = h2r_pipeline_join(ctx->pipe, st);   // also after a failed issue: whatever was queued is ordered
    return rc ? rc : rj;
}

AuxArgs verify_aux_args(const h2r_ctx *ctx, const void *sig, const void *n, const uint64_t *hashed, uint64_t batch, uint32_t flags,
                        void *trace, const h2r_verify_layout &vl, void *powed_out, uint8_t *is_valid_out, uint8_t *status) {
    AuxArgs aa;
    std::memset(&aa, 0, sizeof aa);
    aa.x = sig; aa.n = n; aa.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    aa.hashed = hashed; aa.powed = powed_out; aa.batch = batch; aa.L = ctx->L;
    aa.trace = static_cast<u8 *>(trace); aa.elem_stride = vl.elem_stride; aa.off_in_field = vl.off_in_field; aa.off_em = vl.off_em;
    aa.is_valid = is_valid_out; aa.status = status;
    return aa;
}
int32_t launch_verify_aux(const h2r_ctx *ctx, const void *sig, const void *n, const uint64_t *hashed, uint64_t batch, uint32_t flags,
                          void *trace, const h2r_verify_layout &vl, void *powed_out, uint8_t *is_valid_out, uint8_t *status, hipStream_t st) {
    const AuxArgs aa = verify_aux_args(ctx, sig, n, hashed, batch, flags, trace, vl, powed_out, is_valid_out, status);
    ProfScope ps(H2R_KERNEL_AUX, st, true);   // dispatch-stamped events: no marker packets on the caller's stream
    const AuxGeom ag(ctx->L, 64);
    hipExtLaunchKernelGGL((aux_kernel<64>), dim3((unsigned)batch), dim3(64), (unsigned)(ag.in_field_sz() + ag.em_sz()), st, ps.a, ps.b, 0, aa);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
}

// The assert_in_field(x, n) witness alone (src/chip.rs:106), one element every h2r_fresh_op_layout(IS_IN_FIELD) stride.
int32_t in_field_args(const h2r_ctx *ctx, const void *x, const void *n, uint64_t batch, uint32_t flags, void *in_field_trace, AuxArgs *aa, u32 *lds) {
    u64 es = 0;
    const int32_t rc = h2r_fresh_op_layout(ctx, FRESH_IS_IN_FIELD, &es, nullptr, nullptr);
    if (rc) return rc;
    std::memset(aa, 0, sizeof *aa);
    aa->x = x; aa->n = n; aa->n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    aa->batch = batch; aa->L = ctx->L;
    aa->trace = static_cast<u8 *>(in_field_trace); aa->elem_stride = es; aa->off_in_field = 0;
    const AuxGeom ag(ctx->L, ctx->layout.limb_width);
    *lds = (u32)(ag.in_field_sz() + ag.em_sz());
    return H2R_OK;
}
int32_t launch_in_field(const h2r_ctx *ctx, const void *x, const void *n, uint64_t batch, uint32_t flags, void *in_field_trace, hipStream_t st) {
    if (batch == 0) return H2R_OK;
    AuxArgs aa; u32 lds = 0;
    const int32_t rc = in_field_args(ctx, x, n, batch, flags, in_field_trace, &aa, &lds);
    if (rc) return rc;
    H2R_ON_DEVICE(ctx->params.device);
    ProfScope ps(H2R_KERNEL_AUX, st, true);
    if (ctx->layout.limb_width == 64) hipExtLaunchKernelGGL((aux_kernel<64>), dim3((unsigned)batch), dim3(64), lds, st, ps.a, ps.b, 0, aa);
    else hipExtLaunchKernelGGL((aux_kernel<32>), dim3((unsigned)batch), dim3(64), lds, st, ps.a, ps.b, 0, aa);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
}
}  // namespace

int32_t h2r_pipeline_modpow_public_key(h2r_pipeline *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len,
                                       uint64_t batch, uint32_t flags, void *trace, void *in_field_trace, void *out,
                                       uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!p || !trace || !workspace) return H2R_E_NULL;
    h2r_pow_layout pl;
    const int32_t rc = h2r_pow_fixed_layout(p->ctx, e_le, e_len, &pl);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the in-field witness needs only x and n: its kernel runs on the caller's stream right behind the chain kernel -- or,
    // when the call is issued as one-launch steps, as a role of the launch that writes the call's records
    AuxArgs aa; u32 aux_lds = 0;
    const bool have_aux = in_field_trace && batch && in_field_args(p->ctx, x, n, batch, flags, in_field_trace, &aa, &aux_lds) == H2R_OK;
    return pipeline_issue(p, x, n, e_le, e_len, batch, flags, trace, pl, pl.elem_stride, out, status, workspace, st,
                          [&]() -> int32_t {
                              if (!in_field_trace) return H2R_OK;
                              return launch_in_field(p->ctx, x, n, batch, flags, in_field_trace, st);
                          }, 1, false, have_aux ? &aa : nullptr, aux_lds);
} H2R_CATCH_STATUS

// RSAPubE::Var (src/chip.rs:108-110): per-element exponents; the same pipelining as the fixed-exponent form
int32_t h2r_pipeline_modpow_public_key_var(h2r_pipeline *p, const void *x, const void *e_limbs, uint32_t e_num_limbs, uint32_t exp_limb_bits,
                                           const void *n, uint64_t batch, uint32_t flags, void *trace, void *in_field_trace, void *out,
                                           uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!p || !trace || !workspace || !e_limbs) return H2R_E_NULL;
    h2r_pow_layout pl;
    const int32_t rc = h2r_pow_var_layout(p->ctx, e_num_limbs, exp_limb_bits, &pl);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    AuxArgs aa; u32 aux_lds = 0;
    const bool have_aux = in_field_trace && batch && in_field_args(p->ctx, x, n, batch, flags, in_field_trace, &aa, &aux_lds) == H2R_OK;
    return pipeline_issue(p, x, n, nullptr, 0, batch, flags, trace, pl, pl.elem_stride, out, status, workspace, st,
                          [&]() -> int32_t {
                              if (!in_field_trace) return H2R_OK;
                              return launch_in_field(p->ctx, x, n, batch, flags, in_field_trace, st);
                          }, 1, false, have_aux ? &aa : nullptr, aux_lds, e_limbs, e_num_limbs, exp_limb_bits);
} H2R_CATCH_STATUS

int32_t h2r_pipeline_verify_pkcs1v15(h2r_pipeline *p, const void *sig, const void *n, const uint8_t *e_le, size_t e_len,
                                     const uint64_t *hashed, uint64_t batch, uint32_t flags, void *trace, void *powed_out,
                                     uint8_t *is_valid_out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!p || !sig || !n || !hashed || !trace || !powed_out || !status || !workspace) return H2R_E_NULL;
    h2r_verify_layout vl;
    const int32_t rc = h2r_verify_layout_fixed(p->ctx, e_le, e_len, &vl);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the in-field / encoded-message kernel needs only the chain's result: it runs on the caller's stream right
    // behind the chain kernel and writes the element's in-field and EM regions (disjoint from the records)
    const AuxArgs va = verify_aux_args(p->ctx, sig, n, hashed, batch, flags, trace, vl, powed_out, is_valid_out, status);
    return pipeline_issue(p, sig, n, e_le, e_len, batch, flags, trace, vl.pow, vl.elem_stride, powed_out, status, workspace, st,
                          [&]() -> int32_t {
                              if (batch == 0) return H2R_OK;
                              return launch_verify_aux(p->ctx, sig, n, hashed, batch, flags, trace, vl, powed_out, is_valid_out, status, st);
                          }, 1, false, nullptr, 0, nullptr, 0, 0, nullptr, &va);
} H2R_CATCH_STATUS

// RSAPubE::Var arm of the pipelined verifier (src/chip.rs:108-110)
int32_t h2r_pipeline_verify_pkcs1v15_var(h2r_pipeline *p, const void *sig, const void *n, const void *e_limbs, uint32_t e_num_limbs,
                                         uint32_t exp_limb_bits, const uint64_t *hashed, uint64_t batch, uint32_t flags, void *trace,
                                         void *powed_out, uint8_t *is_valid_out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!p || !sig || !n || !e_limbs || !hashed || !trace || !powed_out || !status || !workspace) return H2R_E_NULL;
    h2r_verify_layout vl;
    const int32_t rc = h2r_verify_layout_var(p->ctx, e_num_limbs, exp_limb_bits, &vl);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const AuxArgs va = verify_aux_args(p->ctx, sig, n, hashed, batch, flags, trace, vl, powed_out, is_valid_out, status);
    return pipeline_issue(p, sig, n, nullptr, 0, batch, flags, trace, vl.pow, vl.elem_stride, powed_out, status, workspace, st,
                          [&]() -> int32_t {
                              if (batch == 0) return H2R_OK;
                              return launch_verify_aux(p->ctx, sig, n, hashed, batch, flags, trace, vl, powed_out, is_valid_out, status, st);
                          }, 1, false, nullptr, 0, e_limbs, e_num_limbs, exp_limb_bits, nullptr, &va);
} H2R_CATCH_STATUS

// RSASignatureVerifier::verify_pkcs1v15_signature (src/lib.rs:183-246) as a pipelined call: the SHA-256 / hashed-message step of this
// call's messages is a role of the call's step launch (hidden next to the records of the previous call) and the chain role of the same
// launch writes the in-field / encoded-message witness from its limbs; where the call is not issued as steps, both are kernels on `stream`.
int32_t h2r_pipeline_signature_verifier(h2r_pipeline *p, const uint8_t *msgs, const uint64_t *msg_off, uint64_t fixed_len, const void *sig,
                                        const void *n, const uint8_t *e_le, size_t e_len, uint64_t batch, uint32_t flags, void *trace,
                                        void *hm_trace, uint64_t hm_stride, uint8_t *digest_out, uint64_t *hashed_out, void *powed_out,
                                        uint8_t *is_valid_out, uint8_t *status, void *workspace, h2r_stream_t stream) try {
    if (!p || !sig || !n || !hashed_out || !trace || !powed_out || !status || !workspace || (!msgs && (msg_off || fixed_len))) return H2R_E_NULL;
    if (hm_trace && hm_stride == 0) hm_stride = HM_REGION;
    if ((reinterpret_cast<u64>(digest_out) | reinterpret_cast<u64>(hashed_out) | reinterpret_cast<u64>(hm_trace) | hm_stride) & 15) return H2R_E_SHAPE;
    if (hm_trace && hm_stride < HM_REGION) return H2R_E_SHAPE;
    h2r_verify_layout vl;
    const int32_t rc = h2r_verify_layout_fixed(p->ctx, e_le, e_len, &vl);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Sha256Args sa;
    sa.msgs = msgs; sa.off = msg_off; sa.fixed_len = fixed_len; sa.batch = batch;
    sa.digest = digest_out; sa.hashed = hashed_out; sa.region = static_cast<u8 *>(hm_trace); sa.region_stride = hm_stride;
    sa.done = nullptr; sa.target = 0;
    const AuxArgs va = verify_aux_args(p->ctx, sig, n, hashed_out, batch, flags, trace, vl, powed_out, is_valid_out, status);
    return pipeline_issue(p, sig, n, e_le, e_len, batch, flags, trace, vl.pow, vl.elem_stride, powed_out, status, workspace, st,
                          [&]() -> int32_t {
                              if (batch == 0) return H2R_OK;
                              return launch_verify_aux(p->ctx, sig, n, hashed_out, batch, flags, trace, vl, powed_out, is_valid_out, status, st);
                          }, 1, false, nullptr, 0, nullptr, 0, 0, &sa, &va);
} H2R_CATCH_STATUS

int32_t h2r_fresh_op_layout(const h2r_ctx *ctx, uint32_t op, uint64_t *elem_stride, uint64_t *stream_bytes, uint32_t *value_limbs) try {
    if (!ctx) return H2R_E_NULL;
    if (op >= FRESH_OP_COUNT) return H2R_E_UNSUPPORTED;
    if (ctx->L + 3 > 64 * AUX_V) return H2R_E_UNSUPPORTED;
    const AuxGeom g(ctx->L, ctx->layout.limb_width);
    u64 dev = 0, sb = 0;
    fresh_sections(g, op, [&](u64 off, u64 len) { sb += len; dev = off + len; });
    if (elem_stride) *elem_stride = round_up(dev, 256);
    if (stream_bytes) *stream_bytes = sb;
    if (value_limbs) *value_limbs = (op == FRESH_ADD || op == FRESH_SUB) ? ctx->L + 1 : ((op == FRESH_ADD_MOD || op == FRESH_SUB_MOD) ? ctx->L : 0);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_fresh_op_batch(const h2r_ctx *ctx, uint32_t op, const void *a, const void *b, const void *n, uint64_t batch,
                           uint32_t flags, void *trace, void *value_out, uint8_t *flag_out, uint8_t *status, h2r_stream_t stream) try {
    if (!ctx || !a || !trace || !status) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    u64 es; u32 vl;
    int32_t rc = h2r_fresh_op_layout(ctx, op, &es, nullptr, &vl);
    if (rc) return rc;
    const bool needs_b = op != FRESH_IS_ZERO, needs_n = op == FRESH_ADD_MOD || op == FRESH_SUB_MOD;
    if ((needs_b && !b) || (needs_n && !n)) return H2R_E_NULL;
    if (batch == 0) return H2R_OK;
    FreshArgs fa;
    std::memset(&fa, 0, sizeof fa);
    fa.a = a; fa.b = b; fa.n = n; fa.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    // the ops without an `n` (comparisons, is_in_field(a, b = modulus)): the flag says that `b` is one integer shared by the batch
    fa.b_stride = (!needs_n && (flags & H2R_F_SHARED_MODULUS)) ? 0 : ctx->L;
    fa.batch = batch; fa.L = ctx->L; fa.op = op; fa.trace = static_cast<u8 *>(trace); fa.elem_stride = es;
    fa.value_out = value_out; fa.value_limbs = vl; fa.flag_out = flag_out; fa.status = status;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps(H2R_KERNEL_AUX, st);
    if (ctx->layout.limb_width == 64) hipLaunchKernelGGL((fresh_kernel<64>), dim3((unsigned)batch), dim3(64), 0, st, fa);
    else hipLaunchKernelGGL((fresh_kernel<32>), dim3((unsigned)batch), dim3(64), 0, st, fa);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_fresh_op_flatten(const h2r_ctx *ctx, uint32_t op, const void *elem_host, void *stream_out) try {
    if (!ctx || !elem_host || !stream_out) return H2R_E_NULL;
    if (op >= FRESH_OP_COUNT) return H2R_E_UNSUPPORTED;
    const u8 *e = static_cast<const u8 *>(elem_host);
    u8 *o = static_cast<u8 *>(stream_out);
    const AuxGeom g(ctx->L, ctx->layout.limb_width);
    fresh_sections(g, op, [&](u64 off, u64 len) { std::memcpy(o, e + off, len); o += len; });
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_range_decompose_batch(const h2r_ctx *ctx, const void *values, uint32_t value_bytes, uint64_t count,
                                  uint32_t bit_len, uint32_t sublimb_bits, uint8_t *sublimbs_out,
                                  uint32_t sub_stride, uint32_t *hist, h2r_stream_t stream) try {
    if (!ctx || !values) return H2R_E_NULL;
    if ((value_bytes != 8 && value_bytes != 16) || bit_len == 0 || bit_len > 8 * value_bytes || sublimb_bits == 0 ||
        sublimb_bits > 8)
        return H2R_E_SHAPE;
    DecompArgs da;
    std::memset(&da, 0, sizeof da);
    da.values = static_cast<const u8 *>(values); da.value_bytes = value_bytes; da.count = count;
    da.bit_len = bit_len; da.sub_bits = sublimb_bits;
    da.has_ov = bit_len % sublimb_bits ? 1 : 0; da.nsub = bit_len / sublimb_bits + da.has_ov;
    if (sublimbs_out && sub_stride < da.nsub) return H2R_E_SHAPE;
    da.sub_out = sublimbs_out; da.sub_stride = sub_stride; da.hist = hist; da.comp_len = 1u << sublimb_bits;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (count == 0) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    u64 blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const u32 shmem = (da.comp_len + 256) * sizeof(u32);
    hipLaunchKernelGGL(decompose_kernel, dim3((unsigned)blocks), dim3(256), shmem, static_cast<hipStream_t>(stream), da);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

uint32_t h2r_hist_len(const h2r_ctx *ctx) try { return ctx ? ctx->hist_len : 0; } H2R_CATCH_ZERO

int32_t h2r_trace_lookup_hist(const h2r_ctx *ctx, const void *trace, uint64_t first_record_off, uint64_t elem_stride,
                              uint64_t num_elems, uint32_t records_per_elem, uint32_t *hist_out, h2r_stream_t stream) try {
    if (!ctx || !trace || !hist_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (num_elems == 0) return H2R_OK;
    const h2r_layout &lo = ctx->layout;
    HistArgs ha;
    std::memset(&ha, 0, sizeof ha);
    ha.trace = static_cast<const u8 *>(trace); ha.first_record_off = first_record_off; ha.elem_stride = elem_stride;
    ha.record_stride = lo.record_stride; ha.num_elems = num_elems; ha.records_per_elem = records_per_elem;
    ha.off_q_sub = lo.plane_off[H2R_PL_Q_SUB]; ha.off_r_sub = lo.plane_off[H2R_PL_R_SUB];
    ha.off_carry_sub = lo.plane_off[H2R_PL_CARRY_SUB];
    ha.L = lo.num_limbs; ha.C = lo.num_cols; ha.carry_nsub = lo.carry_nsub; ha.carry_sub_stride = lo.carry_sub_stride;
    ha.carry_has_ov = (lo.carry_bits % lo.carry_sub_bits) ? 1 : 0;
    ha.tab0_len = ctx->tab0_len; ha.tab1_off = ctx->tab1_off; ha.tab1_len = ctx->tab1_len;
    ha.tab2_off = ctx->tab2_off; ha.tab2_len = ctx->tab2_len; ha.hist_len = ctx->hist_len;
    ha.hist = hist_out;
    H2R_ON_DEVICE(ctx->params.device);
    ProfScope ps(H2R_KERNEL_HIST, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)num_elems), dim3(256), ctx->hist_len * sizeof(u32),
                       static_cast<hipStream_t>(stream), ha);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

uint32_t h2r_lookups_per_record(const h2r_ctx *ctx) try {
    if (!ctx) return 0;
    const h2r_layout &lo = ctx->layout;
    return 2 * lo.num_limbs * lo.limb_nsub + (lo.num_cols - 1) * lo.carry_nsub;
} H2R_CATCH_ZERO

int32_t h2r_trace_lookup_permutation(const h2r_ctx *ctx, const void *trace, uint64_t first_record_off, uint64_t elem_stride,
                                     uint64_t num_elems, uint32_t records_per_elem, uint32_t *perm_out, uint16_t *rows_out,
                                     h2r_stream_t stream) try {
    return h2r_trace_lookup_permutation_hist(ctx, trace, first_record_off, elem_stride, num_elems, records_per_elem, perm_out,
                                             rows_out, nullptr, stream);
} H2R_CATCH_STATUS

int32_t h2r_trace_lookup_permutation_hist(const h2r_ctx *ctx, const void *trace, uint64_t first_record_off, uint64_t elem_stride,
                                          uint64_t num_elems, uint32_t records_per_elem, uint32_t *perm_out, uint16_t *rows_out,
                                          uint32_t *hist_out, h2r_stream_t stream) try {
    if (!ctx || !trace || !perm_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (num_elems == 0 || records_per_elem == 0) return H2R_OK;
    const h2r_layout &lo = ctx->layout;
    if (ctx->hist_len > (u32)PERM_MAX_ROWS || lo.limb_nsub != 8 || lo.carry_sub_stride % 16) return H2R_E_UNSUPPORTED;
    if ((2ull * lo.num_limbs * 8 + (u64)(lo.num_cols - 1) * lo.carry_sub_stride) / 16 > (u64)PERM_STAGE_U4) return H2R_E_UNSUPPORTED;
    PermArgs pa;
    std::memset(&pa, 0, sizeof pa);
    HistArgs &ha = pa.h;
    ha.trace = static_cast<const u8 *>(trace); ha.first_record_off = first_record_off; ha.elem_stride = elem_stride;
    ha.record_stride = lo.record_stride; ha.num_elems = num_elems; ha.records_per_elem = records_per_elem;
    ha.off_q_sub = lo.plane_off[H2R_PL_Q_SUB]; ha.off_r_sub = lo.plane_off[H2R_PL_R_SUB];
    ha.off_carry_sub = lo.plane_off[H2R_PL_CARRY_SUB];
    ha.L = lo.num_limbs; ha.C = lo.num_cols; ha.carry_nsub = lo.carry_nsub; ha.carry_sub_stride = lo.carry_sub_stride;
    ha.carry_has_ov = (lo.carry_bits % lo.carry_sub_bits) ? 1 : 0;
    ha.tab0_len = ctx->tab0_len; ha.tab1_off = ctx->tab1_off; ha.tab1_len = ctx->tab1_len;
    ha.tab2_off = ctx->tab2_off; ha.tab2_len = ctx->tab2_len; ha.hist_len = ctx->hist_len;
    pa.cells_per_record = h2r_lookups_per_record(ctx);
    const u64 n_cells = (u64)pa.cells_per_record * records_per_elem;
    if (n_cells >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    pa.n_cells = (u32)n_cells; pa.perm = perm_out; pa.rows = rows_out;
    ha.hist = hist_out;   // nullable: the rows' multiplicities fall out of the counting pass (= h2r_trace_lookup_hist's output)
    H2R_ON_DEVICE(ctx->params.device);
    const u64 stage_bytes = 4ull * ((2ull * lo.num_limbs * 8 + (u64)(lo.num_cols - 1) * lo.carry_sub_stride) / 16) * 16;
    const u64 same_bytes = perm_same_bytes(ctx->hist_len);   // match masks: per wave, group and table row
    const u64 staged_bytes = stage_bytes + ((2ull * n_cells + 15) & ~15ull) + same_bytes;
    // staged form: 16-bit cell ids in LDS (two workgroups per CU still fit next to the 10 KB of static tables)
    if (n_cells < 65536 && staged_bytes <= 68 * 1024) {
        if (staged_bytes > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&perm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)staged_bytes);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(perm_kernel<true>, dim3((unsigned)num_elems), dim3(256), (unsigned)staged_bytes, static_cast<hipStream_t>(stream), pa);
    } else
        hipLaunchKernelGGL(perm_kernel<false>, dim3((unsigned)num_elems), dim3(256), (unsigned)(stage_bytes + same_bytes), static_cast<hipStream_t>(stream), pa);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- halo2's lookup argument: table, per-argument multiplicities, permuted columns (h2r_lookup.hpp) ---------------------
int32_t h2r_lookup_config_custom(const uint32_t *bit_lens, const uint32_t *tags, uint32_t n, h2r_lookup_config *out) try {
    if (!bit_lens || !tags || !out) return H2R_E_NULL;
    std::memset(out, 0, sizeof *out);
    // RangeChip::configure: sort, de-duplicate, drop zero entries
    std::vector<std::pair<u32, u32>> v;
    for (u32 i = 0; i < n; ++i) {
        if (!bit_lens[i]) continue;
        bool dup = false;
        for (auto &e : v) if (e.first == bit_lens[i]) { dup = true; if (e.second != tags[i]) return H2R_E_SHAPE; }
        if (!dup) v.emplace_back(bit_lens[i], tags[i]);
    }
    std::sort(v.begin(), v.end());
    if (v.empty() || v.size() > H2R_LOOKUP_MAX_LENS) return H2R_E_SHAPE;
    u64 off = 1;
    for (size_t i = 0; i < v.size(); ++i) {
        if (v[i].first > 10 || v[i].second == 0) return H2R_E_SHAPE;   // tag 0 is the lookup-off row
        for (size_t j = 0; j < i; ++j) if (v[j].second == v[i].second) return H2R_E_SHAPE;
        out->bit_len[i] = v[i].first; out->tag[i] = v[i].second; out->row_off[i] = (u32)off;
        off += 1ull << v[i].first;
    }
    if (off > (u64)LOOKUP_MAX_ROWS) return H2R_E_UNSUPPORTED;
    out->n_lens = (u32)v.size(); out->n_rows = (u32)off;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_lookup_config_default(const h2r_ctx *ctx, uint32_t rsa_chip, h2r_lookup_config *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    u32 comp[4] = {0, 0, 0, 0}, over[3] = {0, 0, 0};
    compute_range_lens(ctx->layout.limb_width, ctx->L, comp, over);      // big_integer/chip.rs:1220-1249
    if (rsa_chip) comp[3] = 32 / kNumLookupLimbs;                         // src/chip.rs:252
    u32 lens[7] = {comp[0], comp[1], comp[2], comp[3], over[0], over[1], over[2]}, uniq[7], tags[7], n = 0;
    std::sort(lens, lens + 7);
    for (u32 i = 0; i < 7; ++i) if (lens[i] && (n == 0 || uniq[n - 1] != lens[i])) { uniq[n] = lens[i]; tags[n] = n + 1; ++n; }
    return h2r_lookup_config_custom(uniq, tags, n, out);
} H2R_CATCH_STATUS

int32_t h2r_lookup_table_image(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint64_t *tag_col, uint64_t *value_col) try {
    if (!ctx || !cfg || !tag_col || !value_col) return H2R_E_NULL;
    if (cfg->n_rows == 0 || cfg->n_rows > (u32)LOOKUP_MAX_ROWS) return H2R_E_SHAPE;
    std::memset(tag_col, 0, (size_t)cfg->n_rows * 32); std::memset(value_col, 0, (size_t)cfg->n_rows * 32);
    const bool mont = (ctx->repr.flags & H2R_ADVICE_MONTGOMERY) != 0;
    for (u32 i = 0; i < cfg->n_lens; ++i) {
        const Fe tg = mont ? fe_to_mont(fe_small(cfg->tag[i]), ctx->fc) : fe_small(cfg->tag[i]);   // small integers: canonical as they are
        for (u32 v = 0; v < (1u << cfg->bit_len[i]); ++v) {
            const Fe vv = mont ? fe_to_mont(fe_small(v), ctx->fc) : fe_small(v);
            for (int k = 0; k < 4; ++k) { tag_col[(u64)(cfg->row_off[i] + v) * 4 + k] = tg.v[k]; value_col[(u64)(cfg->row_off[i] + v) * 4 + k] = vv.v[k]; }
        }
    }
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
// sub-limb shape of RangeChip::assign(value, s, bit_len) and the table rows its lookups hit
int32_t range_shape(const h2r_lookup_config &cfg, u32 bit_len, u32 s, RangeShape *out) {
    if (!s || !bit_len || s > 8) return H2R_E_SHAPE;
    RangeShape r;
    r.sub_bits = s; r.ov_bits = bit_len % s; r.nsub = bit_len / s + (r.ov_bits ? 1 : 0);
    r.row_comp = r.row_ov = 0;
    if (r.nsub == 0 || r.nsub > 16) return H2R_E_SHAPE;
    bool fc = false, fo = r.ov_bits == 0;
    for (u32 i = 0; i < cfg.n_lens; ++i) {
        if (cfg.bit_len[i] == s) { r.row_comp = cfg.row_off[i]; fc = true; }
        if (r.ov_bits && cfg.bit_len[i] == r.ov_bits) { r.row_ov = cfg.row_off[i]; fo = true; }
    }
    if (!fc || !fo) return H2R_E_SHAPE;   // RangeChip::assign panics: no table for that bit length
    *out = r;
    return H2R_OK;
}
}  // namespace

int32_t h2r_lookup_hist_records(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const void *trace, uint64_t first_record_off,
                                uint64_t elem_stride, uint64_t num_elems, uint32_t records_per_elem, const uint8_t *status,
                                uint32_t *hist, h2r_stream_t stream) try {
    if (!ctx || !cfg || !trace || !hist) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (cfg->n_rows == 0 || cfg->n_rows > (u32)LOOKUP_MAX_ROWS) return H2R_E_SHAPE;
    const h2r_layout &lo = ctx->layout;
    if (lo.limb_nsub != 8 || lo.carry_nsub > 16) return H2R_E_UNSUPPORTED;
    LookupHistArgs a;
    std::memset(&a, 0, sizeof a);
    int32_t rc = range_shape(*cfg, lo.limb_width, lo.limb_sub_bits, &a.limb);
    if (rc) return rc;
    rc = range_shape(*cfg, lo.carry_bits, lo.carry_sub_bits, &a.carry);
    if (rc) return rc;
    if (num_elems == 0 || records_per_elem == 0) return H2R_OK;
    a.trace = static_cast<const u8 *>(trace); a.first_record_off = first_record_off; a.elem_stride = elem_stride;
    a.record_stride = lo.record_stride; a.num_elems = num_elems; a.records_per_elem = records_per_elem;
    a.off_q_sub = lo.plane_off[H2R_PL_Q_SUB]; a.off_r_sub = lo.plane_off[H2R_PL_R_SUB]; a.off_carry_sub = lo.plane_off[H2R_PL_CARRY_SUB];
    a.L = lo.num_limbs; a.C = lo.num_cols; a.carry_sub_stride = lo.carry_sub_stride;
    a.n_rows = cfg->n_rows; a.hist = hist; a.status = status;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps(H2R_KERNEL_HIST, st);
    hipLaunchKernelGGL(lookup_hist_records_kernel, dim3((unsigned)num_elems), dim3(256), LOOKUP_ARGS * cfg->n_rows * sizeof(u32), st, a);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
int32_t lookup_hist_values_impl(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const void *values, uint32_t value_bytes,
                                uint64_t values_per_elem, uint64_t num_elems, uint64_t elem_stride, uint64_t value_stride,
                                uint32_t bit_len, uint32_t sublimb_bits, const uint8_t *status, uint32_t *hist, h2r_stream_t stream) {
    if (!ctx || !cfg || !values || !hist) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (cfg->n_rows == 0 || cfg->n_rows > (u32)LOOKUP_MAX_ROWS) return H2R_E_SHAPE;
    if ((value_bytes != 4 && value_bytes != 8 && value_bytes != 16) || bit_len > 8 * value_bytes) return H2R_E_SHAPE;
    const u32 align = value_bytes == 4 ? 4 : 8;
    if (value_stride < value_bytes || (value_stride % align) || (elem_stride % align) || (reinterpret_cast<u64>(values) % align)) return H2R_E_SHAPE;
    LookupValuesArgs a;
    std::memset(&a, 0, sizeof a);
    const int32_t rc = range_shape(*cfg, bit_len, sublimb_bits, &a.shape);
    if (rc) return rc;
    if (num_elems == 0 || values_per_elem == 0) return H2R_OK;
    a.values = static_cast<const u8 *>(values); a.value_bytes = value_bytes; a.values_per_elem = values_per_elem; a.num_elems = num_elems;
    a.elem_stride = elem_stride; a.value_stride = value_stride; a.status = status;
    a.n_rows = cfg->n_rows; a.hist = hist;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps(H2R_KERNEL_HIST, st);
    hipLaunchKernelGGL(lookup_hist_values_kernel, dim3((unsigned)num_elems), dim3(256), LOOKUP_ARGS * cfg->n_rows * sizeof(u32), st, a);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
}
}  // namespace

int32_t h2r_lookup_hist_values(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const void *values, uint32_t value_bytes,
                               uint64_t values_per_elem, uint64_t num_elems, uint32_t bit_len, uint32_t sublimb_bits,
                               uint32_t *hist, h2r_stream_t stream) try {
    return lookup_hist_values_impl(ctx, cfg, values, value_bytes, values_per_elem, num_elems, values_per_elem * value_bytes, value_bytes,
                                   bit_len, sublimb_bits, nullptr, hist, stream);
} H2R_CATCH_STATUS

int32_t h2r_lookup_hist_values_strided(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const void *values, uint32_t value_bytes,
                                       uint64_t values_per_elem, uint64_t num_elems, uint64_t elem_stride, uint64_t value_stride,
                                       uint32_t bit_len, uint32_t sublimb_bits, const uint8_t *status, uint32_t *hist,
                                       h2r_stream_t stream) try {
    return lookup_hist_values_impl(ctx, cfg, values, value_bytes, values_per_elem, num_elems, elem_stride, value_stride, bit_len,
                                   sublimb_bits, status, hist, stream);
} H2R_CATCH_STATUS

extern "C++" {
namespace {
int32_t lookup_hist_fresh_impl(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint32_t op, const void *trace, uint64_t first_off,
                               uint64_t elem_stride, uint64_t num_elems, const uint8_t *status, uint32_t *hist, h2r_stream_t stream);
}  // namespace
}  // extern "C++"

// Every lookup of one verify_pkcs1v15_signature element that the witness holds (src/chip.rs:99-199): the range assigns inside
// assert_in_field, the q / r limbs and carries of every mul_mod record, and the two RangeChip::assign(half, 4, 32) of the
// encoded-message check (:170-171; the halves sit at bytes 12 and 24 of the EM region, aux_em).
int32_t h2r_lookup_hist_verify(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const h2r_verify_layout *vl, const void *trace,
                               uint64_t num_elems, const uint8_t *status, uint32_t *hist, h2r_stream_t stream) try {
    if (!ctx || !cfg || !vl || !trace || !hist) return H2R_E_NULL;
    if (vl->pow.off_records == UINT64_MAX) return H2R_E_SHAPE;   // h2r_verify_layout_compact: the q / r limbs and carries are not in this witness (h2r_lookup_hist_advice counts them from the image)
    int32_t rc = lookup_hist_fresh_impl(ctx, cfg, H2R_OP_IS_IN_FIELD, trace, vl->off_in_field, vl->elem_stride, num_elems, status, hist, stream);   // (a failed element counts nothing, as in the two passes below)
    if (!rc) rc = h2r_lookup_hist_records(ctx, cfg, trace, vl->pow.off_records, vl->elem_stride, num_elems, vl->pow.num_mul_mods, status, hist, stream);
    if (!rc) rc = lookup_hist_values_impl(ctx, cfg, static_cast<const u8 *>(trace) + vl->off_em + 12, 4, 2, num_elems, vl->elem_stride, 12, 32, 4,
                                          status, hist, stream);
    return rc;
} H2R_CATCH_STATUS

extern "C++" {
namespace {
// The RangeChip::assign entries of a Fresh-op region, in the region's own geometry (fresh_sections): add(n) holds two per step
// (c and the carry, chip.rs:279-282), sub_unchecked's difference one per limb (chip.rs:1307-1308).
template <typename F>
void fresh_range_runs(const AuxGeom &g, u32 op, F &&run) {
    u64 off = 0;
    const u32 L = g.L;
    auto add = [&](u32 n) { run(off + 2 * g.SB, n, g.STEP); run(off + 2 * g.SB + g.RA, n, g.STEP); off += g.add_sz(n); };
    auto eq = [&](u32 n) { off += g.eq_sz(n); };
    auto subu = [&](u32 n1) { run(off, n1, g.RA); off += g.cl_sz(n1); add(n1); eq(n1 + 1); };
    auto sub = [&](u32 nA, u32 nB) {
        const u32 m = nA > nB ? nA : nB, n1 = m + 1;
        add(m); subu(n1);
        off += 16; off += AuxGeom::a16((u64)n1 * g.LB); off += AuxGeom::a16((u64)m * g.LB);
        subu(n1);
    };
    auto lt = [&]() { sub(L, L); eq(L); off += 16; };
    switch (op) {
        case FRESH_ADD: add(L); break;
        case FRESH_SUB: case FRESH_IS_LESS_THAN_OR_EQUAL: case FRESH_IS_GREATER_THAN: sub(L, L); break;
        case FRESH_ADD_MOD: add(L); sub(L + 1, L); break;
        case FRESH_SUB_MOD: sub(L, L); sub(L, L + 1); break;
        case FRESH_IS_LESS_THAN: case FRESH_IS_IN_FIELD: case FRESH_IS_GREATER_THAN_OR_EQUAL: lt(); break;
        default: break;   // is_zero, is_equal_fresh: no range assign
    }
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
int32_t lookup_hist_fresh_impl(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint32_t op, const void *trace, uint64_t first_off,
                               uint64_t elem_stride, uint64_t num_elems, const uint8_t *status, uint32_t *hist, h2r_stream_t stream) {
    if (!ctx || !cfg || !trace || !hist) return H2R_E_NULL;
    if (ctx->params.device < 0 || op >= FRESH_OP_COUNT) return H2R_E_UNSUPPORTED;
    if (cfg->n_rows == 0 || cfg->n_rows > (u32)LOOKUP_MAX_ROWS) return H2R_E_SHAPE;
    const h2r_layout &lo = ctx->layout;
    LookupFreshArgs a;
    std::memset(&a, 0, sizeof a);
    a.status = status;
    const int32_t rc = range_shape(*cfg, lo.limb_width, lo.limb_sub_bits, &a.limb);
    if (rc) return rc;
    const AuxGeom g(ctx->L, lo.limb_width);
    bool too_many = false;
    fresh_range_runs(g, op, [&](u64 off, u32 n, u32 stride) {
        if (a.n_runs >= 16) { too_many = true; return; }
        a.run_off[a.n_runs] = (u32)off; a.run_n[a.n_runs] = n; a.run_stride[a.n_runs] = stride; ++a.n_runs;
    });
    if (too_many) return H2R_E_UNSUPPORTED;
    if (num_elems == 0 || a.n_runs == 0) return H2R_OK;
    a.trace = static_cast<const u8 *>(trace); a.first_off = first_off; a.elem_stride = elem_stride; a.num_elems = num_elems;
    a.sub_off = lo.limb_bytes; a.n_rows = cfg->n_rows; a.hist = hist;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps(H2R_KERNEL_HIST, st);
    hipLaunchKernelGGL(lookup_hist_fresh_kernel, dim3((unsigned)num_elems), dim3(64), LOOKUP_ARGS * cfg->n_rows * sizeof(u32), st, a);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
}
}  // namespace
}  // extern "C++"

int32_t h2r_lookup_hist_fresh_op(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint32_t op, const void *trace, uint64_t first_off,
                                 uint64_t elem_stride, uint64_t num_elems, uint32_t *hist, h2r_stream_t stream) try {
    return lookup_hist_fresh_impl(ctx, cfg, op, trace, first_off, elem_stride, num_elems, nullptr, hist, stream);
} H2R_CATCH_STATUS

uint64_t h2r_lookup_workspace_bytes(const h2r_lookup_config *cfg, uint64_t num_elems) try {
    if (!cfg || cfg->n_rows == 0 || cfg->n_rows > (u32)LOOKUP_MAX_ROWS) return 0;
    return num_elems * LOOKUP_ARGS * lookup_slot_bytes(cfg->n_rows) + 256;
} H2R_CATCH_ZERO

int32_t h2r_lookup_permuted_columns(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const uint32_t *hist, const uint64_t *theta,
                                    uint64_t num_elems, uint32_t usable_rows, uint32_t arg_mask, void *a_perm_out,
                                    void *s_perm_out, uint64_t out_elem_stride, uint8_t *status, void *workspace,
                                    h2r_stream_t stream) try {
    if (!ctx || !cfg || !hist || !theta || !a_perm_out || !s_perm_out || !workspace) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (cfg->n_rows == 0 || cfg->n_rows > (u32)LOOKUP_MAX_ROWS || cfg->n_lens == 0 || cfg->n_lens > H2R_LOOKUP_MAX_LENS) return H2R_E_SHAPE;
    if (usable_rows < cfg->n_rows || out_elem_stride < (u64)LOOKUP_ARGS * usable_rows * 32 || (out_elem_stride & 15) ||
        (reinterpret_cast<u64>(a_perm_out) & 15) || (reinterpret_cast<u64>(s_perm_out) & 15)) return H2R_E_SHAPE;
    if (num_elems == 0 || !(arg_mask & 31u)) return H2R_OK;
    if (num_elems * LOOKUP_ARGS >= (1ull << 31) || num_elems > 65535) return H2R_E_UNSUPPORTED;   // grid.z
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    u8 *ws = reinterpret_cast<u8 *>(round_up(reinterpret_cast<u64>(workspace), 256));
    if (status) HIP_TRY(hipMemsetAsync(status, 0, num_elems, st));
    LookupSetupArgs sa;
    std::memset(&sa, 0, sizeof sa);
    sa.hist = hist; sa.theta = theta; sa.num_elems = num_elems; sa.usable_rows = usable_rows; sa.n_rows = cfg->n_rows;
    sa.n_lens = cfg->n_lens; sa.arg_mask = arg_mask & 31u;
    for (u32 i = 0; i < cfg->n_lens; ++i) { sa.tag[i] = cfg->tag[i]; sa.row_off[i] = cfg->row_off[i]; sa.bit_len[i] = cfg->bit_len[i]; }
    sa.f = ctx->fc; sa.ws = ws; sa.status = status;
    sa.mont = (ctx->repr.flags & H2R_ADVICE_MONTGOMERY) ? 1u : 0u;   // theta in, A' / S' out: the ctx's representation
    hipLaunchKernelGGL(lookup_setup_kernel, dim3((unsigned)(num_elems * LOOKUP_ARGS)), dim3(256), lookup_setup_lds_bytes(cfg->n_rows), st, sa);
    HIP_TRY(hipGetLastError());
    LookupFillArgs fa;
    std::memset(&fa, 0, sizeof fa);
    fa.ws = ws; fa.num_elems = num_elems; fa.usable_rows = usable_rows; fa.n_rows = cfg->n_rows; fa.arg_mask = arg_mask & 31u;
    // [r6] 256 rows (8 KB of A' and of S') per workgroup, was 8,192: what the workgroups in flight write is then a dense window of each column
    // instead of a comb of 8 KB pieces 256 KB apart, which is what two buffers of one placement class punish most (same buffers, whole call:
    // 1.63 -> 1.56 ms on a pair of different classes, 2.03-2.16 -> 1.76-1.90 ms on a pair of one class; the 15 KB slot every workgroup loads
    // comes from the L2: tools/lookup_geometry_probe.py, profiles/r06_lookup_geometry.txt)
    fa.rows_per_block = 256; fa.round_robin = 0; fa.status = status;
#ifdef H2R_DEV_KNOBS   // developer build: bits 8-15 of arg_mask = rows per workgroup / 256, bit 16 = round-robin blocks (tools/lookup_geometry_probe.py)
    if ((arg_mask >> 8) & 0xffu) fa.rows_per_block = 256u * ((arg_mask >> 8) & 0xffu);
    fa.round_robin = (arg_mask >> 16) & 1u;
#endif
    fa.a_perm = static_cast<u8 *>(a_perm_out); fa.s_perm = static_cast<u8 *>(s_perm_out); fa.out_elem_stride = out_elem_stride;
    const unsigned chunks = (usable_rows + fa.rows_per_block - 1) / fa.rows_per_block;
    const unsigned lds = (unsigned)lookup_slot_bytes(cfg->n_rows);
    ProfScope ps(H2R_KERNEL_LOOKUP, st, true);
    hipExtLaunchKernelGGL(lookup_fill_kernel, dim3(chunks, LOOKUP_ARGS, (unsigned)num_elems), dim3(256), lds, st, ps.a, ps.on ? ps.b : nullptr, 0, fa);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_field_eval(const h2r_ctx *ctx, uint32_t op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) try {
    if (!ctx || !a || !out || (op < 3 && !b)) return H2R_E_NULL;
    Fe x, y = fe_zero(), r;
    for (int k = 0; k < 4; ++k) { x.v[k] = a[k]; if (b) y.v[k] = b[k]; }
    if (ge_p(x.v, ctx->fc.p) || (op < 3 && ge_p(y.v, ctx->fc.p))) return H2R_E_SHAPE;   // canonical elements only
    switch (op) {
        case 0: r = fe_add(x, y, ctx->fc.p); break;
        case 1: r = fe_sub(x, y, ctx->fc.p); break;
        case 2: r = fe_mul(x, y, ctx->fc); break;
        case 3: if (fe_is_zero(x)) return H2R_E_SHAPE; r = fe_inv(x, ctx->fc); break;
        case 4: if (fe_is_zero(x)) return H2R_E_SHAPE; r = fe_inv_fermat(x, ctx->fc); break;
        case 5: if (fe_is_zero(x)) return H2R_E_SHAPE; r = fe_inv_fast(x, ctx->fc); break;
        case 6: {   // x * R mod p the way the kernels convert a cell: the SHORT Montgomery product over the digits x occupies
            u32 d[8], t[8]; int K = 1;
            for (int k = 0; k < 4; ++k) { d[2 * k] = (u32)x.v[k]; d[2 * k + 1] = (u32)(x.v[k] >> 32); }
            for (int k = 0; k < 8; ++k) if (d[k]) K = k + 1;
            switch (K) {
                case 1: { const u32 q[1] = {d[0]}; mont_short<1>(q, ctx->mk.bk[1], ctx->mk.p, ctx->mk.n0inv, t); break; }
                case 2: { const u32 q[2] = {d[0], d[1]}; mont_short<2>(q, ctx->mk.bk[2], ctx->mk.p, ctx->mk.n0inv, t); break; }
                case 3: { const u32 q[3] = {d[0], d[1], d[2]}; mont_short<3>(q, ctx->mk.bk[3], ctx->mk.p, ctx->mk.n0inv, t); break; }
                case 4: { const u32 q[4] = {d[0], d[1], d[2], d[3]}; mont_short<4>(q, ctx->mk.bk[4], ctx->mk.p, ctx->mk.n0inv, t); break; }
                case 5: { const u32 q[5] = {d[0], d[1], d[2], d[3], d[4]}; mont_short<5>(q, ctx->mk.bk[5], ctx->mk.p, ctx->mk.n0inv, t); break; }
                case 6: { const u32 q[6] = {d[0], d[1], d[2], d[3], d[4], d[5]}; mont_short<6>(q, ctx->mk.bk[6], ctx->mk.p, ctx->mk.n0inv, t); break; }
                case 7: { const u32 q[7] = {d[0], d[1], d[2], d[3], d[4], d[5], d[6]}; mont_short<7>(q, ctx->mk.bk[7], ctx->mk.p, ctx->mk.n0inv, t); break; }
                default: mont_short<8>(d, ctx->mk.bk[8], ctx->mk.p, ctx->mk.n0inv, t); break;
            }
            for (int k = 0; k < 4; ++k) r.v[k] = ((u64)t[2 * k + 1] << 32) | t[2 * k];
            break;
        }
        case 7: r = fe_from_mont(x, ctx->fc); break;   // x * R^-1 mod p: a Montgomery-form element back to its canonical integer
        case 8: r = fe_to_mont(x, ctx->fc); break;     // x * R mod p by the generic R^2 product (the cross-check of 6)
        case 9: {   // x * R mod p in radix 2^30 over the digits x occupies: what cells_kernel runs
            u32 d[8], t[8]; int bits = 1;
            for (int k = 0; k < 4; ++k) { d[2 * k] = (u32)x.v[k]; d[2 * k + 1] = (u32)(x.v[k] >> 32); }
            for (int k = 0; k < 256; ++k) if ((d[k / 32] >> (k % 32)) & 1u) bits = k + 1;
            switch ((bits + 29) / 30) {
                case 1: mont_bits<30, 8>(d, ctx->mk, t); break;
                case 2: mont_bits<60, 8>(d, ctx->mk, t); break;
                case 3: mont_bits<90, 8>(d, ctx->mk, t); break;
                case 4: mont_bits<120, 8>(d, ctx->mk, t); break;
                case 5: mont_bits<150, 8>(d, ctx->mk, t); break;
                case 6: mont_bits<180, 8>(d, ctx->mk, t); break;
                case 7: mont_bits<210, 8>(d, ctx->mk, t); break;
                case 8: mont_bits<240, 8>(d, ctx->mk, t); break;
                default: mont_bits<256, 8>(d, ctx->mk, t); break;
            }
            for (int k = 0; k < 4; ++k) r.v[k] = ((u64)t[2 * k + 1] << 32) | t[2 * k];
            break;
        }
        default: return H2R_E_UNSUPPORTED;
    }
    for (int k = 0; k < 4; ++k) out[k] = r.v[k];
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- host-side flatten: planes -> the reference's assignment order -----------------------------------
namespace {
struct Out { u8 *p; };
inline void emit(Out &o, const u8 *src, u32 n) { std::memcpy(o.p, src, n); o.p += n; }
// WIDE value = 16-byte LO entry followed by the (wide_bytes-16)-byte HI entry
inline void emit_wide(Out &o, const h2r_layout &lo, const u8 *rec, int pl_lo, u64 idx) {
    emit(o, rec + lo.plane_off[pl_lo] + idx * 16, lo.wide_bytes < 16 ? lo.wide_bytes : 16);
    if (lo.wide_bytes > 16) emit(o, rec + lo.plane_off[pl_lo + 1] + idx * 8, lo.wide_bytes - 16);
}
// accumulator of (j, i % L) in the interleaved AB/QN region (see h2r.h)
inline void emit_acc(Out &o, const h2r_layout &lo, const u8 *rec, int pl_lo, u32 j, u32 im) {
    const u64 g = j / lo.acc_steps_per_group, st = j % lo.acc_steps_per_group;
    emit(o, rec + lo.plane_off[pl_lo] + g * lo.acc_lo_group_bytes + st * lo.acc_lo_row_bytes + (u64)im * 16, lo.wide_bytes < 16 ? lo.wide_bytes : 16);
    if (lo.wide_bytes > 16) emit(o, rec + lo.plane_off[pl_lo + 1] + (u64)(j >> 1) * lo.acc_hi_group_bytes + (u64)im * 16 + (j & 1) * 8, lo.wide_bytes - 16);
}
inline void emit_plane(Out &o, const h2r_layout &lo, const u8 *rec, int pl, u64 idx, u32 n) {
    emit(o, rec + lo.plane_off[pl] + idx * lo.plane_elem[pl], n);
}
}  // namespace

// parts: 1 = q/r limbs + sub-limbs (T1, T2), 2 = mul(a,b) accumulators (T3), 4 = mul(q,n) accumulators (T4),
//        8 = eq_b (T5), 16 = is_equal_muled steps (T6)
//        32 (modifier) = a_b as the 32-byte canonical field element (modulus fp), H2R_STREAM_FIELD_AB
static u8 *flatten_parts(const h2r_layout &lo, const u8 *rec, u8 *outp, u32 parts, const u64 *fp = nullptr) {
    const u32 L = lo.num_limbs, C = lo.num_cols;
    Out o{outp};
    if (parts & 1)   // T1/T2: q then r, each limb followed by its sub-limbs (chip.rs:588-599)
        for (int which = 0; which < 2; ++which)
            for (u32 k = 0; k < L; ++k) {
                emit_plane(o, lo, rec, which ? H2R_PL_R : H2R_PL_Q, k, lo.limb_bytes);
                emit_plane(o, lo, rec, which ? H2R_PL_R_SUB : H2R_PL_Q_SUB, k, lo.limb_nsub);
            }
    // T3/T4: mul(a,b) then mul(q,n): column i ascending, j ascending (chip.rs:400-412)
    for (int which = 0; which < 2; ++which) {
        if (!(parts & (which ? 4u : 2u))) continue;
        for (u32 i = 0; i < C; ++i) {
            u32 j = (L >= i + 1) ? 0 : i + 1 - L;
            for (; j < L && j <= i; ++j) emit_acc(o, lo, rec, which ? H2R_PL_QN_LO : H2R_PL_AB_LO, j, i % L);
        }
    }
    if (parts & 8)   // T5: eq_b[i] = qn[i] + r[i], i < L (chip.rs:617)
        for (u32 i = 0; i < L; ++i) emit_wide(o, lo, rec, H2R_PL_EQB_LO, i);
    if (parts & 16)  // T6: is_equal_muled steps (chip.rs:857-893)
        for (u32 i = 0; i < C; ++i) {
            if ((parts & 32) && fp) {   // x >= 0 -> x, x < 0 -> p - |x| = p + x (mod 2^256)
                u64 x[4] = {0, 0, 0, 0};
                std::memcpy(x, rec + lo.plane_off[H2R_PL_AMB_LO] + (u64)i * 16, 16);
                bool neg;
                if (lo.wide_bytes > 16) { std::memcpy(&x[2], rec + lo.plane_off[H2R_PL_AMB_HI] + (u64)i * 8, 8); neg = (x[2] >> 63) != 0; x[3] = neg ? ~0ull : 0; }
                else { neg = (x[1] >> 63) != 0; x[2] = x[3] = neg ? ~0ull : 0; }
                if (neg) { u128 cy = 0; for (int k = 0; k < 4; ++k) { cy += (u128)x[k] + fp[k]; x[k] = (u64)cy; cy >>= 64; } }
                emit(o, reinterpret_cast<const u8 *>(x), 32);
            } else emit_wide(o, lo, rec, H2R_PL_AMB_LO, i);
            emit_wide(o, lo, rec, H2R_PL_SUM_LO, i);
            emit_plane(o, lo, rec, H2R_PL_CARRY, i, lo.carry_bytes);
            emit_plane(o, lo, rec, H2R_PL_CMOD, i, lo.limb_bytes);
            emit_wide(o, lo, rec, H2R_PL_NQ1_LO, i);
            emit_plane(o, lo, rec, H2R_PL_AMNQ1, i, lo.limb_bytes);
            emit_wide(o, lo, rec, H2R_PL_ACCX_LO, i);
            emit_plane(o, lo, rec, H2R_PL_QACC, i, lo.carry_bytes);
            emit_plane(o, lo, rec, H2R_PL_MODACC, i, lo.limb_bytes);
            emit_wide(o, lo, rec, H2R_PL_NQ2_LO, i);
            emit_plane(o, lo, rec, H2R_PL_AMNQ2, i, lo.limb_bytes);
            const u8 *fl = rec + lo.plane_off[H2R_PL_FLAGS] + (u64)i * 4;
            emit(o, fl, 2);
            if (i < C - 1) {
                emit_plane(o, lo, rec, H2R_PL_CARRY_DUP, i, lo.carry_bytes);
                emit_plane(o, lo, rec, H2R_PL_CARRY_SUB, i, lo.carry_nsub);
            }
            emit(o, fl + 2, 2);
        }
    return o.p;
}

int32_t h2r_trace_flatten(const h2r_ctx *ctx, const void *record_host, void *stream_out) try {
    if (!ctx || !record_host || !stream_out) return H2R_E_NULL;
    const h2r_layout &lo = ctx->layout;
    u8 *end = flatten_parts(lo, static_cast<const u8 *>(record_host), static_cast<u8 *>(stream_out), 31);
    if ((u64)(end - static_cast<u8 *>(stream_out)) != lo.stream_bytes) return H2R_E_SHAPE;
    return H2R_OK;
} H2R_CATCH_STATUS

uint64_t h2r_stream_bytes(const h2r_ctx *ctx, uint32_t flags) try {
    if (!ctx) return 0;
    const h2r_layout &lo = ctx->layout;
    return lo.stream_bytes + ((flags & H2R_STREAM_FIELD_AB) ? (u64)lo.num_cols * (32 - lo.wide_bytes) : 0);
} H2R_CATCH_ZERO
uint64_t h2r_pow_stream_bytes(const h2r_ctx *ctx, const h2r_pow_layout *pl, uint32_t flags) try {
    if (!ctx || !pl) return 0;
    return pl->stream_bytes + (u64)pl->num_mul_mods * (h2r_stream_bytes(ctx, flags) - ctx->layout.stream_bytes);
} H2R_CATCH_ZERO
int32_t h2r_trace_flatten_ex(const h2r_ctx *ctx, const void *record_host, uint32_t flags, void *stream_out) try {
    if (!ctx || !record_host || !stream_out) return H2R_E_NULL;
    if (flags & ~H2R_STREAM_FIELD_AB) return H2R_E_UNSUPPORTED;
    u8 *end = flatten_parts(ctx->layout, static_cast<const u8 *>(record_host), static_cast<u8 *>(stream_out),
                            31 | ((flags & H2R_STREAM_FIELD_AB) ? 32u : 0u), ctx->field_p);
    if ((u64)(end - static_cast<u8 *>(stream_out)) != h2r_stream_bytes(ctx, flags)) return H2R_E_SHAPE;
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
// The segment table of one record's stream for emit_kernel: q/r block, column ranges of the two accumulator planes
// (each at most EMIT_SEG_CAP bytes), eq_b + the is_equal_muled steps.
int32_t build_emit_segments(const h2r_layout &lo, u32 flags, EmitArgs &ea) {
    const u32 L = lo.num_limbs, C = lo.num_cols, WB = lo.wide_bytes;
    u32 n = 0;
    auto push = [&](u32 kind, u32 c0, u32 c1, u64 off, u64 bytes) -> bool {
        if (n >= (u32)EMIT_MAX_SEGS || bytes > EMIT_SEG_CAP) return false;
        ea.seg[n++] = EmitSeg{kind, c0, c1, (u32)bytes, off};
        return true;
    };
    u64 off = 0;
    const u64 qr = 2ull * L * (lo.limb_bytes + lo.limb_nsub);
    if (lo.limb_nsub != 8 || !push(EMIT_QR, 0, 0, off, qr)) return H2R_E_UNSUPPORTED;
    off += qr;
    for (u32 kind = EMIT_ACC_AB; kind <= EMIT_ACC_QN; ++kind) {
        u32 c0 = 0;
        while (c0 < C) {
            u32 c1 = c0; u64 bytes = 0;
            while (c1 < C) {
                const u64 cb = (u64)(emit_colstart(c1 + 1, L) - emit_colstart(c1, L)) * WB;
                if (bytes + cb > EMIT_SEG_CAP) break;
                bytes += cb; ++c1;
            }
            if (c1 == c0 || !push(kind, c0, c1, off, bytes)) return H2R_E_UNSUPPORTED;
            off += bytes; c0 = c1;
        }
    }
    if (!push(EMIT_EQB, 0, 0, off, (u64)L * WB)) return H2R_E_UNSUPPORTED;
    off += (u64)L * WB;
    const u64 ab = (flags & H2R_STREAM_FIELD_AB) ? 32 : WB;
    const u64 per_col = ab + 4ull * WB + 2ull * lo.carry_bytes + 4ull * lo.limb_bytes + 4;
    const u64 per_col_ra = per_col + lo.carry_bytes + lo.carry_nsub;   // every column but the last range-assigns its carry
    for (u32 c0 = 0; c0 < C;) {
        u32 c1 = c0 + (u32)(EMIT_SEG_CAP / per_col_ra);
        if (c1 > C) c1 = C;
        const u64 bytes = (u64)(c1 - c0) * per_col_ra - (c1 == C ? lo.carry_bytes + lo.carry_nsub : 0);
        if (c1 == c0 || !push(EMIT_STEPS, c0, c1, off, bytes)) return H2R_E_UNSUPPORTED;
        off += bytes; c0 = c1;
    }
    ea.nseg = n; ea.rec_bytes = off;
    return H2R_OK;
}

int32_t launch_emit(const h2r_ctx *ctx, EmitArgs &ea, u32 flags, hipStream_t st) {
    if (flags & ~H2R_STREAM_FIELD_AB) return H2R_E_UNSUPPORTED;
    const h2r_layout &lo = ctx->layout;
    const int32_t rc = build_emit_segments(lo, flags, ea);
    if (rc) return rc;
    if (ea.rec_bytes != h2r_stream_bytes(ctx, flags)) return H2R_E_SHAPE;
    for (int p = 0; p < H2R_PL_COUNT; ++p) ea.off[p] = lo.plane_off[p];
    ea.L = lo.num_limbs; ea.carry_nsub = lo.carry_nsub; ea.carry_sub_stride = lo.carry_sub_stride;
    ea.record_stride = lo.record_stride;
    ea.field_ab = (flags & H2R_STREAM_FIELD_AB) ? 1u : 0u;
    for (int k = 0; k < 4; ++k) ea.p[k] = ctx->field_p[k];
    const u64 blocks = ea.n_elems * (ea.T ? ea.T : 1);
    if (blocks == 0) return H2R_OK;
    if (blocks >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    H2R_ON_DEVICE(ctx->params.device);
    ProfScope ps(H2R_KERNEL_EMIT, st, true);
    const unsigned lds = EMIT_SEG_CAP + 64;
    if (lo.limb_width == 64) hipExtLaunchKernelGGL((emit_kernel<64>), dim3((unsigned)blocks), dim3(256), lds, st, ps.a, ps.on ? ps.b : nullptr, 0, ea);
    else hipExtLaunchKernelGGL((emit_kernel<32>), dim3((unsigned)blocks), dim3(256), lds, st, ps.a, ps.on ? ps.b : nullptr, 0, ea);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
}
}  // namespace

int32_t h2r_trace_emit_stream(const h2r_ctx *ctx, const void *trace, uint64_t num_records, uint32_t flags, void *stream_out,
                              uint64_t out_stride, uint64_t out_off, h2r_stream_t stream) try {
    if (!ctx || !trace || !stream_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (out_stride < out_off + h2r_stream_bytes(ctx, flags)) return H2R_E_SHAPE;
    EmitArgs ea;
    std::memset(&ea, 0, sizeof ea);
    ea.trace = static_cast<const u8 *>(trace); ea.elem_stride = ctx->layout.record_stride; ea.off_records = 0; ea.T = 1;
    ea.n_elems = num_records; ea.out = static_cast<u8 *>(stream_out); ea.out_stride = out_stride; ea.out_off = out_off;
    return launch_emit(ctx, ea, flags, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

int32_t h2r_pow_trace_emit_stream(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *trace, uint64_t elem_stride,
                                  uint64_t batch, uint32_t flags, void *stream_out, uint64_t out_stride, uint64_t out_off,
                                  h2r_stream_t stream) try {
    if (!ctx || !pl || !trace || !stream_out) return H2R_E_NULL;
    if (pl->off_records == UINT64_MAX) return H2R_E_SHAPE;   // a witness-only layout (h2r_pow_layout_compact) holds no records
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (out_stride < out_off + h2r_pow_stream_bytes(ctx, pl, flags)) return H2R_E_SHAPE;
    const h2r_layout &lo = ctx->layout;
    EmitArgs ea;
    std::memset(&ea, 0, sizeof ea);
    ea.trace = static_cast<const u8 *>(trace); ea.elem_stride = elem_stride ? elem_stride : pl->elem_stride;
    ea.off_records = pl->off_records; ea.T = pl->num_mul_mods; ea.n_elems = batch;
    ea.out = static_cast<u8 *>(stream_out); ea.out_stride = out_stride; ea.out_off = out_off;
    ea.var = pl->off_e_bits != UINT64_MAX ? 1u : 0u; ea.nbits = ea.var ? pl->num_exp_bits : 0;
    ea.limbs_bytes = lo.num_limbs * lo.limb_bytes;
    ea.off_e_bits = pl->off_e_bits; ea.off_selected = pl->off_selected; ea.selected_stride = pl->selected_stride;
    ea.off_result = pl->off_result; ea.has_result = 1;
    if (ea.var && ea.T != 2 * ea.nbits) return H2R_E_SHAPE;
    return launch_emit(ctx, ea, flags, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

// ---- advice-column image ---------------------------------------------------------------------------------------
uint32_t h2r_advice_rows(const h2r_ctx *ctx) try { return ctx ? advice_rows_per_record(ctx->L, ctx->layout.carry_nsub) : 0; } H2R_CATCH_ZERO

namespace {
// The caller's image buffer in the ctx's representation (h2r_advice_repr): `rows` rows per element, element e at + e * out_stride.
int32_t advice_dst(const h2r_ctx *ctx, void *advice_out, u64 out_stride, u64 rows, u64 batch, AdviceDst *d) {
    d->base = static_cast<u8 *>(advice_out); d->elem_stride = out_stride;
    d->mont = (ctx->repr.flags & H2R_ADVICE_MONTGOMERY) ? 1u : 0u;
    if (ctx->repr.flags & H2R_ADVICE_COLUMNS) {
        const u64 cs = ctx->repr.col_stride ? ctx->repr.col_stride : rows * 32;   // 0: the element's five columns packed back to back
        if (cs < rows * 32 || (cs & 15) || (out_stride & 15)) return H2R_E_SHAPE;
        // [element][column][row] (out_stride covers five columns) or [column][element][row] (col_stride covers every element)
        const bool elem_major = out_stride >= 4 * cs + rows * 32, col_major = batch == 0 || (out_stride >= rows * 32 && cs >= (batch - 1) * out_stride + rows * 32);
        if (!elem_major && !col_major) return H2R_E_SHAPE;
        d->row_pitch = 32; d->col_pitch = cs;
    } else {
        if (out_stride < rows * ADVICE_ROW_BYTES) return H2R_E_SHAPE;
        d->row_pitch = ADVICE_ROW_BYTES; d->col_pitch = 32;
    }
    return H2R_OK;
}
int32_t launch_advice(const h2r_ctx *ctx, AdviceArgs &aa, hipStream_t st) {
    const h2r_layout &lo = ctx->layout;
    if (lo.num_limbs > 128 || lo.limb_nsub != 8 || lo.carry_nsub > 16) return H2R_E_UNSUPPORTED;
    for (int p = 0; p < H2R_PL_COUNT; ++p) aa.off[p] = lo.plane_off[p];
    aa.L = lo.num_limbs; aa.carry_bits = lo.carry_bits; aa.carry_sub_bits = lo.carry_sub_bits; aa.carry_nsub = lo.carry_nsub;
    aa.carry_sub_stride = lo.carry_sub_stride; aa.record_stride = lo.record_stride;
    aa.rows = h2r_advice_rows(ctx);
    aa.f = ctx->fc; aa.desc = ctx->advice_desc_dev;
    aa.mk = ctx->mk_dev;
    if (aa.n_items == 0) return H2R_OK;
    if (aa.n_items >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    ProfScope ps(H2R_KERNEL_EMIT, st, true);
    if (lo.limb_width == 64) hipExtLaunchKernelGGL((advice_kernel<64>), dim3((unsigned)aa.n_items), dim3(256), 0, st, ps.a, ps.on ? ps.b : nullptr, 0, aa);
    else hipExtLaunchKernelGGL((advice_kernel<32>), dim3((unsigned)aa.n_items), dim3(256), 0, st, ps.a, ps.on ? ps.b : nullptr, 0, aa);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
}
// The same image written directly from the operands (cells_kernel, h2r_cells.hpp): one wave per mul_mod, no record read.
int32_t launch_cells(const h2r_ctx *ctx, CellsArgs &ca, hipStream_t st) {
    const h2r_layout &lo = ctx->layout;
    if (lo.num_limbs > 128 || lo.limb_nsub != 8 || lo.carry_nsub > 16) return H2R_E_UNSUPPORTED;
    ca.ktab = ctx->cells_ktab_dev;
    ca.per_col_magic = (u32)(((1ull << 32) + (ADVICE_COL_ROWS + (lo.carry_nsub + 3) / 4) - 1) / (ADVICE_COL_ROWS + (lo.carry_nsub + 3) / 4));
    ca.L = lo.num_limbs; ca.carry_sub_bits = lo.carry_sub_bits; ca.carry_nsub = lo.carry_nsub;
    ca.rows = h2r_advice_rows(ctx);
    ca.mk = ctx->mk_dev;
    const bool mont = ca.dst.mont != 0;
    if (ca.n_items == 0) return H2R_OK;
    if (ca.n_items >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    // Residency: FOUR waves per CU, one per SIMD (measured: 6.64-6.67 TB/s against 6.47 with the six the RSA-2048 shape's 26 KB would
    // allow, 4.2 with three -- profiles/r04_cells_kernel.txt).  Enforced the way occupancy is enforced on this hardware: by the LDS request.
    const u32 nwv = mont ? ctx->cells_nwv : 1u;
    u32 lds = cells_lds_bytes(lo.limb_width, lo.num_limbs, mont, nwv);
#ifndef H2R_CELLS_WAVES
#define H2R_CELLS_WAVES 4   // (developer variants: 0 = whatever fits)
#endif
    // (Montgomery cells: the kernel is VALU-issue bound, not store bound -- every wave the LDS admits helps: 29 KB per wave = five per CU
    //  for RSA-2048, 2.9 -> 2.5 ms per 1,024 elements against four)
    const u32 quarter = (H2R_CELLS_WAVES && !mont) ? (ctx->lds_per_cu / H2R_CELLS_WAVES - 512) & ~15u : 0u;
    if (lds < quarter) lds = quarter;
    if (lds > ctx->lds_per_cu) return H2R_E_UNSUPPORTED;
    ProfScope ps(H2R_KERNEL_CELLS, st, true);
    HIP_TRY(launch_cells_shape(lo.limb_width, mont, nwv, lds, ca, st, ps.a, ps.on ? ps.b : nullptr));
    return H2R_OK;
}
}  // namespace

int32_t h2r_mul_mod_emit_advice(const h2r_ctx *ctx, const void *a, const void *b, const void *n, uint32_t flags, const void *trace,
                                uint64_t batch, const uint8_t *status, void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    if (!ctx || !a || !b || !n || !trace || !advice_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    AdviceDst dst;
    if (const int32_t rc = advice_dst(ctx, advice_out, out_stride, h2r_advice_rows(ctx), batch, &dst)) return rc;
    H2R_ON_DEVICE(ctx->params.device);
    if (flags & H2R_ADVICE_DIRECT) {   // recomputed from (a, b, n) and the record's q, r limbs; nothing else of the record is read
        const h2r_layout &lo = ctx->layout;
        CellsArgs ca;
        std::memset(&ca, 0, sizeof ca);
        ca.opA = a; ca.opB = b; ca.op_stride = ctx->L;
        ca.opQ = static_cast<const u8 *>(trace) + lo.plane_off[H2R_PL_Q]; ca.opR = static_cast<const u8 *>(trace) + lo.plane_off[H2R_PL_R];
        ca.qr_stride = lo.record_stride / lo.limb_bytes;
        ca.n = n; ca.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
        ca.status = status; ca.T = 1; ca.n_items = batch; ca.dst = dst;
        return launch_cells(ctx, ca, static_cast<hipStream_t>(stream));
    }
    AdviceArgs aa;
    std::memset(&aa, 0, sizeof aa);
    aa.opA = a; aa.opB = b; aa.op_stride = ctx->L; aa.n = n; aa.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    aa.status = status; aa.trace = static_cast<const u8 *>(trace); aa.elem_stride = ctx->layout.record_stride; aa.off_records = 0;
    aa.T = 1; aa.n_items = batch; aa.dst = dst;
    return launch_advice(ctx, aa, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

namespace {
// the pow rows of an element image whose row 0 is dst's (the public export, and a section of the whole-element images)
int32_t pow_emit_advice(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *n, uint32_t flags, const void *trace,
                        uint64_t elem_stride, const void *workspace, uint64_t batch, const uint8_t *status,
                        AdviceDst dst, h2r_stream_t stream) {
    if (!ctx || !pl || !n || !workspace || !dst.base) return H2R_E_NULL;
    const bool var = pl->off_e_bits != UINT64_MAX;
    if (!trace && (var || !(flags & H2R_ADVICE_DIRECT))) return H2R_E_NULL;   // (a Var element's e_bits / selected planes live in the trace)
    if (!(flags & H2R_ADVICE_DIRECT) && pl->off_records == UINT64_MAX) return H2R_E_SHAPE;   // the record-read image needs records: a witness-only layout serves H2R_ADVICE_DIRECT only
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (pl->num_mul_mods == 0 || batch == 0) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    const u64 lb = ctx->layout.limb_bytes;
    const u8 *ws = reinterpret_cast<const u8 *>(round_up(reinterpret_cast<u64>(workspace), 256));   // as run_path carves it
    hipStream_t st = static_cast<hipStream_t>(stream);
    u32 sel_rows = 0;
    if (var) {   // pow_mod: the to_bits rows of every exponent limb in front, the select rows of every bit behind its mul_mod (chip.rs:674-691)
        if (!pl->exp_limb_bits || !pl->e_num_limbs || pl->num_mul_mods != 2 * pl->num_exp_bits) return H2R_E_SHAPE;
        VarRowsArgs va;
        std::memset(&va, 0, sizeof va);
        va.trace = static_cast<const u8 *>(trace); va.elem_stride = elem_stride ? elem_stride : pl->elem_stride;
        va.off_e_bits = pl->off_e_bits; va.off_selected = pl->off_selected; va.selected_stride = pl->selected_stride;
        va.opA = ws; va.opR = ws + 3 * ctx->L * lb; va.op_stride = 4ull * ctx->L;
        va.status = status; va.batch = batch; va.L = ctx->L; va.T = pl->num_mul_mods; va.nbits = pl->num_exp_bits;
        va.exp_limb_bits = pl->exp_limb_bits; va.e_num_limbs = pl->e_num_limbs; va.rows = h2r_advice_rows(ctx);
        va.dst = dst; va.mk = ctx->mk_dev;
        const u64 per_elem = (u64)va.e_num_limbs * var_to_bits_rows(va.exp_limb_bits) + (u64)va.nbits * va.L;
        const u64 blocks = (batch * per_elem + 255) / 256;
        if (blocks >= (1ull << 31)) return H2R_E_UNSUPPORTED;
        if (ctx->layout.limb_width == 64) hipLaunchKernelGGL((var_rows_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, st, va);
        else hipLaunchKernelGGL((var_rows_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, va);
        HIP_TRY(hipGetLastError());
        dst = dst.at_row((u64)va.e_num_limbs * var_to_bits_rows(va.exp_limb_bits));
        sel_rows = ctx->L;
    }
    if (flags & H2R_ADVICE_DIRECT) {   // from the call's operands alone (trace may be NULL: a fixed-exponent call that wrote no records)
        CellsArgs ca;
        std::memset(&ca, 0, sizeof ca);
        ca.opA = ws; ca.opB = ws + ctx->L * lb; ca.opQ = ws + 2 * ctx->L * lb; ca.opR = ws + 3 * ctx->L * lb;
        ca.op_stride = 4ull * ctx->L; ca.qr_stride = ca.op_stride;
        ca.n = n; ca.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
        ca.status = status; ca.T = pl->num_mul_mods; ca.n_items = batch * pl->num_mul_mods;
        ca.dst = dst;
        ca.pre_rows = 2u; ca.sel_rows = sel_rows;   // acc = assign_constant(1, L): [1], [0] (chip.rs:729 resp. :682)
        return launch_cells(ctx, ca, st);
    }
    AdviceArgs aa;
    std::memset(&aa, 0, sizeof aa);
    aa.opA = ws; aa.opB = ws + ctx->L * lb; aa.op_stride = 4ull * ctx->L;
    aa.n = n; aa.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    aa.status = status; aa.trace = static_cast<const u8 *>(trace); aa.elem_stride = elem_stride ? elem_stride : pl->elem_stride;
    aa.off_records = pl->off_records; aa.T = pl->num_mul_mods; aa.n_items = batch * pl->num_mul_mods;
    aa.dst = dst;
    aa.pre_rows = 2u; aa.sel_rows = sel_rows;
    return launch_advice(ctx, aa, st);
}
}  // namespace

int32_t h2r_pow_trace_emit_advice(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *n, uint32_t flags, const void *trace,
                                  uint64_t elem_stride, const void *workspace, uint64_t batch, const uint8_t *status,
                                  void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    if (!ctx || !pl || !n || !workspace || !advice_out) return H2R_E_NULL;
    AdviceDst dst;
    if (const int32_t rc = advice_dst(ctx, advice_out, out_stride, h2r_pow_advice_rows(ctx, pl), batch, &dst)) return rc;
    return pow_emit_advice(ctx, pl, n, flags, trace, elem_stride, workspace, batch, status, dst, stream);
} H2R_CATCH_STATUS

uint64_t h2r_pow_advice_rows(const h2r_ctx *ctx, const h2r_pow_layout *pl) try {
    if (!ctx || !pl) return 0;
    const u64 recs = 2ull + (u64)pl->num_mul_mods * h2r_advice_rows(ctx);   // acc = assign_constant(1): CONST1, CONST0, then the mul_mods
    if (pl->off_e_bits == UINT64_MAX) return recs;
    return (u64)pl->e_num_limbs * var_to_bits_rows(pl->exp_limb_bits) + recs + (u64)pl->num_exp_bits * ctx->L;
} H2R_CATCH_ZERO

int32_t h2r_pow_row_kinds(const h2r_ctx *ctx, const h2r_pow_layout *pl, uint8_t *kinds_out) try {
    if (!ctx || !pl || !kinds_out) return H2R_E_NULL;
    const u32 rows = h2r_advice_rows(ctx), nrc = (ctx->layout.carry_nsub + 3) / 4;
    const bool var = pl->off_e_bits != UINT64_MAX;
    uint8_t *k = kinds_out;
    if (var)
        for (u32 l = 0; l < pl->e_num_limbs; ++l)
            for (u32 i = 0; i < var_to_bits_rows(pl->exp_limb_bits); ++i) *k++ = (uint8_t)var_to_bits_kind(i, pl->exp_limb_bits);
    *k++ = ROWK_CONST1; *k++ = ROWK_CONST0;
    for (u32 t = 0; t < pl->num_mul_mods; ++t) {
        for (u32 r = 0; r < rows; ++r) *k++ = (uint8_t)advice_decode(r, ctx->L, nrc).kind;
        if (var && !(t & 1)) for (u32 j = 0; j < ctx->L; ++j) *k++ = (uint8_t)ROWK_SELECT;
    }
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_advice_row_kinds(const h2r_ctx *ctx, uint8_t *kinds_out) try {
    if (!ctx || !kinds_out) return H2R_E_NULL;
    const u32 rows = h2r_advice_rows(ctx), nrc = (ctx->layout.carry_nsub + 3) / 4;
    for (u32 r = 0; r < rows; ++r) kinds_out[r] = (uint8_t)advice_decode(r, ctx->L, nrc).kind;
    return H2R_OK;
} H2R_CATCH_STATUS

namespace { int32_t fixed_row_repr(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint32_t kind, bool mont, h2r_fixed_row *out); }
int32_t h2r_advice_fixed_row(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint32_t kind, h2r_fixed_row *out) try {
    if (!ctx || !out) return H2R_E_NULL;
    return fixed_row_repr(ctx, cfg, kind, (ctx->repr.flags & H2R_ADVICE_MONTGOMERY) != 0, out);   // the selectors are field elements like the cells: the ctx's representation
} H2R_CATCH_STATUS
namespace {
int32_t fixed_row_repr(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint32_t kind, bool mont, h2r_fixed_row *out) {
    std::memset(out, 0, sizeof *out);
    const h2r_layout &lo = ctx->layout;
    const u64 (&p)[4] = ctx->fc.p;
    auto put = [&](uint64_t (&dst)[4], const Fe &v, bool neg) {
        Fe r = neg ? fe_sub(fe_zero(), v, p) : v;
        if (mont) r = fe_to_mont(r, ctx->fc);
        for (int k = 0; k < 4; ++k) dst[k] = r.v[k];
    };
    const Fe one = fe_small(1);
    Fe Bv = fe_zero(); if (lo.limb_width == 64) Bv.v[1] = 1; else Bv.v[0] = 1ull << 32;
    Fe W; for (int k = 0; k < 4; ++k) W.v[k] = ctx->word_max.v[k];
    auto pow2 = [&](u32 sh) { Fe r = fe_zero(); r.v[sh / 64] = 1ull << (sh % 64); return r; };
    switch (kind) {
        case ROWK_NOP: case ROWK_VALUE: break;
        case ROWK_CONST0: put(out->sa, one, false); break;
        case ROWK_CONST1: put(out->sa, one, false); put(out->s_const, one, true); break;
        case ROWK_CONST_B: put(out->sa, one, false); put(out->s_const, Bv, true); break;
        case ROWK_BIT: case ROWK_MUL: put(out->s_mul_ab, one, false); put(out->sc, one, true); break;
        case ROWK_MUL_ADD: put(out->s_mul_ab, one, false); put(out->sc, one, false); put(out->sd, one, true); break;
        case ROWK_ADD: put(out->sa, one, false); put(out->sb, one, false); put(out->sc, one, true); break;
        case ROWK_SUB: put(out->sa, one, false); put(out->sb, one, true); put(out->sc, one, true); break;
        case ROWK_ADD_WM: put(out->sa, one, false); put(out->sb, one, false); put(out->sc, one, true); put(out->s_const, W, false); break;
        case ROWK_ADDC_WM: put(out->sa, one, false); put(out->sb, one, true); put(out->s_const, W, false); break;
        case ROWK_ASSERT_EQ: put(out->sa, one, false); put(out->sb, one, true); break;
        case ROWK_ISZERO_INV: put(out->s_mul_ab, one, false); put(out->sc, one, false); put(out->s_const, one, true); break;
        case ROWK_ISZERO_RA: put(out->s_mul_ab, one, false); break;
        case ROWK_SELECT: put(out->s_mul_ab, one, false); put(out->s_mul_cd, one, true); put(out->sd, one, false); put(out->se, one, true); break;
        case ROWK_NOT: put(out->sa, one, false); put(out->sb, one, false); put(out->s_const, one, true); break;
        case ROWK_ASSERT_ONE: put(out->sa, one, false); put(out->s_const, one, true); break;
        case ROWK_CONST_BM1: put(out->sa, one, false); put(out->s_const, fe_sub(Bv, one, p), true); break;
        case ROWK_ASSERT_ZERO: put(out->sa, one, false); break;
        case ROWK_CONST_EM: case ROWK_CONST_EM + 1: case ROWK_CONST_EM + 2: case ROWK_CONST_EM + 3: case ROWK_CONST_EM + 4: case ROWK_CONST_EM + 5:
            put(out->sa, one, false); put(out->s_const, fe_small(em_const(kind - ROWK_CONST_EM)), true); break;
        case ROWK_CONST_COEFF8: case ROWK_CONST_COEFF8 + 1: case ROWK_CONST_COEFF8 + 2: case ROWK_CONST_COEFF8 + 3:
        case ROWK_CONST_COEFF8 + 4: case ROWK_CONST_COEFF8 + 5: case ROWK_CONST_COEFF8 + 6: case ROWK_CONST_COEFF8 + 7:
            put(out->sa, one, false); put(out->s_const, pow2(8 * (kind - ROWK_CONST_COEFF8)), true); break;
        case ROWK_RANGE_U32: case ROWK_RANGE_U32 + 1: {   // RangeChip::assign(value, 4, 32): eight 4-bit sub-limbs, two rows
            const bool last = kind == ROWK_RANGE_U32 + 1;
            uint64_t (*sel[4])[4] = {&out->sa, &out->sb, &out->sc, &out->sd};
            for (u32 q = 0; q < 4; ++q) put(*sel[q], pow2((last ? 7 - q : q) * 4), false);
            put(out->se, one, true);
            if (!last) put(out->se_next, one, false);
            if (cfg) {
                for (u32 i = 0; i < cfg->n_lens; ++i) if (cfg->bit_len[i] == 4) out->tag_composition = cfg->tag[i];
                if (!out->tag_composition) return H2R_E_SHAPE;   // a table without RSAChip's 4-bit range (src/chip.rs:252)
            }
            break;
        }
        default: {
            if (kind >= ROWK_BITS_COMPOSE && kind < ROWK_BITS_COMPOSE_LAST + 64) {   // to_bits' compose rows: coefficients 2^(bit index), no lookup
                const bool last = kind >= ROWK_BITS_COMPOSE_LAST;
                const u32 rr = last ? (kind - ROWK_BITS_COMPOSE_LAST) / 4 : kind - ROWK_BITS_COMPOSE, terms = last ? (kind - ROWK_BITS_COMPOSE_LAST) % 4 + 1 : 4;
                uint64_t (*sel[4])[4] = {&out->sa, &out->sb, &out->sc, &out->sd};
                for (u32 q = 0; q < terms; ++q) { const u32 b = last ? 4 * rr + terms - 1 - q : 4 * rr + q; if (b >= 64) return H2R_E_SHAPE; put(*sel[q], pow2(b), false); }
                put(out->se, one, true);
                if (!last) put(out->se_next, one, false);
                break;
            }
            const bool carry = kind >= ROWK_RANGE_CARRY;
            const u32 rr = kind - (carry ? ROWK_RANGE_CARRY : ROWK_RANGE_LIMB);
            const u32 s = carry ? lo.carry_sub_bits : lo.limb_sub_bits, nsub = carry ? lo.carry_nsub : lo.limb_nsub;
            const u32 nrows = (nsub + 3) / 4;
            if (kind < ROWK_RANGE_LIMB || (!carry && kind >= ROWK_RANGE_LIMB + 8) || rr >= nrows) return H2R_E_SHAPE;
            const bool last = rr == nrows - 1;
            uint64_t (*sel[4])[4] = {&out->sa, &out->sb, &out->sc, &out->sd};
            const u32 k0 = 4 * rr, k1 = std::min(4 * rr + 4, nsub);
            for (u32 q = 0; q < k1 - k0; ++q) put(*sel[q], pow2((last ? k1 - 1 - q : k0 + q) * s), false);
            put(out->se, one, true);
            if (!last) put(out->se_next, one, false);
            if (cfg) {   // the lookup tags of the row (0 = off)
                const u32 ov = carry ? lo.carry_bits % s : 0;
                for (u32 i = 0; i < cfg->n_lens; ++i) {
                    if (cfg->bit_len[i] == s) out->tag_composition = cfg->tag[i];
                    if (last && ov && cfg->bit_len[i] == ov) out->tag_overflow = cfg->tag[i];
                }
                if (!out->tag_composition || (last && ov && !out->tag_overflow)) return H2R_E_SHAPE;
            }
            break;
        }
    }
    return H2R_OK;
}
}  // namespace

// ---- advice rows of the Fresh-integer family (h2r_rowprog.hpp) ---------------------------------------------------------------
namespace {
// a row program, built by the symbolic walk on first use and uploaded when the ctx has a device
constexpr u32 kProgEm = 0x1000, kProgVerifyPre = 0x1001;
int32_t row_prog(const h2r_ctx *ctx, u32 key, const std::function<bool(RowProgBuilder &)> &build, const h2r_ctx::RowProg **out) {
    if (ctx->L + 3 > 64 * AUX_V) return H2R_E_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(ctx->prog_mu);
    auto it = ctx->progs.find(key);
    if (it == ctx->progs.end()) {
        RowProgBuilder rb(AuxGeom(ctx->L, ctx->layout.limb_width));
        if (!build(rb)) return H2R_E_UNSUPPORTED;
        h2r_ctx::RowProg rp;
        rp.host = std::move(rb.rows);
        for (u32 r = 0; r < rp.host.size(); ++r) if (rp.host[r].c[1].type == RP_INV34) rp.inv_rows.push_back(r);
        if (ctx->params.device >= 0) {
            DeviceGuard dg(ctx->params.device);
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&rp.dev), rp.host.size() * sizeof(RpRow)));
            if (hipMemcpy(rp.dev, rp.host.data(), rp.host.size() * sizeof(RpRow), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(rp.dev); return H2R_E_HIP; }
            if (!rp.inv_rows.empty()) {
                if (hipMalloc(reinterpret_cast<void **>(&rp.inv_dev), rp.inv_rows.size() * sizeof(u32)) != hipSuccess ||
                    hipMemcpy(rp.inv_dev, rp.inv_rows.data(), rp.inv_rows.size() * sizeof(u32), hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipFree(rp.dev); if (rp.inv_dev) (void)hipFree(rp.inv_dev);
                    return H2R_E_HIP;
                }
            }
        }
        it = ctx->progs.emplace(key, std::move(rp)).first;
    }
    *out = &it->second;
    return H2R_OK;
}
int32_t fresh_row_prog(const h2r_ctx *ctx, uint32_t op, uint32_t flags, const h2r_ctx::RowProg **out) {
    if (op >= FRESH_OP_COUNT) return H2R_E_UNSUPPORTED;
    const bool assert_one = (flags & H2R_ADVICE_ASSERT_ONE) != 0;   // on an op without a bit: unsupported
    return row_prog(ctx, op | (assert_one ? 256u : 0u), [&](RowProgBuilder &rb) { return rb.build(op, assert_one); }, out);
}
int32_t launch_row_prog(const h2r_ctx *ctx, const h2r_ctx::RowProg *rp, RowProgArgs &ra, hipStream_t st) {
    ra.prog = rp->dev; ra.rows = (u32)rp->host.size(); ra.f = ctx->fc; ra.mk = ctx->mk_dev;
    ra.inv_rows = rp->inv_dev; ra.n_inv = (u32)rp->inv_rows.size();
    // Workgroup size.  A Montgomery ctx's cells kernel (256 VGPRs, six one-wave workgroups per CU, LDS full) lets a workgroup of several
    // waves in only when its grid drains: 256-row row programs issued next to it sat there for its whole 2 ms with everything behind them
    // on the stream (the next call's chains) -- one-wave workgroups (64 rows, 10 KB) are placed as cells workgroups retire: the pipelined
    // calls 2.39 -> 2.26 ms (modpow_public_key element), 2.67 -> 2.26 ms (whole verify element); profiles/r05_advice_pipeline_montgomery.txt.
    // Next to the canonical cells kernel both sizes run alike; 256 stays.  (H2R_ROWPROG_STAGE_ROWS = 64 | 128 | 256: developer A/B, -DH2R_DEV_KNOBS build.)
    const long sr_k = knobs().rowprog_stage_rows;
    const u32 sr_env = (sr_k == 64 || sr_k == 128 || sr_k == 256) ? (u32)sr_k : 0u;
    const u32 sr = sr_env ? sr_env : ((ctx->repr.flags & H2R_ADVICE_MONTGOMERY) ? 64u : 256u);
    const u64 blocks = ra.batch * ((ra.rows + sr - 1) / sr);
    if (blocks == 0) return H2R_OK;
    if (blocks >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    ProfScope ps(H2R_KERNEL_EMIT, st, true);
    auto go = [&](auto lw_c, auto sr_c) {
        constexpr int LW = decltype(lw_c)::value; constexpr u32 SR = decltype(sr_c)::value;
        hipExtLaunchKernelGGL((rowprog_kernel<LW, SR>), dim3((unsigned)blocks), dim3(SR), 0, st, ps.a, ps.on ? ps.b : nullptr, 0, ra);
    };
    using I64 = std::integral_constant<int, 64>; using I32 = std::integral_constant<int, 32>;
    const bool w64 = ctx->layout.limb_width == 64;
    if (sr == 64) { if (w64) go(I64{}, std::integral_constant<u32, 64>{}); else go(I32{}, std::integral_constant<u32, 64>{}); }
    else if (sr == 128) { if (w64) go(I64{}, std::integral_constant<u32, 128>{}); else go(I32{}, std::integral_constant<u32, 128>{}); }
    else { if (w64) go(I64{}, std::integral_constant<u32, 256>{}); else go(I32{}, std::integral_constant<u32, 256>{}); }
    HIP_TRY(hipGetLastError());
    if (ra.n_inv) {   // is_zero's inverse witnesses, packed into full waves (rowprog_inv_kernel)
        const u32 nt = sr == 64 ? 64u : 256u;   // (one-wave workgroups with the one-wave stage)
        const u64 ib = (ra.batch * ra.n_inv + nt - 1) / nt;
        if (ib >= (1ull << 31)) return H2R_E_UNSUPPORTED;
        if (nt == 64) {
            if (w64) hipLaunchKernelGGL((rowprog_inv_kernel<64, 64>), dim3((unsigned)ib), dim3(64), 0, st, ra);
            else hipLaunchKernelGGL((rowprog_inv_kernel<32, 64>), dim3((unsigned)ib), dim3(64), 0, st, ra);
        } else {
            if (w64) hipLaunchKernelGGL((rowprog_inv_kernel<64>), dim3((unsigned)ib), dim3(256), 0, st, ra);
            else hipLaunchKernelGGL((rowprog_inv_kernel<32>), dim3((unsigned)ib), dim3(256), 0, st, ra);
        }
        HIP_TRY(hipGetLastError());
    }
    return H2R_OK;
}
}  // namespace

uint32_t h2r_fresh_op_advice_rows(const h2r_ctx *ctx, uint32_t op, uint32_t flags) try {
    const h2r_ctx::RowProg *rp = nullptr;
    if (!ctx || fresh_row_prog(ctx, op, flags, &rp)) return 0;
    return (uint32_t)rp->host.size();
} H2R_CATCH_ZERO

int32_t h2r_fresh_op_row_kinds(const h2r_ctx *ctx, uint32_t op, uint32_t flags, uint8_t *kinds_out) try {
    if (!ctx || !kinds_out) return H2R_E_NULL;
    const h2r_ctx::RowProg *rp = nullptr;
    const int32_t rc = fresh_row_prog(ctx, op, flags, &rp);
    if (rc) return rc;
    for (size_t r = 0; r < rp->host.size(); ++r) kinds_out[r] = (uint8_t)rp->host[r].kind;
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
int32_t fresh_emit_advice(const h2r_ctx *ctx, uint32_t op, uint32_t flags, const void *a, const void *b, const void *n,
                          const void *trace, uint64_t first_off, uint64_t elem_stride, uint64_t batch, const uint8_t *status,
                          const AdviceDst *dst_in, void *advice_out, uint64_t out_stride, h2r_stream_t stream) {
    if (!ctx || !a || !trace || (!dst_in && !advice_out)) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    const h2r_ctx::RowProg *rp = nullptr;
    int32_t rc = fresh_row_prog(ctx, op, flags, &rp);
    if (rc) return rc;
    const bool needs_b = op != FRESH_IS_ZERO, needs_n = op == FRESH_ADD_MOD || op == FRESH_SUB_MOD;
    if ((needs_b && !b) || (needs_n && !n)) return H2R_E_NULL;
    u64 es = 0;
    rc = h2r_fresh_op_layout(ctx, op, &es, nullptr, nullptr);
    if (rc) return rc;
    if (elem_stride == 0) elem_stride = es;
    const u64 rows = rp->host.size();
    if ((first_off & 15) || (elem_stride & 15)) return H2R_E_SHAPE;
    AdviceDst dst;
    if (dst_in) dst = *dst_in;
    else if ((rc = advice_dst(ctx, advice_out, out_stride, rows, batch, &dst))) return rc;
    if (batch == 0) return H2R_OK;
    RowProgArgs ra;
    std::memset(&ra, 0, sizeof ra);
    ra.a = a; ra.b = b ? b : a; ra.n = n ? n : a;
    ra.a_stride = ctx->L;
    ra.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    ra.b_stride = (!needs_n && (flags & H2R_F_SHARED_MODULUS)) ? 0 : ctx->L;   // as h2r_fresh_op_batch
    ra.trace = static_cast<const u8 *>(trace); ra.elem_stride = elem_stride; ra.first_off = first_off;
    ra.status = status; ra.batch = batch; ra.dst = dst;
    H2R_ON_DEVICE(ctx->params.device);
    return launch_row_prog(ctx, rp, ra, static_cast<hipStream_t>(stream));
}
}  // namespace

int32_t h2r_fresh_op_emit_advice(const h2r_ctx *ctx, uint32_t op, uint32_t flags, const void *a, const void *b, const void *n,
                                 const void *trace, uint64_t first_off, uint64_t elem_stride, uint64_t batch, const uint8_t *status,
                                 void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    return fresh_emit_advice(ctx, op, flags, a, b, n, trace, first_off, elem_stride, batch, status, nullptr, advice_out, out_stride, stream);
} H2R_CATCH_STATUS

// ---- the whole verify_pkcs1v15_signature element as advice rows ------------------------------------------------------------
namespace {
// Fork / join around the short kernels of a whole-element image.  Holds the ctx's side-stream mutex for the duration of the enqueue
// (the two events are re-recorded by every call: their meaning is fixed at enqueue time, so enqueues must not interleave).
struct SideFork {
    const h2r_ctx *ctx; hipStream_t main; std::unique_lock<std::mutex> lk; bool ok = false;
    SideFork(const h2r_ctx *c, hipStream_t st) : ctx(c), main(st), lk(c->side_mu) {
        // A CAPTURING caller does not fork: the ctx has ONE side stream, and once a capture has forked into it the stream stays part of
        // that capture until EndCapture -- a second emit on the same ctx before then (another thread, or this thread on a plain stream)
        // would record / wait events across the capture's boundary and invalidate it.  Inside a capture the row programs run in order
        // on the caller's stream (a graph has no use for the overlap of a 2 % tail anyway).
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
        if (cs != hipStreamCaptureStatusNone) return;
        if (!ctx->side_stream) {
            if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
        }
        ok = hipEventRecord(ctx->side_fork, main) == hipSuccess && hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    hipStream_t side() const { return ok ? ctx->side_stream : main; }   // (no side stream: everything in order on the caller's)
    int32_t join() {
        if (!ok) return H2R_OK;
        ok = false;
        HIP_TRY(hipEventRecord(ctx->side_join, ctx->side_stream));
        HIP_TRY(hipStreamWaitEvent(main, ctx->side_join, 0));
        return H2R_OK;
    }
    // an export that fails after the fork still joins: what it has queued on the side stream writes into the caller's image, and the
    // caller's stream is what the caller will order the image's release behind
    ~SideFork() { if (ok) (void)join(); }
    SideFork(const SideFork &) = delete;
    SideFork &operator=(const SideFork &) = delete;
};

int32_t verify_progs(const h2r_ctx *ctx, const h2r_ctx::RowProg **pre, const h2r_ctx::RowProg **inf, const h2r_ctx::RowProg **em) {
    if (ctx->layout.limb_width != 64 || ctx->L < 8) return H2R_E_UNSUPPORTED;   // RSAChip::LIMB_WIDTH
    int32_t rc = row_prog(ctx, kProgVerifyPre, [](RowProgBuilder &rb) { rb.build_verify_preamble(); return true; }, pre);
    if (!rc) rc = fresh_row_prog(ctx, FRESH_IS_IN_FIELD, H2R_ADVICE_ASSERT_ONE, inf);
    if (!rc) rc = row_prog(ctx, kProgEm, [](RowProgBuilder &rb) { rb.build_em(); return true; }, em);
    return rc;
}
}  // namespace

uint64_t h2r_verify_advice_rows(const h2r_ctx *ctx, const h2r_verify_layout *vl, uint64_t section_rows[4]) try {
    if (!ctx || !vl) return 0;
    const h2r_ctx::RowProg *pre, *inf, *em;
    if (verify_progs(ctx, &pre, &inf, &em)) return 0;
    const u64 r[4] = {pre->host.size(), inf->host.size(), h2r_pow_advice_rows(ctx, &vl->pow), em->host.size()};
    if (section_rows) for (int k = 0; k < 4; ++k) section_rows[k] = r[k];
    return r[0] + r[1] + r[2] + r[3];
} H2R_CATCH_ZERO

int32_t h2r_verify_row_kinds(const h2r_ctx *ctx, const h2r_verify_layout *vl, uint8_t *kinds_out) try {
    if (!ctx || !vl || !kinds_out) return H2R_E_NULL;
    const h2r_ctx::RowProg *pre, *inf, *em;
    const int32_t rc = verify_progs(ctx, &pre, &inf, &em);
    if (rc) return rc;
    uint8_t *k = kinds_out;
    for (const RpRow &r : pre->host) *k++ = (uint8_t)r.kind;
    for (const RpRow &r : inf->host) *k++ = (uint8_t)r.kind;
    if (int32_t rk = h2r_pow_row_kinds(ctx, &vl->pow, k)) return rk;
    k += h2r_pow_advice_rows(ctx, &vl->pow);
    for (const RpRow &r : em->host) *k++ = (uint8_t)r.kind;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_verify_emit_advice(const h2r_ctx *ctx, const h2r_verify_layout *vl, const void *sig, const void *n, const uint64_t *hashed,
                               const void *powed, uint32_t flags, const void *trace, const void *workspace, uint64_t batch,
                               const uint8_t *status, void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    if (!ctx || !vl || !sig || !n || !hashed || !powed || !trace || !workspace || !advice_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    const h2r_ctx::RowProg *pre, *inf, *em;
    int32_t rc = verify_progs(ctx, &pre, &inf, &em);
    if (rc) return rc;
    u64 sec[4];
    const u64 rows = h2r_verify_advice_rows(ctx, vl, sec);
    AdviceDst dst;
    if ((rc = advice_dst(ctx, advice_out, out_stride, rows, batch, &dst))) return rc;
    if (batch == 0) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the three row programs (seed row, assert_in_field, encoded-message check: 2 % of the bytes, latency-bound inverse launches among them)
    // run on the ctx's side stream NEXT to the pow rows' kernel -- other rows of the same image -- and are joined before returning
    SideFork fork(ctx, st);
    hipStream_t ss = fork.side();
    RowProgArgs ra;
    std::memset(&ra, 0, sizeof ra);
    ra.a = sig; ra.b = n; ra.n = n; ra.a_stride = ctx->L; ra.b_stride = ra.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    ra.trace = static_cast<const u8 *>(trace); ra.elem_stride = vl->elem_stride; ra.first_off = vl->off_in_field;
    ra.status = status; ra.batch = batch;
    ra.dst = dst;                                   // is_eq = assign_constant(1), src/chip.rs:137
    if ((rc = launch_row_prog(ctx, pre, ra, ss))) return rc;
    ra.dst = dst.at_row(sec[0]);                    // assert_in_field(sig, n), :106
    if ((rc = launch_row_prog(ctx, inf, ra, ss))) return rc;
    ra.a = powed; ra.b = hashed; ra.b_stride = 4; ra.first_off = vl->off_em;
    ra.dst = dst.at_row(sec[0] + sec[1] + sec[2]);                                                     // :138-198
    if ((rc = launch_row_prog(ctx, em, ra, ss))) return rc;
    rc = pow_emit_advice(ctx, &vl->pow, n, flags, trace, vl->elem_stride, workspace, batch, status,
                         dst.at_row(sec[0] + sec[1]), stream);   // pow_mod_fixed_exp / pow_mod, :108-111
    if (rc) return rc;
    return fork.join();
} H2R_CATCH_STATUS

// ---- copy constraints of the image, and its layout as data (h2r_copymap.hpp) -------------------------------------------------
uint32_t h2r_advice_copy_map(const h2r_ctx *ctx, h2r_copy *out, uint32_t cap) try {
    if (!ctx) return 0;
    std::vector<h2r_copy> v;
    copy_map_record(ctx->L, (ctx->layout.carry_nsub + 3) / 4, v);
    for (u32 i = 0; out && i < cap && i < v.size(); ++i) out[i] = v[i];
    return (uint32_t)v.size();
} H2R_CATCH_ZERO

int32_t h2r_pow_operand_sources(const h2r_ctx *ctx, const h2r_pow_layout *pl, const uint8_t *e_le_bytes, size_t e_len, int32_t *a_src, int32_t *b_src) try {
    if (!ctx || !pl || !a_src || !b_src) return H2R_E_NULL;
    if (pl->off_e_bits != UINT64_MAX) return H2R_E_UNSUPPORTED;   // (a Var element: acc comes from the bit's select rows, squared from record 2 bit - 1)
    ExpBits eb; u32 T;
    const int32_t rc = exp_to_bits(e_le_bytes, e_len, &eb, &T);
    if (rc) return rc;
    if (T != pl->num_mul_mods) return H2R_E_SHAPE;
    int32_t cur = H2R_SRC_X, acc = H2R_SRC_ONE;
    u32 t = 0;
    for (u32 bi = 0; bi < eb.nbits; ++bi) {   // square first, then the multiply with the value BEFORE the squaring (chip.rs:731-740)
        const int32_t sq = (int32_t)t;
        a_src[t] = b_src[t] = cur; ++t;
        if (exp_bit(eb.bytes, bi)) { a_src[t] = acc; b_src[t] = cur; acc = (int32_t)t; ++t; }
        cur = sq;
    }
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
int32_t layout_valid(const h2r_advice_layout *layout) {   // every entry a permutation of 0..4 (the table is public data: a hand-filled one must not index past a row)
    if (layout->version != H2R_ADVICE_LAYOUT_VERSION) return H2R_E_UNSUPPORTED;
    for (int k = 0; k < 256; ++k) {
        u32 seen = 0;
        for (int c = 0; c < 5; ++c) { const u32 v = layout->column_of[k][c]; if (v > 4 || (seen >> v) & 1u) return H2R_E_SHAPE; seen |= 1u << v; }
    }
    return H2R_OK;
}
// the kinds whose rows are RangeChip::assign decomposition rows (lookup-enabled): e carries "what remains", the lookups read a..d, the overflow lookup reads a
bool is_decompose_kind(u32 k) {
    return (k >= ROWK_RANGE_LIMB && k < ROWK_RANGE_CARRY + 8) || (k >= ROWK_RANGE_U32 && k < ROWK_RANGE_U32 + 8) ||
           (k >= ROWK_BITS_COMPOSE && k < ROWK_BITS_COMPOSE_LAST + 64);
}
// What h2r_advice_check / h2r_lookup_hist_advice also need of a (possibly hand-filled) layout: their lookup passes read the PHYSICAL columns
// 0..3 (composition) and 0 (overflow) of a decomposition row, so such a row must keep a at column 0 and e at column 4 -- the rule
// h2r_advice_layout_custom enforces (layout_perm_valid); any other table would give false violations or wrong multiplicities silently.
int32_t layout_lookup_valid(const h2r_advice_layout *layout) {
    if (const int32_t rc = layout_valid(layout)) return rc;
    for (u32 k = 0; k < 256; ++k)
        if (is_decompose_kind(k) && (layout->column_of[k][0] != 0 || layout->column_of[k][4] != 4)) return H2R_E_SHAPE;
    return H2R_OK;
}
}  // namespace

int32_t h2r_advice_layout_default(h2r_advice_layout *out) try {
    if (!out) return H2R_E_NULL;
    out->version = H2R_ADVICE_LAYOUT_VERSION;
    for (int k = 0; k < 256; ++k) for (int c = 0; c < 5; ++c) out->column_of[k][c] = (uint8_t)c;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_advice_layout_custom(const h2r_ctx *ctx, const uint8_t *kinds, const uint8_t (*column_of)[5], uint32_t n_kinds, h2r_advice_layout *out) try {
    if (!ctx || !out || (n_kinds && (!kinds || !column_of))) return H2R_E_NULL;
    h2r_advice_layout_default(out);
    for (u32 i = 0; i < n_kinds; ++i) {
        h2r_fixed_row f;
        const int32_t rc = h2r_advice_fixed_row(ctx, nullptr, kinds[i], &f);
        if (rc) return rc;
        const u32 k = kinds[i];
        const bool decompose = is_decompose_kind(k);
        u8 col[5];
        for (int c = 0; c < 5; ++c) col[c] = column_of[i][c];
        if (!layout_perm_valid(col, f, decompose)) return H2R_E_SHAPE;
        for (int c = 0; c < 5; ++c) out->column_of[k][c] = col[c];
    }
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_advice_fixed_row_ex(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const h2r_advice_layout *layout, uint32_t kind, h2r_fixed_row *out) try {
    if (!ctx || !layout || !out) return H2R_E_NULL;
    if (kind > 255) return H2R_E_SHAPE;
    if (const int32_t rv = layout_valid(layout)) return rv;
    h2r_fixed_row f;
    const int32_t rc = h2r_advice_fixed_row(ctx, cfg, kind, &f);
    if (rc) return rc;
    u8 col[5];
    for (int c = 0; c < 5; ++c) col[c] = layout->column_of[kind][c];
    layout_permute_fixed(col, f, out);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_advice_apply_layout(const h2r_ctx *ctx, const h2r_advice_layout *layout, const uint8_t *kinds_dev, uint64_t rows, void *image,
                                uint64_t out_stride, uint64_t batch, const uint8_t *status, h2r_stream_t stream) try {
    if (!ctx || !layout || !kinds_dev || !image) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (const int32_t rv = layout_valid(layout)) return rv;
    LayoutArgs la;
    std::memset(&la, 0, sizeof la);
    if (const int32_t rc = advice_dst(ctx, image, out_stride, rows, batch, &la.dst)) return rc;
    if (!rows || !batch) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    la.kinds = kinds_dev; la.rows = rows; la.batch = batch; la.status = status;
    std::memcpy(la.perm, layout->column_of, sizeof la.perm);
    const u64 blocks = (rows * batch + 255) / 256;
    if (blocks >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    hipLaunchKernelGGL(advice_layout_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), la);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- one RSAChip::modpow_public_key element as advice rows: [assert_in_field(x, n)] [pow_mod_fixed_exp] (src/chip.rs:106-111) ----
uint64_t h2r_modpow_public_key_advice_rows(const h2r_ctx *ctx, const h2r_pow_layout *pl, uint64_t section_rows[2]) try {
    if (!ctx || !pl) return 0;
    const u64 r[2] = {h2r_fresh_op_advice_rows(ctx, FRESH_IS_IN_FIELD, H2R_ADVICE_ASSERT_ONE), h2r_pow_advice_rows(ctx, pl)};
    if (!r[0]) return 0;
    if (section_rows) { section_rows[0] = r[0]; section_rows[1] = r[1]; }
    return r[0] + r[1];
} H2R_CATCH_ZERO

int32_t h2r_modpow_public_key_emit_advice(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *x, const void *n, uint32_t flags,
                                          const void *in_field_trace, const void *trace, const void *workspace, uint64_t batch,
                                          const uint8_t *status, void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    if (!ctx || !pl || !x || !n || !in_field_trace || !workspace || !advice_out) return H2R_E_NULL;
    u64 sec[2];
    const u64 rows = h2r_modpow_public_key_advice_rows(ctx, pl, sec);
    if (!rows) return H2R_E_UNSUPPORTED;
    AdviceDst dst;
    int32_t rc = advice_dst(ctx, advice_out, out_stride, rows, batch, &dst);
    if (rc) return rc;
    H2R_ON_DEVICE(ctx->params.device);
    SideFork fork(ctx, static_cast<hipStream_t>(stream));   // the in-field rows next to the pow rows (see h2r_verify_emit_advice)
    rc = fresh_emit_advice(ctx, FRESH_IS_IN_FIELD, (flags & H2R_F_SHARED_MODULUS) | H2R_ADVICE_ASSERT_ONE, x, n, nullptr, in_field_trace, 0, 0,
                           batch, status, &dst, nullptr, 0, static_cast<h2r_stream_t>(fork.side()));
    if (rc) return rc;
    rc = pow_emit_advice(ctx, pl, n, flags | (trace ? 0u : H2R_ADVICE_DIRECT), trace, 0, workspace, batch, status, dst.at_row(sec[0]), stream);
    if (rc) return rc;
    return fork.join();
} H2R_CATCH_STATUS

// The same element, PIPELINED (see h2r_pipeline_create): the chains and the assert_in_field witness of call k and its in-field rows on the
// caller's stream, its pow rows (cells_kernel, from the operands in the workspace) on a side stream of the pipeline, next to the chains
// of call k + 1.  No records are written.  The moduli the cells kernel needs are copied into the workspace inside the call, so x and n
// are read in `stream` order inside the call like every other pipelined form's inputs.
int32_t h2r_pipeline_modpow_public_key_advice(h2r_pipeline *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len, uint64_t batch,
                                              uint32_t flags, void *in_field_trace, void *out, uint8_t *status, void *workspace,
                                              void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    if (!p || !x || !n || !e_le || !in_field_trace || !status || !workspace || !advice_out) return H2R_E_NULL;
    const h2r_ctx *ctx = p->ctx;
    h2r_pow_layout pl;
    int32_t rc = h2r_pow_fixed_layout(ctx, e_le, e_len, &pl);
    if (rc) return rc;
    u64 sec[2];
    const u64 rows = h2r_modpow_public_key_advice_rows(ctx, &pl, sec);
    if (!rows) return H2R_E_UNSUPPORTED;
    AdviceDst dst;
    if ((rc = advice_dst(ctx, advice_out, out_stride, rows, batch, &dst))) return rc;
    if (batch == 0) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->pending) {   // records still owed by a call of another form: they go out alone, `st` behind them
        rc = pipeline_flush(p, st);
        if (rc) return rc;
    }
    const u32 slot = p->k % p->depth;
    p->done[slot] = DoneRef{};
    rc = h2r_modpow_public_key_batch(ctx, x, n, e_le, e_len, batch, flags, nullptr, in_field_trace, out, status, workspace, stream);
    if (rc) return rc;
    // the elements' moduli, where the record writers of the other forms find them too (Workspace::off_n)
    const h2r_layout &lo = ctx->layout;
    const Workspace wp = workspace_plan(lo.limb_bytes, ctx->L, batch, pl.num_mul_mods ? pl.num_mul_mods : 1);
    u8 *ws = reinterpret_cast<u8 *>(round_up(reinterpret_cast<u64>(workspace), 256));
    const bool shared = (flags & H2R_F_SHARED_MODULUS) != 0;
    HIP_TRY(hipMemcpyAsync(ws + wp.off_n, n, (shared ? 1ull : batch) * ctx->L * lo.limb_bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipEventRecord(p->chain_done[slot], st));
    rc = fresh_emit_advice(ctx, FRESH_IS_IN_FIELD, (flags & H2R_F_SHARED_MODULUS) | H2R_ADVICE_ASSERT_ONE, x, n, nullptr, in_field_trace, 0, 0,
                           batch, status, &dst, nullptr, 0, stream);
    if (rc) return rc;
    hipStream_t side = p->aux[p->k & 1];
    HIP_TRY(hipStreamWaitEvent(side, p->chain_done[slot], 0));
    rc = pow_emit_advice(ctx, &pl, ws + wp.off_n, (flags & H2R_F_SHARED_MODULUS) | H2R_ADVICE_DIRECT, nullptr, 0, workspace, batch, status,
                         dst.at_row(sec[0]), static_cast<h2r_stream_t>(side));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->trace_done[slot], side));
    p->done[slot] = DoneRef{p->trace_done[slot], 0, false};
    p->done_stream[slot] = side;
    p->k += 1;
    // lazy join, as in the other forms: the NEXT call reuses the buffers of call k - depth
    for (; p->joined + p->depth <= p->k; ++p->joined) {
        rc = pipeline_wait_slot(p, p->joined % p->depth, st);
        if (rc) return rc;
    }
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- the whole RSAChip::verify_pkcs1v15_signature element as advice rows, no records, pipelined --------------------------------
// The witness-only form of a verify layout: the element keeps its in-field and encoded-message witness (what the row programs read)
// and nothing else -- no record planes.  `pow` keeps its counts (rows, the number of mul_mods); pow.off_records = UINT64_MAX marks the absence.
int32_t h2r_verify_layout_compact(const h2r_ctx *ctx, const h2r_verify_layout *full, h2r_verify_layout *out) try {
    if (!ctx || !full || !out) return H2R_E_NULL;
    if (ctx->layout.limb_width != 64 || ctx->L < 9) return H2R_E_SHAPE;
    const AuxGeom g(ctx->L, 64);
    *out = *full;
    out->off_in_field = 0;
    out->off_em = round_up(g.in_field_sz(), 256);
    out->elem_stride = round_up(out->off_em + g.em_sz(), 256);
    out->pow.off_records = UINT64_MAX;   // no record planes: the record exports refuse this layout (H2R_E_SHAPE)
    if (full->pow.off_e_bits != UINT64_MAX) {   // a Var element: its pow witness (selected operands, result, exponent bits) behind the EM region
        h2r_pow_layout pc;
        const int32_t rc = h2r_pow_layout_compact(ctx, &full->pow, &pc);
        if (rc) return rc;
        const u64 base = out->elem_stride;
        pc.off_selected += base; pc.off_result += base; pc.off_e_bits += base;
        out->elem_stride = base + pc.elem_stride;
        pc.elem_stride = out->elem_stride;
        out->pow = pc;
    }
    return H2R_OK;
} H2R_CATCH_STATUS

// Pipelined like h2r_pipeline_modpow_public_key_advice: the chains, powed_out, the in-field / encoded-message witness, is_valid and the
// three short row programs (is_eq seed, assert_in_field, the encoded-message check) of call k on the caller's stream; its pow rows
// (cells_kernel, from the operands in the workspace) on a side stream of the pipeline, next to the chains of call k + 1.
namespace {
int32_t pipeline_verify_advice(h2r_pipeline *p, const void *sig, const void *n, const uint8_t *e_le, size_t e_len, const void *e_limbs,
                               uint32_t e_num_limbs, uint32_t exp_limb_bits, const uint64_t *hashed, uint64_t batch, uint32_t flags, void *witness,
                               void *powed_out, uint8_t *is_valid_out, uint8_t *status, void *workspace, void *advice_out, uint64_t out_stride,
                               h2r_stream_t stream) {
    if (!p || !sig || !n || (!e_le && !e_limbs) || !hashed || !witness || !powed_out || !status || !workspace || !advice_out) return H2R_E_NULL;
    if (reinterpret_cast<u64>(witness) & 15) return H2R_E_SHAPE;   // (16-byte stores into the witness sections)
    const h2r_ctx *ctx = p->ctx;
    const bool var = e_limbs != nullptr;
    h2r_verify_layout full, vl;
    int32_t rc = var ? h2r_verify_layout_var(ctx, e_num_limbs, exp_limb_bits, &full) : h2r_verify_layout_fixed(ctx, e_le, e_len, &full);
    if (rc) return rc;
    if ((rc = h2r_verify_layout_compact(ctx, &full, &vl))) return rc;
    const h2r_ctx::RowProg *pre, *inf, *em;
    if ((rc = verify_progs(ctx, &pre, &inf, &em))) return rc;
    u64 sec[4];
    const u64 rows = h2r_verify_advice_rows(ctx, &vl, sec);
    if (!rows) return H2R_E_UNSUPPORTED;
    AdviceDst dst;
    if ((rc = advice_dst(ctx, advice_out, out_stride, rows, batch, &dst))) return rc;
    if (batch == 0) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->pending) {   // records still owed by a call of another form: they go out alone, `st` behind them
        rc = pipeline_flush(p, st);
        if (rc) return rc;
    }
    const u32 slot = p->k % p->depth;
    p->done[slot] = DoneRef{};
    if (var) rc = run_path(ctx, CHAIN_POW_VAR, sig, nullptr, n, e_limbs, e_num_limbs, exp_limb_bits, nullptr, 1, batch, flags, vl.pow.num_mul_mods, witness,
                           vl.elem_stride, UINT64_MAX, &vl.pow, powed_out, status, workspace, st);
    else rc = pow_fixed_impl(ctx, sig, n, e_le, e_len, batch, flags, nullptr, powed_out, status, workspace, stream, 1);
    if (rc) return rc;
    if ((rc = launch_verify_aux(ctx, sig, n, hashed, batch, flags, witness, vl, powed_out, is_valid_out, status, st))) return rc;
    const h2r_layout &lo = ctx->layout;
    const Workspace wp = workspace_plan(lo.limb_bytes, ctx->L, batch, vl.pow.num_mul_mods ? vl.pow.num_mul_mods : 1);
    u8 *ws = reinterpret_cast<u8 *>(round_up(reinterpret_cast<u64>(workspace), 256));
    const bool shared = (flags & H2R_F_SHARED_MODULUS) != 0;
    HIP_TRY(hipMemcpyAsync(ws + wp.off_n, n, (shared ? 1ull : batch) * ctx->L * lo.limb_bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipEventRecord(p->chain_done[slot], st));
    RowProgArgs ra;
    std::memset(&ra, 0, sizeof ra);
    ra.a = sig; ra.b = n; ra.n = n; ra.a_stride = ctx->L; ra.b_stride = ra.n_stride = shared ? 0 : ctx->L;
    ra.trace = static_cast<const u8 *>(witness); ra.elem_stride = vl.elem_stride; ra.first_off = vl.off_in_field;
    ra.status = status; ra.batch = batch;
    ra.dst = dst;                                   // is_eq = assign_constant(1), src/chip.rs:137
    if ((rc = launch_row_prog(ctx, pre, ra, st))) return rc;
    ra.dst = dst.at_row(sec[0]);                    // assert_in_field(sig, n), :106
    if ((rc = launch_row_prog(ctx, inf, ra, st))) return rc;
    ra.a = powed_out; ra.b = hashed; ra.b_stride = 4; ra.first_off = vl.off_em;
    ra.dst = dst.at_row(sec[0] + sec[1] + sec[2]);  // :138-198
    if ((rc = launch_row_prog(ctx, em, ra, st))) return rc;
    hipStream_t side = p->aux[p->k & 1];
    HIP_TRY(hipStreamWaitEvent(side, p->chain_done[slot], 0));
    rc = pow_emit_advice(ctx, &vl.pow, ws + wp.off_n, (flags & H2R_F_SHARED_MODULUS) | H2R_ADVICE_DIRECT, var ? witness : nullptr, var ? vl.elem_stride : 0,
                         workspace, batch, status, dst.at_row(sec[0] + sec[1]), static_cast<h2r_stream_t>(side));   // pow_mod_fixed_exp :111 / pow_mod :109
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->trace_done[slot], side));
    p->done[slot] = DoneRef{p->trace_done[slot], 0, false};
    p->done_stream[slot] = side;
    p->k += 1;
    for (; p->joined + p->depth <= p->k; ++p->joined) {
        rc = pipeline_wait_slot(p, p->joined % p->depth, st);
        if (rc) return rc;
    }
    return H2R_OK;
}
}  // namespace

int32_t h2r_pipeline_verify_pkcs1v15_advice(h2r_pipeline *p, const void *sig, const void *n, const uint8_t *e_le, size_t e_len,
                                            const uint64_t *hashed, uint64_t batch, uint32_t flags, void *witness, void *powed_out,
                                            uint8_t *is_valid_out, uint8_t *status, void *workspace, void *advice_out, uint64_t out_stride,
                                            h2r_stream_t stream) try {
    if (!e_le) return H2R_E_NULL;
    return pipeline_verify_advice(p, sig, n, e_le, e_len, nullptr, 0, 0, hashed, batch, flags, witness, powed_out, is_valid_out, status, workspace,
                                  advice_out, out_stride, stream);
} H2R_CATCH_STATUS

// the RSAPubE::Var arm (src/chip.rs:108-110): the witness also keeps the pow_mod's exponent bits, selected operands and result
int32_t h2r_pipeline_verify_pkcs1v15_var_advice(h2r_pipeline *p, const void *sig, const void *n, const void *e_limbs, uint32_t e_num_limbs,
                                                uint32_t exp_limb_bits, const uint64_t *hashed, uint64_t batch, uint32_t flags, void *witness,
                                                void *powed_out, uint8_t *is_valid_out, uint8_t *status, void *workspace, void *advice_out,
                                                uint64_t out_stride, h2r_stream_t stream) try {
    if (!e_limbs) return H2R_E_NULL;
    return pipeline_verify_advice(p, sig, n, nullptr, 0, e_limbs, e_num_limbs, exp_limb_bits, hashed, batch, flags, witness, powed_out, is_valid_out, status,
                                  workspace, advice_out, out_stride, stream);
} H2R_CATCH_STATUS

// ---- RSAPubE::Var (src/chip.rs:108-110) without records -------------------------------------------------------------------
// The witness-only form of a pow layout: a Var element keeps its exponent bits, its selected operands and its result (what the to_bits /
// select rows read) and no record planes; a Fix element keeps its result.  off_records = UINT64_MAX marks the absence.
int32_t h2r_pow_layout_compact(const h2r_ctx *ctx, const h2r_pow_layout *full, h2r_pow_layout *out) try {
    if (!ctx || !full || !out) return H2R_E_NULL;
    const u64 limbs_bytes = (u64)ctx->L * ctx->layout.limb_bytes;
    *out = *full;
    out->off_records = UINT64_MAX;
    const bool var = full->off_e_bits != UINT64_MAX;
    u64 o = 0;
    if (var) { out->off_selected = 0; out->selected_stride = limbs_bytes; o = (u64)full->num_exp_bits * limbs_bytes; }
    out->off_result = o; o += limbs_bytes;
    if (var) { out->off_e_bits = o; o += full->num_exp_bits; }
    out->elem_stride = round_up(o, 256);
    return H2R_OK;
} H2R_CATCH_STATUS

// h2r_pipeline_modpow_public_key_advice for the Var arm: per-element exponents (pow_mod, big_integer/chip.rs:664-696).  `witness`: batch *
// h2r_pow_layout_compact(h2r_pow_var_layout(...)).elem_stride bytes.  The to_bits / select rows are written next to the cells kernel.
int32_t h2r_pipeline_modpow_public_key_var_advice(h2r_pipeline *p, const void *x, const void *e_limbs, uint32_t e_num_limbs, uint32_t exp_limb_bits,
                                                  const void *n, uint64_t batch, uint32_t flags, void *in_field_trace, void *witness, void *out,
                                                  uint8_t *status, void *workspace, void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    if (!p || !x || !e_limbs || !n || !in_field_trace || !witness || !status || !workspace || !advice_out) return H2R_E_NULL;
    if (reinterpret_cast<u64>(witness) & 15) return H2R_E_SHAPE;
    const h2r_ctx *ctx = p->ctx;
    h2r_pow_layout full, pl;
    int32_t rc = h2r_pow_var_layout(ctx, e_num_limbs, exp_limb_bits, &full);
    if (rc) return rc;
    if ((rc = h2r_pow_layout_compact(ctx, &full, &pl))) return rc;
    u64 sec[2];
    const u64 rows = h2r_modpow_public_key_advice_rows(ctx, &pl, sec);
    if (!rows) return H2R_E_UNSUPPORTED;
    AdviceDst dst;
    if ((rc = advice_dst(ctx, advice_out, out_stride, rows, batch, &dst))) return rc;
    if (batch == 0) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->pending) {   // records still owed by a call of another form: they go out alone, `st` behind them
        rc = pipeline_flush(p, st);
        if (rc) return rc;
    }
    const u32 slot = p->k % p->depth;
    p->done[slot] = DoneRef{};
    rc = run_path(ctx, CHAIN_POW_VAR, x, nullptr, n, e_limbs, e_num_limbs, exp_limb_bits, nullptr, 1, batch, flags, pl.num_mul_mods, witness,
                  pl.elem_stride, pl.off_records, &pl, out, status, workspace, st);
    if (rc) return rc;
    if ((rc = launch_in_field(ctx, x, n, batch, flags, in_field_trace, st))) return rc;
    const h2r_layout &lo = ctx->layout;
    const Workspace wp = workspace_plan(lo.limb_bytes, ctx->L, batch, pl.num_mul_mods ? pl.num_mul_mods : 1);
    u8 *ws = reinterpret_cast<u8 *>(round_up(reinterpret_cast<u64>(workspace), 256));
    const bool shared = (flags & H2R_F_SHARED_MODULUS) != 0;
    HIP_TRY(hipMemcpyAsync(ws + wp.off_n, n, (shared ? 1ull : batch) * ctx->L * lo.limb_bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipEventRecord(p->chain_done[slot], st));
    rc = fresh_emit_advice(ctx, FRESH_IS_IN_FIELD, (flags & H2R_F_SHARED_MODULUS) | H2R_ADVICE_ASSERT_ONE, x, n, nullptr, in_field_trace, 0, 0,
                           batch, status, &dst, nullptr, 0, stream);
    if (rc) return rc;
    hipStream_t side = p->aux[p->k & 1];
    HIP_TRY(hipStreamWaitEvent(side, p->chain_done[slot], 0));
    rc = pow_emit_advice(ctx, &pl, ws + wp.off_n, (flags & H2R_F_SHARED_MODULUS) | H2R_ADVICE_DIRECT, witness, pl.elem_stride, workspace, batch, status,
                         dst.at_row(sec[0]), static_cast<h2r_stream_t>(side));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->trace_done[slot], side));
    p->done[slot] = DoneRef{p->trace_done[slot], 0, false};
    p->done_stream[slot] = side;
    p->k += 1;
    for (; p->joined + p->depth <= p->k; ++p->joined) {
        rc = pipeline_wait_slot(p, p->joined % p->depth, st);
        if (rc) return rc;
    }
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- the hashed-message limbs of RSASignatureVerifier as advice rows (src/lib.rs:225-239) ---------------------------------
namespace {
constexpr u32 kProgHashedMsg = 0x1002;
int32_t hashed_msg_prog(const h2r_ctx *ctx, const h2r_ctx::RowProg **out) {
    if (ctx->layout.limb_width != 64) return H2R_E_UNSUPPORTED;   // RSAChip::LIMB_WIDTH (limb_bytes = 8, src/lib.rs:215)
    return row_prog(ctx, kProgHashedMsg, [](RowProgBuilder &rb) { rb.build_hashed_msg(); return true; }, out);
}
}  // namespace

uint32_t h2r_hashed_msg_advice_rows(const h2r_ctx *ctx) try {
    const h2r_ctx::RowProg *rp = nullptr;
    if (!ctx || hashed_msg_prog(ctx, &rp)) return 0;
    return (uint32_t)rp->host.size();
} H2R_CATCH_ZERO

int32_t h2r_hashed_msg_row_kinds(const h2r_ctx *ctx, uint8_t *kinds_out) try {
    if (!ctx || !kinds_out) return H2R_E_NULL;
    const h2r_ctx::RowProg *rp = nullptr;
    const int32_t rc = hashed_msg_prog(ctx, &rp);
    if (rc) return rc;
    for (size_t r = 0; r < rp->host.size(); ++r) kinds_out[r] = (uint8_t)rp->host[r].kind;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_hashed_msg_emit_advice(const h2r_ctx *ctx, const void *hm_trace, uint64_t hm_stride, uint64_t batch, const uint8_t *status,
                                   void *advice_out, uint64_t out_stride, h2r_stream_t stream) try {
    if (!ctx || !hm_trace || !advice_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    const h2r_ctx::RowProg *rp = nullptr;
    const int32_t rc = hashed_msg_prog(ctx, &rp);
    if (rc) return rc;
    if (hm_stride == 0) hm_stride = HM_REGION;
    if ((hm_stride & 15) || hm_stride < HM_REGION) return H2R_E_SHAPE;
    RowProgArgs ra;
    std::memset(&ra, 0, sizeof ra);
    if (const int32_t rd = advice_dst(ctx, advice_out, out_stride, rp->host.size(), batch, &ra.dst)) return rd;
    if (batch == 0) return H2R_OK;
    ra.a = ra.b = ra.n = hm_trace;   // no operand cells in this program
    ra.trace = static_cast<const u8 *>(hm_trace); ra.elem_stride = hm_stride; ra.first_off = 0;
    ra.status = status; ra.batch = batch;
    H2R_ON_DEVICE(ctx->params.device);
    return launch_row_prog(ctx, rp, ra, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

// ---- audit of an advice image: the device-side MockProver (h2r_check.hpp) ------------------------------------------------------
namespace {
int32_t check_table(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const h2r_advice_layout *layout, const CheckKind **out) {
    std::vector<u8> key(sizeof(h2r_lookup_config) + sizeof layout->column_of, 0);
    if (cfg) std::memcpy(key.data(), cfg, sizeof *cfg);
    std::memcpy(key.data() + sizeof(h2r_lookup_config), layout->column_of, sizeof layout->column_of);
    std::lock_guard<std::mutex> lk(ctx->check_mu);
    auto it = ctx->check_tabs.find(key);
    if (it != ctx->check_tabs.end()) { *out = it->second; return H2R_OK; }
    std::vector<CheckKind> tab(256);
    std::memset(static_cast<void *>(tab.data()), 0, tab.size() * sizeof(CheckKind));
    for (u32 k = 0; k < 256; ++k) {
        h2r_fixed_row f, g;
        if (fixed_row_repr(ctx, cfg, k, true, &f)) continue;   // (an unknown kind, or one whose lookup the configuration has no table for: rows of it are flagged)
        u8 col[5];
        for (int c = 0; c < 5; ++c) col[c] = layout->column_of[k][c];
        layout_permute_fixed(col, f, &g);
        const uint64_t (*sel[9])[4] = {&g.sa, &g.sb, &g.sc, &g.sd, &g.se, &g.s_mul_ab, &g.s_mul_cd, &g.se_next, &g.s_const};
        CheckKind &ck = tab[k];
        for (int q = 0; q < 9; ++q) { for (int w = 0; w < 4; ++w) ck.s[q].v[w] = (*sel[q])[w]; if (!fe_is_zero(ck.s[q])) ck.nz |= 1u << q; }
        auto bits_of = [&](u32 tag) -> u32 { if (!tag || !cfg) return 0; for (u32 i = 0; i < cfg->n_lens; ++i) if (cfg->tag[i] == tag) return cfg->bit_len[i]; return 0; };
        ck.comp_bits = bits_of(g.tag_composition); ck.ov_bits = bits_of(g.tag_overflow);
        ck.valid = 1;
    }
    CheckKind *dev = nullptr;
    DeviceGuard dg(ctx->params.device);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dev), tab.size() * sizeof(CheckKind)));
    if (hipMemcpy(dev, tab.data(), tab.size() * sizeof(CheckKind), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); return H2R_E_HIP; }
    ctx->check_tabs.emplace(std::move(key), dev);
    *out = dev;
    return H2R_OK;
}
}  // namespace

int32_t h2r_advice_check(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const h2r_advice_layout *layout, const uint8_t *kinds_dev, uint64_t rows,
                         const void *image, uint64_t out_stride, uint64_t batch, const uint8_t *status, const h2r_copy *copies_dev, uint64_t n_copies,
                         const void *src_a, const void *src_b, const void *src_n, uint32_t flags, uint32_t *bad_out, uint64_t *first_bad_out,
                         h2r_stream_t stream) try {
    if (!ctx || !kinds_dev || !image || !bad_out || !first_bad_out || (n_copies && !copies_dev)) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (flags & ~H2R_F_SHARED_MODULUS) return H2R_E_UNSUPPORTED;
    h2r_advice_layout ident;
    if (!layout) { h2r_advice_layout_default(&ident); layout = &ident; }
    else if (const int32_t rc = layout_lookup_valid(layout)) return rc;
    AdviceCheckArgs ca;
    std::memset(static_cast<void *>(&ca), 0, sizeof ca);
    if (const int32_t rc = advice_dst(ctx, const_cast<void *>(image), out_stride, rows, batch, &ca.img)) return rc;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(bad_out, 0, batch * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(first_bad_out, 0, batch * sizeof(uint64_t), st));
    if (!rows || !batch) return H2R_OK;
    if (const int32_t rc = check_table(ctx, cfg, layout, &ca.tab)) return rc;
    ca.kinds = kinds_dev; ca.rows = rows; ca.batch = batch; ca.status = status; ca.f = ctx->fc;
    ca.bad = bad_out; ca.first = reinterpret_cast<unsigned long long *>(first_bad_out);
    ca.copies = copies_dev; ca.n_copies = n_copies;
    ca.ext[0] = src_a; ca.ext[1] = src_b; ca.ext[2] = src_n;
    ca.ext_stride[0] = ca.ext_stride[1] = ctx->L; ca.ext_stride[2] = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    ca.limb_bytes = ctx->layout.limb_bytes;
    std::memcpy(ca.perm, layout->column_of, sizeof ca.perm);
    u64 blocks = (rows * batch + 255) / 256;
    if (blocks >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    hipLaunchKernelGGL(advice_check_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ca);
    HIP_TRY(hipGetLastError());
    if (n_copies) {
        blocks = (n_copies * batch + 255) / 256;
        if (blocks >= (1ull << 31)) return H2R_E_UNSUPPORTED;
        hipLaunchKernelGGL(advice_check_copies_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ca);
        HIP_TRY(hipGetLastError());
    }
    return H2R_OK;
} H2R_CATCH_STATUS

// The lookup multiplicities of an advice image (advice_hist_kernel): hist[elem][5][n_rows] uint32, ADDED to like every h2r_lookup_hist_*.
int32_t h2r_lookup_hist_advice(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const h2r_advice_layout *layout, const uint8_t *kinds_dev,
                               uint64_t rows, const void *image, uint64_t image_stride, uint64_t batch, const uint8_t *status, uint32_t *hist,
                               h2r_stream_t stream) try {
    if (!ctx || !cfg || !kinds_dev || !image || !hist) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (cfg->n_rows == 0 || cfg->n_rows > (u32)LOOKUP_MAX_ROWS || cfg->n_lens == 0 || cfg->n_lens > H2R_LOOKUP_MAX_LENS) return H2R_E_SHAPE;
    h2r_advice_layout ident;
    if (!layout) { h2r_advice_layout_default(&ident); layout = &ident; }
    else if (const int32_t rc = layout_lookup_valid(layout)) return rc;
    AdviceHistArgs ha;
    std::memset(static_cast<void *>(&ha), 0, sizeof ha);
    if (const int32_t rc = advice_dst(ctx, const_cast<void *>(image), image_stride, rows, batch, &ha.img)) return rc;
    if (!rows || !batch) return H2R_OK;
    H2R_ON_DEVICE(ctx->params.device);
    if (const int32_t rc = check_table(ctx, cfg, layout, &ha.tab)) return rc;
    ha.kinds = kinds_dev; ha.rows = rows; ha.batch = batch; ha.status = status; ha.f = ctx->fc;
    ha.hist = hist; ha.n_rows = cfg->n_rows; ha.n_lens = cfg->n_lens;
    for (u32 i = 0; i < cfg->n_lens; ++i) { ha.bit_len[i] = cfg->bit_len[i]; ha.row_off[i] = cfg->row_off[i]; }
    const u64 blocks = batch * ((rows + HIST_ROWS_PER_WG - 1) / HIST_ROWS_PER_WG);
    if (blocks >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    hipLaunchKernelGGL(advice_hist_kernel, dim3((unsigned)blocks), dim3(256), 5u * cfg->n_rows * sizeof(u32), static_cast<hipStream_t>(stream), ha);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

// The copy constraints of one fixed-exponent pow element (h2r_pow_trace_emit_advice's image: CONST1, CONST0, then the records), rows
// counted from `row_offset` (the pow section's first row inside a larger element image): every record's own pairs, and its operand
// limbs tied to where pow_mod_fixed_exp takes them from (big_integer/chip.rs:729-740).
uint64_t h2r_pow_copy_map(const h2r_ctx *ctx, const h2r_pow_layout *pl, const uint8_t *e_le_bytes, size_t e_len, uint64_t row_offset,
                          h2r_copy *out, uint64_t cap) try {
    if (!ctx || !pl || pl->off_e_bits != UINT64_MAX) return 0;
    std::vector<int32_t> as(pl->num_mul_mods), bs(pl->num_mul_mods);
    if (h2r_pow_operand_sources(ctx, pl, e_le_bytes, e_len, as.data(), bs.data())) return 0;
    std::vector<h2r_copy> rec;
    const u32 L = ctx->L, rows = h2r_advice_rows(ctx);
    copy_map_record(L, (ctx->layout.carry_nsub + 3) / 4, rec);
    u64 n = 0;
    auto push = [&](u64 row, u32 col, u64 src_row, u32 src_col) {
        if (out && n < cap) out[n] = h2r_copy{(uint32_t)row, col, (uint32_t)src_row, src_col};
        ++n;
    };
    if (row_offset + 2 + (u64)pl->num_mul_mods * rows >= 0xFFFFFF00ull) return 0;
    for (u32 t = 0; t < pl->num_mul_mods; ++t) {
        const u64 base = row_offset + 2 + (u64)t * rows;
        auto operand = [&](int32_t src, u32 limb, u64 *srow, u32 *scol) {   // where limb `limb` of an operand lives
            if (src == H2R_SRC_X) { *srow = H2R_COPY_SRC_A; *scol = limb; }
            else if (src == H2R_SRC_ONE) { *srow = row_offset + (limb ? 1 : 0); *scol = 0; }   // acc = assign_constant(1): CONST1, then the shared CONST0 cell
            else { *srow = row_offset + 2 + (u64)src * rows + 2 * (L + limb); *scol = 4; }      // the r limbs of record `src`: column e of the assign's first row
        };
        for (const h2r_copy &c : rec) {
            u64 srow; u32 scol;
            if (c.src_row == H2R_COPY_SRC_A) operand(as[t], c.src_col, &srow, &scol);
            else if (c.src_row == H2R_COPY_SRC_B) operand(bs[t], c.src_col, &srow, &scol);
            else if (c.src_row == H2R_COPY_SRC_N) { srow = H2R_COPY_SRC_N; scol = c.src_col; }
            else { srow = base + c.src_row; scol = c.src_col; }
            push(base + c.row, c.col, srow, scol);
        }
    }
    return n;
} H2R_CATCH_ZERO

// ---- in-place audit ----------------------------------------------------------------------------------------
namespace {
int32_t launch_check(const h2r_ctx *ctx, CheckArgs &ca, const uint8_t *status, uint32_t *bad_out, uint32_t *first_bad_out,
                     u64 n_elems, hipStream_t st) {
    const h2r_layout &lo = ctx->layout;
    if (lo.num_limbs > 128) return H2R_E_UNSUPPORTED;
    for (int p = 0; p < H2R_PL_COUNT; ++p) ca.off[p] = lo.plane_off[p];
    ca.wm[0] = ctx->word_max.v[0]; ca.wm[1] = ctx->word_max.v[1]; ca.wm[2] = ctx->word_max.v[2];
    ca.L = lo.num_limbs; ca.carry_bits = lo.carry_bits; ca.carry_sub_bits = lo.carry_sub_bits; ca.carry_nsub = lo.carry_nsub;
    ca.carry_sub_stride = lo.carry_sub_stride; ca.record_stride = lo.record_stride;
    ca.status = status; ca.bad = bad_out; ca.first_bad = first_bad_out;
    HIP_TRY(hipMemsetAsync(bad_out, 0, n_elems * sizeof(uint32_t), st));
    if (first_bad_out) HIP_TRY(hipMemsetAsync(first_bad_out, 0, n_elems * sizeof(uint32_t), st));
    if (ca.n_items == 0) return H2R_OK;
    if (ca.n_items >= (1ull << 31)) return H2R_E_UNSUPPORTED;
    if (lo.limb_width == 64) hipLaunchKernelGGL((check_kernel<64>), dim3((unsigned)ca.n_items), dim3(256), 0, st, ca);
    else hipLaunchKernelGGL((check_kernel<32>), dim3((unsigned)ca.n_items), dim3(256), 0, st, ca);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
}
}  // namespace

int32_t h2r_mul_mod_trace_check(const h2r_ctx *ctx, const void *a, const void *b, const void *n, uint32_t flags, const void *trace,
                                uint64_t batch, const uint8_t *status, uint32_t *bad_out, uint32_t *first_bad_out, h2r_stream_t stream) try {
    if (!ctx || !a || !b || !n || !trace || !bad_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    H2R_ON_DEVICE(ctx->params.device);
    CheckArgs ca;
    std::memset(&ca, 0, sizeof ca);
    ca.opA = a; ca.opB = b; ca.op_stride = ctx->L;
    ca.n = n; ca.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    ca.trace = static_cast<const u8 *>(trace); ca.elem_stride = ctx->layout.record_stride; ca.off_records = 0; ca.T = 1; ca.n_items = batch;
    return launch_check(ctx, ca, status, bad_out, first_bad_out, batch, static_cast<hipStream_t>(stream));
} H2R_CATCH_STATUS

int32_t h2r_pow_trace_check(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *x, const void *n, const uint8_t *e_le,
                            size_t e_len, uint32_t flags, const void *trace, uint64_t elem_stride, const void *workspace,
                            uint64_t batch, const uint8_t *status, uint32_t *bad_out, uint32_t *first_bad_out, h2r_stream_t stream) try {
    if (!ctx || !pl || !x || !n || !trace || !workspace || !bad_out) return H2R_E_NULL;
    if (pl->off_records == UINT64_MAX) return H2R_E_SHAPE;   // a witness-only layout (h2r_pow_layout_compact) holds no records to audit
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    const bool var = pl->off_e_bits != UINT64_MAX;
    if (!var && !e_le && e_len) return H2R_E_NULL;
    LinkArgs la;
    std::memset(&la, 0, sizeof la);
    if (!var) {
        u32 T = 0;
        const int32_t rc = exp_to_bits(e_le, e_len, &la.e, &T);
        if (rc) return rc;
        if (T != pl->num_mul_mods) return H2R_E_SHAPE;
    }
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const h2r_layout &lo = ctx->layout;
    const u64 lb = lo.limb_bytes;
    const u8 *ws = reinterpret_cast<const u8 *>(round_up(reinterpret_cast<u64>(workspace), 256));   // as run_path carves it
    CheckArgs ca;
    std::memset(&ca, 0, sizeof ca);
    ca.opA = ws; ca.opB = ws + ctx->L * lb; ca.opQ = ws + 2 * ctx->L * lb; ca.opR = ws + 3 * ctx->L * lb; ca.op_stride = 4ull * ctx->L;
    ca.n = n; ca.n_stride = (flags & H2R_F_SHARED_MODULUS) ? 0 : ctx->L;
    ca.trace = static_cast<const u8 *>(trace); ca.elem_stride = elem_stride ? elem_stride : pl->elem_stride;
    ca.off_records = pl->off_records; ca.T = pl->num_mul_mods ? pl->num_mul_mods : 1;
    ca.n_items = pl->num_mul_mods ? batch * pl->num_mul_mods : 0;
    int32_t rc = launch_check(ctx, ca, status, bad_out, first_bad_out, batch, st);
    if (rc || batch == 0) return rc;
    la.x = x; la.ops = ws; la.op_stride = 4ull * ctx->L; la.L = ctx->L; la.T = pl->num_mul_mods; la.var = var ? 1u : 0u;
    la.nbits = var ? pl->num_exp_bits : la.e.nbits;
    la.status = status; la.trace = ca.trace; la.elem_stride = ca.elem_stride;
    la.off_e_bits = pl->off_e_bits; la.off_selected = pl->off_selected; la.selected_stride = pl->selected_stride; la.off_result = pl->off_result;
    la.bad = bad_out; la.first_bad = first_bad_out;
    if (lo.limb_width == 64) hipLaunchKernelGGL((link_kernel<64>), dim3((unsigned)batch), dim3(64), 0, st, la);
    else hipLaunchKernelGGL((link_kernel<32>), dim3((unsigned)batch), dim3(64), 0, st, la);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

// ---- BigIntInstructions::mul / square, is_equal_muled, refresh -----------------
uint64_t h2r_mul_stream_bytes(const h2r_ctx *ctx) try { return ctx ? (u64)ctx->L * ctx->L * ctx->layout.wide_bytes : 0; } H2R_CATCH_ZERO
uint64_t h2r_is_equal_muled_stream_bytes(const h2r_ctx *ctx) try {
    if (!ctx) return 0;
    const h2r_layout &lo = ctx->layout;
    return (u64)lo.num_cols * (5ull * lo.wide_bytes + 2ull * lo.carry_bytes + 4ull * lo.limb_bytes + 4) +
           (u64)(lo.num_cols - 1) * (lo.carry_bytes + lo.carry_nsub);
} H2R_CATCH_ZERO
uint64_t h2r_refresh_stream_bytes(const h2r_ctx *ctx) try {
    if (!ctx) return 0;
    const h2r_layout &lo = ctx->layout;
    u64 b = 0;
    for (u32 i = 0; i < ctx->refresh_nf; ++i)
        b += (u64)(ctx->refresh_inc[i] + 1) * (lo.carry_bytes + lo.limb_bytes + lo.wide_bytes + lo.limb_bytes) + (u64)ctx->refresh_inc[i] * lo.wide_bytes;
    return b + (u64)ctx->refresh_nf * (lo.limb_bytes + lo.limb_nsub);
} H2R_CATCH_ZERO

int32_t h2r_mul_batch(const h2r_ctx *ctx, const void *a, const void *b, uint64_t batch, void *trace, uint64_t *muled_out,
                      h2r_stream_t stream) try {
    if (!ctx || !a || !b || !trace || !muled_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (batch == 0) return H2R_OK;
    if (batch >= (1ull << 32)) return H2R_E_UNSUPPORTED;
    TraceArgs ta;
    fill_trace_args(ctx, ta);
    ta.mode = TRACE_MUL; ta.opA = a; ta.opB = b; ta.op_stride = ctx->L; ta.n_items = batch; ta.T = 1;
    ta.trace = static_cast<u8 *>(trace); ta.elem_stride = ctx->layout.record_stride; ta.off_records = 0;
    ta.muled_out = muled_out;
    H2R_ON_DEVICE(ctx->params.device);
    ProfScope ps(H2R_KERNEL_TRACE, static_cast<hipStream_t>(stream));
    HIP_TRY(launch_trace(ctx, ta, static_cast<hipStream_t>(stream)));
    return H2R_OK;
} H2R_CATCH_STATUS
int32_t h2r_mul_trace_flatten(const h2r_ctx *ctx, const void *record_host, void *stream_out) try {
    if (!ctx || !record_host || !stream_out) return H2R_E_NULL;
    flatten_parts(ctx->layout, static_cast<const u8 *>(record_host), static_cast<u8 *>(stream_out), 2);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_is_equal_muled_batch(const h2r_ctx *ctx, const uint64_t *muled_a, const uint64_t *muled_b, uint64_t batch,
                                 void *trace, uint8_t *eq_out, h2r_stream_t stream) try {
    if (!ctx || !muled_a || !muled_b || !trace) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (batch == 0) return H2R_OK;
    if (batch >= (1ull << 32)) return H2R_E_UNSUPPORTED;
    TraceArgs ta;
    fill_trace_args(ctx, ta);
    ta.mode = TRACE_EQ; ta.n_items = batch; ta.T = 1;
    ta.trace = static_cast<u8 *>(trace); ta.elem_stride = ctx->layout.record_stride; ta.off_records = 0;
    ta.muled_a = muled_a; ta.muled_b = muled_b; ta.eq_out = eq_out;
    H2R_ON_DEVICE(ctx->params.device);
    ProfScope ps(H2R_KERNEL_TRACE, static_cast<hipStream_t>(stream));
    HIP_TRY(launch_trace(ctx, ta, static_cast<hipStream_t>(stream)));
    return H2R_OK;
} H2R_CATCH_STATUS
int32_t h2r_is_equal_muled_flatten(const h2r_ctx *ctx, const void *record_host, void *stream_out) try {
    if (!ctx || !record_host || !stream_out) return H2R_E_NULL;
    flatten_parts(ctx->layout, static_cast<const u8 *>(record_host), static_cast<u8 *>(stream_out), 16);
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
// RefreshAux::new(w, n_l, n_r) and the stream geometry of BigIntChip::refresh with it
struct RefreshPlan { u32 nf = 0; u8 inc[MULED_MAX] = {}; u32 off[MULED_MAX] = {}; u32 range_off = 0, stream_bytes = 0; };
int32_t refresh_plan(const h2r_ctx *ctx, u32 n_l, u32 n_r, RefreshPlan &rp) {
    const h2r_layout &lo = ctx->layout;
    if (n_l == 0 || n_r == 0 || n_l > ctx->L || n_r > ctx->L) return H2R_E_SHAPE;   // the stream widths are the ctx's: values of longer operands need not fit
    u8 inc[2 * 128 + 8] = {};
    rp.nf = refresh_aux_increased_limbs(lo.limb_width, n_l, n_r, inc);
    if (rp.nf > (u32)MULED_MAX) return H2R_E_UNSUPPORTED;
    u64 off = 0;
    for (u32 i = 0; i < rp.nf; ++i) {
        if (inc[i] > 2) return H2R_E_UNSUPPORTED;
        rp.inc[i] = inc[i]; rp.off[i] = (u32)off;
        off += (u64)(inc[i] + 1) * (lo.carry_bytes + 2ull * lo.limb_bytes + lo.wide_bytes) + (u64)inc[i] * lo.wide_bytes;
    }
    rp.range_off = (u32)off;
    rp.stream_bytes = (u32)(off + (u64)rp.nf * (lo.limb_bytes + lo.limb_nsub));
    return H2R_OK;
}
}  // namespace

int32_t h2r_refresh_layout(const h2r_ctx *ctx, uint32_t num_limbs_l, uint32_t num_limbs_r, uint32_t *num_limbs_fresh,
                           uint64_t *stream_bytes, uint64_t *elem_stride) try {
    if (!ctx) return H2R_E_NULL;
    RefreshPlan rp;
    const int32_t rc = refresh_plan(ctx, num_limbs_l, num_limbs_r, rp);
    if (rc) return rc;
    if (num_limbs_fresh) *num_limbs_fresh = rp.nf;
    if (stream_bytes) *stream_bytes = rp.stream_bytes;
    if (elem_stride) *elem_stride = round_up(rp.stream_bytes, 256);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_refresh_batch_ex(const h2r_ctx *ctx, const uint64_t *muled, uint64_t muled_stride_cols, uint32_t num_limbs_l,
                             uint32_t num_limbs_r, uint64_t batch, void *trace, void *fresh_out, uint8_t *status, h2r_stream_t stream) try {
    if (!ctx || !muled || !trace || !status) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    RefreshPlan rp;
    const int32_t rc = refresh_plan(ctx, num_limbs_l, num_limbs_r, rp);
    if (rc) return rc;
    if (muled_stride_cols < num_limbs_l + num_limbs_r - 1) return H2R_E_SHAPE;
    if (batch == 0) return H2R_OK;
    const h2r_layout &lo = ctx->layout;
    RefreshArgs ra;
    std::memset(&ra, 0, sizeof ra);
    ra.muled = muled; ra.muled_stride = muled_stride_cols; ra.batch = batch; ra.d = num_limbs_l + num_limbs_r - 1; ra.nf = rp.nf; ra.w = lo.limb_width;
    std::memcpy(ra.inc, rp.inc, sizeof ra.inc); std::memcpy(ra.off, rp.off, sizeof ra.off);
    ra.range_off = rp.range_off; ra.trace = static_cast<u8 *>(trace); ra.elem_stride = round_up(rp.stream_bytes, 256);
    ra.fresh_out = fresh_out; ra.status = status; ra.LB = lo.limb_bytes; ra.WB = lo.wide_bytes; ra.CB = lo.carry_bytes; ra.stream_bytes = rp.stream_bytes;
    const unsigned stage = (unsigned)round_up(rp.stream_bytes, 16) + 32;
    if (stage > 60 * 1024) return H2R_E_UNSUPPORTED;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stage > 40 * 1024) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&refresh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage); (void)hipGetLastError(); }
    ProfScope ps(H2R_KERNEL_AUX, st);
    hipLaunchKernelGGL(refresh_kernel, dim3((unsigned)batch), dim3(256), stage, st, ra);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_refresh_batch(const h2r_ctx *ctx, const uint64_t *muled, uint64_t batch, void *trace, void *fresh_out,
                          uint8_t *status, h2r_stream_t stream) try {
    if (!ctx) return H2R_E_NULL;
    return h2r_refresh_batch_ex(ctx, muled, 2ull * ctx->L, ctx->L, ctx->L, batch, trace, fresh_out, status, stream);
} H2R_CATCH_STATUS

// ---- general operand shapes: mul(d0, d1), is_equal_muled(n_l, n_r) --------------------------------------------------------
uint64_t h2r_mul_stream_bytes_ex(const h2r_ctx *ctx, uint32_t d0, uint32_t d1) try { return ctx ? (u64)d0 * d1 * ctx->layout.wide_bytes : 0; } H2R_CATCH_ZERO

int32_t h2r_mul_batch_ex(const h2r_ctx *ctx, const void *a, uint32_t d0, const void *b, uint32_t d1, uint64_t batch, void *trace,
                         uint64_t *muled_out, h2r_stream_t stream) try {
    if (!ctx || !a || !b || !trace || !muled_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    if (d0 == 0 || d1 == 0 || d0 > ctx->L || d1 > ctx->L) return H2R_E_SHAPE;
    if (d0 == ctx->L && d1 == ctx->L) return h2r_mul_batch(ctx, a, b, batch, trace, muled_out, stream);
    if (batch == 0) return H2R_OK;
    // zero-padded to num_limbs limbs the products are the same (padding adds zero terms only); the reference's partial
    // accumulators are the entries (j, i) with j < d0, i - j < d1 of the padded record (h2r_mul_trace_flatten_ex walks those)
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const u64 lb = ctx->layout.limb_bytes, one = batch * ctx->L * lb;
    u8 *pad = nullptr;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&pad), 2 * one, st));
    for (int k = 0; k < 2; ++k) {
        PadArgs pa{static_cast<const u8 *>(k ? b : a), pad + k * one, batch, k ? d1 : d0, ctx->L, (u32)lb};
        hipLaunchKernelGGL(pad_limbs_kernel, dim3((unsigned)std::min<u64>(2048, (batch * ctx->L + 255) / 256)), dim3(256), 0, st, pa);
    }
    const int32_t rc = hipGetLastError() == hipSuccess ? h2r_mul_batch(ctx, pad, pad + one, batch, trace, muled_out, stream) : H2R_E_HIP;
    (void)hipFreeAsync(pad, st);
    return rc;
} H2R_CATCH_STATUS

int32_t h2r_mul_trace_flatten_ex(const h2r_ctx *ctx, const void *record_host, uint32_t d0, uint32_t d1, void *stream_out) try {
    if (!ctx || !record_host || !stream_out) return H2R_E_NULL;
    if (d0 == 0 || d1 == 0 || d0 > ctx->L || d1 > ctx->L) return H2R_E_SHAPE;
    const h2r_layout &lo = ctx->layout;
    Out o{static_cast<u8 *>(stream_out)};
    const u8 *rec = static_cast<const u8 *>(record_host);
    for (u32 i = 0; i < d0 + d1 - 1; ++i) {   // chip.rs:400-412 with (d0, d1)
        u32 j = (d1 >= i + 1) ? 0 : i + 1 - d1;
        for (; j < d0 && j <= i; ++j) emit_acc(o, lo, rec, H2R_PL_AB_LO, j, i % lo.num_limbs);
    }
    return H2R_OK;
} H2R_CATCH_STATUS

namespace {
struct EqPlan { U256 wm; u32 carry_bits, sub_bits, nsub, per_col; u64 stream_bytes; };
int32_t eq_plan(const h2r_ctx *ctx, u32 n_l, u32 n_r, u32 flags, EqPlan &ep) {
    const h2r_layout &lo = ctx->layout;
    if (n_l == 0 || n_r == 0 || n_l > ctx->L || n_r > ctx->L || n_l + n_r - 1 > 255) return H2R_E_SHAPE;
    if (flags & ~H2R_STREAM_FIELD_AB) return H2R_E_UNSUPPORTED;
    ep.wm = compute_mul_word_max(lo.limb_width, std::min(n_l, n_r));                 // chip.rs:838
    ep.carry_bits = (ep.wm + ep.wm).bits() - lo.limb_width;                           // :841-842
    ep.sub_bits = sublimb_bit_len(ep.carry_bits); ep.nsub = n_sublimbs(ep.carry_bits);
    const u32 AB = (flags & H2R_STREAM_FIELD_AB) ? 32 : lo.wide_bytes;
    ep.per_col = AB + 4 * lo.wide_bytes + 2 * lo.carry_bytes + 4 * lo.limb_bytes + 4 + lo.carry_bytes + ep.nsub;
    ep.stream_bytes = (u64)(n_l + n_r - 1) * ep.per_col - (lo.carry_bytes + ep.nsub);
    return H2R_OK;
}
}  // namespace

uint64_t h2r_is_equal_muled_stream_bytes_ex(const h2r_ctx *ctx, uint32_t num_limbs_l, uint32_t num_limbs_r, uint32_t flags) try {
    EqPlan ep;
    return (ctx && eq_plan(ctx, num_limbs_l, num_limbs_r, flags, ep) == H2R_OK) ? ep.stream_bytes : 0;
} H2R_CATCH_ZERO

int32_t h2r_is_equal_muled_batch_ex(const h2r_ctx *ctx, const uint64_t *muled_a, const uint64_t *muled_b, uint64_t muled_stride_cols,
                                    uint32_t num_limbs_l, uint32_t num_limbs_r, uint64_t batch, uint32_t flags, void *stream_out,
                                    uint64_t out_stride, uint8_t *eq_out, h2r_stream_t stream) try {
    if (!ctx || !muled_a || !muled_b || !stream_out) return H2R_E_NULL;
    if (ctx->params.device < 0) return H2R_E_UNSUPPORTED;
    EqPlan ep;
    const int32_t rc = eq_plan(ctx, num_limbs_l, num_limbs_r, flags, ep);
    if (rc) return rc;
    if (muled_stride_cols < num_limbs_l + num_limbs_r - 1 || out_stride < ep.stream_bytes || (out_stride & 15)) return H2R_E_SHAPE;
    if (batch == 0) return H2R_OK;
    const h2r_layout &lo = ctx->layout;
    EqMuledArgs ea;
    std::memset(&ea, 0, sizeof ea);
    ea.ma = muled_a; ea.mb = muled_b; ea.muled_stride = muled_stride_cols; ea.batch = batch; ea.C = num_limbs_l + num_limbs_r - 1; ea.w = lo.limb_width;
    ea.wm[0] = ep.wm.v[0]; ea.wm[1] = ep.wm.v[1]; ea.wm[2] = ep.wm.v[2];
    ea.carry_bits = ep.carry_bits; ea.carry_sub_bits = ep.sub_bits; ea.carry_nsub = ep.nsub;
    ea.LB = lo.limb_bytes; ea.WB = lo.wide_bytes; ea.CB = lo.carry_bytes; ea.AB = (flags & H2R_STREAM_FIELD_AB) ? 32 : lo.wide_bytes;
    for (int k = 0; k < 4; ++k) ea.p[k] = ctx->field_p[k];
    ea.per_col = ep.per_col; ea.stream_bytes = (u32)ep.stream_bytes; ea.trace = static_cast<u8 *>(stream_out); ea.elem_stride = out_stride; ea.eq_out = eq_out;
    const unsigned stage = (unsigned)round_up((u64)ea.C * ep.per_col, 16) + 32;
    if (stage > 60 * 1024) return H2R_E_UNSUPPORTED;
    H2R_ON_DEVICE(ctx->params.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stage > 36 * 1024) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&is_equal_muled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage); (void)hipGetLastError(); }
    ProfScope ps(H2R_KERNEL_AUX, st);
    hipLaunchKernelGGL(is_equal_muled_kernel, dim3((unsigned)batch), dim3(256), stage, st, ea);
    HIP_TRY(hipGetLastError());
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_pow_trace_flatten(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *elem_host, void *stream_out) try {
    return h2r_pow_trace_flatten_ex(ctx, pl, elem_host, 0, stream_out);
} H2R_CATCH_STATUS

int32_t h2r_pow_trace_flatten_ex(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *elem_host, uint32_t flags, void *stream_out) try {
    if (!ctx || !pl || !elem_host || !stream_out) return H2R_E_NULL;
    if (pl->off_records == UINT64_MAX) return H2R_E_SHAPE;   // a witness-only layout (h2r_pow_layout_compact) holds no records
    if (flags & ~H2R_STREAM_FIELD_AB) return H2R_E_UNSUPPORTED;
    const h2r_layout &lo = ctx->layout;
    const u64 rsb = h2r_stream_bytes(ctx, flags);
    const u8 *e = static_cast<const u8 *>(elem_host);
    u8 *o = static_cast<u8 *>(stream_out);
    const u32 limbs_bytes = lo.num_limbs * lo.limb_bytes;
    const bool var = pl->off_e_bits != UINT64_MAX;
    if (var) {
        std::memcpy(o, e + pl->off_e_bits, pl->num_exp_bits); o += pl->num_exp_bits;
        for (u32 b = 0; b < pl->num_exp_bits; ++b) {
            int32_t rc = h2r_trace_flatten_ex(ctx, e + pl->off_records + (u64)(2 * b) * lo.record_stride, flags, o);
            if (rc) return rc;
            o += rsb;
            std::memcpy(o, e + pl->off_selected + (u64)b * pl->selected_stride, limbs_bytes); o += limbs_bytes;
            rc = h2r_trace_flatten_ex(ctx, e + pl->off_records + (u64)(2 * b + 1) * lo.record_stride, flags, o);
            if (rc) return rc;
            o += rsb;
        }
    } else {
        for (u32 t = 0; t < pl->num_mul_mods; ++t) {
            int32_t rc = h2r_trace_flatten_ex(ctx, e + pl->off_records + (u64)t * lo.record_stride, flags, o);
            if (rc) return rc;
            o += rsb;
        }
    }
    std::memcpy(o, e + pl->off_result, limbs_bytes); o += limbs_bytes;
    if ((u64)(o - static_cast<u8 *>(stream_out)) != h2r_pow_stream_bytes(ctx, pl, flags)) return H2R_E_SHAPE;
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_profile_enable(uint32_t capacity) try {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    ++g_prof_gen;
    g_prof_cap = capacity;
    if (capacity) g_prof.reserve(capacity);
    return H2R_OK;
} H2R_CATCH_STATUS

int32_t h2r_profile_read(uint32_t kernel, float *ms_out, uint32_t max_count, uint32_t *count) try {
    if (!count) return H2R_E_NULL;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    u32 n = 0;
    for (auto &r : g_prof) {
        if (r.kernel != kernel) continue;
        if (ms_out && n < max_count) {
            HIP_TRY(hipEventSynchronize(r.b));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
            ms_out[n] = ms;
        }
        ++n;
    }
    *count = n;
    return H2R_OK;
} H2R_CATCH_STATUS

const char *h2r_status_str(int32_t s) try {
    switch (s) {
        case H2R_OK: return "ok";
        case H2R_E_SHAPE: return "shape";
        case H2R_E_ZERO_MODULUS: return "zero modulus";
        case H2R_E_NOT_REDUCED: return "quotient does not fit num_limbs limbs";
        case H2R_E_FIELD_TOO_SMALL: return "field too small for the un-carried column bound";
        case H2R_E_HIP: return "HIP runtime error";
        case H2R_E_UNSUPPORTED: return "unsupported shape";
        case H2R_E_NULL: return "null pointer";
        case H2R_E_NOT_IN_FIELD: return "x >= n";
        case H2R_E_ASSERTION: return "an assert_* constraint does not hold";
        case H2R_E_NOMEM: return "a host allocation failed inside the library";
        case H2R_E_INTERNAL: return "an unexpected C++ exception was stopped at the boundary";
        default: return "unknown";
    }
} H2R_CATCH_STR
const char *h2r_last_hip_error(void) try { return g_hip_err; } H2R_CATCH_STR

}  // extern "C"


