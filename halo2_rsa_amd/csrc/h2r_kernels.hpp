// gfx950 (MI355X, CDNA4, wave64) kernels of libh2r.  Hand-written HIP; no CUDA paths.
//
//   chain_kernel<K,NW> K2/K3/K5: one workgroup of NW waves per element.  Runs the element's dependent
//                     chain of modular multiplications on 32-bit digits (K = bits/32), producing for
//                     every mul_mod t the operands and the TRUE quotient/remainder (q_t, r_t) of
//                     BigIntChip::mul_mod (reference big_integer/chip.rs:562-584).  Wave 0 holds the
//                     numbers one digit per lane and owns the serial logic (Barrett with a
//                     per-modulus reciprocal computed in-kernel by wave-parallel Knuth D; carries
//                     resolved with wavefront ballots); all NW waves share the partial products.
//   trace_kernel<W,L> K1: 2L threads per mul_mod, 256-thread workgroups.  Emits the whole witness
//                     record of one mul_mod (mul :386-419 twice, eq_b :614-623, is_equal_muled
//                     :822-895 incl. div_mod_main_gate :1323-1349 and the range-check sub-limbs)
//                     as planes whose entries are written by consecutive lanes (coalesced).
//                     Bound: HBM writes (64,338 algorithmic bytes per RSA-2048 mul_mod).
//   hist_kernel       K4: lookup-table multiplicities of the range-check sub-limbs (LDS atomics).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "h2r.h"
#include "h2r_field.hpp"

namespace h2r {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;
using u128 = unsigned __int128;

// ------------------------------------------------------------------------------------------------
// wavefront carry resolution
//   Position i (lane i of a 64-lane group) generates a carry (g) or propagates an incoming one (p);
//   g and p are mutually exclusive.  carry-in(i) = ((G << 1 | cin) + P) ^ P, one 64-bit scalar add.
// ------------------------------------------------------------------------------------------------
struct CarryGroup {
    u64 cin_mask;  // bit i = carry into position i
    bool cout;     // carry out of position width-1
};
__device__ __forceinline__ CarryGroup carry_group(u64 G, u64 P, bool cin, int width) {
    u64 g1 = (G << 1) | (u64)cin;
    u64 S = g1 + P;
    CarryGroup r;
    r.cin_mask = S ^ P;
    if (width >= 64) r.cout = ((G >> 63) != 0) || (S < g1);
    else r.cout = ((r.cin_mask >> width) & 1) != 0;
    return r;
}

// ================================================================================================
// K2: chain kernel
// ================================================================================================
enum { CHAIN_MULMOD = 0, CHAIN_POW_FIXED = 1, CHAIN_POW_VAR = 2 };

struct ExpBits {
    u32 nbits;
    union {           // e.to_bytes_le(), up to 4096 bits; the kernel reads whole 32-bit words (little-endian host)
        u8 bytes[512];
        u32 words[128];
    };
};

struct ChainArgs {
    const u32 *a;        // MULMOD: a operands; POW: x       [elem][kreal]
    const u32 *b;        // MULMOD: b operands
    const u32 *n;        // moduli [elem][K] (stride 0 when shared)
    const u32 *e_limbs;  // POW_VAR: [elem][e_num_limbs * (limb_width/32)] digits
    u64 n_stride;        // digits between consecutive moduli (0 = shared)
    u64 batch;
    u32 kreal;           // digits of the integers in memory (<= the kernel's K: a shape between two compiled sizes runs as the
                         // next larger one with zero high digits -- Barrett's normalisation shift absorbs them)
    u32 mode, T;         // T = mul_mods per element
    u32 e_num_limbs, exp_limb_bits, digits_per_limb;
    u32 check_in_field;  // modpow_public_key: status NOT_IN_FIELD when x >= n (src/chip.rs:106)
    u32 *ops;            // [elem*T + t][4][kreal]: a, b, q, r of every mul_mod (one contiguous 16*kreal-byte run per item)
    u32 *out;            // nullable: result [elem][K]
    u8 *status;          // [elem]
    // POW_VAR extras written straight into the element traces
    u8 *trace; u64 elem_stride, off_e_bits, off_selected, selected_stride, off_result;
    u32 write_result_to_trace;
    u32 prio;            // s_setprio level of the chain's waves (pipeline mode: they share CUs with record kernels)
    const u32 *pre;      // nullable: the shared modulus' precomputed Barrett constants (recip_kernel)
    u32 *n_copy;         // nullable: [elem][kreal] -- the element's modulus, kept next to the operands for the record writer
                         // (which may run after the call has returned and the caller has refilled its n buffer)
    u64 *dbg_time;       // debug: s_memtime stamps of block 0 / wave 0 (nullable)
    // A long exponent walked as SEGMENTS of its bits (nullable `state`: the whole exponent in one launch): this launch covers bits
    // [bit_lo, bit_hi) of every element, its first mul_mod is item t_base of the element, and the running (squared, acc) pair
    // crosses launches in state[elem][2][kreal].  The record kernel of a segment then runs next to the chains of the next one.
    u32 *state; u32 bit_lo, bit_hi, t_base;
    ExpBits e;
};

// Block geometry of the chain kernel: NW waves (64*NW threads) per element.
//   The 2K product columns are split into CG = ceil(2K/64) groups of 64 columns (one column per
//   lane) and the K inner-product steps into SS = NW/CG slices, so wave w accumulates the slice
//   ss = w / CG of column group cg = w % CG.  The B operand is zero-padded on both sides so the
//   inner loop has no bounds logic:  col[c] = sum_j A[j] * Bpad[K + c - j].
template <int K, int NW>
struct Geo {
    static constexpr int V = (K + 63) / 64;       // digits per lane (every wave holds whole numbers)
    static constexpr int GW = K < 64 ? K : 64;    // positions per ballot group
    static constexpr int CG = (2 * K + 63) / 64;  // column groups
    static constexpr int SS = NW / CG;            // step slices
    static constexpr int SL = K / SS;             // steps per slice
    static constexpr int CGH = (K + 63) / 64;     // column groups of the half products (a window of 64 * CGH >= K columns)
    static constexpr int SSMAX = (K >= 64) ? NW / CGH : SS;        // slices of the half products
    static constexpr int LK = (K - 1) % 64;       // lane of digit K-1 in its group (63 unless the last group is partial)
    // K >= 64 that is not a multiple of 64 (K = 96: RSA-3072): the last group of a number is partial -- its idle lanes
    // PROPAGATE in every carry/borrow chain so that the carry out of digit K-1 reaches the group's carry-out
    static constexpr bool PASS = K >= 64 && K % 64 != 0;
    static_assert(NW % CG == 0 && SS >= 1 && K % SS == 0 && (K < 64 || NW % CGH == 0), "bad chain geometry");
};

template <int K, int NW>
struct ChainLds {
    alignas(16) u32 opa[K];      // A operand of the running multiplication
    u32 bpad[3 * K];             // B operand, data at [K, 2K), zeros elsewhere
    u32 nnpad[3 * K];            // normalised modulus n' = n << s, padded like bpad
    u32 mupad[3 * K];            // mu' = floor((2^(64K) - 1) / n') - 2^(32K), padded
    u32 part[Geo<K, NW>::SSMAX][3][2 * K];  // per-slice column partial sums (3 words)
    u32 x0[2 * K + 4];                   // shift scratch; x0[K+3] = low word of column K for the low half product
    alignas(16) u32 stage[4 * K];        // a, b, q, r of one mul_mod on their way to the ops buffer
    u64 *dbg; u32 dbg_n;                 // debug timing (nullable)
};

// intra-wave ordering of LDS traffic (ds ops of one wave execute in order; stop compiler motion)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Sum of a 32-bit value over the 64 lanes (all active), result valid in lane 63: an inclusive DPP scan
// (row_shr 1/2/4/8, row_bcast 15/31) -- six VALU instructions instead of six LDS-routed butterfly shuffles.
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// K x K digit product A (K digits at `A`) x B (padded at `Bpad`) computed by the whole block.  All
// waves accumulate column partial sums; WAVE 0 alone receives the normalised digits (lane holds
// columns v = lane + 64m in plo[m] and v + K in phi[m]) -- it owns the serial carry/decision logic
// while the other waves only help with the products.  Two block barriers (operands published / partial
// sums published); wave 0 publishes A/Bpad before the call.
//   MUL_FULL  all 2K digits.
//   MUL_HIGH  only phi, computed from the columns >= K-2 (never larger than the true high half and at
//             most 1 ulp smaller): Barrett's  floor(x1 * mu' / 2^(32K))  needs no more.
//   MUL_LOW   only plo and digit K (returned in dk): Barrett's  R = x' - q^ * n'  is < 2^(32(K+1)).
// The half forms run K columns instead of 2K (half the multiply-accumulates); they fall back to the
// full product when K < 64 (a single 64-column group already covers everything).
enum { MUL_FULL = 0, MUL_HIGH = 1, MUL_LOW = 2 };

// Developer build only (-DH2R_CHAIN_TIMING, tools/chain_timing.py): s_memtime stamps of block 0 / thread 0.
#ifdef H2R_CHAIN_TIMING
#define H2R_STAMP(s_) do { if ((s_).dbg && threadIdx.x == 0 && (s_).dbg_n < 4000) (s_).dbg[(s_).dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define H2R_STAMP(s_) do { } while (0)
#endif
#ifndef H2R_CHAIN_UNROLL
#define H2R_CHAIN_UNROLL 2     // iterations (of four products) of the throughput build's product loop unrolled together
#endif
#define H2R_PRAGMA_(x) _Pragma(#x)
#define H2R_PRAGMA(x) H2R_PRAGMA_(x)
#define H2R_CHAIN_UNROLL_PRAGMA H2R_PRAGMA(unroll H2R_CHAIN_UNROLL)
#ifndef H2R_CHAIN_MINB
#define H2R_CHAIN_MINB 6   // blocks per CU the register budget is sized for (K <= 64): 79 VGPRs, no scratch (8 => 64 VGPRs + spills whose reloads wait on vmcnt(0))
#endif

template <int K, int NW, int MODE, bool DEEP>
__device__ __forceinline__ void block_mul(const u32 *A, const u32 *Bpad, ChainLds<K, NW> &s, int lane, int wave,
                                          u32 (&plo)[Geo<K, NW>::V], u32 (&phi)[Geo<K, NW>::V], u32 &dk) {
    using G = Geo<K, NW>;
    constexpr int V = G::V;
    constexpr bool HALF = (MODE != MUL_FULL) && (K >= 64);
    constexpr int CGA = HALF ? G::CGH : G::CG;           // active 64-column groups
    constexpr int SSA = NW / CGA;                         // step slices
#ifdef H2R_ABL_HALF_FULL_PRODUCT
    // Developer ablation (WRONG results, timing only): the FULL product's loop at half its length -- the ceiling of what a squaring computed
    // from half its limb products (a_j a_{c-j} for j < c - j, doubled, plus the diagonal) could save (profiles/r05_chain_accounting.txt).
    constexpr int SLA = (MODE == MUL_FULL) ? K / SSA / 2 : K / SSA;
#else
    constexpr int SLA = K / SSA;                          // steps per slice
#endif
    constexpr int CB = (HALF && MODE == MUL_HIGH) ? K - 2 : 0;   // first column of the window
    static_assert(NW % CGA == 0 && K % SSA == 0 && SSA <= Geo<K, NW>::SSMAX, "bad half-product geometry");
    H2R_STAMP(s);
    __syncthreads();  // operands published; previous readers of part/x* are done
    H2R_STAMP(s);
    {
        const int cg = wave % CGA, ss = wave / CGA;
        const int c = CB + 64 * cg + lane;
        const int j0 = ss * SLA;
        u64 acc = 0;
        u32 ov = 0;
        // a window wider than the columns that exist (partial last group): the lanes beyond column 2K-1 walk the zero
        // padding of column 2K-1 and store nothing
        const u32 *bp = Bpad + K + (G::PASS && c > 2 * K - 1 ? 2 * K - 1 : c) - j0;
        const u32 *ap = A + j0;
        if constexpr (DEEP && SLA % 4 == 0) {
            // Latency build: one wave per SIMD, so nothing hides the LDS round trip but this wave's own instructions.
            // Operands are preloaded CH products ahead (double buffered) and four independent accumulators advance by
            // 4 x v_mad_u64_u32 followed by their 4 x v_addc (which also covers the VALU-writes-SGPR hazard).
            u64 acc4[4] = {0, 0, 0, 0};
            u32 ov4[4] = {0, 0, 0, 0};
            auto mac4 = [&](const uint4 &av, u32 b0, u32 b1, u32 b2, u32 b3) {
                u64 c0, c1, c2, c3;
                asm volatile(
                    "v_mad_u64_u32 %0, %8, %12, %16, %0\n\t"
                    "v_mad_u64_u32 %1, %9, %13, %17, %1\n\t"
                    "v_mad_u64_u32 %2, %10, %14, %18, %2\n\t"
                    "v_mad_u64_u32 %3, %11, %15, %19, %3\n\t"
                    "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"
                    "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"
                    "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"
                    "v_addc_co_u32_e64 %7, %11, 0, %7, %11"
                    : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]), "+v"(ov4[0]), "+v"(ov4[1]), "+v"(ov4[2]), "+v"(ov4[3]),
                      "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                    : "v"(av.x), "v"(av.y), "v"(av.z), "v"(av.w), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            };
            constexpr int CH = SLA < 16 ? SLA : 16;   // products per preloaded chunk
            uint4 ac[CH / 4]; u32 bc[CH];
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) ac[q] = *reinterpret_cast<const uint4 *>(ap + 4 * q);   // broadcast 16-byte reads
#pragma unroll
            for (int q = 0; q < CH; ++q) bc[q] = bp[-q];
#pragma unroll
            for (int ch = 0; ch < SLA / CH; ++ch) {
                uint4 an[CH / 4]; u32 bn[CH];
                if (ch + 1 < SLA / CH) {   // prefetch the next chunk before consuming this one
#pragma unroll
                    for (int q = 0; q < CH / 4; ++q) an[q] = *reinterpret_cast<const uint4 *>(ap + (ch + 1) * CH + 4 * q);
#pragma unroll
                    for (int q = 0; q < CH; ++q) bn[q] = bp[-(ch + 1) * CH - q];
                }
#pragma unroll
                for (int q = 0; q < CH / 4; ++q) mac4(ac[q], bc[4 * q], bc[4 * q + 1], bc[4 * q + 2], bc[4 * q + 3]);
                if (ch + 1 < SLA / CH) {
#pragma unroll
                    for (int q = 0; q < CH / 4; ++q) ac[q] = an[q];
#pragma unroll
                    for (int q = 0; q < CH; ++q) bc[q] = bn[q];
                }
            }
            acc = acc4[0];
            ov = ov4[0] + ov4[1] + ov4[2] + ov4[3];
#pragma unroll
            for (int k = 1; k < 4; ++k) { acc += acc4[k]; ov += (acc < acc4[k]) ? 1u : 0u; }
        } else {
            // acc(64) += a*b with the multiplier's own carry-out feeding the overflow word: 2 VALU per product
            // (v_mad_u64_u32 writes the carry to an SGPR pair; gfx950 needs 2 wait states before a VALU reads it).
            // Tried and measured slower on the same box (tools/ab_chain.sh): four independent accumulators with
            // operands preloaded 16 products ahead (4 x mad then 4 x addc, no s_nop) -- 0.117 vs 0.110 ms alone and
            // 0.339 vs 0.279 ms pipelined, because it needs 128 VGPRs and halves the waves that hide LDS latency.
            auto mac = [&](u32 av, u32 bv) {
                u64 carry;
                asm volatile("v_mad_u64_u32 %0, %2, %3, %4, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
                             : "+v"(acc), "+v"(ov), "=&s"(carry) : "v"(av), "v"(bv));
            };
            if constexpr (SLA % 4 == 0) {
H2R_CHAIN_UNROLL_PRAGMA
                for (int j = 0; j < SLA; j += 4) {
                    const uint4 a4 = *reinterpret_cast<const uint4 *>(ap + j);  // broadcast 16-byte read
                    const u32 b0 = bp[-j], b1 = bp[-j - 1], b2 = bp[-j - 2], b3 = bp[-j - 3];
                    mac(a4.x, b0); mac(a4.y, b1); mac(a4.z, b2); mac(a4.w, b3);
                }
            } else {
#pragma unroll
                for (int j = 0; j < SLA; ++j) mac(ap[j], bp[-j]);
            }
        }
        if (c < 2 * K) { s.part[ss][0][c] = (u32)acc; s.part[ss][1][c] = (u32)(acc >> 32); s.part[ss][2][c] = ov; }
        if constexpr (HALF && MODE == MUL_HIGH && K % 64 == 0) {
            // column 2K-2 lies one past the window: its single product A[K-1]*B[K-1]
            if (lane == 0) {
                const u64 p = (wave == 0) ? (u64)A[K - 1] * Bpad[K + K - 1] : 0;
                if (wave < SSA) { s.part[wave][0][2 * K - 2] = (u32)p; s.part[wave][1][2 * K - 2] = (u32)(p >> 32); s.part[wave][2][2 * K - 2] = 0; }
            }
        }
        if constexpr (HALF && MODE == MUL_LOW) {
            // digit K needs column K only modulo 2^32: sum_j lo32(A[j] * B[K-j]), j = 1..K-1 (wave 0, butterfly add)
            if (wave == 0) {
                u32 t = 0;
#pragma unroll
                for (int m = 0; m < V; ++m) { const int j = lane + 64 * m; if (j >= 1 && j < K) t += A[j] * Bpad[K + K - j]; }
                t = wave_sum_u32(t);   // DPP scan: the total sits in lane 63
                if (lane == 63) s.x0[K + 3] = t;   // x0 of column K (published by the barrier below; the reduce skips c = K)
            }
        }
    }
    H2R_STAMP(s);
    __syncthreads();
    H2R_STAMP(s);
    // Wave 0 alone finishes the product: it sums the slices of its own columns straight from `part`, gets the
    // neighbouring columns' words with wavefront DPP shifts (no second LDS round trip, no third barrier) and resolves
    // the carries by ballot.  The other waves are done; they wait at the next block_mul's first barrier, which is also
    // what keeps `part` intact until wave 0 has read it.
    //   x0|x1|x2 (c) = the 96-bit sum of column c as 32-bit words;  t(c) = x0(c) + x1(c-1) + x2(c-2);
    //   digit d(c) = lo32 t(c) + hi32 t(c-1) + carry.
    if (wave != 0) return;
    constexpr int GWL = K < 64 ? K : 64;        // lanes of a column group
    u32 prev_x1 = 0, prev_x2 = 0, prev_r1 = 0, prev_thi = 0;   // previous group's words, for the seam lanes
    int seam = GWL - 1;   // lane of the previous group's last column
    auto shr1 = [&](u32 cur, u32 prev) -> u32 {   // value of lane-1; lane 0 takes the previous group's last lane
        const u32 up = __builtin_amdgcn_readlane(prev, seam);
        return (u32)__builtin_amdgcn_update_dpp((int)up, (int)cur, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    };
    auto group = [&](int c, bool active) -> u64 {   // returns lo32 t(c) + hi32 t(c-1)
        // unconditional reads at a clamped column (inactive lanes are zeroed afterwards): lets the loads of all
        // groups issue back to back instead of one exec-masked LDS round trip per group
        const int cc = c < 0 ? 0 : (c > 2 * K - 1 ? 2 * K - 1 : c);
        u64 s0 = 0, s1 = 0; u32 s2 = 0;
#pragma unroll
        for (int k = 0; k < SSA; ++k) { s0 += s.part[k][0][cc]; s1 += s.part[k][1][cc]; s2 += s.part[k][2][cc]; }
        if (!active) { s0 = 0; s1 = 0; s2 = 0; }
        const u64 t1 = s1 + (s0 >> 32);
        const u32 x0 = (u32)s0, x1 = (u32)t1, x2 = s2 + (u32)(t1 >> 32);
        const u32 a1 = shr1(x1, prev_x1);       // x1(c-1)
        const u32 r1 = shr1(x2, prev_x2);       // x2(c-1)
        const u32 b2 = shr1(r1, prev_r1);       // x2(c-2)
        const u64 t = (u64)x0 + a1 + b2;
        const u32 thi = (u32)(t >> 32);
        const u32 tph = shr1(thi, prev_thi);    // hi32 t(c-1)
        prev_x1 = x1; prev_x2 = x2; prev_r1 = r1; prev_thi = thi;
        return (u64)(u32)t + tph;
    };
    if constexpr (HALF && MODE == MUL_HIGH) (void)group(K - 64 + lane, lane >= 62);   // columns K-2, K-1 feed digit K
    bool cin = false;
#pragma unroll
    for (int g = 0; g < 2 * V; ++g) {
        if (HALF && MODE == MUL_HIGH && g < V) continue;   // low half not needed
        if (HALF && MODE == MUL_LOW && g >= V) continue;   // high half not needed (digit K below)
        const int vv = lane + 64 * (g % V);
        const int c = vv + (g >= V ? K : 0);
        const bool in_col = vv < K;
        // the group before this one: full (lane 63), or the partial last group of the low half (lane of column K-1); the
        // high half product enters its first group from the two-column pre-step above, which sits in lanes 62, 63
        seam = (g % V == 0 && !(HALF && MODE == MUL_HIGH)) ? G::LK : GWL - 1;
        const u64 d = group(c, in_col && !(HALF && MODE == MUL_HIGH && c >= 2 * K - 1));   // column 2K-1 of the window is zero
        const bool gen = in_col && (d >> 32) != 0;
        const bool prop = in_col ? ((u32)d == 0xffffffffu) : G::PASS;
        const CarryGroup cgp = carry_group(__ballot(gen), __ballot(prop), cin, G::GW);
        cin = cgp.cout;
        const u32 digit = in_col ? (u32)d + (u32)((cgp.cin_mask >> lane) & 1) : 0u;
        if (g < V) plo[g] = digit; else phi[g - V] = digit;
    }
    if constexpr (HALF && MODE == MUL_LOW) {
        // digit K = lo32( x0(K) + x1(K-1) + x2(K-2) + hi32 t(K-1) + carry out of digit K-1 ); x0(K) was left in x0[K+3]
        dk = s.x0[K + 3] + __builtin_amdgcn_readlane(prev_x1, G::LK) + __builtin_amdgcn_readlane(prev_r1, G::LK) +
             __builtin_amdgcn_readlane(prev_thi, G::LK) + (cin ? 1u : 0u);
    } else {
        dk = 0;
    }
    H2R_STAMP(s);
}

// K-digit a + b: returns carry out.
template <int K>
__device__ __forceinline__ bool wave_add(u32 (&r)[(K + 63) / 64], const u32 (&a)[(K + 63) / 64],
                                         const u32 (&b)[(K + 63) / 64], int lane) {
    constexpr int V = (K + 63) / 64;
    constexpr int GW = K < 64 ? K : 64;
    bool cin = false;
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const bool act = lane + 64 * m < K;
        const u64 d = act ? (u64)a[m] + b[m] : 0;
        const CarryGroup cg = carry_group(__ballot((d >> 32) != 0), __ballot(act ? (u32)d == 0xffffffffu : (K >= 64)), cin, GW);
        cin = cg.cout;
        r[m] = act ? (u32)d + (u32)((cg.cin_mask >> lane) & 1) : 0u;
    }
    return cin;
}
// K-digit a - b: returns borrow out.
template <int K>
__device__ __forceinline__ bool wave_sub(u32 (&r)[(K + 63) / 64], const u32 (&a)[(K + 63) / 64],
                                         const u32 (&b)[(K + 63) / 64], int lane, bool bin = false) {
    constexpr int V = (K + 63) / 64;
    constexpr int GW = K < 64 ? K : 64;
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const bool act = lane + 64 * m < K;
        const CarryGroup cg = carry_group(__ballot(act && a[m] < b[m]), __ballot(act ? a[m] == b[m] : (K >= 64)), bin, GW);
        bin = cg.cout;
        r[m] = act ? a[m] - b[m] - (u32)((cg.cin_mask >> lane) & 1) : 0u;
    }
    return bin;
}
// a >= b over K digits.
template <int K>
__device__ __forceinline__ bool wave_ge(const u32 (&a)[(K + 63) / 64], const u32 (&b)[(K + 63) / 64], int lane) {
    constexpr int V = (K + 63) / 64;
#pragma unroll
    for (int m = V - 1; m >= 0; --m) {
        const bool act = lane + 64 * m < K;
        const u64 ne = __ballot(act && a[m] != b[m]);
        if (ne) {
            const u64 gt = __ballot(act && a[m] > b[m]);
            const int top = 63 - __builtin_clzll(ne);
            return ((gt >> top) & 1) != 0;
        }
    }
    return true;
}
// a += 1: returns carry out.
template <int K>
__device__ __forceinline__ bool wave_inc(u32 (&a)[(K + 63) / 64], int lane) {
    constexpr int V = (K + 63) / 64;
    constexpr int GW = K < 64 ? K : 64;
    bool cin = true;
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const bool act = lane + 64 * m < K;
        const CarryGroup cg = carry_group(0, __ballot(act ? a[m] == 0xffffffffu : (K >= 64)), cin, GW);
        cin = cg.cout;
        if (act) a[m] += (u32)((cg.cin_mask >> lane) & 1);
    }
    return cin;
}

template <int K>
__device__ __forceinline__ void lds_store(u32 *dst, const u32 (&r)[(K + 63) / 64], int lane) {
#pragma unroll
    for (int m = 0; m < (K + 63) / 64; ++m) if (lane + 64 * m < K) dst[lane + 64 * m] = r[m];
}
template <int K>
__device__ __forceinline__ void lds_store_n(u32 *dst, const u32 (&r)[(K + 63) / 64], int lane, u32 n) {   // first n digits only
#pragma unroll
    for (int m = 0; m < (K + 63) / 64; ++m) if ((u32)(lane + 64 * m) < n) dst[lane + 64 * m] = r[m];
}
template <int K>
__device__ __forceinline__ void lds_load(u32 (&r)[(K + 63) / 64], const u32 *src, int lane) {
#pragma unroll
    for (int m = 0; m < (K + 63) / 64; ++m) r[m] = (lane + 64 * m < K) ? src[lane + 64 * m] : 0;
}
template <int K>
__device__ __forceinline__ void glb_store(u32 *dst, const u32 (&r)[(K + 63) / 64], int lane, u32 n) {   // first n digits
#pragma unroll
    for (int m = 0; m < (K + 63) / 64; ++m) if ((u32)(lane + 64 * m) < n) dst[lane + 64 * m] = r[m];
}

// mu' = floor(((~n') * 2^(32K) + 2^(32K) - 1) / n') by wave-parallel Knuth algorithm D (n'
// normalised).  Run by wave 0 only (wave-local LDS exchanges, no block barriers); returns the K
// quotient digits distributed one per lane.
template <int K, int NW>
__device__ __forceinline__ void wave_reciprocal(ChainLds<K, NW> &s, const u32 (&nn)[Geo<K, NW>::V], int lane, int wave,
                                                u32 (&mu)[Geo<K, NW>::V]) {
    using G = Geo<K, NW>;
    constexpr int V = G::V;
    u32 rem[V];
    (void)wave; (void)s;
#pragma unroll
    for (int m = 0; m < V; ++m) { rem[m] = ~nn[m]; mu[m] = 0; }
    // digit K-1 / K-2 of a lane-distributed number, and the number shifted up by one digit (digit v-1 in lane v):
    // register traffic only (readlane, DPP wave_shr) -- the LDS round trips of an earlier version cost 40 % of the loop
    constexpr int MT = (K - 1) / 64, LT = (K - 1) % 64, MS = K >= 2 ? (K - 2) / 64 : 0, LS = K >= 2 ? (K - 2) % 64 : 0;
    auto shift_up = [&](u32 (&out)[V], const u32 (&in)[V], u32 fill) {
        u32 up = fill;
#pragma unroll
        for (int m = 0; m < V; ++m) {
            out[m] = (u32)__builtin_amdgcn_update_dpp((int)up, (int)in[m], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            up = __builtin_amdgcn_readlane(in[m], 63);
        }
    };
    const u32 ntop = __builtin_amdgcn_readlane(nn[MT], LT);
    // vrec = floor((2^64 - 1) / ntop) - 2^32: the 2-by-1 division reciprocal of the normalised top digit
    const u32 vrec = ntop ? (u32)(0xffffffffffffffffull / ntop - 0x100000000ull) : 0u;
    for (int j = K - 1; j >= 0; --j) {
        const u32 top = __builtin_amdgcn_readlane(rem[MT], LT);
        const u32 second = (K >= 2) ? __builtin_amdgcn_readlane(rem[MS], LS) : 0xffffffffu;
        u64 qd;  // 2-by-1 estimate of the quotient digit (exact for the two leading digits)
        if (top >= ntop) qd = 0xffffffffull;
        else {
            // floor((top * 2^32 + second) / ntop) with the precomputed reciprocal (Moeller-Granlund, exact for top < ntop)
            const u64 qq = (u64)vrec * top + (((u64)top << 32) | second);
            u32 q1 = (u32)(qq >> 32) + 1u;
            const u32 q0 = (u32)qq;
            u32 rr = second - q1 * ntop;
            if (rr > q0) { q1 -= 1u; rr += ntop; }
            if (rr >= ntop) { q1 += 1u; }
            qd = q1;
        }
        u32 plo[V], phi[V], phi_prev[V], remsh[V], e[V];
#pragma unroll
        for (int m = 0; m < V; ++m) {
            const u64 p = (u64)(u32)qd * nn[m];
            plo[m] = (u32)p; phi[m] = (lane + 64 * m < K) ? (u32)(p >> 32) : 0u;
        }
        shift_up(phi_prev, phi, 0u);
        shift_up(remsh, rem, 0xffffffffu);
        bool cin = false;
#pragma unroll
        for (int m = 0; m < V; ++m) {
            const int v = lane + 64 * m;
            const bool act = v < K;
            if (!act) remsh[m] = 0;
            const u64 d = act ? (u64)plo[m] + phi_prev[m] : 0;
            const CarryGroup cg = carry_group(__ballot((d >> 32) != 0), __ballot(act ? (u32)d == 0xffffffffu : (K >= 64)), cin, G::GW);
            cin = cg.cout;
            e[m] = act ? (u32)d + (u32)((cg.cin_mask >> lane) & 1) : 0u;
        }
        const u32 etop = __builtin_amdgcn_readlane(phi[MT], LT) + (cin ? 1u : 0u);
        u32 diff[V];
        const bool bout = wave_sub<K>(diff, remsh, e, lane);
        i64 dtop = (i64)top - (i64)etop - (bout ? 1 : 0);
        for (int it = 0; it < 3 && dtop < 0; ++it) {  // qd too large (at most twice): add n' back
            u32 t2[V];
            const bool c = wave_add<K>(t2, diff, nn, lane);
#pragma unroll
            for (int m = 0; m < V; ++m) diff[m] = t2[m];
            dtop += c ? 1 : 0;
            qd -= 1;
        }
#pragma unroll
        for (int m = 0; m < V; ++m) {
            rem[m] = diff[m];
            if (lane + 64 * m == j) mu[m] = (u32)qd;
        }
    }
}

// (lo, hi) = (2K-digit number held as digits lo/hi) << sh.  Uses s.x0 as block-shared scratch.
// Returns true when non-zero bits are shifted out.
template <int K, int NW>
__device__ __forceinline__ bool block_shl2k(ChainLds<K, NW> &s, u32 sh, int lane, int wave,
                                            u32 (&lo)[Geo<K, NW>::V], u32 (&hi)[Geo<K, NW>::V]) {
    constexpr int V = Geo<K, NW>::V;
    __syncthreads();
    if (wave == 0) { lds_store<K>(s.x0, lo, lane); lds_store<K>(s.x0 + K, hi, lane); }
    __syncthreads();
    if (wave != 0) return false;
    const int ws = (int)(sh >> 5), bs = (int)(sh & 31);
    bool lost = false;
#pragma unroll
    for (int g = 0; g < 2 * V; ++g) {
        const int vv = lane + 64 * (g % V);
        const int c = vv + (g >= V ? K : 0);
        u32 d = 0;
        if (vv < K) {
            const int i0 = c - ws, i1 = c - ws - 1;
            const u32 a0 = (i0 >= 0) ? s.x0[i0] : 0, a1 = (i1 >= 0) ? s.x0[i1] : 0;
            d = bs ? ((a0 << bs) | (a1 >> (32 - bs))) : a0;
            bool l = false;  // this lane also checks whether source digit c loses bits
            if (c + ws >= 2 * K) l = s.x0[c] != 0;
            else if (c + ws == 2 * K - 1 && bs) l = (s.x0[c] >> (32 - bs)) != 0;
            lost = lost || l;
        }
        if (g < V) lo[g] = d; else hi[g - V] = d;
    }
    return __ballot(lost) != 0;
}
// r = (K-digit number in `a`) >> sh.
template <int K, int NW>
__device__ __forceinline__ void block_shr(ChainLds<K, NW> &s, u32 sh, int lane, int wave,
                                          const u32 (&a)[Geo<K, NW>::V], u32 (&r)[Geo<K, NW>::V]) {
    constexpr int V = Geo<K, NW>::V;
    __syncthreads();
    if (wave == 0) { lds_store<K>(s.x0, a, lane); if (lane == 0) s.x0[K] = 0; }
    __syncthreads();
    if (wave != 0) return;
    const int ws = (int)(sh >> 5), bs = (int)(sh & 31);
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const int v = lane + 64 * m;
        u32 d = 0;
        if (v < K) {
            const int i0 = v + ws, i1 = v + ws + 1;
            const u32 a0 = (i0 <= K) ? s.x0[i0] : 0, a1 = (i1 <= K) ? s.x0[i1] : 0;
            d = bs ? ((a0 >> bs) | (a1 << (32 - bs))) : a0;
        }
        r[m] = d;
    }
}

// One BigIntChip::mul_mod's off-circuit arithmetic (reference big_integer/chip.rs:562-584):
// (q, r) = divmod(a * b, n) by Barrett reduction with the per-modulus mu'.  a, b, nn, q, r and the
// returned status are meaningful in WAVE 0 only; every wave must call it (block barriers inside, and
// the control flow never depends on the data).
// Workgroup barriers block_mulmod executes, as a function of the (block-uniform) normalisation shift: every block_mul has two
// (operands published / partial sums published), block_shl2k and block_shr two each.  chain_element_dual's idle group meets exactly
// these barriers on a zero bit -- ONE definition, next to the code it counts: change it together with any barrier in block_mul,
// block_shl2k, block_shr or block_mulmod.
constexpr int BLOCK_MUL_BARRIERS = 2, BLOCK_SHIFT_BARRIERS = 2;
__device__ __forceinline__ constexpr int block_mulmod_barriers(bool shifted) { return 3 * BLOCK_MUL_BARRIERS + (shifted ? 2 * BLOCK_SHIFT_BARRIERS : 0); }

template <int K, int NW, bool DEEP>
__device__ __forceinline__ int block_mulmod(ChainLds<K, NW> &s, u32 shift, int lane, int wave,
                                            const u32 (&a)[Geo<K, NW>::V], const u32 (&b)[Geo<K, NW>::V],
                                            const u32 (&nn)[Geo<K, NW>::V],
                                            u32 (&q)[Geo<K, NW>::V], u32 (&r)[Geo<K, NW>::V]) {
    constexpr int V = Geo<K, NW>::V;
    const bool w0 = wave == 0;
    int status = H2R_OK;
    u32 xlo[V], xhi[V];
    // (no barrier needed before re-publishing opa/bpad: every wave has passed the two barriers that follow the
    //  product loop of the previous block_mul, so nobody still reads them; block_mul's first barrier publishes)
    if (w0) { lds_store<K>(s.opa, a, lane); lds_store<K>(s.bpad + K, b, lane); }
    u32 dk;
    block_mul<K, NW, MUL_FULL, DEEP>(s.opa, s.bpad, s, lane, wave, xlo, xhi, dk);
    if (shift) {  // x' = x << s  (n' = n << s); block-uniform branch
        if (block_shl2k<K, NW>(s, shift, lane, wave, xlo, xhi)) status = H2R_E_NOT_REDUCED;
    }
    // q^ = x1 + floor(x1 * mu' / 2^(32K)),  x1 = floor(x' / 2^(32K))
    if (w0) lds_store<K>(s.opa, xhi, lane);
    u32 ylo[V], yhi[V];
    block_mul<K, NW, MUL_HIGH, DEEP>(s.opa, s.mupad, s, lane, wave, ylo, yhi, dk);
    if (w0 && wave_add<K>(q, xhi, yhi, lane)) status = H2R_E_NOT_REDUCED;
    // R = x' - q^ * n'   (0 <= R < 7 n': q^ may be up to 6 short of the true quotient)
    if (w0) lds_store<K>(s.opa, q, lane);
    u32 zlo[V], zhi[V];
    block_mul<K, NW, MUL_LOW, DEEP>(s.opa, s.nnpad, s, lane, wave, zlo, zhi, dk);
    if (K < 64) dk = __builtin_amdgcn_readfirstlane(zhi[0]);   // full product was computed: digit K is its first high digit
    u32 rl[V];
#pragma unroll
    for (int m = 0; m < V; ++m) rl[m] = 0;
    if (w0) {
        const bool b0 = wave_sub<K>(rl, xlo, zlo, lane);
        u32 rtop = __builtin_amdgcn_readfirstlane(xhi[0]) - dk - (b0 ? 1u : 0u);
        for (int it = 0; it < 8; ++it) {
            if (rtop == 0 && !wave_ge<K>(rl, nn, lane)) break;
            u32 t[V];
            const bool bo = wave_sub<K>(t, rl, nn, lane);
#pragma unroll
            for (int m = 0; m < V; ++m) rl[m] = t[m];
            rtop -= bo ? 1u : 0u;
            if (wave_inc<K>(q, lane)) status = H2R_E_NOT_REDUCED;
        }
    }
    if (shift) block_shr<K, NW>(s, shift, lane, wave, rl, r);  // r = R >> s
    else {
#pragma unroll
        for (int m = 0; m < V; ++m) r[m] = rl[m];
    }
    return status;
}

// Per-modulus constants of the Barrett reduction: the normalisation shift, n' = n << shift and mu' (see block_mulmod).
// Computed by every chain block for its own modulus, or ONCE per call by recip_kernel when the batch shares one modulus
// (H2R_F_SHARED_MODULUS: one key, many signatures -- the Knuth-D reciprocal is 17 of the 58 us of a single RSA-2048
// chain).  Layout of the precomputed form in memory: u32 [0] = shift, [1] = status, [4, 4+K) = n', [4+K, 4+2K) = mu'.
constexpr u32 CHAIN_PRE_HDR = 4;
__host__ __device__ constexpr u32 chain_pre_words(u32 K) { return CHAIN_PRE_HDR + 2 * K; }

// Fills s.nnpad / s.mupad and returns (status, shift, nn in wave 0's registers) for the modulus `nraw`.  Every wave
// calls it (block barriers inside); the control flow depends only on block-uniform values.
template <int K, int NW>
__device__ __forceinline__ int chain_modulus_setup(ChainLds<K, NW> &s, const u32 (&nraw)[Geo<K, NW>::V], int lane, int wave, u32 &shift,
                                                   u32 (&nn)[Geo<K, NW>::V]) {
    constexpr int V = Geo<K, NW>::V;
    // normalisation shift: leading zero bits of n within 32K bits (every wave computes it: block-uniform)
    int top_digit = -1;
#pragma unroll
    for (int m = V - 1; m >= 0; --m) {
        const u64 nz = __ballot(nraw[m] != 0);
        if (nz && top_digit < 0) top_digit = 64 * m + (63 - __builtin_clzll(nz));
    }
    shift = 0;
#pragma unroll
    for (int m = 0; m < V; ++m) nn[m] = nraw[m];
    if (top_digit < 0) return H2R_E_ZERO_MODULUS;  // reference divides by zero, chip.rs:566
    const u32 topv = __shfl(top_digit >= 64 ? nraw[V - 1] : nraw[0], top_digit & 63);
    shift = 32u * (u32)(K - 1 - top_digit) + (u32)__builtin_clz(topv);
    if (shift) {
        u32 zero[V];
#pragma unroll
        for (int m = 0; m < V; ++m) zero[m] = 0;
        block_shl2k<K, NW>(s, shift, lane, wave, nn, zero);
    }
    __syncthreads();
    if (wave == 0) {
        lds_store<K>(s.nnpad + K, nn, lane);
        wave_sync();
        u32 mu[V];
        wave_reciprocal<K, NW>(s, nn, lane, wave, mu);
        lds_store<K>(s.mupad + K, mu, lane);
    }
    __syncthreads();
    return H2R_OK;
}

}  // namespace h2r
#include "h2r_chain_wave.hpp"   // one wavefront per element (K <= 32): wave_mulmod, wave_modulus_setup
namespace h2r {

// One element's chain.  Returns through `status_out` semantics of the reference's panics (see h2r.h).
// SEG = false (the step launches' chain role, whose register budget is tight): the code for segments of a long exponent is compiled out.
// WAVE (K <= 32, NW = 1): the element belongs to ONE wavefront of a workgroup of independent waves -- `s` is that wave's own LDS, the
// mul_mods run register-resident (h2r_chain_wave.hpp) and nothing below synchronises beyond the wave (a wave may leave early).
template <int K, int NW, bool DEEP, bool SEG = true, bool WAVE = false>
__device__ __forceinline__ void chain_element(const ChainArgs &args, ChainLds<K, NW> &s, const u64 elem) {
    using G = Geo<K, NW>;
    constexpr int V = G::V;
    static_assert(!WAVE || (NW == 1 && K <= 32), "the one-wave chain holds 2K <= 64 product columns");
    const int lane = threadIdx.x & 63, wave = WAVE ? 0 : (int)(threadIdx.x >> 6);
    const int tid = WAVE ? lane : (int)threadIdx.x;   // this element's thread index among its NT threads
    constexpr int NT = WAVE ? 64 : 64 * NW;
    const bool w0 = wave == 0;
    const u32 *n_g = args.n + elem * args.n_stride;
    const u32 KR = args.kreal;   // digits in memory; digits [KR, K) are zero
    u32 nraw[V];
#pragma unroll
    for (int m = 0; m < V; ++m) nraw[m] = ((u32)(lane + 64 * m) < KR) ? n_g[lane + 64 * m] : 0;
    if constexpr (!WAVE) { for (int i = tid; i < 3 * K; i += NT) { s.bpad[i] = 0; s.nnpad[i] = 0; s.mupad[i] = 0; } }
    if (tid == 0) { s.dbg = (blockIdx.x == 0 && threadIdx.x == 0) ? args.dbg_time : nullptr; s.dbg_n = 0; }
    if (w0 && args.n_copy) glb_store<K>(args.n_copy + elem * KR, nraw, lane, KR);
    int status = H2R_OK;
    {   // n = 0: the reference divides by zero (chip.rs:566); block-uniform
        bool nz = false;
#pragma unroll
        for (int m = 0; m < V; ++m) nz = nz || __ballot(nraw[m] != 0) != 0;
        if (!nz) status = H2R_E_ZERO_MODULUS;
    }
    // operands of the first step (all waves load x so that the in-field predicate is block-uniform)
    u32 cur[V], acc[V], bop[V];
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const int v = lane + 64 * m;
        cur[m] = (u32)v < KR ? args.a[elem * KR + v] : 0;
        bop[m] = (args.mode == CHAIN_MULMOD && (u32)v < KR) ? args.b[elem * KR + v] : 0;
        acc[m] = (v == 0) ? 1u : 0u;  // acc = const 1 padded to num_limbs (:729 / :682)
    }
    const bool seg = SEG && args.state != nullptr;
    const bool resumed = seg && args.bit_lo > 0;   // a later segment of a long exponent: block-uniform
    if (resumed) {
        if (args.status[elem] != 0) return;   // the element failed in an earlier segment: its status stands
#pragma unroll
        for (int m = 0; m < V; ++m) {
            const int v = lane + 64 * m;
            cur[m] = (u32)v < KR ? args.state[(elem * 2 + 0) * KR + v] : 0;
            acc[m] = (u32)v < KR ? args.state[(elem * 2 + 1) * KR + v] : 0;
        }
    }
    if (!resumed && status == H2R_OK && args.mode != CHAIN_MULMOD && args.check_in_field && wave_ge<K>(cur, nraw, lane))
        status = H2R_E_NOT_IN_FIELD;  // src/chip.rs:106
    if (!resumed && args.mode == CHAIN_POW_VAR && args.exp_limb_bits < 32 * args.digits_per_limb) {
        // main_gate.to_bits(limb, exp_limb_bits) (chip.rs:677) cannot be satisfied by a limb with bits at or above
        // exp_limb_bits: the reference's circuit fails, so the element gets a status instead of a plausible trace
        bool wide = false;
        for (u32 l = tid; l < args.e_num_limbs; l += NT) {
            const u32 *ed = args.e_limbs + (elem * args.e_num_limbs + l) * args.digits_per_limb;
            const u64 v = args.digits_per_limb == 2 ? (((u64)ed[1] << 32) | ed[0]) : (u64)ed[0];
            wide = wide || (v >> args.exp_limb_bits) != 0;
        }
        bool any_wide;
        if constexpr (WAVE) any_wide = __ballot(wide) != 0; else any_wide = __syncthreads_or(wide ? 1 : 0) != 0;
        if (any_wide && status == H2R_OK) status = H2R_E_SHAPE;
    }
    if (status != H2R_OK) {  // block-uniform early exit
        if (tid == 0) args.status[elem] = (u8)status;
        return;
    }
    u32 shift;
    u32 nn[V];
    u32 mu_w = 0;     // WAVE: mu' in registers (lanes 0..K-1)
    if (args.pre) {   // shared modulus: constants computed once by recip_kernel
        shift = args.pre[0];
#pragma unroll
        for (int m = 0; m < V; ++m) nn[m] = (lane + 64 * m < K) ? args.pre[CHAIN_PRE_HDR + lane + 64 * m] : 0;
        if constexpr (WAVE) mu_w = lane < K ? args.pre[CHAIN_PRE_HDR + K + lane] : 0u;
        else {
            __syncthreads();   // the zero fill above is complete
            for (int i = tid; i < K; i += NT) { s.nnpad[K + i] = args.pre[CHAIN_PRE_HDR + i]; s.mupad[K + i] = args.pre[CHAIN_PRE_HDR + K + i]; }
            __syncthreads();
        }
    } else {
        if constexpr (WAVE) {
            const int rs = wave_modulus_setup<K>(s, nraw[0], lane, shift, nn[0], mu_w);   // (n != 0 was established above: only the developer check can fail)
            if (rs != H2R_OK) { if (tid == 0) args.status[elem] = (u8)rs; return; }
        }
        else (void)chain_modulus_setup<K, NW>(s, nraw, lane, wave, shift, nn);   // n != 0 was established above
    }
    u32 q[V], r[V];
    // one mul_mod, by the workgroup (four-wave form) or by this wave alone
    auto mulmod = [&](const u32 (&oa)[V], const u32 (&ob)[V]) -> int {
        if constexpr (WAVE) return wave_mulmod<K>(shift, lane, oa[0], ob[0], nn[0], mu_w, q[0], r[0]);
        else return block_mulmod<K, NW, DEEP>(s, shift, lane, wave, oa, ob, nn, q, r);
    };
    const u64 item0 = elem * args.T;
    auto emit = [&](u32 t, const u32 (&oa)[V], const u32 (&ob)[V]) {
        if (w0 && status == H2R_OK) {
            const u64 it = item0 + t;
            // staged through LDS so that the four K-digit values leave as 16-byte stores (one instruction for K = 64):
            // a store instruction of this wave queues behind the co-running record kernel's stores, so fewer is faster
            wave_sync();
            lds_store_n<K>(s.stage, oa, lane, KR); lds_store_n<K>(s.stage + KR, ob, lane, KR);
            lds_store_n<K>(s.stage + 2 * KR, q, lane, KR); lds_store_n<K>(s.stage + 3 * KR, r, lane, KR);
            wave_sync();
            uint4 *dst = reinterpret_cast<uint4 *>(args.ops + it * (4ull * KR));   // 4 * KR digits = KR 16-byte units
            for (u32 v = lane; v < KR; v += 64) dst[v] = reinterpret_cast<const uint4 *>(s.stage)[v];
        }
    };
    // a quotient that needs more than KR digits does not fit num_limbs limbs (chip.rs:583-584); only possible when KR < K
    auto fold = [&](int st) {
        if (KR < (u32)K && w0) {
            bool hi = false;
#pragma unroll
            for (int m = 0; m < V; ++m) hi = hi || ((u32)(lane + 64 * m) >= KR && q[m] != 0);
            if (__ballot(hi) && st == H2R_OK) st = H2R_E_NOT_REDUCED;
        }
        if (st != H2R_OK && status == H2R_OK) status = st;
    };
    if (args.mode == CHAIN_MULMOD) {
        fold(mulmod(cur, bop));
        emit(0, cur, bop);
        if (w0 && status == H2R_OK && args.out) glb_store<K>(args.out + elem * KR, r, lane, KR);
    } else {
        // pow_mod_fixed_exp (chip.rs:710-742) / pow_mod (chip.rs:664-696)
        u32 t = seg ? args.t_base : 0;
        const bool var = args.mode == CHAIN_POW_VAR;
        const u32 nbits = var ? args.e_num_limbs * args.exp_limb_bits : args.e.nbits;
        const u32 b_lo = seg ? args.bit_lo : 0, b_hi = seg ? args.bit_hi : nbits;
        u8 *etrace = args.trace ? args.trace + elem * args.elem_stride : nullptr;
        // Exponent bits are fetched one 32-bit word at a time: a per-bit load would put an s_waitcnt vmcnt(0) into
        // every iteration, which also waits for the previous mul_mod's operand stores (slow while a record kernel
        // saturates HBM next to this one).
        u32 eword = 0;
        for (u32 bi = b_lo; bi < b_hi; ++bi) {
            u32 bit;
            if (var) {  // main_gate.to_bits per e-limb, LSB first (chip.rs:674-681)
                const u32 limb = bi / args.exp_limb_bits, pos = bi % args.exp_limb_bits;
                if ((pos & 31) == 0 || (SEG && bi == b_lo)) eword = (args.e_limbs + (elem * args.e_num_limbs + limb) * args.digits_per_limb)[pos >> 5];
                bit = (eword >> (pos & 31)) & 1u;
                if (etrace && tid == 0) etrace[args.off_e_bits + bi] = (u8)bit;
            } else {
                if ((bi & 31) == 0 || (SEG && bi == b_lo)) eword = args.e.words[bi >> 5];
                bit = (eword >> (bi & 31)) & 1u;
            }
            if (var) {
                // muled = mul_mod(acc, squared) ALWAYS (:686); acc[j] = select(muled[j], acc[j], bit) (:688-691)
                fold(mulmod(acc, cur));
                emit(t, acc, cur);
                ++t;
#pragma unroll
                for (int m = 0; m < V; ++m) acc[m] = bit ? r[m] : acc[m];
                if (etrace && w0 && status == H2R_OK)
                    glb_store<K>((u32 *)(etrace + args.off_selected + (u64)bi * args.selected_stride), acc, lane, KR);
            }
            // squared = square_mod(cur) (:734 resp. :693)
            fold(mulmod(cur, cur));
            emit(t, cur, cur);
            ++t;
            u32 sq[V];
#pragma unroll
            for (int m = 0; m < V; ++m) sq[m] = r[m];
            if (!var && bit) {  // acc = mul_mod(acc, cur_sq) with the value BEFORE this squaring (:732-739)
                fold(mulmod(acc, cur));
                emit(t, acc, cur);
                ++t;
#pragma unroll
                for (int m = 0; m < V; ++m) acc[m] = r[m];
            }
#pragma unroll
            for (int m = 0; m < V; ++m) cur[m] = sq[m];
        }
        if (status == H2R_OK && w0) {
            if (SEG && b_hi < nbits) {   // not the last segment: the pair the next launch resumes from
                glb_store<K>(args.state + (elem * 2 + 0) * KR, cur, lane, KR);
                glb_store<K>(args.state + (elem * 2 + 1) * KR, acc, lane, KR);
            } else {
                if (args.out) glb_store<K>(args.out + elem * KR, acc, lane, KR);
                if (etrace && args.write_result_to_trace) glb_store<K>((u32 *)(etrace + args.off_result), acc, lane, KR);
            }
        }
    }
    if (tid == 0) args.status[elem] = (u8)status;
}

// DEEP: the latency build for small batches (about one block per CU, nothing else to hide LDS latency behind): the
// product loop preloads its operands 16 products ahead and runs four accumulators; it needs twice the registers.
// The grid may be smaller than the batch: block b then runs elements b, b + gridDim.x, ... one after the other, which
// bounds the kernel's footprint on the CUs whatever the batch size (a co-running record kernel keeps its store rate).
#ifndef H2R_CHAIN_MINB_BIG
#define H2R_CHAIN_MINB_BIG (H2R_CHAIN_MINB / 2)   // waves per SIMD the register budget of the K > 64 builds is sized for
#endif
// SEG: the build that can walk a segment of a long exponent (ChainArgs::state); the calls without one keep the build whose registers are
// what they were (the 64-digit throughput build: 80 VGPRs, no scratch).
template <int K, int NW, bool DEEP, bool SEG = false>
__global__ __launch_bounds__(64 * NW, DEEP ? 2 : (K <= 64 ? H2R_CHAIN_MINB : H2R_CHAIN_MINB_BIG)) void chain_kernel(ChainArgs args) {
    __shared__ ChainLds<K, NW> s;
    if (args.prio) __builtin_amdgcn_s_setprio(3);
    for (u64 elem = blockIdx.x; elem < args.batch; elem += gridDim.x) {
        if (elem != blockIdx.x) __syncthreads();   // every wave is done with the previous element's LDS
        chain_element<K, NW, DEEP, SEG>(args, s, elem);
    }
}

// The one-wave form (K <= 32, h2r_chain_wave.hpp): a workgroup is CHAIN_WAVE_WPB independent waves, one element each; no workgroup barrier
// anywhere.  Workgroup b walks the elements (b * WPB + wave) + k * gridDim.x * WPB.
constexpr int CHAIN_WAVE_WPB = 4;
template <int K, bool SEG = false>
__global__ __launch_bounds__(64 * CHAIN_WAVE_WPB) void chain_wave_kernel(ChainArgs args) {
    __shared__ ChainLds<K, 1> s[CHAIN_WAVE_WPB];
    const int wv = threadIdx.x >> 6;
    if (args.prio) __builtin_amdgcn_s_setprio(3);
    for (u64 elem = (u64)blockIdx.x * CHAIN_WAVE_WPB + wv; elem < args.batch; elem += (u64)gridDim.x * CHAIN_WAVE_WPB)
        chain_element<K, 1, false, SEG, true>(args, s[wv], elem);
}

// ---- two chains per element, side by side (round 3) -----------------------------------------------------------------
// pow_mod_fixed_exp (chip.rs:731-740) is TWO dependent chains, not one: the squarings s[i+1] = s[i]^2 mod n and the running product
// acc <- acc * s[i] mod n for the set bits -- and the multiply of bit i needs s[i], not s[i+1].  pow_mod (Var, chip.rs:684-694) likewise:
// muled = acc * s[i] (always) and s[i+1] = s[i]^2 are independent given (acc, s[i]).  This build gives an element EIGHT waves in two groups
// of four: group 0 walks the squarings, group 1 the multiplies of the same exponent bit, each with its own LDS and owner wave, in
// lockstep (block_mulmod's barriers are data-independent, so both groups meet at every one).  The critical path of an element is one
// mul_mod per exponent bit instead of 1 + bit (fixed) or 2 (Var): BASELINE config 5's 3,072 dependent mul_mods become 2,048 steps,
// the Var path's 4,096 become 2,048.  On a zero bit of a fixed exponent group 1 only meets the barriers (it used to run a mul_mod whose
// result was dropped, which took issue slots from the squaring next to it).  The build is chosen for latency-bound batches (at most two
// elements per CU) and, for fixed exponents, dense ones.  The values, the items of the operand
// buffer and every status are those of chain_element.
template <int K, bool DEEP>
__device__ __forceinline__ void chain_element_dual(const ChainArgs &args, ChainLds<K, 4> (&s2)[2], u32 (&xch)[K], int (&xst)[2], const u64 elem) {
    constexpr int NW = 4;
    using G = Geo<K, NW>;
    constexpr int V = G::V;
    const int g = threadIdx.x >> 8, tg = threadIdx.x & 255;          // group, thread within the group
    const int lane = tg & 63, wave = tg >> 6;
    const bool w0 = wave == 0;
    ChainLds<K, NW> &s = s2[g];
    const u32 *n_g = args.n + elem * args.n_stride;
    const u32 KR = args.kreal;
    u32 nraw[V];
#pragma unroll
    for (int m = 0; m < V; ++m) nraw[m] = ((u32)(lane + 64 * m) < KR) ? n_g[lane + 64 * m] : 0;
    for (int i = tg; i < 3 * K; i += 256) { s.bpad[i] = 0; s.nnpad[i] = 0; s.mupad[i] = 0; }
    if (tg == 0) { s.dbg = nullptr; s.dbg_n = 0; }
    if (g == 0 && w0 && args.n_copy) glb_store<K>(args.n_copy + elem * KR, nraw, lane, KR);
    int status = H2R_OK;
    {
        bool nz = false;
#pragma unroll
        for (int m = 0; m < V; ++m) nz = nz || __ballot(nraw[m] != 0) != 0;
        if (!nz) status = H2R_E_ZERO_MODULUS;
    }
    u32 cur[V], acc[V];
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const int v = lane + 64 * m;
        cur[m] = (u32)v < KR ? args.a[elem * KR + v] : 0;
        acc[m] = (v == 0) ? 1u : 0u;
    }
    const bool resumed = args.state && args.bit_lo > 0;   // a later segment of a long exponent (chain_element)
    if (resumed) {
        if (args.status[elem] != 0) return;
#pragma unroll
        for (int m = 0; m < V; ++m) {
            const int v = lane + 64 * m;
            cur[m] = (u32)v < KR ? args.state[(elem * 2 + 0) * KR + v] : 0;
            acc[m] = (u32)v < KR ? args.state[(elem * 2 + 1) * KR + v] : 0;
        }
    }
    if (!resumed && status == H2R_OK && args.check_in_field && wave_ge<K>(cur, nraw, lane)) status = H2R_E_NOT_IN_FIELD;
    const bool var = args.mode == CHAIN_POW_VAR;
    if (!resumed && var && args.exp_limb_bits < 32 * args.digits_per_limb) {
        bool wide = false;
        for (u32 l = threadIdx.x; l < args.e_num_limbs; l += 512) {
            const u32 *ed = args.e_limbs + (elem * args.e_num_limbs + l) * args.digits_per_limb;
            const u64 v = args.digits_per_limb == 2 ? (((u64)ed[1] << 32) | ed[0]) : (u64)ed[0];
            wide = wide || (v >> args.exp_limb_bits) != 0;
        }
        if (__syncthreads_or(wide ? 1 : 0) && status == H2R_OK) status = H2R_E_SHAPE;
    }
    if (status != H2R_OK) {   // block-uniform early exit
        if (threadIdx.x == 0) args.status[elem] = (u8)status;
        return;
    }
    u32 shift, nn[V];
    (void)chain_modulus_setup<K, NW>(s, nraw, lane, wave, shift, nn);   // both groups, each into its own LDS (same barriers)
    u32 q[V], r[V];
    const u64 item0 = elem * args.T;
    auto emit = [&](u32 t, const u32 (&oa)[V], const u32 (&ob)[V]) {
        if (w0 && status == H2R_OK) {
            wave_sync();
            lds_store_n<K>(s.stage, oa, lane, KR); lds_store_n<K>(s.stage + KR, ob, lane, KR);
            lds_store_n<K>(s.stage + 2 * KR, q, lane, KR); lds_store_n<K>(s.stage + 3 * KR, r, lane, KR);
            wave_sync();
            uint4 *dst = reinterpret_cast<uint4 *>(args.ops + (item0 + t) * (4ull * KR));
            for (u32 v = lane; v < KR; v += 64) dst[v] = reinterpret_cast<const uint4 *>(s.stage)[v];
        }
    };
    auto fold = [&](int st) {
        if (KR < (u32)K && w0) {
            bool hi = false;
#pragma unroll
            for (int m = 0; m < V; ++m) hi = hi || ((u32)(lane + 64 * m) >= KR && q[m] != 0);
            if (__ballot(hi) && st == H2R_OK) st = H2R_E_NOT_REDUCED;
        }
        if (st != H2R_OK && status == H2R_OK) status = st;
    };
    u32 t = args.state ? args.t_base : 0;
    const u32 nbits = var ? args.e_num_limbs * args.exp_limb_bits : args.e.nbits;
    const u32 b_lo = args.state ? args.bit_lo : 0, b_hi = args.state ? args.bit_hi : nbits;
    u8 *etrace = args.trace ? args.trace + elem * args.elem_stride : nullptr;
    u32 eword = 0;
    for (u32 bi = b_lo; bi < b_hi; ++bi) {
        u32 bit;
        if (var) {
            const u32 limb = bi / args.exp_limb_bits, pos = bi % args.exp_limb_bits;
            if ((pos & 31) == 0 || bi == b_lo) eword = (args.e_limbs + (elem * args.e_num_limbs + limb) * args.digits_per_limb)[pos >> 5];
            bit = (eword >> (pos & 31)) & 1u;
            if (etrace && threadIdx.x == 0) etrace[args.off_e_bits + bi] = (u8)bit;
        } else {
            if ((bi & 31) == 0 || bi == b_lo) eword = args.e.words[bi >> 5];
            bit = (eword >> (bi & 31)) & 1u;
        }
        // group 0: squared = square_mod(cur) (:734 / :693);  group 1: acc * cur (:686 always for Var; :739 for a set bit of a fixed
        // exponent -- for a zero bit the same arithmetic runs and is dropped: the barriers inside are what both groups share)
        if (g == 0) fold(block_mulmod<K, NW, DEEP>(s, shift, lane, wave, cur, cur, nn, q, r));
        else if (var || bit) fold(block_mulmod<K, NW, DEEP>(s, shift, lane, wave, acc, cur, nn, q, r));
        else {
            // a zero bit of a fixed exponent: no multiply (chip.rs:735-739).  The group only keeps the workgroup's barrier count --
            // block_mulmod's is 2 per product + 2 per shift when the modulus needs normalising (block-uniform) -- and leaves the
            // SIMDs to the squaring, which is the critical path
            const int nbar = block_mulmod_barriers(shift != 0);
            for (int b = 0; b < nbar; ++b) __syncthreads();
        }
        // items of the operand buffer in the reference's call order: Var: multiply 2 bi, square 2 bi + 1; fixed: square t, multiply t + 1
        if (g == 0) emit(var ? 2 * bi + 1 : t, cur, cur);
        else if (var || bit) emit(var ? 2 * bi : t + 1, acc, cur);
        t += 1 + ((var || bit) ? 1u : 0u);
        if (g == 1) {
            if (var) {
#pragma unroll
                for (int m = 0; m < V; ++m) acc[m] = bit ? r[m] : acc[m];            // select(muled, acc, bit), :688-691
                if (etrace && w0 && status == H2R_OK)
                    glb_store<K>((u32 *)(etrace + args.off_selected + (u64)bi * args.selected_stride), acc, lane, KR);
            } else if (bit) {
#pragma unroll
                for (int m = 0; m < V; ++m) acc[m] = r[m];
            }
        } else if (w0) lds_store<K>(xch, r, lane);                                   // s[i+1] for both groups
        __syncthreads();
        if (w0) lds_load<K>(cur, xch, lane);   // (xch is rewritten only behind the next mul_mod, whose barriers every wave passes first)
    }
    // the two groups' statuses; the result is group 1's acc
    if (tg == 0) xst[g] = status;
    __syncthreads();
    const int st_all = xst[0] != H2R_OK ? xst[0] : xst[1];
    if (b_hi < nbits) {   // not the last segment: the pair the next launch resumes from (group 0 holds the squarings, group 1 the product)
        if (w0 && st_all == H2R_OK) glb_store<K>(args.state + (elem * 2 + g) * KR, g == 0 ? cur : acc, lane, KR);
    } else if (g == 1 && w0 && st_all == H2R_OK) {
        if (args.out) glb_store<K>(args.out + elem * KR, acc, lane, KR);
        if (etrace && args.write_result_to_trace) glb_store<K>((u32 *)(etrace + args.off_result), acc, lane, KR);
    }
    if (threadIdx.x == 0) args.status[elem] = (u8)st_all;
}

template <int K, bool DEEP>
__global__ __launch_bounds__(512, 2) void chain_dual_kernel(ChainArgs args) {
    __shared__ ChainLds<K, 4> s2[2];
    __shared__ u32 xch[K];
    __shared__ int xst[2];
    if (args.prio) __builtin_amdgcn_s_setprio(3);
    for (u64 elem = blockIdx.x; elem < args.batch; elem += gridDim.x) {
        if (elem != blockIdx.x) __syncthreads();
        chain_element_dual<K, DEEP>(args, s2, xch, xst, elem);
    }
}

// The shared modulus' Barrett constants, once per call (one workgroup).
template <int K, int NW>
__global__ __launch_bounds__(64 * NW) void recip_kernel(const u32 *n, u32 kreal, u32 *pre) {
    constexpr int V = Geo<K, NW>::V;
    __shared__ ChainLds<K, NW> s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 nraw[V], nn[V];
#pragma unroll
    for (int m = 0; m < V; ++m) nraw[m] = ((u32)(lane + 64 * m) < kreal) ? n[lane + 64 * m] : 0;
    for (int i = threadIdx.x; i < 3 * K; i += 64 * NW) { s.bpad[i] = 0; s.nnpad[i] = 0; s.mupad[i] = 0; }
    if (threadIdx.x == 0) { s.dbg = nullptr; s.dbg_n = 0; }
    __syncthreads();
    u32 shift;
    const int status = chain_modulus_setup<K, NW>(s, nraw, lane, wave, shift, nn);
    if (threadIdx.x == 0) { pre[0] = shift; pre[1] = (u32)status; pre[2] = 0; pre[3] = 0; }
    if (status == H2R_OK)
        for (int i = threadIdx.x; i < K; i += 64 * NW) { pre[CHAIN_PRE_HDR + i] = s.nnpad[K + i]; pre[CHAIN_PRE_HDR + K + i] = s.mupad[K + i]; }
}

// ================================================================================================
// K1: trace kernel
// ================================================================================================
// Workgroup -> work mapping of the big streaming kernels.  The hardware hands consecutive workgroups to the 8 XCDs in
// turn (blockIdx % 8); mapped one to one, the XCDs sweep the output together, a few hundred MB wide.  Giving every XCD a
// CONTIGUOUS EIGHTH of the work instead keeps the eight store streams far apart in the address space, and the memory system
// likes that: the record kernel of the 44 GB config-4 trace goes from 5.3-5.5 to 6.6-6.9 TB/s, RSA-3072's from 5.3-5.5 to
// 5.7-6.1 TB/s, RSA-2048's by 0-6 % depending on where the buffers landed (same-box A/B runs in
// profiles/r02_xcd_mapping.txt; eighths are what matters -- chunks of 16-256 workgroups per XCD change nothing).
// A bijection on [0, n): the n % 8 trailing workgroups keep their index.  Small launches are left alone.
__device__ __forceinline__ u32 xcd_contiguous_block(u32 b, u32 n) {
    const u32 n8 = n >> 3;
    if (n < 2048 || b >= (n8 << 3)) return b;
    return (b & 7) * n8 + (b >> 3);
}

template <int LW> struct LimbT;
template <> struct LimbT<64> { using type = u64; };
template <> struct LimbT<32> { using type = u32; };

// Wide values: up to 160 bits for 64-bit limbs (u128 + u32), 128 bits for 32-bit limbs.
template <int LW> struct Wide;
template <> struct Wide<64> {
    u128 lo; u32 hi;
    __device__ __forceinline__ static Wide zero() { return Wide{0, 0}; }
    __device__ __forceinline__ static Wide from(u128 x) { return Wide{x, 0}; }
    __device__ __forceinline__ Wide operator+(const Wide &o) const { Wide r; r.lo = lo + o.lo; r.hi = hi + o.hi + (r.lo < lo ? 1u : 0u); return r; }
    __device__ __forceinline__ Wide operator-(const Wide &o) const { Wide r; r.lo = lo - o.lo; r.hi = hi - o.hi - (lo < o.lo ? 1u : 0u); return r; }
    __device__ __forceinline__ u64 low_limb() const { return (u64)lo; }
    __device__ __forceinline__ Wide shr_limb() const { Wide r; r.lo = (lo >> 64) | ((u128)hi << 64); r.hi = 0; return r; }
    __device__ __forceinline__ Wide shl_limb() const { Wide r; r.lo = lo << 64; r.hi = (u32)(lo >> 64); return r; }
    __device__ __forceinline__ u64 hi_word() const { return (u64)(i64)(i32)hi; }  // sign-extended third word
};
template <> struct Wide<32> {
    u128 lo;
    __device__ __forceinline__ static Wide zero() { return Wide{0}; }
    __device__ __forceinline__ static Wide from(u128 x) { return Wide{x}; }
    __device__ __forceinline__ Wide operator+(const Wide &o) const { return Wide{lo + o.lo}; }
    __device__ __forceinline__ Wide operator-(const Wide &o) const { return Wide{lo - o.lo}; }
    __device__ __forceinline__ u64 low_limb() const { return (u64)(u32)lo; }
    __device__ __forceinline__ Wide shr_limb() const { return Wide{lo >> 32}; }
    __device__ __forceinline__ Wide shl_limb() const { return Wide{lo << 32}; }
    __device__ __forceinline__ u64 hi_word() const { return 0; }
};

enum { TRACE_FULL = 0, TRACE_MUL = 1, TRACE_EQ = 2 };

struct TraceArgs {
    const void *opA, *opB, *opQ, *opR;  // limbs of item k at [k * op_stride, k * op_stride + L)
    u64 op_stride;                      // L for caller-owned arrays, 4L inside the chain kernel's ops buffer
    const void *n;                      // [elem][L] limbs
    u64 n_stride;                       // limbs between moduli (0 = shared)
    const u8 *status;                   // [elem]; nonzero => skip the element's items
    u64 n_items; u32 T;                 // item = elem*T + t
    u32 t_lo, T_ops;                    // T_ops != 0 (a segment of a long exponent): item = elem*T + (t - t_lo) covers the mul_mods
                                        // [t_lo, t_lo + T) of every element, whose operands sit at item elem*T_ops + t of the ops buffer
    u8 *trace; u64 elem_stride, off_records, record_stride;
    const u8 *const_rec;                // a record whose ACCX/QACC/MODACC/NQ2/AMNQ2 planes hold the (w,L) constants
    u64 off[H2R_PL_COUNT];
    u64 wm[3];                          // word_max (chip.rs:838)
    u32 carry_bits, carry_sub_bits, carry_nsub, carry_sub_stride;
    u32 ablate;                         // timing experiments only (H2R_ABLATE); 0 in production
    u32 dyn_lds;                        // developer override: raw dynamic LDS per block (sweeps)
    u32 residency;                      // host only: workgroups per CU the launch is capped to (0 = uncapped)
    u32 prio;                           // raise wave priority (pipeline co-scheduling)
    u32 acc_spg, acc_lo_row; u64 acc_lo_group, acc_hi_group;   // accumulator-plane addressing (h2r_layout)
    u32 mode;                           // TRACE_FULL (mul_mod), TRACE_MUL (BigIntChip::mul only), TRACE_EQ (is_equal_muled only)
    const u64 *muled_a, *muled_b;       // TRACE_EQ inputs: [item][2L] x 4 u64 (256-bit columns)
    u64 *muled_out;                     // TRACE_MUL output, same format
    u8 *eq_out;                         // TRACE_EQ: final eq_bit per item
};

template <int LW, int L>
struct TraceLds {
    using limb_t = typename LimbT<LW>::type;
    limb_t A[2][L];  // [0] = a, [1] = q
    limb_t B[2][L];  // [0] = b, [1] = n
    limb_t r[L];
    u64 c0[2][2 * L], c1[2][2 * L];  // final columns: [0] ab, [1] eq_b   (words 0, 1)
    u32 c2[2][2 * L];                //                                    (word 2)
    u64 dhi0[2 * L]; u32 dhi1[2 * L]; u32 shi[2 * L];
};

// Record stores carry the non-temporal (streaming) cache policy: the trace is written once and not read again by
// these kernels, and keeping it from displacing L2 lines is worth +6..8 % on the pipelined path and +1..4 % alone
// (same-box A/B, tools/ab_nt.sh; the sc0/sc1 scope bits make no difference with or without nt).
// -DH2R_STORE_PLAIN restores ordinary stores for such A/B runs.
#if defined(H2R_STORE_BITS)   // developer A/B of the cache-policy bits: -DH2R_STORE_BITS='"sc1 nt"'
typedef u32 h2r_v4u32 __attribute__((ext_vector_type(4)));
typedef u32 h2r_v2u32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(u8 *p, u64 a, u64 b) {
    h2r_v4u32 v; v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)b; v.w = (u32)(b >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off " H2R_STORE_BITS :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st8(u8 *p, u64 a) {
    h2r_v2u32 v; v.x = (u32)a; v.y = (u32)(a >> 32);
    asm volatile("global_store_dwordx2 %0, %1, off " H2R_STORE_BITS :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st4(u8 *p, u32 a) { asm volatile("global_store_dword %0, %1, off " H2R_STORE_BITS :: "v"(p), "v"(a) : "memory"); }
#elif !defined(H2R_STORE_PLAIN)
typedef u64 h2r_v2u64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(u8 *p, u64 a, u64 b) {
    h2r_v2u64 v; v.x = a; v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<h2r_v2u64 *>(p));
}
__device__ __forceinline__ void st8(u8 *p, u64 a) { __builtin_nontemporal_store(a, reinterpret_cast<u64 *>(p)); }
__device__ __forceinline__ void st4(u8 *p, u32 a) { __builtin_nontemporal_store(a, reinterpret_cast<u32 *>(p)); }
#else
__device__ __forceinline__ void st16(u8 *p, u64 a, u64 b) {
    *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(a, b);
}
__device__ __forceinline__ void st8(u8 *p, u64 a) { *reinterpret_cast<u64 *>(p) = a; }
__device__ __forceinline__ void st4(u8 *p, u32 a) { *reinterpret_cast<u32 *>(p) = a; }
#endif

// Plain (cacheable) stores for the small array-of-structs regions of the aux / Fresh-op / refresh kernels: their 4- and
// 8-byte stores at an 80-byte stride must merge in L2 -- with the streaming policy every one of them became a partial
// HBM write, and a co-running record kernel lost 15 % (verify path: 0.262 vs 0.237 ms/step).
__device__ __forceinline__ void pst16(u8 *p, u64 a, u64 b) { *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(a, b); }
__device__ __forceinline__ void pst8(u8 *p, u64 a) { *reinterpret_cast<u64 *>(p) = a; }
__device__ __forceinline__ void pst4(u8 *p, u32 a) { *reinterpret_cast<u32 *>(p) = a; }

template <int LW>
__device__ __forceinline__ void store_wide(u8 *rec, const u64 *off, int pl_lo, u64 idx, const Wide<LW> &v) {
    st16(rec + off[pl_lo] + idx * 16, (u64)v.lo, (u64)(v.lo >> 64));
    if constexpr (LW == 64) st8(rec + off[pl_lo + 1] + idx * 8, v.hi_word());
}
template <int LW>
__device__ __forceinline__ void store_limb(u8 *rec, const u64 *off, int pl, u64 idx, u64 v) {
    if constexpr (LW == 64) st8(rec + off[pl] + idx * 8, v); else st4(rec + off[pl] + idx * 4, (u32)v);
}
// CARRY-class value (carry_bits bits): 16 bytes for 64-bit limbs, 8 for 32-bit limbs.
template <int LW>
__device__ __forceinline__ void store_carry(u8 *rec, const u64 *off, int pl, u64 idx, const Wide<LW> &v) {
    if constexpr (LW == 64) st16(rec + off[pl] + idx * 16, (u64)v.lo, (u64)(v.lo >> 64)); else st8(rec + off[pl] + idx * 8, (u64)v.lo);
}
// one byte per sub-limb (RangeChip::assign decomposition of a limb: limb_width/8 bits each, 8 of them)
template <int LW>
__device__ __forceinline__ u64 limb_sub_bytes(u64 v) {
    if constexpr (LW == 64) return v;  // 8-bit sub-limbs: the bytes themselves
    else {                             // 4-bit sub-limbs -> one byte each
        u64 x = v & 0xffffffffull;
        x = (x | (x << 16)) & 0x0000ffff0000ffffull;
        x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
        x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
        return x;
    }
}

// Running accumulator of a product column: 160 bits (5 words) for 64-bit limbs, 96 bits (3 words) for
// 32-bit limbs, with explicit carry chains (v_add_co / v_addc_co) and branch-free reset / capture.
template <int LW> struct ColAcc;
template <> struct ColAcc<64> {
    u32 w[5];
    __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = w[3] = w[4] = 0; }
    __device__ __forceinline__ void keep_if(bool k) { const u32 m = k ? ~0u : 0u; w[0] &= m; w[1] &= m; w[2] &= m; w[3] &= m; w[4] &= m; }
    __device__ __forceinline__ void add_product(u64 x, u64 y) {  // += x * y (64 x 64 -> 128)
        const u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
        const u64 t0 = (u64)x0 * y0;
        const u64 t1 = (u64)x1 * y0 + (t0 >> 32);
        const u64 t2 = (u64)x0 * y1 + (u32)t1;
        const u64 hi = (u64)x1 * y1 + (t1 >> 32) + (t2 >> 32);
        u32 c;
        w[0] = __builtin_addc(w[0], (u32)t0, 0u, &c);
        w[1] = __builtin_addc(w[1], (u32)t2, c, &c);
        w[2] = __builtin_addc(w[2], (u32)hi, c, &c);
        w[3] = __builtin_addc(w[3], (u32)(hi >> 32), c, &c);
        w[4] += c;
    }
    __device__ __forceinline__ void take_if(bool t, const ColAcc &o) { for (int k = 0; k < 5; ++k) w[k] = t ? o.w[k] : w[k]; }
    __device__ __forceinline__ u64 lo0() const { return ((u64)w[1] << 32) | w[0]; }
    __device__ __forceinline__ u64 lo1() const { return ((u64)w[3] << 32) | w[2]; }
    __device__ __forceinline__ u64 hi64() const { return (u64)w[4]; }
    __device__ __forceinline__ Wide<64> wide() const { Wide<64> r; r.lo = ((u128)lo1() << 64) | lo0(); r.hi = w[4]; return r; }
};
template <> struct ColAcc<32> {
    u32 w[3];
    __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = 0; }
    __device__ __forceinline__ void keep_if(bool k) { const u32 m = k ? ~0u : 0u; w[0] &= m; w[1] &= m; w[2] &= m; }
    __device__ __forceinline__ void add_product(u64 x, u64 y) {  // += x * y (32 x 32 -> 64)
        const u64 t = (u64)(u32)x * (u32)y;
        u32 c;
        w[0] = __builtin_addc(w[0], (u32)t, 0u, &c);
        w[1] = __builtin_addc(w[1], (u32)(t >> 32), c, &c);
        w[2] += c;
    }
    __device__ __forceinline__ void take_if(bool t, const ColAcc &o) { for (int k = 0; k < 3; ++k) w[k] = t ? o.w[k] : w[k]; }
    __device__ __forceinline__ u64 lo0() const { return ((u64)w[1] << 32) | w[0]; }
    __device__ __forceinline__ u64 lo1() const { return (u64)w[2]; }
    __device__ __forceinline__ u64 hi64() const { return 0; }
    __device__ __forceinline__ Wide<32> wide() const { Wide<32> r; r.lo = ((u128)lo1() << 64) | lo0(); return r; }
};

// Timing experiments (tools/ablate.sh, developer build -DH2R_ABLATION): skip parts of the kernel at run time.
#ifdef H2R_ABLATION
#define H2R_ABLATE(bit_) ((args.ablate & (bit_)) != 0)
#else
#define H2R_ABLATE(bit_) false
#endif

// Thread geometry of the record kernel for num_limbs = L: an item (one mul_mod) is worked on by 2L threads -- lanes
// [0, L) the a*b columns, lanes [L, 2L) the q*n columns, then one thread per un-carried column.  The item's thread
// group is padded to TPI threads so that it either divides a wavefront (a power of two <= 64) or is a whole number of
// wavefronts; the padding threads (t >= 2L, none when L is a power of two) only take part in barriers and ballots.
template <int L>
struct TraceGeo {
    // (one exception to the padding: two items of 96 threads fill three wavefronts exactly -- RSA-3072 as 48 x 64-bit limbs
    //  runs its items back to back, straddling wavefronts, instead of leaving a quarter of the lanes idle)
    static constexpr int pad(int t) {
        if (t > 64 && t % 64 != 0 && (2 * t) % 64 == 0 && 2 * t <= 256) return t;
        if (t >= 64) return (t + 63) / 64 * 64;
        int p = 1; while (p < t) p <<= 1; return p;
    }
    static constexpr int TPI = pad(2 * L);                 // threads per item
    static constexpr int IPB = TPI >= 256 ? 1 : 256 / TPI;  // items per workgroup
    static constexpr int BT = IPB * TPI;                   // threads per workgroup (256, or 192 for 64 < L <= 96)
    static_assert(TPI <= 256, "num_limbs > 128 not supported");
};

// BT = threads per workgroup (a multiple of the padded thread group of an item, at most 256)
// The LDS of one workgroup of the record kernel
template <int LW, int L, int BT = TraceGeo<L>::BT>
struct TraceShared {
    static constexpr int IPB = TraceGeo<L>::TPI >= BT ? 1 : BT / TraceGeo<L>::TPI;   // items per block
    TraceLds<LW, L> lds_all[IPB];
    u64 xg[8], xp[8], xbad[8];  // per-wave carry masks for multi-wave items (a workgroup has at most eight waves: the step launch of the 4096-bit shapes)
};
// The work of workgroup `block` of `n_blocks` (the kernel below; also callable as one role of a larger launch)
// TSEG = false (the step launches' record role, whose register allocation is tuned): no segment addressing (t_lo / T_ops) in the code.
template <int LW, int L, int BT = TraceGeo<L>::BT, bool TSEG = true>
__device__ __forceinline__ void trace_block(const TraceArgs &args, const u32 block, const u32 n_blocks, TraceShared<LW, L, BT> &sh) {
    using limb_t = typename LimbT<LW>::type;
    using W = Wide<LW>;
    constexpr int TPI = TraceGeo<L>::TPI;        // threads per item (>= 2L)
    constexpr int IPB = TPI >= BT ? 1 : BT / TPI;  // items per block
    static_assert(BT % 64 == 0 && (TPI >= BT ? TPI == BT : BT % TPI == 0), "block shape");
    constexpr int C = 2 * L - 1;
    constexpr int WPI = TPI / 64 > 0 ? TPI / 64 : 1;  // waves per item (when TPI >= 64)
    constexpr bool POW2 = (L & (L - 1)) == 0;
    TraceLds<LW, L> (&lds_all)[IPB] = sh.lds_all;
    u64 (&xg)[8] = sh.xg; u64 (&xp)[8] = sh.xp; u64 (&xbad)[8] = sh.xbad;

    if (args.prio) __builtin_amdgcn_s_setprio(3);  // co-scheduled with chain_kernel: keep the store stream fed
    const int tid = threadIdx.x;
    const int slot = tid / TPI, t = tid % TPI;
    const int h = t / L, i = t % L;
    const int lane = tid & 63, wave = tid >> 6;
    TraceLds<LW, L> &s = lds_all[slot];
    const u32 bid = xcd_contiguous_block(block, n_blocks);
    const u32 item = bid * IPB + slot;  // n_items < 2^32 (checked by the host)
    const bool in_range = item < args.n_items;
    const u32 elem32 = in_range ? item / args.T : 0;
    const u32 tt = in_range ? item - elem32 * args.T + (TSEG ? args.t_lo : 0u) : 0;
    const u64 elem = elem32;
    const u64 op_item = (TSEG && args.T_ops) ? elem * args.T_ops + tt : (u64)item;
    const bool live = in_range && t < 2 * L && (args.status == nullptr || args.status[elem] == 0);
    u8 *rec = args.trace + elem * args.elem_stride + args.off_records + (u64)tt * args.record_stride;
    const u64 *off = args.off;
    // An item that fits one wave (2L <= 64) never needs a workgroup barrier: its LDS traffic is
    // wave-local, so waves of a block run independently.
    auto item_sync = [&]() { if constexpr (TPI <= 64) wave_sync(); else __syncthreads(); };

    const u32 mode = args.mode;
    const bool prod = live && mode != TRACE_EQ && (mode == TRACE_FULL || h == 0);   // this thread runs a product column
    // ---- stage operands in LDS; emit q, r and their sub-limbs (chip.rs:588-599) -------------------
    if (live && mode == TRACE_MUL && h == 0) {   // BigIntChip::mul(a, b) alone: only the a*b half is active
        s.A[0][i] = reinterpret_cast<const limb_t *>(args.opA)[(u64)item * args.op_stride + i];
        s.B[0][i] = reinterpret_cast<const limb_t *>(args.opB)[(u64)item * args.op_stride + i];
    }
    if (live && mode == TRACE_FULL) {
        const u64 ib = op_item * args.op_stride;
        const limb_t *gQ = reinterpret_cast<const limb_t *>(args.opQ) + ib;
        const limb_t *gA = h == 0 ? reinterpret_cast<const limb_t *>(args.opA) + ib : gQ;
        const limb_t *gB = h == 0 ? reinterpret_cast<const limb_t *>(args.opB) + ib
                                  : reinterpret_cast<const limb_t *>(args.n) + elem * args.n_stride;
        const limb_t av = gA[i], bv = gB[i];
        // half 0 emits q and its sub-limbs, half 1 emits r (each store instruction covers both planes)
        const limb_t ov = h == 0 ? gQ[i] : reinterpret_cast<const limb_t *>(args.opR)[ib + i];
        s.A[h][i] = av; s.B[h][i] = bv;
        if (h == 1) s.r[i] = ov;
        store_limb<LW>(rec, off, h == 0 ? H2R_PL_Q : H2R_PL_R, i, ov);
        st8(rec + off[h == 0 ? H2R_PL_Q_SUB : H2R_PL_R_SUB] + (u64)i * 8, limb_sub_bytes<LW>(ov));
    }
    item_sync();

    // ---- BigIntChip::mul twice (chip.rs:386-419): lanes [0,L) a*b, lanes [L,2L) q*n ---------------
    // Lane i owns column i (steps s <= i) and then column i+L (steps s > i); step s multiplies
    // A[s] * B[(i-s) mod L], so every product a[j]*b[k] is visited once, in ascending j per column.
    ColAcc<LW> acc, first;
    acc.clear(); first.clear();
    if (prod && !H2R_ABLATE(2)) {
        // accumulator addressing (h2r.h): planar rows, or interleaved [ab half | qn half] rows with a shared HI row
        // row strides as layout_compute derives them from (limb_width, L): compile-time, so the stores use immediate offsets
        constexpr u64 lo_row = LW == 64 ? 2ull * L * 16 : 0, lo_group = LW == 64 ? 3ull * (2 * L * 16) : (u64)L * 16, hi_group = lo_group;
        constexpr bool two = LW == 64;   // layout_compute: interleaved rows, two steps per group, iff the HI word exists
        u8 *plo = rec + off[h == 0 ? H2R_PL_AB_LO : H2R_PL_QN_LO] + (u64)i * 16;
        u8 *phi = rec + off[h == 0 ? H2R_PL_AB_HI : H2R_PL_QN_HI] + (u64)i * 16;
        const limb_t *Ah = s.A[h], *Bh = s.B[h];
        limb_t x = Ah[0], y = Bh[i];
        u64 hi_even = 0;
#pragma unroll 4
        for (int st = 0; st < L; ++st) {
            limb_t xn, yn;
            if (H2R_ABLATE(4)) { xn = x + 3; yn = y ^ (limb_t)st; }
            else if constexpr (POW2) { xn = Ah[(st + 1) & (L - 1)]; yn = Bh[(i - st - 1) & (L - 1)]; }  // prefetch next step
            else { xn = Ah[st + 1 == L ? 0 : st + 1]; const int k = i - st - 1; yn = Bh[k < 0 ? k + L : k]; }   // (i - st - 1) mod L
            acc.keep_if(st != i + 1);       // column i is complete: start column i+L from zero
            if (H2R_ABLATE(8)) { acc.w[0] += (u32)x; acc.w[1] ^= (u32)y; }
            else acc.add_product(x, y);
            const u64 lo_off = two ? (u64)(st >> 1) * lo_group + (u64)(st & 1) * lo_row : (u64)st * lo_group;
            st16(plo + lo_off, acc.lo0(), acc.lo1());
            if constexpr (LW == 64) {       // third words of steps (2p, 2p+1) share one 16-byte slot
                if (st & 1) st16(phi + (u64)(st >> 1) * hi_group, hi_even, acc.hi64());
                else hi_even = acc.hi64();
            }
            first.take_if(st == i, acc);
            x = xn; y = yn;
        }
        // final columns -> LDS; eq_b[i] = qn[i] + r[i] for i < L (chip.rs:614-623)
        W fw = first.wide();
        if (mode == TRACE_MUL) {   // hand the un-carried columns to the caller (AssignedInteger<Muled>)
            u64 *mo = args.muled_out + (u64)item * (2 * L) * 4;
            mo[4 * i] = first.lo0(); mo[4 * i + 1] = first.lo1(); mo[4 * i + 2] = first.hi64(); mo[4 * i + 3] = 0;
            const bool has2 = i < L - 1;
            mo[4 * (i + L)] = has2 ? acc.lo0() : 0; mo[4 * (i + L) + 1] = has2 ? acc.lo1() : 0;
            mo[4 * (i + L) + 2] = has2 ? acc.hi64() : 0; mo[4 * (i + L) + 3] = 0;
        }
        if (h == 1) {
            fw = fw + W::from((u128)s.r[i]);
            store_wide<LW>(rec, off, H2R_PL_EQB_LO, i, fw);
        }
        s.c0[h][i] = (u64)fw.lo; s.c1[h][i] = (u64)(fw.lo >> 64);
        if constexpr (LW == 64) s.c2[h][i] = fw.hi;
        if (i < L - 1) {
            s.c0[h][i + L] = acc.lo0(); s.c1[h][i + L] = acc.lo1();
            if constexpr (LW == 64) s.c2[h][i + L] = acc.w[4];
        }
    }
    item_sync();
    if (H2R_ABLATE(1) || mode == TRACE_MUL) return;

    // ---- BigIntChip::is_equal_muled (chip.rs:822-895): thread t = column c -------------------------
    // Thread 2L-1 has no column; it still stores (zeros) so that every store instruction of this
    // phase covers whole 128-byte lines (the planes reserve 2L entries).
    const int c = t;
    const bool col = live && c < C;
    W wm; wm.lo = ((u128)args.wm[1] << 64) | args.wm[0];
    if constexpr (LW == 64) wm.hi = (u32)args.wm[2];
    W a_b = W::zero(), D = W::zero();
    u64 dlo = 0; W dhi = W::zero();
    if (col) {
        W A, Bq;
        if (mode == TRACE_EQ) {   // stand-alone is_equal_muled: the two Muled integers come from the caller
            const u64 *pa = args.muled_a + ((u64)item * (2 * L) + c) * 4, *pb = args.muled_b + ((u64)item * (2 * L) + c) * 4;
            A.lo = ((u128)pa[1] << 64) | pa[0]; Bq.lo = ((u128)pb[1] << 64) | pb[0];
            if constexpr (LW == 64) { A.hi = (u32)pa[2]; Bq.hi = (u32)pb[2]; }
        } else {
            A.lo = ((u128)s.c1[0][c] << 64) | s.c0[0][c];
            Bq.lo = ((u128)s.c1[1][c] << 64) | s.c0[1][c];
            if constexpr (LW == 64) { A.hi = s.c2[0][c]; Bq.hi = s.c2[1][c]; }
        }
        a_b = A - Bq;              // :859 (two's complement)
        D = a_b + wm;              // >= 0
        dlo = D.low_limb();
        dhi = D.shr_limb();
        s.dhi0[c] = (u64)dhi.lo; s.dhi1[c] = (u32)(dhi.lo >> 64);
    }
    item_sync();
    W dhi_prev = W::zero();
    u64 slo = 0; u32 shi = 0;
    if (col) {
        if (c > 0) dhi_prev.lo = ((u128)s.dhi1[c - 1] << 64) | s.dhi0[c - 1];
        const W S = W::from((u128)dlo) + dhi_prev;
        slo = S.low_limb();
        shi = (u32)S.shr_limb().lo;
        s.shi[c] = shi;
    }
    item_sync();
    u32 shi_prev = 0; bool gen = false, prop = false;
    if (col) {
        shi_prev = c > 0 ? s.shi[c - 1] : 0;
        const u128 U = (u128)slo + shi_prev;
        constexpr u64 mask = LW == 64 ? ~0ull : 0xffffffffull;
        gen = (U >> LW) != 0;
        prop = ((u64)U & mask) == mask;
    }
    // carry-in bit f[c]: within-wave ballots, chained across the waves of a multi-wave item
    const u64 G = __ballot(gen), P = __ballot(prop);
    bool f;
    if constexpr (TPI <= 64) {
        const CarryGroup cg = carry_group(G, P, false, 64);
        f = ((cg.cin_mask >> lane) & 1) != 0;
    } else {
        if (lane == 0) { xg[wave] = G; xp[wave] = P; }
        __syncthreads();
        // chain from the item's first wave; when items straddle wavefronts, from the workgroup's first wave -- thread 2L-1
        // of every item has no column (G = P = 0) and stops the chain between items
        const int w0 = (TPI % 64 == 0) ? (wave / WPI) * WPI : 0;
        bool cin = false;
        for (int k = w0; k < wave; ++k) cin = carry_group(xg[k], xp[k], cin, 64).cout;
        const CarryGroup cg = carry_group(G, P, cin, 64);
        f = ((cg.cin_mask >> lane) & 1) != 0;
    }
    bool f1 = true, f2 = true;
    W carry_out = W::zero(), sum = W::zero(), nq = W::zero(), qacc = W::zero();
    u64 cmod = 0, modacc = 0;
    if (col) {
        const W carry_in = dhi_prev + W::from((u128)shi_prev + (f ? 1u : 0u));
        sum = D + carry_in;                               // :860-861
        cmod = sum.low_limb();                            // :864 div_mod r
        carry_out = sum.shr_limb();                       //            q
        nq = carry_out.shl_limb();                        // :1345
    }
    if (live) {
        store_wide<LW>(rec, off, H2R_PL_AMB_LO, c, a_b);
        store_wide<LW>(rec, off, H2R_PL_SUM_LO, c, sum);
        store_carry<LW>(rec, off, H2R_PL_CARRY, c, carry_out);
        store_limb<LW>(rec, off, H2R_PL_CMOD, c, cmod);
        store_wide<LW>(rec, off, H2R_PL_NQ1_LO, c, nq);
        store_limb<LW>(rec, off, H2R_PL_AMNQ1, c, (sum - nq).low_limb());   // :1346
        // input-independent part of the step (acc_extra chain, :869-871): copy from the constant record
        // (whose entry 2L-1 is zero)
        const u8 *cr = args.const_rec;
        const ulonglong2 ax = *reinterpret_cast<const ulonglong2 *>(cr + off[H2R_PL_ACCX_LO] + (u64)c * 16);
        const ulonglong2 n2 = *reinterpret_cast<const ulonglong2 *>(cr + off[H2R_PL_NQ2_LO] + (u64)c * 16);
        st16(rec + off[H2R_PL_ACCX_LO] + (u64)c * 16, ax.x, ax.y);
        st16(rec + off[H2R_PL_NQ2_LO] + (u64)c * 16, n2.x, n2.y);
        if constexpr (LW == 64) {
            st8(rec + off[H2R_PL_ACCX_HI] + (u64)c * 8, *reinterpret_cast<const u64 *>(cr + off[H2R_PL_ACCX_HI] + (u64)c * 8));
            st8(rec + off[H2R_PL_NQ2_HI] + (u64)c * 8, *reinterpret_cast<const u64 *>(cr + off[H2R_PL_NQ2_HI] + (u64)c * 8));
            const ulonglong2 qa = *reinterpret_cast<const ulonglong2 *>(cr + off[H2R_PL_QACC] + (u64)c * 16);
            qacc.lo = ((u128)qa.y << 64) | qa.x;
            modacc = *reinterpret_cast<const u64 *>(cr + off[H2R_PL_MODACC] + (u64)c * 8);
            st8(rec + off[H2R_PL_AMNQ2] + (u64)c * 8, *reinterpret_cast<const u64 *>(cr + off[H2R_PL_AMNQ2] + (u64)c * 8));
        } else {
            qacc.lo = *reinterpret_cast<const u64 *>(cr + off[H2R_PL_QACC] + (u64)c * 8);
            modacc = *reinterpret_cast<const u32 *>(cr + off[H2R_PL_MODACC] + (u64)c * 4);
            st4(rec + off[H2R_PL_AMNQ2] + (u64)c * 4, *reinterpret_cast<const u32 *>(cr + off[H2R_PL_AMNQ2] + (u64)c * 4));
        }
        store_carry<LW>(rec, off, H2R_PL_QACC, c, qacc);
        store_limb<LW>(rec, off, H2R_PL_MODACC, c, modacc);
        f1 = cmod == modacc;                               // cs_acc_eq, :873
        // range-assign the carry (:879-885): duplicate value + sub-limbs; range_eq == 1 (:886).
        // Columns C-1 (no range check, :888-892) and 2L-1 (no column) store zeros.
        W dup = W::zero();
        u32 wds[4] = {0, 0, 0, 0};
        if (c < C - 1) {
            dup = carry_out;
            const u32 sb = args.carry_sub_bits, ns = args.carry_nsub;
            u128 v = carry_out.lo;
            const u32 m = (1u << sb) - 1;
            for (u32 k = 0; k < ns; ++k) { wds[k >> 2] |= ((u32)v & m) << (8 * (k & 3)); v >>= sb; }
        } else if (c == C - 1) {
            f2 = carry_out.lo == qacc.lo;                  // final_carry_eq, :890 (acc_extra == q_acc)
        }
        store_carry<LW>(rec, off, H2R_PL_CARRY_DUP, c, dup);
        st16(rec + off[H2R_PL_CARRY_SUB] + (u64)c * 16, ((u64)wds[1] << 32) | wds[0], ((u64)wds[3] << 32) | wds[2]);
    }
    // eq_bit is the running AND over (cs_acc_eq, range_eq | final_carry_eq) in column order (:874, :887, :891)
    const u64 bad = __ballot(col && !(f1 && f2));
    bool prev_ok;
    if constexpr (TPI <= 64) {
        const u64 seg = (TPI == 64) ? ~0ull : (((1ull << (TPI & 63)) - 1) << (lane - t));
        prev_ok = (bad & seg & ((1ull << lane) - 1)) == 0;
    } else {
        if (lane == 0) xbad[wave] = bad;
        __syncthreads();
        if constexpr (TPI % 64 == 0) {
            const int w0 = (wave / WPI) * WPI;
            prev_ok = (bad & ((1ull << lane) - 1)) == 0;
            for (int k = w0; k < wave; ++k) prev_ok = prev_ok && xbad[k] == 0;
        } else {   // items straddle wavefronts: no failed column among the item's threads [slot * TPI, tid)
            const int s0 = slot * TPI, ws = s0 >> 6, ls = s0 & 63;
            prev_ok = true;
            for (int k = ws; k <= wave; ++k) {
                u64 m = k == wave ? bad & ((1ull << lane) - 1) : xbad[k];
                if (k == ws) m &= ~((1ull << ls) - 1);
                prev_ok = prev_ok && m == 0;
            }
        }
    }
    if (live) {
        const u32 e1 = (prev_ok && f1) ? 1u : 0u, e2 = (e1 && f2) ? 1u : 0u;
        const u32 fl = (f1 ? 1u : 0u) | (e1 << 8) | ((f2 ? 1u : 0u) << 16) | (e2 << 24);
        st4(rec + off[H2R_PL_FLAGS] + (u64)c * 4, col ? fl : 0u);
        if (mode == TRACE_EQ && c == C - 1 && args.eq_out) args.eq_out[item] = (u8)e2;   // is_equal_muled's result bit
    }
}

template <int LW, int L, int BT = TraceGeo<L>::BT>
__global__ __launch_bounds__(BT) void trace_kernel(TraceArgs args) {
    __shared__ TraceShared<LW, L, BT> sh;
    trace_block<LW, L, BT>(args, blockIdx.x, gridDim.x, sh);
}

// ================================================================================================
// K6: auxiliary witness of RSAChip::verify_pkcs1v15_signature around the pow path
//   (a) BigIntChip::assert_in_field(x, n)  src/chip.rs:106 -> big_integer/chip.rs:1150 -> 998 -> 908-919:
//       is_less_than = sub (:310-373: add :245-297, sub_unchecked :1286-1318 twice, selects) and
//       is_equal_fresh (:780-805).
//   (b) the encoded-message check after the modpow, src/chip.rs:136-198.
// One wave per element; limb positions are spread over the lanes and every carry / borrow / running
// AND is resolved with ballots.  The output is the flat stream itself, section by section, each
// section starting on a 16-byte boundary of the device buffer (AuxGeom); ~10 KB per element.
// ================================================================================================
struct AuxGeom {
    u32 L, LB, SB, RA, STEP;  // limbs, LIMB bytes, LIMB+8 bytes (a_b/sum/c+carry*B), range-assigned limb (LB+8), add step
    __host__ __device__ static u64 a16(u64 x) { return (x + 15) & ~15ull; }
    __host__ __device__ AuxGeom(u32 L_, u32 lw) : L(L_), LB(lw / 8), SB(lw / 8 + 8), RA(lw / 8 + 8), STEP(3 * (lw / 8 + 8) + 2 * (lw / 8 + 8)) {}
    __host__ __device__ u64 add_sz(u32 n) const { return a16((u64)n * STEP); }
    __host__ __device__ u64 eq_sz(u32 n) const { return a16(2ull * n); }
    __host__ __device__ u64 cl_sz(u32 n) const { return a16((u64)n * RA); }
    __host__ __device__ u64 subu_sz(u32 n1) const { return cl_sz(n1) + add_sz(n1) + eq_sz(n1 + 1); }
    __host__ __device__ u64 sub_sz(u32 nA, u32 nB) const {
        const u32 m = nA > nB ? nA : nB, n1 = m + 1;
        return add_sz(m) + subu_sz(n1) + 16 + a16((u64)n1 * LB) + a16((u64)m * LB) + subu_sz(n1);
    }
    __host__ __device__ u64 sub_sz() const { return sub_sz(L, L); }
    __host__ __device__ u64 in_field_sz() const { return sub_sz() + eq_sz(L) + 16; }
    // largest Fresh-op region: sub_mod = sub(L, L) + sub(L, L+1) + (L+2) limbs
    __host__ __device__ u64 fresh_max_sz() const { return sub_sz(L, L) + sub_sz(L, L + 1) + add_sz(L) + a16((u64)(L + 2) * LB) + eq_sz(L) + 64; }
    __host__ __device__ u64 em_sz() const { return a16(2ull * L + 34); }
};

struct AuxArgs {
    const void *x, *n; u64 n_stride;   // limbs
    const u64 *hashed;                 // [elem][4] 64-bit limbs of the SHA-256 digest (nullable: no EM check)
    const void *powed;                 // [elem][L] result of the pow path
    u64 batch; u32 L;
    u8 *trace; u64 elem_stride, off_in_field, off_em;
    u8 *is_valid;                      // [elem] (nullable)
    const u8 *status;                  // [elem]: EM check skipped (is_valid = 0) when nonzero
};

constexpr int AUX_V = 3;  // positions per lane: L + 2 <= 192

template <int LW>
struct AuxW {
    static constexpr u64 MASK = LW == 64 ? ~0ull : 0xffffffffull;
    static constexpr int LB = LW / 8;
    // store a (LW+1)-bit value (lo, hi bit) in LB+8 bytes / a limb in LB bytes (4-byte granular)
    __device__ static void put_sb(u8 *p, u64 lo, u32 hi) {
        if constexpr (LW == 64) { pst8(p, lo); pst8(p + 8, hi); }
        else { pst4(p, (u32)lo); pst4(p + 4, (u32)(lo >> 32) | (hi << 0)); pst4(p + 8, 0); }
    }
    __device__ static void put_limb(u8 *p, u64 v) { if constexpr (LW == 64) pst8(p, v); else pst4(p, (u32)v); }
    __device__ static void put_ra(u8 *p, u64 v) {  // RangeChip::assign(limb): value + its 8 sub-limb bytes
        put_limb(p, v);
        const u64 sb = limb_sub_bytes<LW>(v);
        if constexpr (LW == 64) pst8(p + 8, sb); else { pst4(p + 4, (u32)sb); pst4(p + 8, (u32)(sb >> 32)); }
    }
};

// BigIntChip::add (chip.rs:245-297) of A (nA limbs) and B (nB limbs); OUT gets max_n + 1 limbs.
template <int LW>
__device__ __forceinline__ void aux_add(u8 *sec, const AuxGeom &g, const u64 (&A)[AUX_V], u32 nA, const u64 (&B)[AUX_V], u32 nB,
                                        u64 (&OUT)[AUX_V], int lane) {
    using X = AuxW<LW>;
    const u32 max_n = nA > nB ? nA : nB;
    bool cin = false;
#pragma unroll
    for (int m = 0; m < AUX_V; ++m) {
        const u32 p = lane + 64 * m;
        const u64 ai = p < nA ? A[m] : 0, bi = p < nB ? B[m] : 0;   // :258-263 zero padding
        const u64 ab_lo = (ai + bi) & X::MASK;
        const bool ab_hi = LW == 64 ? (ab_lo < ai) : (((ai + bi) >> 32) != 0);
        const bool act = p < max_n;
        const CarryGroup cg = carry_group(__ballot(act && ab_hi), __ballot(act && !ab_hi && ab_lo == X::MASK), cin, 64);
        cin = cg.cout;
        const u32 ci = (u32)((cg.cin_mask >> lane) & 1);
        const u64 c = (ab_lo + ci) & X::MASK;                          // :276
        const bool sum_hi = ab_hi || (ci && ab_lo == X::MASK);         // carry out of this limb, :277
        if (act) {
            u8 *q = sec + (u64)p * g.STEP;
            X::put_sb(q, ab_lo, ab_hi ? 1u : 0u);                       // a_b  :272
            X::put_sb(q + g.SB, c, sum_hi ? 1u : 0u);                   // sum  :273
            X::put_ra(q + 2 * g.SB, c);                                 // c    :279-280
            X::put_ra(q + 2 * g.SB + g.RA, sum_hi ? 1u : 0u);           // carry :281-282
            X::put_sb(q + 2 * g.SB + 2 * g.RA, c, sum_hi ? 1u : 0u);    // c + carry * 2^w  :283
        }
        OUT[m] = act ? c : (p == max_n ? (u64)ci : 0);                  // :286, :290
    }
}

// BigIntChip::is_equal_fresh (chip.rs:780-805); returns the final eq_bit.
template <int LW>
__device__ __forceinline__ bool aux_eq(u8 *sec, const u64 (&A)[AUX_V], u32 n1, const u64 (&B)[AUX_V], u32 n2, int lane) {
    const bool a_larger = n1 > n2;
    const u32 max_n = a_larger ? n1 : n2;
    bool all_prev = true;
#pragma unroll
    for (int m = 0; m < AUX_V; ++m) {
        const u32 p = lane + 64 * m;
        const bool act = p < max_n;
        bool flag;
        if (a_larger && p >= n2) flag = A[m] == 0;
        else if (!a_larger && p >= n1) flag = B[m] == 0;
        else flag = A[m] == B[m];
        const u64 bad = __ballot(act && !flag);
        const bool eq = all_prev && (bad & ((2ull << lane) - 1)) == 0;   // AND over positions <= p
        if (act) *reinterpret_cast<uint16_t *>(sec + 2ull * p) = (uint16_t)((flag ? 1u : 0u) | ((eq ? 1u : 0u) << 8));
        all_prev = all_prev && bad == 0;
    }
    return all_prev;
}

// BigIntChip::sub_unchecked (chip.rs:1286-1318): C = A - B (A >= B), then add(B, C) and is_equal_fresh(A, added).
template <int LW>
__device__ __forceinline__ void aux_subu(u8 *sec, const AuxGeom &g, const u64 (&A)[AUX_V], u32 n1, const u64 (&B)[AUX_V], u32 n2,
                                         u64 (&C)[AUX_V], int lane) {
    using X = AuxW<LW>;
    bool bin = false;
#pragma unroll
    for (int m = 0; m < AUX_V; ++m) {
        const u32 p = lane + 64 * m;
        const bool act = p < n1;
        const u64 ai = act ? A[m] : 0, bi = p < n2 ? B[m] : 0;
        const CarryGroup cg = carry_group(__ballot(act && ai < bi), __ballot(act && ai == bi), bin, 64);
        bin = cg.cout;
        C[m] = act ? ((ai - bi - ((cg.cin_mask >> lane) & 1)) & X::MASK) : 0;   // :1300, :1304-1311
        if (act) X::put_ra(sec + (u64)p * g.RA, C[m]);                           // :1307-1308
    }
    u64 added[AUX_V];
    aux_add<LW>(sec + g.cl_sz(n1), g, B, n2, C, n1, added, lane);                // :1315
    aux_eq<LW>(sec + g.cl_sz(n1) + g.add_sz(n1), A, n1, added, n1 + 1, lane);    // :1316
}

// limb `idx` of a distributed integer, broadcast to the wave
__device__ __forceinline__ u64 aux_limb(const u64 (&X)[AUX_V], u32 idx) {
    const u32 m = idx >> 6;
    return __shfl(m == 0 ? X[0] : (m == 1 ? X[1] : X[2]), (int)(idx & 63));
}

// BigIntChip::sub (chip.rs:310-373): REAL = |A - B| (max(nA,nB)+1 limbs), returns is_overflowed (1 iff A <= B).
// Advances `sec` past its sections.
template <int LW>
__device__ __forceinline__ bool aux_sub(u8 *&sec, const AuxGeom &g, const u64 (&A)[AUX_V], u32 nA, const u64 (&B)[AUX_V], u32 nB,
                                        u64 (&REAL)[AUX_V], int lane) {
    using X = AuxW<LW>;
    const u32 m = nA > nB ? nA : nB, n1 = m + 1;
    u64 MAXI[AUX_V], IA[AUX_V], IS[AUX_V], SL[AUX_V], SR[AUX_V];
#pragma unroll
    for (int k = 0; k < AUX_V; ++k) MAXI[k] = (u32)(lane + 64 * k) < nB ? X::MASK : 0;   // max_value(n2), :319 -> :138-154
    aux_add<LW>(sec, g, A, nA, MAXI, nB, IA, lane);                          // inflated_a, :321
    sec += g.add_sz(m);
    aux_subu<LW>(sec, g, IA, n1, B, nB, IS, lane);                           // inflated_subed, :323
    sec += g.subu_sz(n1);
    const bool not_ov = aux_limb(IS, nB) == 1;                               // :330
    if (lane == 0) *reinterpret_cast<uint16_t *>(sec) = (uint16_t)((not_ov ? 1u : 0u) | ((not_ov ? 0u : 1u) << 8));   // :330-331
    sec += 16;
#pragma unroll
    for (int k = 0; k < AUX_V; ++k) {                                        // selects, :345-367
        const u32 p = lane + 64 * k;
        SL[k] = p < n1 ? (p >= nB ? (not_ov ? IS[k] : 0) : (not_ov ? IS[k] : B[k])) : 0;
        u64 r = 0;
        if (p < m) {
            if (p >= nA) r = not_ov ? MAXI[k] : 0;
            else if (p >= nB) r = not_ov ? 0 : A[k];
            else r = not_ov ? MAXI[k] : A[k];
        }
        SR[k] = r;
        if (p < n1) X::put_limb(sec + (u64)p * g.LB, SL[k]);
        if (p < m) X::put_limb(sec + AuxGeom::a16((u64)n1 * g.LB) + (u64)p * g.LB, SR[k]);
    }
    sec += AuxGeom::a16((u64)n1 * g.LB) + AuxGeom::a16((u64)m * g.LB);
    aux_subu<LW>(sec, g, SL, n1, SR, m, REAL, lane);                         // real_subed, :371
    sec += g.subu_sz(n1);
    return !not_ov;
}

// BigIntChip::is_less_than (chip.rs:908-919)
template <int LW>
__device__ __forceinline__ bool aux_less_than(u8 *&sec, const AuxGeom &g, const u64 (&A)[AUX_V], const u64 (&B)[AUX_V], int lane) {
    u64 REAL[AUX_V];
    const bool ov = aux_sub<LW>(sec, g, A, g.L, B, g.L, REAL, lane);         // is_less_than_or_equal, :915 -> :939
    const bool is_eq = aux_eq<LW>(sec, A, g.L, B, g.L, lane);                // :916
    sec += g.eq_sz(g.L);
    const bool lt = ov && !is_eq;                                            // :917-918
    if (lane == 0) *reinterpret_cast<uint16_t *>(sec) = (uint16_t)((is_eq ? 0u : 1u) | ((lt ? 1u : 0u) << 8));
    sec += 16;
    return lt;
}

// OWN: powed[elem] and status[elem] were stored by THIS wave a moment ago (the step launch's chain role): they are read back with
// agent-scope loads, i.e. from the L2 the stores went to, not from a possibly older L1 line.
template <int LW, bool OWN = false>
__device__ __forceinline__ void aux_wave(const AuxArgs &a, const u64 elem, const int lane, uint4 *aux_stage) {
    using X = AuxW<LW>;
    using limb_t = typename LimbT<LW>::type;
    const u32 L = a.L;
    const AuxGeom g(L, LW);
    u64 Xv[AUX_V], Nv[AUX_V];
#pragma unroll
    for (int m = 0; m < AUX_V; ++m) {
        const u32 p = lane + 64 * m;
        Xv[m] = p < L ? (u64) reinterpret_cast<const limb_t *>(a.x)[elem * L + p] : 0;
        Nv[m] = p < L ? (u64) reinterpret_cast<const limb_t *>(a.n)[elem * a.n_stride + p] : 0;
    }
    u8 *et = a.trace + elem * a.elem_stride;
    // Both regions are assembled in LDS and leave as 16-byte streaming stores: written in place, their 4/8-byte stores
    // at an 80-byte stride slowed a co-running record kernel by 8 % (pipelined verify 0.255 vs 0.237 ms/step).
    const u32 if_u4 = (u32)(g.in_field_sz() / 16), em_u4 = (u32)(g.em_sz() / 16);
    for (u32 k = lane; k < if_u4 + em_u4; k += 64) aux_stage[k] = make_uint4(0, 0, 0, 0);
    wave_sync();
    u8 *sec = reinterpret_cast<u8 *>(aux_stage);
    (void)aux_less_than<LW>(sec, g, Xv, Nv, lane);   // assert_in_field = is_less_than(x, n), chip.rs:998-1006
    (void)sizeof(X);
    // ---- encoded-message check (src/chip.rs:136-198; LIMB_WIDTH = 64 only) ----------------------------
    // One flag per lane, in the reference's order: hash limbs 0-3, the two DigestInfo limbs, the low and high half of limb 6,
    // the 0xff.. limbs 7 .. L-2, the last limb -- L + 1 flags; the running AND (is_eq) is a ballot prefix.
    if constexpr (LW == 64) {
        if (a.hashed != nullptr) {
            u8 *e = reinterpret_cast<u8 *>(aux_stage + if_u4);
            if (OWN) __threadfence_block();   // the result / status stores of this wave have left it
            auto ld8 = [](const u8 *q) -> u8 { return OWN ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q; };
            const bool ok_status = a.status == nullptr || ld8(a.status + elem) == 0;
            const u64 *pwp = reinterpret_cast<const u64 *>(a.powed) + elem * L;
            struct PW { const u64 *p; __device__ u64 operator[](u32 i) const { return OWN ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i]; } };
            const PW pw{pwp};
            const PW hm{a.hashed + elem * 4};   // (OWN: possibly written by this launch's SHA role on another XCD)
            const u32 S = L + 1;
            bool all_prev = true;
            if (ok_status) {
                for (u32 base = 0; base < S; base += 64) {
                    const u32 sidx = base + lane;
                    const bool act = sidx < S;
                    bool flag = true; u32 pos_f = 0, pos_r = 0;
                    if (act) {
                        if (sidx < 4) { flag = pw[sidx] == hm[sidx]; pos_f = 2 * sidx; pos_r = 2 * sidx + 1; }                                  // :141-144
                        else if (sidx == 4) { flag = pw[4] == 217300885422736416ull; pos_f = 8; pos_r = 10; }                                 // :150-156
                        else if (sidx == 5) { flag = pw[5] == 938447882527703397ull; pos_f = 9; pos_r = 11; }
                        else if (sidx == 6) { flag = (u32)pw[6] == 3158320u; pos_f = 44; pos_r = 45; }                                        // :175-177
                        else if (sidx == 7) { flag = (u32)(pw[6] >> 32) == 4294967295u; pos_f = 46; pos_r = 47; }                             // :180-182
                        else if (sidx < L) { flag = pw[sidx - 1] == 18446744073709551615ull; pos_f = 48 + 2 * (sidx - 8); pos_r = pos_f + 1; }   // :185-188
                        else { flag = pw[L - 1] == 562949953421311ull; pos_f = 48 + 2 * (L - 8); pos_r = pos_f + 1; }                         // :191-197
                    }
                    const u64 bad = __ballot(act && !flag);
                    const bool run = all_prev && (bad & ((2ull << lane) - 1)) == 0;   // AND over flags <= sidx
                    if (act) { e[pos_f] = flag ? 1 : 0; e[pos_r] = run ? 1 : 0; }
                    all_prev = all_prev && bad == 0;
                }
                if (lane == 6) {   // limb 6 is split with two 32-bit range assigns and recomposed (:159-173)
                    const u32 low = (u32)pw[6], high = (u32)(pw[6] >> 32);
                    auto ra32 = [&](u8 *q, u32 v) { pst4(q, v); const u64 sb = limb_sub_bytes<32>(v); pst4(q + 4, (u32)sb); pst4(q + 8, (u32)(sb >> 32)); };
                    ra32(e + 12, low); ra32(e + 24, high);
                    pst4(e + 36, low); pst4(e + 40, high);
                }
            } else all_prev = false;
            if (a.is_valid && lane == 0) a.is_valid[elem] = all_prev ? 1 : 0;
        }
    }
    wave_sync();
    for (u32 k = lane; k < if_u4; k += 64) { const uint4 v = aux_stage[k]; st16(et + a.off_in_field + 16ull * k, ((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z); }
    if (LW == 64 && a.hashed != nullptr)
        for (u32 k = lane; k < em_u4; k += 64) { const uint4 v = aux_stage[if_u4 + k]; st16(et + a.off_em + 16ull * k, ((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z); }
}
// one wave per element; dynamic LDS: AuxGeom::in_field_sz() + em_sz() bytes
template <int LW>
__global__ __launch_bounds__(64) void aux_kernel(AuxArgs a) {
    extern __shared__ uint4 aux_stage_dyn[];
    aux_wave<LW>(a, blockIdx.x, threadIdx.x, aux_stage_dyn);
}

// ONE launch per pipeline step, three roles: workgroups [0, n_chain) run the chains of call k+1 (each walks elements b,
// b + n_chain, ...), the next n_rec write the records of call k, the last n_aux (one wave each) write call k's
// assert_in_field witness.  Consecutive steps then sit on ONE queue, 5 us apart, instead of on two queues with a barrier
// packet in front of every record kernel -- and the record role keeps the store rate the record kernel has ALONE: chain
// workgroups are dispatched first (lowest indices) and take their four slots per CU, record workgroups fill what is left
// (80 VGPRs => six 4-wave workgroups per CU) and every slot a finished chain frees.
// Measured (tools/fused_probe.py, 1,024 RSA-2048 signatures): 0.181 ms per step against 0.208-0.218 ms on two queues.
// The roles share one workgroup size; their LDS is overlaid.
// The workgroup size is the chain role's (64 * NW threads); the record role packs as many items into it as fit
// (trace_block<LW, L, 64 * NW>: RSA-2048 four items in 256 threads as ever; 128 x 32-bit limbs two 256-thread items in the
// 512 threads of an eight-wave chain workgroup; RSA-3072 four 96-thread items in the 384 threads of a six-wave one).
// SHA-256 of ragged messages + the hashed-message limbs (h2r_sha256.hpp; RSASignatureVerifier, reference src/lib.rs:205-239).  The
// arguments live here because the step launch can carry that work as one more role.
struct Sha256Args {
    const u8 *msgs; const u64 *off; u64 fixed_len;   // off == nullptr: message e = msgs[e * fixed_len, (e + 1) * fixed_len)
    u64 batch;
    u8 *digest;          // nullable, 32 bytes per element
    u64 *hashed;         // nullable, 4 limbs per element
    u8 *region; u64 region_stride;   // nullable
    u32 *done; u32 target;           // step launch only (nullable): messages hashed so far on this pipeline / the count that means "this launch's are done"
};
template <int NT> __device__ void sha256_role(const Sha256Args &a, u32 blk, u32 *w);   // h2r_sha256.hpp

template <int K, int NW, int LW, int L, bool WAVE = false>
union StepShared {
    ChainLds<K, NW> chain; TraceShared<LW, L, 64 * NW> trace; uint4 aux[sizeof(TraceShared<LW, L, 64 * NW>) / 16];
    ChainLds<(WAVE ? K : 2), 1> chainw[WAVE ? NW : 1];   // the one-wave chain form (K <= 32): every wave of the chain role's workgroup owns one
    __device__ StepShared() {}
};
// FOLD: the verifier's build -- the chain role also writes the verifier's in-field + encoded-message witness (va), and the launch may
// carry the SHA-256 role (sa).  A build of its own: with that code in it the kernel spills 41 registers instead of 2 (RSA-2048), so the
// launches of modpow_public_key calls keep the build without it, whose registers and scratch are what they were.
// WAVE (K <= 32): the chain role's workgroup is NW independent one-wave chains (chain_element<.., WAVE>) -- workgroup b of the role walks the
// elements (b * NW + wave) + k * n_chain * NW; the record role is unchanged.  The verifier's folded witness (va) is not built in this form.
template <int K, int NW, int LW, int L, bool FOLD, bool WAVE = false>
__global__ __launch_bounds__(64 * NW, H2R_CHAIN_MINB) void step_kernel(ChainArgs ca, TraceArgs ta, AuxArgs aa, AuxArgs va, Sha256Args sa, u32 n_sha, u32 n_chain, u32 n_rec) {
    static_assert((64 * NW) % TraceGeo<L>::TPI == 0, "the record role's items tile the chain role's workgroup");
    __shared__ StepShared<K, NW, LW, L, WAVE> sh;
    if (FOLD && blockIdx.x < n_sha) {
        // FIRST in dispatch order (n_sha is a multiple of 8, like n_chain): the verifier's SHA-256 of THIS call's messages, one thread
        // per message -- a long, latency-bound role (three compressions of 64 dependent rounds), so it has to start with the launch;
        // dispatched last it stretched the launch's tail by 40 us.  Its consumer (the encoded-message kernel) runs behind the launch.
        sha256_role<64 * NW>(sa, blockIdx.x, reinterpret_cast<u32 *>(&sh));
        return;
    }
    const u32 b = blockIdx.x - (FOLD ? n_sha : 0u);
    if constexpr (WAVE) {
        if (b < n_chain) {
            const int wv = threadIdx.x >> 6;
            for (u64 elem = (u64)b * NW + wv; elem < ca.batch; elem += (u64)n_chain * NW)
                chain_element<(WAVE ? K : 2), 1, false, false, true>(ca, sh.chainw[wv], elem);
            return;
        }
    }
    if (b < n_chain) {
        for (u64 elem = b; elem < ca.batch; elem += n_chain) {
            if (elem != b) __syncthreads();   // every wave is done with the previous element's LDS
            chain_element<K, NW, false, false>(ca, sh.chain, elem);
            if (FOLD && va.batch) {
                // the verifier's assert_in_field + encoded-message witness of THIS element (src/chip.rs:106, 136-198), by the wave that
                // has just stored its result and status: no kernel of its own behind the launch.  Wave 0 reads its own stores back
                // through the L2 (aux_wave: agent-scope loads of powed / status).
                __syncthreads();
                if (threadIdx.x < 64) {
                    bool ready = true;
                    if (sa.done) {
                        // the hashed limbs come from THIS launch's SHA role (other workgroups, other XCDs): wait for its count.  Those
                        // workgroups have the lowest indices of the launch, so they were dispatched before this one and wait for
                        // nothing -- the spin ends; it is bounded all the same (a chain takes >= 90 us, the hashing ~40).
                        u32 it = 0;
                        if (threadIdx.x == 0)
                            while ((int)(__hip_atomic_load(sa.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - sa.target) < 0 && ++it < (1u << 20)) __builtin_amdgcn_s_sleep(64);
                        ready = __shfl((int)(it < (1u << 20)), 0) != 0;
                        // acquire, once, and without the write-back a full fence would add (a thousand chain workgroups flushing
                        // their XCD's L2 next to the record role cost the launch 50 us); the limbs themselves are then read with
                        // agent-scope loads (aux_wave<.., OWN>)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    if (ready) aux_wave<LW, true>(va, elem, (int)threadIdx.x, reinterpret_cast<uint4 *>(&sh));
                    else if (threadIdx.x == 0) { ca.status[elem] = (u8)H2R_E_HIP; if (va.is_valid) va.is_valid[elem] = 0; }
                }
            }
        }
    } else if (b < n_chain + n_rec) {
        // (a record role of a few workgroups per CU that WALK the records was tried: inlined into a loop the body spills 25
        //  registers at this launch's 80, as a real call it runs at 4.1 TB/s -- one workgroup per four records it is)
        trace_block<LW, L, 64 * NW, false>(ta, b - n_chain, n_rec, sh.trace);
    } else if (b - n_chain - n_rec < aa.batch) {
        // last in dispatch order: these short workgroups fill the slots the record role's tail leaves (in front of the record
        // role they cost the step 3-5 us)
        if (threadIdx.x < 64) aux_wave<LW>(aa, b - n_chain - n_rec, (int)threadIdx.x, sh.aux);
    }
}


// The Fresh-integer family of BigIntInstructions as one batch op: one wave per element,
// flat stream written section by section like aux_kernel.
enum { FRESH_ADD = 0, FRESH_SUB, FRESH_ADD_MOD, FRESH_SUB_MOD, FRESH_IS_ZERO, FRESH_IS_EQUAL_FRESH, FRESH_IS_LESS_THAN,
       FRESH_IS_LESS_THAN_OR_EQUAL, FRESH_IS_GREATER_THAN, FRESH_IS_GREATER_THAN_OR_EQUAL, FRESH_IS_IN_FIELD, FRESH_OP_COUNT };

struct FreshArgs {
    const void *a, *b, *n; u64 n_stride, b_stride;   // b_stride = 0: one b for every element (the modulus of is_in_field)
    u64 batch; u32 L, op;
    u8 *trace; u64 elem_stride;
    void *value_out; u32 value_limbs;   // [elem][value_limbs] (nullable)
    u8 *flag_out;                       // [elem] (nullable)
    u8 *status;                         // [elem]
};

template <int LW>
__global__ __launch_bounds__(64) void fresh_kernel(FreshArgs f) {
    using X = AuxW<LW>;
    using limb_t = typename LimbT<LW>::type;
    const int lane = threadIdx.x;
    const u64 elem = blockIdx.x;
    const u32 L = f.L;
    const AuxGeom g(L, LW);
    u64 A[AUX_V], B[AUX_V], N[AUX_V], R1[AUX_V], R2[AUX_V], OUT[AUX_V];
#pragma unroll
    for (int m = 0; m < AUX_V; ++m) {
        const u32 p = lane + 64 * m;
        A[m] = p < L ? (u64) reinterpret_cast<const limb_t *>(f.a)[elem * L + p] : 0;
        B[m] = (f.b && p < L) ? (u64) reinterpret_cast<const limb_t *>(f.b)[elem * f.b_stride + p] : 0;
        N[m] = (f.n && p < L) ? (u64) reinterpret_cast<const limb_t *>(f.n)[elem * f.n_stride + p] : 0;
        OUT[m] = 0;
    }
    u8 *sec = f.trace + elem * f.elem_stride;
    int status = H2R_OK;
    int flag = 0;
    u32 nv = 0;
    switch (f.op) {  // block-uniform
        case FRESH_ADD: aux_add<LW>(sec, g, A, L, B, L, OUT, lane); nv = L + 1; break;                     // chip.rs:245-297
        case FRESH_SUB: flag = aux_sub<LW>(sec, g, A, L, B, L, OUT, lane); nv = L + 1; break;              // chip.rs:310-373
        case FRESH_ADD_MOD: {                                                                              // chip.rs:452-481
            aux_add<LW>(sec, g, A, L, B, L, R1, lane); sec += g.add_sz(L);                                 // added, :462
            const bool ov = aux_sub<LW>(sec, g, R1, L + 1, N, L, R2, lane);                                // :464
            bool bad = false;
#pragma unroll
            for (int m = 0; m < AUX_V; ++m) {                                                              // select(added, subed, ov), :469-474
                const u32 p = lane + 64 * m;
                const u64 v = p < L + 2 ? (ov ? (p < L + 1 ? R1[m] : 0) : R2[m]) : 0;
                if (p < L + 2) X::put_limb(sec + (u64)p * g.LB, v);
                OUT[m] = p < L ? v : 0;
                bad = bad || (p >= L && v != 0);                                                           // assert_zero, :475-478
            }
            if (__ballot(bad)) status = H2R_E_NOT_REDUCED;
            nv = L; break;
        }
        case FRESH_SUB_MOD: {                                                                              // chip.rs:495-528
            const bool ov1 = aux_sub<LW>(sec, g, A, L, B, L, R1, lane);                                    // subed1 (L+1 limbs), :506
            const bool ov2 = aux_sub<LW>(sec, g, N, L, R1, L + 1, R2, lane);                               // subed2 (L+2 limbs), :509
            if (ov2) status = H2R_E_NOT_IN_FIELD;                                                          // assert_zero(is_overflowed2), :510
            bool bad = false;
#pragma unroll
            for (int m = 0; m < AUX_V; ++m) {                                                              // select(subed2, subed1, ov1), :516-521
                const u32 p = lane + 64 * m;
                const u64 v = p < L + 2 ? (ov1 ? R2[m] : (p < L + 1 ? R1[m] : 0)) : 0;
                if (p < L + 2) X::put_limb(sec + (u64)p * g.LB, v);
                OUT[m] = p < L ? v : 0;
                bad = bad || (p >= L && v != 0);
            }
            if (status == H2R_OK && __ballot(bad)) status = H2R_E_NOT_REDUCED;
            nv = L; break;
        }
        case FRESH_IS_ZERO: {                                                                              // chip.rs:754-767
            u64 Z[AUX_V];
#pragma unroll
            for (int m = 0; m < AUX_V; ++m) Z[m] = 0;
            flag = aux_eq<LW>(sec, A, L, Z, L, lane);   // same (flag, running AND) pair sequence as is_equal against zero
            break;
        }
        case FRESH_IS_EQUAL_FRESH: flag = aux_eq<LW>(sec, A, L, B, L, lane); break;                        // chip.rs:780-805
        case FRESH_IS_LESS_THAN: case FRESH_IS_IN_FIELD: flag = aux_less_than<LW>(sec, g, A, B, lane); break;
        case FRESH_IS_LESS_THAN_OR_EQUAL: flag = aux_sub<LW>(sec, g, A, L, B, L, R1, lane); break;         // chip.rs:932-941
        case FRESH_IS_GREATER_THAN:                                                                        // chip.rs:954-963
            flag = !aux_sub<LW>(sec, g, A, L, B, L, R1, lane);
            if (lane == 0) sec[0] = (u8)flag;
            break;
        case FRESH_IS_GREATER_THAN_OR_EQUAL:                                                               // chip.rs:976-985
            flag = !aux_less_than<LW>(sec, g, A, B, lane);
            if (lane == 0) sec[0] = (u8)flag;
            break;
        default: status = H2R_E_UNSUPPORTED; break;
    }
    if (f.value_out && nv) {
#pragma unroll
        for (int m = 0; m < AUX_V; ++m) {
            const u32 p = lane + 64 * m;
            if (p < nv && p < f.value_limbs) reinterpret_cast<limb_t *>(f.value_out)[elem * f.value_limbs + p] = (limb_t)OUT[m];
        }
    }
    if (lane == 0) { if (f.flag_out) f.flag_out[elem] = (u8)flag; f.status[elem] = (u8)status; }
}

// (BigIntChip::refresh and the general-shape is_equal_muled live in h2r_muled.hpp)

// ================================================================================================
// K4: lookup multiplicities of the range-check sub-limbs of a set of records
// ================================================================================================
struct HistArgs {
    const u8 *trace; u64 first_record_off, elem_stride, record_stride, num_elems; u32 records_per_elem;
    u64 off_q_sub, off_r_sub, off_carry_sub;
    u32 L, C, carry_nsub, carry_sub_stride, carry_has_ov;
    u32 tab0_len, tab1_off, tab1_len, tab2_off, tab2_len, hist_len;  // table row offsets in the histogram
    u32 *hist;  // [elem][hist_len]
};

#ifdef H2R_TU_API   // a plain (non-template) kernel: defined in the one translation unit that launches it
__global__ __launch_bounds__(256) void hist_kernel(HistArgs a) {
    extern __shared__ u32 h[];
    const u64 elem = blockIdx.x;
    for (u32 k = threadIdx.x; k < a.hist_len; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const u8 *base = a.trace + elem * a.elem_stride + a.first_record_off;
    const u32 limb_bytes_per_rec = 2 * a.L * 8;
    const u32 ncomp = a.carry_nsub - a.carry_has_ov;
    for (u32 rcd = 0; rcd < a.records_per_elem; ++rcd) {
        const u8 *rec = base + (u64)rcd * a.record_stride;
        // limb sub-limbs: Q_SUB and R_SUB are adjacent planes of L*8 bytes each
        for (u32 k = threadIdx.x * 4; k < limb_bytes_per_rec; k += blockDim.x * 4) {
            const u8 *p = (k < a.L * 8) ? rec + a.off_q_sub + k : rec + a.off_r_sub + (k - a.L * 8);
            const u32 wv = *reinterpret_cast<const u32 *>(p);
            atomicAdd(&h[wv & 0xff], 1u); atomicAdd(&h[(wv >> 8) & 0xff], 1u);
            atomicAdd(&h[(wv >> 16) & 0xff], 1u); atomicAdd(&h[wv >> 24], 1u);
        }
        for (u32 k = threadIdx.x; k < (a.C - 1) * a.carry_nsub; k += blockDim.x) {
            const u32 cc = k / a.carry_nsub, j = k % a.carry_nsub;
            const u32 v = rec[a.off_carry_sub + (u64)cc * a.carry_sub_stride + j];
            if (j < ncomp) atomicAdd(&h[a.tab1_off + v], 1u); else atomicAdd(&h[a.tab2_off + v], 1u);
        }
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < a.hist_len; k += blockDim.x) a.hist[elem * a.hist_len + k] = h[k];
}
#endif

// Grouped ("sorted") arrangement of an element's lookup inputs: a stable counting sort of its sub-limb
// cells by lookup-table row.  Cell ids follow the flat stream: record t contributes, in order, the q
// sub-limbs, the r sub-limbs and the carry sub-limbs (cells_per_record = 16 L + (C-1) carry_nsub).
// perm[k] = id of the cell at sorted position k; rows[k] = its table row (the halo2 lookup argument's
// permuted input column A' is the row values in this order; the theta-compression is [3P]).
struct PermArgs {
    HistArgs h;
    u32 cells_per_record, n_cells;
    u32 *perm;     // [elem][n_cells]
    uint16_t *rows;  // nullable [elem][n_cells]
};
constexpr int PERM_MAX_ROWS = 512;   // table rows: at most 256 (8-bit) + 128 (7-bit overflow) on this path

constexpr int PERM_STAGE_U4 = 384;   // 16-byte units of lookup bytes per record (L = 128: 2 * 1024 + 254 * 16 bytes)
constexpr int PERM_STAGE_NR = PERM_STAGE_U4 / 64;

// One workgroup per element, four waves; wave w owns a contiguous range of the element's records, so a stable
// sort only needs per-wave row counts.  A record's lookup bytes (q / r sub-limb planes, carry sub-limb slots) are
// staged in the wave's LDS buffer with 16-byte loads -- the next record's loads are in flight while the current
// one is processed -- because byte-granular reads straight from HBM made the kernel latency bound
// (0.73 ms per 1024 RSA-2048 traces; tools/lookup_timing.py).
// STAGED (an element's cells fit 16-bit ids and LDS): positions are scattered into an LDS copy of the permutation and
// written out as full lines, rows[] is rebuilt from the row starts -- the scattered 4- and 2-byte global stores of
// the direct form (64 partial lines per store instruction) were the next bound after the key reads.
constexpr int PERM_G = 2;   // 64-cell groups a wave works on at once (their LDS round trips overlap)
__host__ __device__ constexpr u64 perm_same_bytes(u32 rows) { return 4ull * PERM_G * ((rows + 1) & ~1u) * 8; }

template <bool STAGED>
__global__ __launch_bounds__(256) void perm_kernel(PermArgs p) {
    __shared__ u32 cnt[4][PERM_MAX_ROWS];   // per-wave row counts, then per-wave running bases
    __shared__ u32 start[PERM_MAX_ROWS + 1];
    extern __shared__ uint4 dyn_lds[];      // [4][stage_u4] record staging, (STAGED) u16 perm_s[n_cells], u64 same[4][PERM_G][R2]
    uint4 *const stage_all = dyn_lds;
    const HistArgs &a = p.h;
    const u64 elem = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 R = a.hist_len, R2 = (R + 1) & ~1u;
    const u8 *base = a.trace + elem * a.elem_stride + a.first_record_off;
    const u32 T = a.records_per_elem, per = (T + 3) / 4;
    const u32 r0 = min(T, wave * per), r1 = min(T, r0 + per);
    const u32 nl = a.L * 8, nl4 = nl / 16, nc4 = (a.C - 1) * a.carry_sub_stride / 16, n_u4 = 2 * nl4 + nc4;
    const u32 cpr = p.cells_per_record, ncomp = a.carry_nsub - a.carry_has_ov;
    uint4 *const stage_w = stage_all + (u64)wave * n_u4;
    uint16_t *const perm_s = reinterpret_cast<uint16_t *>(stage_all + 4ull * n_u4);
    // per wave and group: the lanes of the group's 64 cells that hit each row
    u64 *const same_all = reinterpret_cast<u64 *>(reinterpret_cast<u8 *>(perm_s) + (STAGED ? ((2ull * p.n_cells + 15) & ~15ull) : 0));
    u64 *const same_w = same_all + (u64)wave * PERM_G * R2;
    for (u32 k = threadIdx.x; k < 4 * PERM_MAX_ROWS; k += 256) (&cnt[0][0])[k] = 0;
    for (u32 k = threadIdx.x; k < 4 * PERM_G * R2; k += 256) same_all[k] = 0;
    __syncthreads();
    uint4 regs[PERM_STAGE_NR];
    auto fetch = [&](u32 rcd) {   // 16-byte loads of record rcd's lookup bytes into registers
        const u8 *rec = base + (u64)rcd * a.record_stride;
#pragma unroll
        for (int k = 0; k < PERM_STAGE_NR; ++k) {
            const u32 idx = lane + 64 * k;
            if (idx < n_u4) {
                const u8 *src = idx < nl4 ? rec + a.off_q_sub + 16ull * idx
                              : idx < 2 * nl4 ? rec + a.off_r_sub + 16ull * (idx - nl4) : rec + a.off_carry_sub + 16ull * (idx - 2 * nl4);
                regs[k] = *reinterpret_cast<const uint4 *>(src);
            }
        }
    };
    auto publish = [&]() {
#pragma unroll
        for (int k = 0; k < PERM_STAGE_NR; ++k) { const u32 idx = lane + 64 * k; if (idx < n_u4) stage_w[idx] = regs[k]; }
    };
    const u8 *sb = reinterpret_cast<const u8 *>(stage_w);
    // table row of cell li of the staged record (cells in stream order: q sub-limbs, r sub-limbs, carry sub-limbs)
    auto key_of = [&](u32 li) -> u32 {
        if (li < 2 * nl) return sb[li];
        const u32 cj = li - 2 * nl, cc = cj / a.carry_nsub, j = cj - cc * a.carry_nsub;
        const u32 v = sb[2 * nl + cc * a.carry_sub_stride + j];
        return (j < ncomp) ? a.tab1_off + v : a.tab2_off + v;
    };
    u32 *perm = p.perm + elem * (u64)p.n_cells;
    uint16_t *rows = p.rows ? p.rows + elem * (u64)p.n_cells : nullptr;
    const u64 lane_bit = 1ull << lane, below = lane_bit - 1;
    for (int pass = 0; pass < 2; ++pass) {
        if (r0 < r1) fetch(r0);
        for (u32 rcd = r0; rcd < r1; ++rcd) {
            publish();
            wave_sync();
            if (rcd + 1 < r1) fetch(rcd + 1);
            // PERM_G groups of 64 cells per iteration: every LDS round trip below (key bytes, match masks, counts) is
            // issued for all of them before the first result is needed -- one group at a time the loop was a chain of
            // eight dependent LDS round trips per 64 cells (0.228 ms per 1,024 RSA-2048 traces)
            for (u32 lb = 0; lb < cpr; lb += 64 * PERM_G) {
                u32 key[PERM_G]; bool valid[PERM_G];
#pragma unroll
                for (int g = 0; g < PERM_G; ++g) {
                    const u32 li = lb + 64 * g + lane;
                    valid[g] = li < cpr;
                    key[g] = valid[g] ? key_of(li) : 0;
                }
                if (pass == 0) {
#pragma unroll
                    for (int g = 0; g < PERM_G; ++g) if (valid[g]) atomicAdd(&cnt[wave][key[g]], 1u);
                } else {
                    // Multi-split in cell order.  "Which lanes of group g hold my row" is one LDS atomic OR of the lane bit
                    // into same[g][row] -- a match-any in O(1) instead of one ballot per distinct row (with ~300 rows nearly
                    // every lane holds a different one); the rank is the row's count so far, plus the cells of the row in the
                    // earlier groups of this iteration, plus a popcount below the lane.
#pragma unroll
                    for (int g = 0; g < PERM_G; ++g)
                        if (valid[g]) atomicOr(reinterpret_cast<unsigned long long *>(&same_w[g * R2 + key[g]]), (unsigned long long)lane_bit);
                    wave_sync();
                    u64 mask[PERM_G]; u32 pos[PERM_G];
#pragma unroll
                    for (int g = 0; g < PERM_G; ++g) {
                        mask[g] = 0; pos[g] = 0;
                        if (valid[g]) {
                            mask[g] = same_w[g * R2 + key[g]];
                            u32 b = cnt[wave][key[g]];
#pragma unroll
                            for (int e = 0; e < g; ++e) b += (u32)__builtin_popcountll(same_w[e * R2 + key[g]]);
                            pos[g] = b + (u32)__builtin_popcountll(mask[g] & below);
                        }
                    }
                    wave_sync();
#pragma unroll
                    for (int g = 0; g < PERM_G; ++g) {
                        if (valid[g]) {
                            if ((mask[g] & below) == 0) {   // lowest lane of the row in this group
                                atomicAdd(&cnt[wave][key[g]], (u32)__builtin_popcountll(mask[g]));
                                same_w[g * R2 + key[g]] = 0;
                            }
                            const u32 id = rcd * cpr + lb + 64 * g + lane;
                            if constexpr (STAGED) perm_s[pos[g]] = (uint16_t)id;
                            else { perm[pos[g]] = id; if (rows) rows[pos[g]] = (uint16_t)key[g]; }
                        }
                    }
                    wave_sync();
                }
            }
            wave_sync();   // everyone is done with the staged record before the next publish
        }
        if (pass == 0) {
            __syncthreads();
            {   // exclusive scan over the (at most 512) table rows: two rows per thread, wave scan, four wave totals
                // (one thread walking the rows cost 11 us per element: ~340 dependent LDS round trips)
                __shared__ u32 wtot[4];
                const u32 ra = 2 * threadIdx.x, rb = ra + 1;
                const u32 ta = ra < R ? cnt[0][ra] + cnt[1][ra] + cnt[2][ra] + cnt[3][ra] : 0;
                const u32 tb = rb < R ? cnt[0][rb] + cnt[1][rb] + cnt[2][rb] + cnt[3][rb] : 0;
                u32 inc = ta + tb;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const u32 up = __shfl_up(inc, d); if (lane >= d) inc += up; }
                if (lane == 63) wtot[wave] = inc;
                __syncthreads();
                u32 off = 0;
                for (int w = 0; w < wave; ++w) off += wtot[w];
                const u32 ex = off + inc - (ta + tb);
                if (ra < R) start[ra] = ex;
                if (rb < R) start[rb] = ex + ta;
                if (threadIdx.x == 255) start[R] = off + inc;   // rows beyond R contribute 0: the grand total
            }
            __syncthreads();
            for (u32 r = threadIdx.x; r < R; r += 256) {  // per-wave bases keep the sort stable across the 4 record ranges
                u32 bb = start[r];
                for (int w = 0; w < 4; ++w) { const u32 t = cnt[w][r]; cnt[w][r] = bb; bb += t; }
            }
            __syncthreads();
        }
    }
    if (a.hist)   // multiplicities of the table rows: what the counting pass found (start[] is final since the row scan)
        for (u32 r = threadIdx.x; r < R; r += 256) a.hist[elem * R + r] = start[r + 1] - start[r];
    if constexpr (STAGED) {
        __syncthreads();
        // rows[k] = the row whose [start[r], start[r+1]) holds k.  A per-cell binary search over `start` (nine dependent LDS
        // reads) was half of the kernel; instead the row of every 64th position is tabulated once (in the match-mask
        // area, free now) and each cell walks on from there -- one or two steps.
        uint16_t *const rowmap = reinterpret_cast<uint16_t *>(same_all);
        if (rows) {
            for (u32 r = threadIdx.x; r < R; r += 256)
                for (u32 b = (start[r] + 63) / 64; 64 * b < start[r + 1]; ++b) rowmap[b] = (uint16_t)r;
            __syncthreads();
        }
        for (u32 k = threadIdx.x; k < p.n_cells; k += 256) {
            perm[k] = perm_s[k];
            if (rows) {
                u32 r = rowmap[k >> 6];
                while (start[r + 1] <= k) ++r;
                rows[k] = (uint16_t)r;
            }
        }
    }
}

// ================================================================================================
// K7: device-side flatten -- planes -> the reference's assignment order (the flat op-trace stream), in HBM
//   What h2r_trace_flatten does on a host copy of one record, for every record of a batch, at HBM speed: a layouter
//   shim (or a prover's advice-column builder) reads the values in the order the reference assigns them
//   (big_integer/chip.rs:588-599 q/r + sub-limbs, :400-412 accumulators column by column, :617 eq_b, :857-893 the
//   is_equal_muled steps) without walking the planes on the host.
//   One workgroup per record.  The record's stream is cut into segments of at most EMIT_SEG_CAP bytes (host-built
//   table: q/r block, column ranges of the two accumulator planes, eq_b + carry steps).  Per segment: every thread
//   reads source entries in SOURCE order (coalesced 16-byte loads from the planes) and scatters them into an LDS image
//   of the segment at their stream offsets; the image then leaves as 16-byte stores aligned to the OUTPUT address --
//   streams are byte-packed, so a segment starts at an arbitrary byte of the output and the LDS image is funnel-shifted
//   (v_alignbyte) on its way out.  Bound: HBM (reads one record, writes one stream: 2 x 64 KB per RSA-2048 mul_mod).
// ================================================================================================
constexpr u32 EMIT_SEG_CAP = 24 * 1024;   // bytes of stream staged in LDS at a time (6 workgroups per CU; one RSA-2048 accumulator plane)
constexpr int EMIT_MAX_SEGS = 40;
enum { EMIT_QR = 0, EMIT_ACC_AB = 1, EMIT_ACC_QN = 2, EMIT_EQB = 3, EMIT_STEPS = 4 };
struct EmitSeg { u32 kind, c0, c1, bytes; u64 off; };   // columns [c0, c1) (accumulator and step kinds); off/bytes within the record's stream

// entries of the reference's column order before column i (column c has min(c + 1, 2L - 1 - c) accumulators)
__host__ __device__ inline u32 emit_colstart(u32 i, u32 L) {
    if (i <= L) return i * (i + 1) / 2;
    return L * (L + 1) / 2 + (i - L) * (2 * L - 1) - ((i - 1) * i / 2 - (L - 1) * L / 2);
}

struct EmitArgs {
    const u8 *trace; u64 elem_stride, off_records, record_stride; u32 T; u64 n_elems;
    u8 *out; u64 out_stride, out_off;   // element e's stream starts at out + e * out_stride + out_off
    u64 rec_bytes;                      // stream bytes of one record (depends on field_ab)
    u32 var, nbits, limbs_bytes;        // pow_mod layout: e bits first, selected limbs between the records of a bit
    u64 off_e_bits, off_selected, selected_stride, off_result;
    u32 has_result;
    u64 off[H2R_PL_COUNT];
    u32 L, carry_nsub, carry_sub_stride;
    u32 field_ab;                       // a_b as a 32-byte canonical field element (p - |x| when negative, chip.rs:859)
    u64 p[4];                           // the field modulus
    u32 nseg; EmitSeg seg[EMIT_MAX_SEGS];
};

template <int LW>
__global__ __launch_bounds__(256) void emit_kernel(EmitArgs a) {
    constexpr u32 LB = LW / 8, WB = LW == 64 ? 24 : 16, CB = LW == 64 ? 16 : 8;
    extern __shared__ uint4 emit_lds4[];
    u8 *const lds = reinterpret_cast<u8 *>(emit_lds4);
    u32 *const lds32 = reinterpret_cast<u32 *>(emit_lds4);
    const u32 tid = threadIdx.x;
    const u32 Tn = a.T ? a.T : 1;
    const u32 bid = xcd_contiguous_block(blockIdx.x, gridDim.x);   // every XCD reads and writes a contiguous eighth
    const u64 elem = bid / Tn;
    const u32 t = (u32)(bid - elem * Tn);
    const u8 *et = a.trace + elem * a.elem_stride;
    u8 *eo = a.out + elem * a.out_stride + a.out_off;
    const u32 L = a.L, C = 2 * L - 1;
    // stream offset of record t within the element
    u64 roff;
    if (a.var) roff = (u64)a.nbits + (u64)(t >> 1) * (2 * a.rec_bytes + a.limbs_bytes) + (u64)(t & 1) * (a.rec_bytes + a.limbs_bytes);
    else roff = (u64)t * a.rec_bytes;
    // small extras (byte copies): e bits, the selected limbs that follow a bit's first record, the result limbs
    if (a.var && t == 0) for (u32 k = tid; k < a.nbits; k += 256) eo[k] = et[a.off_e_bits + k];
    if (a.var && a.T && (t & 1) == 0)
        for (u32 k = tid; k < a.limbs_bytes; k += 256) eo[roff + a.rec_bytes + k] = et[a.off_selected + (u64)(t >> 1) * a.selected_stride + k];
    if (a.has_result && t == Tn - 1) {
        const u64 res_off = a.var ? (u64)a.nbits + (u64)(a.T >> 1) * (2 * a.rec_bytes + a.limbs_bytes) : (u64)a.T * a.rec_bytes;
        for (u32 k = tid; k < a.limbs_bytes; k += 256) eo[res_off + k] = et[a.off_result + k];
    }
    if (a.T == 0) return;
    const u8 *rec = et + a.off_records + (u64)t * a.record_stride;
    const u32 ABB = a.field_ab ? 32u : WB;                                    // bytes of a_b in the stream
    const u32 per_col = ABB + 4 * WB + 2 * CB + 4 * LB + 4;                   // one is_equal_muled step without the range assign
    const u32 per_col_ra = per_col + CB + a.carry_nsub;                       // ... with it (every column but the last)
    auto put_bytes = [&](u32 pos, u64 v, u32 n) { for (u32 k = 0; k < n; ++k) lds[pos + k] = (u8)(v >> (8 * k)); };
    for (u32 sg = 0; sg < a.nseg; ++sg) {
        const EmitSeg &S = a.seg[sg];
        // ---- scatter: source order in, stream order in LDS -----------------------------------------------------
        if (S.kind == EMIT_QR) {   // T1/T2: limb + its 8 sub-limb bytes, q then r (chip.rs:588-599)
            for (u32 k = tid; k < 2 * L; k += 256) {
                const u32 which = k >= L, kk = which ? k - L : k;
                const u8 *pl = rec + a.off[which ? H2R_PL_R : H2R_PL_Q] + (u64)kk * LB;
                const u64 sub = *reinterpret_cast<const u64 *>(rec + a.off[which ? H2R_PL_R_SUB : H2R_PL_Q_SUB] + (u64)kk * 8);
                const u32 pos = k * (LB + 8);
                if constexpr (LW == 64) { *reinterpret_cast<u64 *>(lds + pos) = *reinterpret_cast<const u64 *>(pl); *reinterpret_cast<u64 *>(lds + pos + 8) = sub; }
                else { lds32[pos / 4] = *reinterpret_cast<const u32 *>(pl); lds32[pos / 4 + 1] = (u32)sub; lds32[pos / 4 + 2] = (u32)(sub >> 32); }
            }
        } else if (S.kind == EMIT_ACC_AB || S.kind == EMIT_ACC_QN) {   // T3/T4: column i ascending, then j ascending (chip.rs:400-412)
            const bool qn = S.kind == EMIT_ACC_QN;
            const u8 *plo = rec + a.off[qn ? H2R_PL_QN_LO : H2R_PL_AB_LO];
            const u8 *phi = rec + a.off[qn ? H2R_PL_QN_HI : H2R_PL_AB_HI];
            const u32 e0 = emit_colstart(S.c0, L);
            // source rows j that hold accumulators of columns [c0, c1), and the rectangle (j, i) that covers them; a
            // whole plane is walked as its L x L source entries instead (no holes)
            const bool whole = S.c0 == 0 && S.c1 == C;
            const u32 jlo = S.c0 >= L ? S.c0 - L + 1 : 0, jhi = S.c1 - 1 < L - 1 ? S.c1 - 1 : L - 1, cw = S.c1 - S.c0;
            const u32 n_idx = whole ? L * L : (jhi - jlo + 1) * cw;
            for (u32 sidx = tid; sidx < n_idx; sidx += 256) {
                u32 j, i, im;
                if (whole) { j = sidx / L; im = sidx - j * L; i = im >= j ? im : im + L; }   // column of entry (j, i % L == im)
                else {
                    j = jlo + sidx / cw; i = S.c0 + sidx % cw;
                    if (i < j || i > j + L - 1) continue;       // a[j] * b[i - j] exists for 0 <= i - j < L
                    im = i >= L ? i - L : i;
                }
                const u32 jmin = i >= L ? i - L + 1 : 0;
                const u32 pos = (emit_colstart(i, L) - e0 + (j - jmin)) * WB;
                if constexpr (LW == 64) {   // interleaved rows: two steps per group, shared HI row (h2r_layout)
                    const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(plo + (u64)(j >> 1) * (3ull * 2 * L * 16) + (u64)(j & 1) * (2ull * L * 16) + (u64)im * 16);
                    const u64 hi = *reinterpret_cast<const u64 *>(phi + (u64)(j >> 1) * (3ull * 2 * L * 16) + (u64)im * 16 + (j & 1) * 8);
                    u64 *d = reinterpret_cast<u64 *>(lds + pos);
                    d[0] = lo.x; d[1] = lo.y; d[2] = hi;
                } else {
                    const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(plo + (u64)j * ((u64)L * 16) + (u64)im * 16);
                    u64 *d = reinterpret_cast<u64 *>(lds + pos);
                    d[0] = lo.x; d[1] = lo.y;
                }
            }
        } else if (S.kind == EMIT_EQB) {   // T5 eq_b (chip.rs:617)
            for (u32 i = tid; i < L; i += 256) {
                const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(rec + a.off[H2R_PL_EQB_LO] + (u64)i * 16);
                u64 *d = reinterpret_cast<u64 *>(lds + i * WB);
                d[0] = lo.x; d[1] = lo.y;
                if constexpr (LW == 64) d[2] = *reinterpret_cast<const u64 *>(rec + a.off[H2R_PL_EQB_HI] + (u64)i * 8);
            }
        } else {   // T6: the is_equal_muled steps of columns [c0, c1) (chip.rs:857-893); every column but C-1 has the range assign
            for (u32 c = S.c0 + tid; c < S.c1; c += 256) {
                u32 pos = (c - S.c0) * per_col_ra;
                auto wide = [&](int pl_lo, u32 nb) {   // WIDE value: 16-byte LO entry + (64-bit limbs) 8-byte HI entry
                    const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(rec + a.off[pl_lo] + (u64)c * 16);
                    put_bytes(pos, lo.x, 8); put_bytes(pos + 8, lo.y, 8);
                    if constexpr (LW == 64) put_bytes(pos + 16, *reinterpret_cast<const u64 *>(rec + a.off[pl_lo + 1] + (u64)c * 8), 8);
                    pos += nb;
                };
                auto limb = [&](int pl) {
                    if constexpr (LW == 64) put_bytes(pos, *reinterpret_cast<const u64 *>(rec + a.off[pl] + (u64)c * 8), 8);
                    else put_bytes(pos, *reinterpret_cast<const u32 *>(rec + a.off[pl] + (u64)c * 4), 4);
                    pos += LB;
                };
                auto carry = [&](int pl) {
                    put_bytes(pos, *reinterpret_cast<const u64 *>(rec + a.off[pl] + (u64)c * CB), 8);
                    if constexpr (LW == 64) put_bytes(pos + 8, *reinterpret_cast<const u64 *>(rec + a.off[pl] + (u64)c * CB + 8), 8);
                    pos += CB;
                };
                if (a.field_ab) {   // a_b as the canonical element of F: x >= 0 -> x, x < 0 -> p - |x| = p + x (mod 2^256)
                    const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(rec + a.off[H2R_PL_AMB_LO] + (u64)c * 16);
                    u64 x[4] = {lo.x, lo.y, 0, 0};
                    bool neg;
                    if constexpr (LW == 64) { x[2] = *reinterpret_cast<const u64 *>(rec + a.off[H2R_PL_AMB_HI] + (u64)c * 8); neg = (x[2] >> 63) != 0; x[3] = neg ? ~0ull : 0; }
                    else { neg = (x[1] >> 63) != 0; x[2] = x[3] = neg ? ~0ull : 0; }
                    if (neg) {
                        u64 cy = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const u64 s1 = x[k] + a.p[k]; const u64 c1 = s1 < x[k]; const u64 s2 = s1 + cy; cy = c1 | (u64)(s2 < s1); x[k] = s2; }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) put_bytes(pos + 8 * k, x[k], 8);
                    pos += 32;
                } else wide(H2R_PL_AMB_LO, WB);
                wide(H2R_PL_SUM_LO, WB); carry(H2R_PL_CARRY); limb(H2R_PL_CMOD); wide(H2R_PL_NQ1_LO, WB); limb(H2R_PL_AMNQ1);
                wide(H2R_PL_ACCX_LO, WB); carry(H2R_PL_QACC); limb(H2R_PL_MODACC); wide(H2R_PL_NQ2_LO, WB); limb(H2R_PL_AMNQ2);
                const u32 fl = *reinterpret_cast<const u32 *>(rec + a.off[H2R_PL_FLAGS] + (u64)c * 4);
                put_bytes(pos, fl & 0xffffu, 2); pos += 2;
                if (c < C - 1) {
                    carry(H2R_PL_CARRY_DUP);
                    const u8 *sb = rec + a.off[H2R_PL_CARRY_SUB] + (u64)c * a.carry_sub_stride;
                    const ulonglong2 sv = *reinterpret_cast<const ulonglong2 *>(sb);
                    for (u32 k = 0; k < a.carry_nsub; ++k) lds[pos + k] = (u8)((k < 8 ? sv.x : sv.y) >> (8 * (k & 7)));
                    pos += a.carry_nsub;
                }
                put_bytes(pos, fl >> 16, 2);
            }
        }
        __syncthreads();
        // ---- stream out: 16-byte stores aligned to the output address; the LDS image is shifted by the misalignment --
        u8 *g = eo + roff + S.off;                           // output address of image byte 0
        const u32 mis = (u32)(reinterpret_cast<u64>(g) & 15u);
        const u32 head = mis ? (16u - mis < S.bytes ? 16u - mis : S.bytes) : 0u;   // bytes before the first aligned unit
        if (tid < head) g[tid] = lds[tid];
        const u32 body = (S.bytes - head) / 16;              // whole aligned 16-byte units
        const u32 sh = head & 3u;                            // byte phase of the image relative to 32-bit LDS words
        for (u32 u = tid; u < body; u += 256) {
            const u32 o = head + 16 * u;                     // image offset of this unit
            const u32 w0 = o >> 2;
            u32 d[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) d[k] = lds32[w0 + k];
            u32 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = (u32)(((((u64)d[k + 1]) << 32) | d[k]) >> (8 * sh));
            st16(g + o, ((u64)r[1] << 32) | r[0], ((u64)r[3] << 32) | r[2]);
        }
        const u32 done = head + 16 * body;
        if (tid < S.bytes - done) g[done + tid] = lds[done + tid];
        __syncthreads();   // the image is reused by the next segment
    }
}

// ================================================================================================
// K8: in-place witness checker (test / audit instrument, not on the product path)
//   Verifies SURVEY Appendix C invariants 1-4 on EVERY record of a batch where it lies in HBM -- traces of 10-50 GB
//   cannot be walked on the host.  Deliberately NOT the producer's algorithm: every relation is checked pointwise from
//   the stored values (each accumulator against its predecessor plus one product; each carry step against the stored
//   previous carry), so nothing here shares the record kernel's scans, ballots or index maps beyond the documented
//   plane layout.  One workgroup per record; bad[elem] counts violated relations, first_bad[elem] keeps one (t, code).
//     code 1  q/r limb vs the operands buffer, sub-limb recomposition (chip.rs:588-599)
//          2  accumulator chain: acc(j, i) = acc(j-1, i) + a[j] * b[i-j], first of a column = the product (chip.rs:400-412)
//          3  eq_b[i] = qn[i] + r[i] (chip.rs:617)
//          4  a_b = ab[i] - eq_b[i] (chip.rs:859)         5  sum = a_b + carry[i] + word_max (chip.rs:860-861)
//          6  div_mod of sum: carry, c, nq, a - nq (chip.rs:864, 1323-1349)
//          7  accumulated_extra chain and its div_mod (chip.rs:869-875)
//          8  flags: cs_acc_eq, range_eq / final_carry_eq and the running eq_bit (chip.rs:873-892)
//          9  range-assigned carry: duplicate, sub-limbs, width (chip.rs:877-885)
//         10  r < n (chip.rs:567)                          11  final eq_bit != 1 (assert_equal_muled, chip.rs:1062)
// ================================================================================================
struct CheckArgs {
    const void *opA, *opB, *opQ, *opR; u64 op_stride;   // limbs of item k at [k * op_stride, ...): the chain kernel's operands buffer
    const void *n; u64 n_stride;
    const u8 *status;                                   // [elem]; nonzero => element skipped
    const u8 *trace; u64 elem_stride, off_records, record_stride; u32 T; u64 n_items;
    u64 off[H2R_PL_COUNT]; u64 wm[3];
    u32 L, carry_bits, carry_sub_bits, carry_nsub, carry_sub_stride;
    u32 *bad; u32 *first_bad;
};

struct U192 {
    u64 w[3];
    __device__ __forceinline__ static U192 make(u64 a, u64 b, u64 c) { U192 r; r.w[0] = a; r.w[1] = b; r.w[2] = c; return r; }
    __device__ __forceinline__ U192 operator+(const U192 &o) const {
        U192 r; const u64 s0 = w[0] + o.w[0]; const u64 c0 = s0 < w[0];
        const u64 s1 = w[1] + o.w[1]; const u64 c1 = s1 < w[1]; const u64 s1b = s1 + c0; const u64 c1b = s1b < s1;
        r.w[0] = s0; r.w[1] = s1b; r.w[2] = w[2] + o.w[2] + (c1 | c1b); return r;
    }
    __device__ __forceinline__ U192 operator-(const U192 &o) const {
        U192 r; const u64 d0 = w[0] - o.w[0]; const u64 b0 = w[0] < o.w[0];
        const u64 d1 = w[1] - o.w[1]; const u64 b1 = w[1] < o.w[1]; const u64 d1b = d1 - b0; const u64 b1b = d1 < b0;
        r.w[0] = d0; r.w[1] = d1b; r.w[2] = w[2] - o.w[2] - (b1 | b1b); return r;
    }
    __device__ __forceinline__ bool operator==(const U192 &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2]; }
    __device__ __forceinline__ U192 shr(u32 s) const {   // s = 32 or 64
        if (s == 64) return make(w[1], w[2], 0);
        return make((w[0] >> 32) | (w[1] << 32), (w[1] >> 32) | (w[2] << 32), w[2] >> 32);
    }
    __device__ __forceinline__ U192 shl(u32 s) const {
        if (s == 64) return make(0, w[0], w[1]);
        return make(w[0] << 32, (w[0] >> 32) | (w[1] << 32), (w[1] >> 32) | (w[2] << 32));
    }
};

template <int LW>
__global__ __launch_bounds__(256) void check_kernel(CheckArgs a) {
    using limb_t = typename LimbT<LW>::type;
    constexpr u32 CB = LW == 64 ? 16 : 8;
    constexpr u64 LMASK = LW == 64 ? ~0ull : 0xffffffffull;
    __shared__ u32 s_bad, s_code;
    __shared__ u64 sa[128], sb_[128], sq[128], sn[128], sr[128];
    const u32 tid = threadIdx.x;
    const u32 item = blockIdx.x;
    const u32 elem = item / a.T, t = item - elem * a.T;
    if (a.status && a.status[elem]) return;
    if (tid == 0) { s_bad = 0; s_code = 0; }
    const u32 L = a.L, C = 2 * L - 1;
    const u8 *rec = a.trace + (u64)elem * a.elem_stride + a.off_records + (u64)t * a.record_stride;
    u32 nbad = 0, code = 0;
    auto fail = [&](u32 c) { ++nbad; if (!code) code = c; };
    // sign-extending reader of a WIDE value stored as 16-byte LO (+ 8-byte HI for 64-bit limbs)
    auto rd_wide = [&](int pl_lo, u32 idx) -> U192 {
        const u64 *lo = reinterpret_cast<const u64 *>(rec + a.off[pl_lo] + (u64)idx * 16);
        if constexpr (LW == 64) return U192::make(lo[0], lo[1], *reinterpret_cast<const u64 *>(rec + a.off[pl_lo + 1] + (u64)idx * 8));
        else return U192::make(lo[0], lo[1], (u64)((i64)lo[1] >> 63));
    };
    auto rd_limb = [&](int pl, u32 idx) -> u64 {
        if constexpr (LW == 64) return *reinterpret_cast<const u64 *>(rec + a.off[pl] + (u64)idx * 8);
        else return *reinterpret_cast<const u32 *>(rec + a.off[pl] + (u64)idx * 4);
    };
    auto rd_carry = [&](int pl, u32 idx) -> U192 {
        const u64 *p = reinterpret_cast<const u64 *>(rec + a.off[pl] + (u64)idx * CB);
        if constexpr (LW == 64) return U192::make(p[0], p[1], 0); else return U192::make(p[0], 0, 0);
    };
    // accumulator entry (j, i % L) per the documented addressing (include/h2r.h)
    auto rd_acc = [&](bool qn, u32 j, u32 im) -> U192 {
        if constexpr (LW == 64) {
            const u64 *lo = reinterpret_cast<const u64 *>(rec + a.off[qn ? H2R_PL_QN_LO : H2R_PL_AB_LO] + (u64)(j >> 1) * (3ull * 2 * L * 16) + (u64)(j & 1) * (2ull * L * 16) + (u64)im * 16);
            const u64 hi = *reinterpret_cast<const u64 *>(rec + a.off[qn ? H2R_PL_QN_HI : H2R_PL_AB_HI] + (u64)(j >> 1) * (3ull * 2 * L * 16) + (u64)im * 16 + (j & 1) * 8);
            return U192::make(lo[0], lo[1], hi);
        } else {
            const u64 *lo = reinterpret_cast<const u64 *>(rec + a.off[qn ? H2R_PL_QN_LO : H2R_PL_AB_LO] + (u64)j * ((u64)L * 16) + (u64)im * 16);
            return U192::make(lo[0], lo[1], 0);
        }
    };
    // ---- operands; q, r and their sub-limbs ------------------------------------------------------------------
    for (u32 k = tid; k < L; k += 256) {
        const u64 ib = (u64)item * a.op_stride + k;
        sa[k] = reinterpret_cast<const limb_t *>(a.opA)[ib]; sb_[k] = reinterpret_cast<const limb_t *>(a.opB)[ib];
        sn[k] = reinterpret_cast<const limb_t *>(a.n)[(u64)elem * a.n_stride + k];
        const u64 q = rd_limb(H2R_PL_Q, k), r = rd_limb(H2R_PL_R, k);
        sq[k] = q; sr[k] = r;
        if (a.opQ && (q != (u64) reinterpret_cast<const limb_t *>(a.opQ)[ib] || r != (u64) reinterpret_cast<const limb_t *>(a.opR)[ib])) fail(1);
        for (int which = 0; which < 2; ++which) {   // RangeChip::assign(v, w/8, w): 8 sub-limbs of w/8 bits
            const u64 sub = *reinterpret_cast<const u64 *>(rec + a.off[which ? H2R_PL_R_SUB : H2R_PL_Q_SUB] + (u64)k * 8);
            u64 v = 0; bool wide = false;
            for (u32 s8 = 0; s8 < 8; ++s8) { const u64 sv = (sub >> (8 * s8)) & 0xff; wide = wide || (sv >> (LW / 8)) != 0; v |= sv << (s8 * (LW / 8)); }
            if (wide || v != (which ? r : q)) fail(1);
        }
    }
    __syncthreads();
    // r < n (most significant differing limb)
    if (tid == 0) {
        bool lt = false;
        for (int k = (int)L - 1; k >= 0; --k) if (sr[k] != sn[k]) { lt = sr[k] < sn[k]; break; }
        if (!lt) fail(10);
    }
    // ---- accumulator chains of both products ------------------------------------------------------------------
    for (u32 sidx = tid; sidx < 2 * L * L; sidx += 256) {
        const bool qn = sidx >= L * L;
        const u32 e = qn ? sidx - L * L : sidx;
        const u32 j = e / L, im = e - j * L;
        const u32 i = im >= j ? im : im + L;
        const u32 jmin = i >= L ? i - L + 1 : 0;
        const u64 x = qn ? sq[j] : sa[j], y = qn ? sn[i - j] : sb_[i - j];
        const U192 prod = U192::make(x * y, __umul64hi(x, y), 0);
        const U192 cur = rd_acc(qn, j, im);
        const U192 want = j == jmin ? prod : rd_acc(qn, j - 1, im) + prod;
        if (!(cur == want)) fail(2);
    }
    // ---- eq_b and the is_equal_muled steps, thread = column ---------------------------------------------------
    const U192 W = U192::make(a.wm[0], a.wm[1], a.wm[2]);
    for (u32 c = tid; c < C; c += 256) {
        const u32 jmax = c < L ? c : L - 1, im = c < L ? c : c - L;
        const U192 ab = rd_acc(false, jmax, im), qnv = rd_acc(true, jmax, im);
        U192 eqb = qnv;
        if (c < L) {
            eqb = rd_wide(H2R_PL_EQB_LO, c);
            if (!(eqb == qnv + U192::make(sr[c], 0, 0))) fail(3);
        }
        const U192 a_b = rd_wide(H2R_PL_AMB_LO, c);
        if (!(a_b == ab - eqb)) fail(4);
        const U192 cprev = c ? rd_carry(H2R_PL_CARRY, c - 1) : U192::make(0, 0, 0);
        const U192 sum = rd_wide(H2R_PL_SUM_LO, c);
        if (!(sum == a_b + cprev + W)) fail(5);
        const U192 cy = rd_carry(H2R_PL_CARRY, c);
        const u64 cmod = rd_limb(H2R_PL_CMOD, c);
        const U192 nq1 = rd_wide(H2R_PL_NQ1_LO, c);
        if (!(cy == sum.shr(LW)) || cmod != (sum.w[0] & LMASK) || !(nq1 == cy.shl(LW)) || rd_limb(H2R_PL_AMNQ1, c) != cmod) fail(6);
        const U192 xprev = c ? rd_carry(H2R_PL_QACC, c - 1) : U192::make(0, 0, 0);
        const U192 accx = rd_wide(H2R_PL_ACCX_LO, c), qacc = rd_carry(H2R_PL_QACC, c), nq2 = rd_wide(H2R_PL_NQ2_LO, c);
        const u64 modacc = rd_limb(H2R_PL_MODACC, c);
        if (!(accx == xprev + W) || !(qacc == accx.shr(LW)) || modacc != (accx.w[0] & LMASK) || !(nq2 == qacc.shl(LW)) ||
            rd_limb(H2R_PL_AMNQ2, c) != modacc) fail(7);
        const u32 fl = *reinterpret_cast<const u32 *>(rec + a.off[H2R_PL_FLAGS] + (u64)c * 4);
        const u32 f1 = fl & 0xff, e1 = (fl >> 8) & 0xff, f2 = (fl >> 16) & 0xff, e2 = fl >> 24;
        const u32 eprev = c ? (*reinterpret_cast<const u32 *>(rec + a.off[H2R_PL_FLAGS] + (u64)(c - 1) * 4) >> 24) : 1u;
        const u32 want_f2 = c < C - 1 ? 1u : (cy == qacc ? 1u : 0u);   // range_eq is 1 by construction; last: carry == acc_extra
        if (f1 != (cmod == modacc ? 1u : 0u) || f2 != want_f2 || e1 != (eprev & f1) || e2 != (e1 & f2)) fail(8);
        if (c == C - 1 && e2 != 1) fail(11);
        if (c < C - 1) {   // RangeChip::assign(carry, sublimb_bit_len(carry_bits), carry_bits)
            const U192 dup = rd_carry(H2R_PL_CARRY_DUP, c);
            const u8 *sbp = rec + a.off[H2R_PL_CARRY_SUB] + (u64)c * a.carry_sub_stride;
            u64 v0 = 0, v1 = 0; bool wide = false;
            const u32 ovb = a.carry_bits % a.carry_sub_bits;
            for (u32 k = 0; k < a.carry_nsub; ++k) {
                const u64 sv = sbp[k];
                const u32 width = (ovb && k == a.carry_nsub - 1) ? ovb : a.carry_sub_bits;
                wide = wide || (sv >> width) != 0;
                const u32 sh = k * a.carry_sub_bits;
                if (sh < 64) { v0 |= sv << sh; if (sh && sh + 8 > 64) v1 |= sv >> (64 - sh); } else v1 |= sv << (sh - 64);
            }
            if (wide || !(dup == cy) || v0 != cy.w[0] || v1 != cy.w[1] || cy.w[2] != 0) fail(9);
        }
    }
    if (nbad) { atomicAdd(&s_bad, nbad); atomicMax(&s_code, code); }
    __syncthreads();
    if (tid == 0 && s_bad) {
        atomicAdd(&a.bad[elem], s_bad);
        if (a.first_bad) atomicCAS(&a.first_bad[elem], 0u, (t << 8) | s_code);
    }
}

// Chain linkage of a pow trace, one wave per element: the operands of every mul_mod are the
// results the reference's control flow feeds it (pow_mod_fixed_exp chip.rs:729-740; pow_mod :682-694), the first base
// is x, the result limbs are the final acc.  bad[elem] += violated links.
struct LinkArgs {
    const void *x; const void *ops; u64 op_stride; u32 L, T, var, nbits;   // ops: [item][a | b | q | r] limbs
    ExpBits e;                                                              // fixed exponent
    const u8 *status; const u8 *trace; u64 elem_stride, off_e_bits, off_selected, selected_stride, off_result;
    u32 *bad; u32 *first_bad;
};
template <int LW>
__global__ __launch_bounds__(64) void link_kernel(LinkArgs a) {
    using limb_t = typename LimbT<LW>::type;
    const u32 lane = threadIdx.x, elem = blockIdx.x;
    if (a.status && a.status[elem]) return;
    const u32 L = a.L;
    const limb_t *ops = reinterpret_cast<const limb_t *>(a.ops) + (u64)elem * a.T * a.op_stride;
    const u8 *et = a.trace + (u64)elem * a.elem_stride;
    u32 nbad = 0, where = 0;
    // compare two L-limb integers held in memory / registers, two limbs per lane (L <= 128)
    auto ld = [&](const limb_t *p, u64 (&v)[2]) { for (int m = 0; m < 2; ++m) { const u32 k = lane + 64 * m; v[m] = k < L ? (u64)p[k] : 0; } };
    auto same = [&](const u64 (&u)[2], const u64 (&v)[2]) { return __ballot(u[0] != v[0] || u[1] != v[1]) == 0; };
    u64 cur[2], acc[2], va[2], vb[2], vr[2];
    ld(reinterpret_cast<const limb_t *>(a.x) + (u64)elem * L, cur);
    acc[0] = lane == 0 ? 1 : 0; acc[1] = 0;
    u32 t = 0;
    auto item = [&](u32 tt, int which) { return ops + (u64)tt * a.op_stride + (u64)which * L; };
    auto expect = [&](u32 tt, const u64 (&ea)[2], const u64 (&eb)[2]) {
        ld(item(tt, 0), va); ld(item(tt, 1), vb);
        if (!same(va, ea) || !same(vb, eb)) { ++nbad; if (!where) where = (tt << 8) | 20; }
        ld(item(tt, 3), vr);
    };
    for (u32 bi = 0; bi < a.nbits && t < a.T; ++bi) {
        if (a.var) {
            const u32 bit = et[a.off_e_bits + bi];
            if (bit > 1) { ++nbad; if (!where) where = (t << 8) | 21; }
            expect(t, acc, cur); ++t;                                      // muled = mul_mod(acc, squared)  (:686)
            if (bit) { acc[0] = vr[0]; acc[1] = vr[1]; }                   // select (:688-691)
            u64 sel[2]; ld(reinterpret_cast<const limb_t *>(et + a.off_selected + (u64)bi * a.selected_stride), sel);
            if (!same(sel, acc)) { ++nbad; if (!where) where = (t << 8) | 22; }
            if (t >= a.T) break;
            expect(t, cur, cur); ++t;                                      // squared = square_mod(squared)  (:693)
            cur[0] = vr[0]; cur[1] = vr[1];
        } else {
            const u32 bit = (a.e.words[bi >> 5] >> (bi & 31)) & 1u;
            expect(t, cur, cur); ++t;                                      // squared = square_mod(cur_sq)   (:734)
            const u64 s0 = vr[0], s1 = vr[1];
            if (bit) {
                if (t >= a.T) break;
                expect(t, acc, cur); ++t;                                  // acc = mul_mod(acc, cur_sq)     (:739)
                acc[0] = vr[0]; acc[1] = vr[1];
            }
            cur[0] = s0; cur[1] = s1;
        }
    }
    if (t != a.T) { ++nbad; if (!where) where = (t << 8) | 23; }
    u64 res[2]; ld(reinterpret_cast<const limb_t *>(et + a.off_result), res);
    if (!same(res, acc)) { ++nbad; if (!where) where = (a.T << 8) | 24; }
    if (lane == 0 && nbad) { atomicAdd(&a.bad[elem], nbad); if (a.first_bad) atomicCAS(&a.first_bad[elem], 0u, where); }
}

// ================================================================================================
// K9: advice-column image -- the witness as rows of the main gate's five advice columns
//   The reference's prover consumes ADVICE COLUMNS of field elements: every main-gate op is a row of five cells
//   (maingate's columns a..e), every RangeChip::assign a run of rows holding four sub-limbs and what remains to be composed.
//   This kernel writes that image for every mul_mod record of a batch, in HBM, in the reference's op order, as canonical
//   32-byte little-endian elements of the ctx's field -- EVERY cell the ops assign: the flat stream's values, the constants
//   (assign_constant), the literal bits (assign_bit(1)) and main_gate.is_zero's internal witnesses (difference, its inverse).
//   The third-party row shapes are NOT in /root/reference (maingate / halo2wrong, rev 63bde545): they are restated in
//   DESIGN.md section 2b -- the VALUES are pinned by the flat-stream parity, the PLACEMENT is unpinned but self-consistent:
//   every row satisfies the main-gate equation with the fixed row h2r_advice_fixed_row gives for its kind (tests).
//   Rows of one mul_mod (nr = 2 rows per range-assigned limb, nrc = ceil(carry_nsub / 4) per carry), kinds H2R_ROW_*:
//     q limbs, r limbs     2L x nr   RANGE_LIMB  [four sub-limbs (last row reversed), remaining]      (chip.rs:588-599)
//     mul(a,b), mul(q,n)   per column i: CONST0 [0], then MUL_ADD [x_j, y_{i-j}, acc_prev, acc]       (chip.rs:400-412)
//     eq_b                 L         ADD [qn_i, r_i, eq_b_i]                                          (chip.rs:617)
//     is_equal_muled       CONST_B [2^w], CONST0, CONST0, BIT [1,1,1]                                  (chip.rs:851-856)
//     per column i (chip.rs:857-893), 23 rows + the carry's range assign:
//        0 SUB [ab_i, eqb_i, a_b]  1 ADD_WM [a_b, carry_i, sum]  2 VALUE [q]  3 VALUE [r]  4 MUL [2^w, q, nq]  5 SUB [sum, nq, sum-nq]
//        6 ASSERT_EQ [r, sum-nq]   7 ADDC_WM [acc_extra, acc_extra+W]   8..12 the same div_mod of it
//        13..16 is_equal(c, mod_acc): SUB [c, mod_acc, d], BIT [f,f,f], ISZERO_INV [d, 1/d or 1, f], ISZERO_RA [f, d]   17 MUL (and) [eq_bit, f, eq_bit']
//        i < C-1: nrc RANGE_CARRY rows of carry_{i+1}; then is_equal(carry, dup) (4 rows) and MUL (and)
//        i = C-1: is_equal(carry, acc_extra) (4 rows) and MUL (and)
//   T7  assert_equal_muled's main_gate.assert_one(eq_bit) (:1062): ASSERT_ONE [eq_bit] -- the record's last row
//   One workgroup per record, one thread per row (ten 16-byte stores of 160 contiguous bytes).  Bound: HBM writes
//   (635,840 bytes per RSA-2048 mul_mod: 9.9 x the flat stream -- what materialising field-element cells costs).
// ================================================================================================
enum : u32 { ROWK_NOP = 0, ROWK_CONST0, ROWK_CONST1, ROWK_CONST_B, ROWK_BIT, ROWK_VALUE, ROWK_MUL_ADD, ROWK_ADD, ROWK_SUB, ROWK_ADD_WM,
             ROWK_ADDC_WM, ROWK_MUL, ROWK_ASSERT_EQ, ROWK_ISZERO_INV, ROWK_ISZERO_RA, ROWK_ASSERT_ONE = 17 /* [a], a - 1 = 0 (15, 16: h2r_rowprog.hpp) */,
             ROWK_RANGE_LIMB = 32, ROWK_RANGE_CARRY = 40 };

template <int LW>
struct RecView {   // reads of one record through the documented plane layout (include/h2r.h)
    const u8 *rec; const u64 *off; u32 L;
    static constexpr u32 CB = LW == 64 ? 16 : 8;
    __device__ __forceinline__ u64 limb(int pl, u32 idx) const {
        if constexpr (LW == 64) return *reinterpret_cast<const u64 *>(rec + off[pl] + (u64)idx * 8);
        else return *reinterpret_cast<const u32 *>(rec + off[pl] + (u64)idx * 4);
    }
};

#ifndef H2R_ADVICE_INV
#define H2R_ADVICE_INV 1
#endif
constexpr u32 ADVICE_ROW_BYTES = 160;
// Where the cells of an advice image go, in either representation (h2r_advice_repr): cell (element e, row r, column c) lies at
//     base + e * elem_stride + r * row_pitch + c * col_pitch.
// Row-major image (the default): row_pitch = 160, col_pitch = 32.  Planar columns (H2R_ADVICE_COLUMNS, what a prover holds: one
// contiguous vector per advice column): row_pitch = 32, col_pitch = the caller's column stride.  mont (H2R_ADVICE_MONTGOMERY):
// the cells are x * R mod p (the in-memory form of halo2curves / pasta field elements) instead of canonical integers.
struct AdviceDst {
    u8 *base; u64 elem_stride, col_pitch; u32 row_pitch, mont;
    __host__ __device__ bool planar() const { return row_pitch != ADVICE_ROW_BYTES; }
    __host__ __device__ AdviceDst at_row(u64 r) const { AdviceDst d = *this; d.base += r * row_pitch; return d; }   // the image that starts r rows further down
    __device__ __forceinline__ u8 *elem(u64 e) const { return base + e * elem_stride; }
};
// canonical cell -> the destination's form (zero and one-limb values, nearly every cell of the small kernels, take the short product)
__device__ __forceinline__ void advice_cell_repr(uint4 &lo, uint4 &hi, const MontK *mk) {
    if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0) return;
    u32 t[8];
    if ((lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0) { const u32 x[2] = {lo.x, lo.y}; mont_short<2>(x, mk->bk[2], mk->p, mk->n0inv, t); }
    else { const u32 x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}; mont_short<8>(x, mk->bk[8], mk->p, mk->n0inv, t); }
    lo = make_uint4(t[0], t[1], t[2], t[3]); hi = make_uint4(t[4], t[5], t[6], t[7]);
}
// one cell of the element image at `img` (= dst.elem(e))
__device__ __forceinline__ void advice_put_cell(const AdviceDst &d, const MontK *mk, u8 *img, u64 row, u32 col, uint4 lo, uint4 hi) {
    if (d.mont) advice_cell_repr(lo, hi, mk);
    u8 *p = img + row * d.row_pitch + (u64)col * d.col_pitch;
    *reinterpret_cast<uint4 *>(p) = lo; *reinterpret_cast<uint4 *>(p + 16) = hi;
}
// The rows a workgroup of NT threads staged in LDS (row-major, canonical cells, 160 bytes each) leave for rows [r0, r0 + n_rows) of the
// element image at `img`.  The default representation IS the stage: full 16-byte-per-lane lines.  Otherwise one thread per cell;
// planar: consecutive threads take consecutive rows of one column (32 contiguous bytes each).
template <u32 NT>
__device__ __forceinline__ void advice_flush(const AdviceDst &d, const MontK *mk, u8 *img, u64 r0, u32 n_rows, const uint4 *stage, u32 tid) {
    if (!d.mont && !d.planar()) {
        u8 *dst = img + r0 * ADVICE_ROW_BYTES;
        for (u32 k = tid; k < n_rows * (ADVICE_ROW_BYTES / 16); k += NT) {
            const uint4 v = stage[k];
            st16(dst + 16ull * k, ((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z);
        }
        return;
    }
    const bool planar = d.planar();
    for (u32 k = tid; k < n_rows * 5; k += NT) {
        const u32 row = planar ? k % n_rows : k / 5, col = planar ? k / n_rows : k - 5 * (k / 5);
        uint4 lo = stage[row * 10 + 2 * col], hi = stage[row * 10 + 2 * col + 1];
        if (d.mont) advice_cell_repr(lo, hi, mk);
        u8 *p = img + (r0 + row) * d.row_pitch + (u64)col * d.col_pitch;
        st16(p, ((u64)lo.y << 32) | lo.x, ((u64)lo.w << 32) | lo.z);
        st16(p + 16, ((u64)hi.y << 32) | hi.x, ((u64)hi.w << 32) | hi.z);
    }
}
constexpr u32 ADVICE_STAGE_ROWS = 256;   // rows built in LDS per stage (= the workgroup size)
constexpr u32 ADVICE_COL_ROWS = 23;    // main-gate rows of one is_equal_muled column besides the carry's range assign
__host__ __device__ inline u32 advice_rows_per_record(u32 L, u32 carry_nsub) {
    const u32 C = 2 * L - 1, nrc = (carry_nsub + 3) / 4;
    return 2 * L * 2 + 2 * (C + L * L) + L + 4 + (C - 1) * (ADVICE_COL_ROWS + nrc) + ADVICE_COL_ROWS + 1;   // + assert_equal_muled's assert_one(eq_bit), chip.rs:1062
}

// Row r of a mul_mod's image: which op it belongs to.  Shared by the kernel and the host export of the row kinds.
struct AdviceRowId {
    u32 kind;      // ROWK_*
    u32 sect;      // 0 q/r range rows, 1 mul rows, 2 eq_b, 3 is_equal_muled preamble, 4 is_equal_muled column rows, 5 the closing assert_one(eq_bit)
    u32 i, j;      // sect 0: i = limb (L.. = r limbs), j = row of the assign; sect 1: column i, step j (kind CONST0: the column's head);
                   // sect 2: i; sect 3: i = 0..3; sect 4: column i, j = row within the column (range rows: j = 18 + row of the assign)
    u32 qn;        // sect 1: 0 = mul(a, b), 1 = mul(q, n)
};
__host__ __device__ inline u32 advice_mul_colstart(u32 i, u32 L) {   // rows of a mul() before column i: its accumulators + one head row per column
    const u32 e = i <= L ? i * (i + 1) / 2 : L * (L + 1) / 2 + (i - L) * (2 * L - 1) - ((i - 1) * i / 2 - (L - 1) * L / 2);
    return e + i;
}
__host__ __device__ inline AdviceRowId advice_decode(u32 r, u32 L, u32 nrc) {
    const u32 C = 2 * L - 1, mul_rows = C + L * L;
    const u32 r_T3 = 4 * L, r_T5 = r_T3 + 2 * mul_rows, r_T6p = r_T5 + L, r_T6 = r_T6p + 4, per_col = ADVICE_COL_ROWS + nrc;
    AdviceRowId id; id.kind = ROWK_NOP; id.sect = 0; id.i = id.j = id.qn = 0;
    if (r < r_T3) { id.sect = 0; id.i = r >> 1; id.j = r & 1; id.kind = ROWK_RANGE_LIMB + id.j; return id; }
    if (r < r_T5) {
        id.sect = 1; id.qn = r >= r_T3 + mul_rows ? 1u : 0u;
        const u32 e = r - r_T3 - id.qn * mul_rows;
        // Columns 0 .. L-1 hold 2, 3, ..., L + 1 rows (head + accumulators): column i starts at i (i + 3) / 2.  Columns
        // L .. 2L-2 mirror columns L-2 .. 0, so counted from the END of the section the same closed form applies.
        const bool back = e >= advice_mul_colstart(L, L);
        const u32 ee = back ? mul_rows - 1 - e : e;
        u32 ii = (u32)((sqrtf(8.f * (float)ee + 9.f) - 3.f) * 0.5f);
        if ((ii + 1) * (ii + 4) / 2 <= ee) ++ii;          // the float estimate is off by at most one
        else if (ii * (ii + 3) / 2 > ee) --ii;
        const u32 kk = ee - ii * (ii + 3) / 2;            // position within the (mirrored) column, 0 .. ii + 1
        const u32 i = back ? C - 1 - ii : ii, k = back ? ii + 1 - kk : kk;
        id.i = i;
        if (k == 0) { id.kind = ROWK_CONST0; id.j = 0; }
        else { id.kind = ROWK_MUL_ADD; id.j = (i >= L ? i - L + 1 : 0) + (k - 1); }
        return id;
    }
    if (r < r_T6p) { id.sect = 2; id.i = r - r_T5; id.kind = ROWK_ADD; return id; }
    if (r < r_T6) { id.sect = 3; id.i = r - r_T6p; id.kind = id.i == 0 ? ROWK_CONST_B : (id.i == 3 ? ROWK_BIT : ROWK_CONST0); return id; }
    const u32 rr = r - r_T6;
    if (rr == (C - 1) * per_col + ADVICE_COL_ROWS) {   // assert_equal_muled: main_gate.assert_one(eq_bit)  :1062 -- the record's last row
        id.sect = 5; id.i = C - 1; id.kind = ROWK_ASSERT_ONE; return id;
    }
    const u32 c = rr / per_col < C - 1 ? rr / per_col : C - 1;
    u32 k = rr - c * per_col;
    id.sect = 4; id.i = c;
    if (c < C - 1 && k >= 18 && k < 18 + nrc) { id.kind = ROWK_RANGE_CARRY + (k - 18); id.j = k; return id; }
    if (c < C - 1 && k >= 18 + nrc) k -= nrc;   // the second is_equal and its `and`: 18..22
    id.j = k;
    switch (k) {
        case 0: case 5: case 11: case 13: case 18: id.kind = ROWK_SUB; break;
        case 1: id.kind = ROWK_ADD_WM; break;
        case 2: case 3: case 8: case 9: id.kind = ROWK_VALUE; break;
        case 4: case 10: case 17: case 22: id.kind = ROWK_MUL; break;
        case 6: case 12: id.kind = ROWK_ASSERT_EQ; break;
        case 7: id.kind = ROWK_ADDC_WM; break;
        case 14: case 19: id.kind = ROWK_BIT; break;
        case 15: case 20: id.kind = ROWK_ISZERO_INV; break;
        default: id.kind = ROWK_ISZERO_RA; break;   // 16, 21
    }
    return id;
}

// the decode of every row is input-independent: the ctx keeps it as a table (one 32-bit word per row, L2-resident) so that the
// kernel's per-row decode is one coalesced load instead of ~100 integer instructions (3.3 -> see profiles/r03_emit_timing.txt)
__host__ __device__ inline u32 advice_pack(const AdviceRowId &id) { return id.kind | (id.sect << 6) | (id.qn << 9) | (id.i << 10) | (id.j << 18); }
__host__ __device__ inline AdviceRowId advice_unpack(u32 v) {
    AdviceRowId id; id.kind = v & 63u; id.sect = (v >> 6) & 7u; id.qn = (v >> 9) & 1u; id.i = (v >> 10) & 255u; id.j = v >> 18;
    return id;
}

// Sources of the is_equal_muled column rows 1..22 (row 0 reads accumulator entries and is planned on its own): per row three
// codes {type, plane, index = column or column - 1}.  type 1: 16-byte LO entry + 8-byte HI plane behind it (s_wide), 2: a carry-sized
// entry, 3: a limb-sized entry, 4: the range-assigned carry (CARRY_DUP; the last column compares with QACC instead).
__host__ __device__ constexpr u32 adv_src(u32 type, u32 plane, u32 minus1 = 0) { return type | (plane << 3) | (minus1 << 9); }
__host__ __device__ constexpr u32 advice_col_src(u32 j, u32 k) {
    switch (j * 3 + k) {
        case 1 * 3 + 0: return adv_src(1, H2R_PL_AMB_LO); case 1 * 3 + 1: return adv_src(2, H2R_PL_CARRY, 1); case 1 * 3 + 2: return adv_src(1, H2R_PL_SUM_LO);
        case 2 * 3 + 0: return adv_src(2, H2R_PL_CARRY);
        case 3 * 3 + 0: return adv_src(3, H2R_PL_CMOD);
        case 4 * 3 + 1: return adv_src(2, H2R_PL_CARRY); case 4 * 3 + 2: return adv_src(1, H2R_PL_NQ1_LO);
        case 5 * 3 + 0: return adv_src(1, H2R_PL_SUM_LO); case 5 * 3 + 1: return adv_src(1, H2R_PL_NQ1_LO); case 5 * 3 + 2: return adv_src(3, H2R_PL_AMNQ1);
        case 6 * 3 + 0: return adv_src(3, H2R_PL_CMOD); case 6 * 3 + 1: return adv_src(3, H2R_PL_AMNQ1);
        case 7 * 3 + 0: return adv_src(2, H2R_PL_QACC, 1); case 7 * 3 + 1: return adv_src(1, H2R_PL_ACCX_LO);
        case 8 * 3 + 0: return adv_src(2, H2R_PL_QACC);
        case 9 * 3 + 0: return adv_src(3, H2R_PL_MODACC);
        case 10 * 3 + 1: return adv_src(2, H2R_PL_QACC); case 10 * 3 + 2: return adv_src(1, H2R_PL_NQ2_LO);
        case 11 * 3 + 0: return adv_src(1, H2R_PL_ACCX_LO); case 11 * 3 + 1: return adv_src(1, H2R_PL_NQ2_LO); case 11 * 3 + 2: return adv_src(3, H2R_PL_AMNQ2);
        case 12 * 3 + 0: return adv_src(3, H2R_PL_MODACC); case 12 * 3 + 1: return adv_src(3, H2R_PL_AMNQ2);
        case 13 * 3 + 0: case 15 * 3 + 0: case 16 * 3 + 0: return adv_src(3, H2R_PL_CMOD);         // is_equal(c, mod_acc)
        case 13 * 3 + 1: case 15 * 3 + 1: case 16 * 3 + 1: return adv_src(3, H2R_PL_MODACC);
        case 18 * 3 + 0: case 20 * 3 + 0: case 21 * 3 + 0: return adv_src(2, H2R_PL_CARRY);         // is_equal(carry, dup | acc_extra)
        case 18 * 3 + 1: case 20 * 3 + 1: case 21 * 3 + 1: return adv_src(4, H2R_PL_CARRY_DUP);
        default: return 0;                                                                          // 14, 17, 19, 22: flag bytes only
    }
}

struct AdviceArgs {
    const u32 *desc;                    // [rows] advice_pack(advice_decode(r))
    const void *opA, *opB; u64 op_stride; const void *n; u64 n_stride;
    const u8 *status;
    const u8 *trace; u64 elem_stride, off_records, record_stride; u32 T; u64 n_items;
    AdviceDst dst; const MontK *mk;     // element e's image: pre_rows rows, then record t from row pre_rows + t * rows (+ the select rows) on
    u64 off[H2R_PL_COUNT];
    u32 L, carry_bits, carry_sub_bits, carry_nsub, carry_sub_stride;
    u32 rows;                           // rows of one record
    u32 pre_rows;                       // pow_mod_fixed_exp's acc = assign_constant(1, L) (chip.rs:729 -> :1272-1276): CONST1 [1], CONST0 [0]
    u32 sel_rows;                       // pow_mod (Var): rows left free behind every EVEN record for the bit's select rows (chip.rs:688-691)
    FieldConsts f;                      // field modulus + Montgomery constants (is_zero's inverse witness)
};

#ifndef H2R_ADV_ABL
#define H2R_ADV_ABL 0   // developer ablations: 1 no build, 2 no plan, no build, 3 no write-out, 4 range / column rows built as zero rows, 5 their loads dropped
#endif
template <int LW>
__global__ __launch_bounds__(256) void advice_kernel(AdviceArgs a) {
    using limb_t = typename LimbT<LW>::type;
    __shared__ u64 sa[128], sb_[128], sq[128], sn[128], sr[128];
    constexpr u32 SR = ADVICE_STAGE_ROWS;                         // rows per stage (40 KB of LDS; 128-208 rows, i.e. four or more workgroups per CU, measured within +-5 % of it)
    __shared__ uint4 stage[SR * (ADVICE_ROW_BYTES / 16)];    // SR rows are built in LDS, then leave as full 16-byte-per-lane lines
    __shared__ u64 s_off[H2R_PL_COUNT];                      // plane offsets and the column rows' source codes, indexed per LANE below
    __shared__ u32 s_col_src[ADVICE_COL_ROWS * 3];
    const u32 tid = threadIdx.x;
    if (tid < H2R_PL_COUNT) s_off[tid] = a.off[tid];
    if (tid >= 64 && tid < 64 + ADVICE_COL_ROWS * 3) s_col_src[tid - 64] = advice_col_src((tid - 64) / 3, (tid - 64) % 3);
    const u32 item = xcd_contiguous_block(blockIdx.x, gridDim.x);   // every XCD reads and writes a contiguous eighth
    const u32 elem = item / a.T, t = item - elem * a.T;
    if (a.status && a.status[elem]) return;
    const u32 L = a.L, C = 2 * L - 1, nrc = (a.carry_nsub + 3) / 4;
    const RecView<LW> rv{a.trace + (u64)elem * a.elem_stride + a.off_records + (u64)t * a.record_stride, a.off, L};
    for (u32 k = tid; k < L; k += 256) {
        const u64 ib = (u64)item * a.op_stride + k;
        sa[k] = reinterpret_cast<const limb_t *>(a.opA)[ib]; sb_[k] = reinterpret_cast<const limb_t *>(a.opB)[ib];
        sn[k] = reinterpret_cast<const limb_t *>(a.n)[(u64)elem * a.n_stride + k];
        sq[k] = rv.limb(H2R_PL_Q, k); sr[k] = rv.limb(H2R_PL_R, k);
    }
    __syncthreads();
    u8 *img = a.dst.elem(elem);
    const u64 row_base = (u64)a.pre_rows + (u64)t * a.rows + (u64)((t + 1) >> 1) * a.sel_rows;   // the record's first row in the element image
    if (t == 0 && tid < a.pre_rows * 5) {   // the constant limbs of pow_mod_fixed_exp's acc = 1: [1, 0, 0, 0, 0] then [0, ...]
        advice_put_cell(a.dst, a.mk, img, tid / 5, tid % 5, make_uint4(tid == 0 ? 1u : 0u, 0, 0, 0), make_uint4(0, 0, 0, 0));
    }
    const U192 Z = U192::make(0, 0, 0);
    const U192 B = LW == 64 ? U192::make(0, 1, 0) : U192::make(1ull << 32, 0, 0);   // 2^w
    auto cell = [&](u8 *p, const U192 &v, bool is_signed) {   // canonical field element, 32 bytes little-endian
        u64 x[4] = {v.w[0], v.w[1], v.w[2], 0};
        if (is_signed && (v.w[2] >> 63)) {   // x < 0 -> p + x (mod 2^256)
            x[3] = ~0ull;
            u64 cy = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u64 s1 = x[k] + a.f.p[k]; const u64 c1 = s1 < x[k]; const u64 s2 = s1 + cy; cy = c1 | (u64)(s2 < s1); x[k] = s2; }
        }
        reinterpret_cast<uint4 *>(p)[0] = make_uint4((u32)x[0], (u32)(x[0] >> 32), (u32)x[1], (u32)(x[1] >> 32));
        reinterpret_cast<uint4 *>(p)[1] = make_uint4((u32)x[2], (u32)(x[2] >> 32), (u32)x[3], (u32)(x[3] >> 32));
    };
    auto row = [&](u32 r, const U192 &c0, const U192 &c1, const U192 &c2, const U192 &c3, const U192 &c4, bool c2_signed = false, bool c0_signed = false, bool c1_signed = false) {
        u8 *p = reinterpret_cast<u8 *>(stage) + (u64)(r % SR) * ADVICE_ROW_BYTES;   // r - r0 == tid (r0 is a multiple of SR)
        cell(p, c0, c0_signed); cell(p + 32, c1, c1_signed); cell(p + 64, c2, c2_signed); cell(p + 96, c3, false); cell(p + 128, c4, false);
    };
    auto lim = [&](u64 v) { return U192::make(v, 0, 0); };
    // 1 / d in the field for the (signed) difference d != 0 -- main_gate.is_zero's witness; rare on this path (every
    // comparison of a valid mul_mod is between equal values), so the 380 Montgomery products sit in a divergent branch
    auto inverse_cell = [&](u8 *p, const U192 &d) {
        Fe x; x.v[0] = d.w[0]; x.v[1] = d.w[1]; x.v[2] = d.w[2]; x.v[3] = 0;
        if (d.w[2] >> 63) {
            x.v[3] = ~0ull;
            u64 cy = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u64 s1 = x.v[k] + a.f.p[k]; const u64 c1 = s1 < x.v[k]; const u64 s2 = s1 + cy; cy = c1 | (u64)(s2 < s1); x.v[k] = s2; }
        }
        const Fe iv = fe_inv_fast(x, a.f);
        reinterpret_cast<uint4 *>(p)[0] = make_uint4((u32)iv.v[0], (u32)(iv.v[0] >> 32), (u32)iv.v[1], (u32)(iv.v[1] >> 32));
        reinterpret_cast<uint4 *>(p)[1] = make_uint4((u32)iv.v[2], (u32)(iv.v[2] >> 32), (u32)iv.v[3], (u32)(iv.v[3] >> 32));
    };
    // a range assign's row (main_gate.decompose): four sub-limbs in columns a..d -- the LAST row reversed, so that the last
    // (overflow) term is in column a, and padded with zero terms -- and, in column e, what remains to be composed
    auto range_vals = [&](u64 s_lo, u64 s_hi, u32 nsub, u32 sub_bits, u32 rr, U192 &c0, U192 &c1, U192 &c2, U192 &c3, U192 &rem) {   // sub-limb bytes in (s_lo, s_hi), row rr of the assign
        c0 = Z; c1 = Z; c2 = Z; c3 = Z; rem = Z;        // (no indexed array: it would live in scratch)
        const u32 last = (nsub - 1) / 4;
        if (sub_bits == 8) {   // byte sub-limbs (every 64-bit-limb shape): the bytes ARE the value -- what remains = the value with its low 4 rr bytes cleared
            const u64 hi = nsub > 8 ? s_hi & ((1ull << (8 * (nsub - 8))) - 1) : 0;
            auto byte = [&](u32 k) -> u64 { return k < nsub ? ((k < 8 ? s_lo >> (8 * k) : hi >> (8 * (k - 8))) & 0xff) : 0; };
            const u32 k0 = rr < last ? 4 * rr : nsub - 1;        // the last row is reversed: cell q holds sub-limb nsub - 1 - q
            const u32 n_last = nsub - 4 * last;                  // terms of the last row
            c0 = lim(byte(k0));
            c1 = lim(rr < last ? byte(k0 + 1) : (n_last > 1 ? byte(k0 - 1) : 0));
            c2 = lim(rr < last ? byte(k0 + 2) : (n_last > 2 ? byte(k0 - 2) : 0));
            c3 = lim(rr < last ? byte(k0 + 3) : (n_last > 3 ? byte(k0 - 3) : 0));
            rem = rr == 0 ? U192::make(s_lo, hi, 0) : (rr == 1 ? U192::make(s_lo & ~0xffffffffull, hi, 0) : (rr == 2 ? U192::make(0, hi, 0) : U192::make(0, hi & ~0xffffffffull, 0)));
            return;
        }
#pragma unroll
        for (u32 k = 0; k < 12; ++k) {                   // at most 9 sub-limbs (8 + overflow), three rows
            if (k >= 4 * rr && k < nsub) {
                const u64 sv = ((k < 8 ? s_lo : s_hi) >> (8 * (k & 7))) & 0xff;
                const u32 sh = k * sub_bits;
                const U192 term = sh < 64 ? U192::make(sv << sh, sh ? sv >> (64 - sh) : 0, 0) : U192::make(0, sv << (sh - 64), 0);
                rem = rem + term;
                if (k < 4 * (rr + 1)) {
                    const u32 q = rr < last ? k - 4 * rr : nsub - 1 - k;
                    if (q == 0) c0 = lim(sv); else if (q == 1) c1 = lim(sv); else if (q == 2) c2 = lim(sv); else c3 = lim(sv);
                }
            }
        }
    };
    // A row has five cells; at most three of them come from the record, the others are constants, flag bytes or staged
    // operands.  Every row is therefore described the same way -- up to three sources (a 16-, 8- or 4-byte load plus an
    // optional 8-byte third word) -- so that the lanes of a wave, which sit in ~25 different rows of 3 columns in the
    // is_equal_muled part, issue the SAME few load instructions with different addresses (a switch over loads serialised
    // ~20 dependent global round trips per wave; loading all values of the column in every lane cost ~40 load
    // instructions per wave).  The loads of the NEXT 256 rows are issued before the current 256 leave LDS for HBM, so
    // their latency hides behind the write-out.
    // The loads themselves are unconditional and of one width (16 bytes at `lo`, 8 at `hi`; what the source does not have is
    // masked off afterwards, an absent source points at the record's first bytes): a load inside a divergent branch gets an
    // s_waitcnt vmcnt(0) at the end of its block, which would serialise the fetches and defeat the prefetch.  A 16-byte read
    // of an 8- or 4-byte plane entry ends at most 12 bytes behind the plane -- inside the record (its stride is padded).
    struct Src { const u8 *lo; const u8 *hi; u32 mode; };   // mode & 3: 0 none, 1 = 4 B, 2 = 8 B, 3 = 16 B at lo; mode & 4: third word = sign of word 1; mode & 8: third word at hi
    constexpr u32 CB = LW == 64 ? 16 : 8;
    auto s_none = [&]() -> Src { return Src{rv.rec, rv.rec, 0u}; };
    auto s_wide = [&](int pl_lo, u32 idx) -> Src {
        if constexpr (LW == 64) return Src{rv.rec + a.off[pl_lo] + (u64)idx * 16, rv.rec + a.off[pl_lo + 1] + (u64)idx * 8, 3u | 8u};
        else return Src{rv.rec + a.off[pl_lo] + (u64)idx * 16, rv.rec, 3u | 4u};
    };
    auto s_acc = [&](bool qn, u32 j, u32 im) -> Src {   // accumulator entry (j, i % L)
        if constexpr (LW == 64)
            return Src{rv.rec + a.off[qn ? H2R_PL_QN_LO : H2R_PL_AB_LO] + (u64)(j >> 1) * (3ull * 2 * L * 16) + (u64)(j & 1) * (2ull * L * 16) + (u64)im * 16,
                       rv.rec + a.off[qn ? H2R_PL_QN_HI : H2R_PL_AB_HI] + (u64)(j >> 1) * (3ull * 2 * L * 16) + (u64)im * 16 + (j & 1) * 8, 3u | 8u};
        else return Src{rv.rec + a.off[qn ? H2R_PL_QN_LO : H2R_PL_AB_LO] + (u64)j * ((u64)L * 16) + (u64)im * 16, rv.rec, 3u};
    };
    auto s_carry = [&](int pl, u32 idx) -> Src { return Src{rv.rec + a.off[pl] + (u64)idx * CB, rv.rec, LW == 64 ? 3u : 2u}; };
    auto s_limb = [&](int pl, u32 idx) -> Src { return Src{rv.rec + a.off[pl] + (u64)idx * (LW / 8), rv.rec, LW == 64 ? 2u : 1u}; };
    // per-row state that lives across the write-out of the previous rows
    AdviceRowId id; id.kind = ROWK_NOP; id.sect = 9; id.i = id.j = id.qn = 0;
    u32 m0 = 0, m1 = 0, m2 = 0, fl = 0, eprev = 1, has_prev = 0, rcur = 0;   // rcur: the row planned last (built next)
    u64 imm0 = 0, imm1 = 0, h0 = 0, h1 = 0, h2 = 0;
    ulonglong2 l0 = make_ulonglong2(0, 0), l1 = l0, l2 = l0;
    auto fetch = [&](const Src &sc, ulonglong2 &lo, u64 &hi, u32 &mode) {
        mode = sc.mode;
        const u64 *p = reinterpret_cast<const u64 *>(sc.lo);   // 8-byte aligned at least (4 for the 32-bit limb planes)
        lo = make_ulonglong2(p[0], p[1]);
        hi = *reinterpret_cast<const u64 *>(sc.hi);
    };
    auto plan_and_load = [&](u32 r) {
        rcur = r;
        Src s0 = s_none(), s1 = s_none(), s2 = s_none();
        const u8 *fp = rv.rec, *fpp = rv.rec;   // flag words of the column and of the one before it
        has_prev = 0;
        if (r >= a.rows) { id.kind = ROWK_NOP; id.sect = 9; }
        else {
            id = advice_unpack(a.desc[r]);
            if (id.sect == 0) {                                 // q then r limbs: RangeChip::assign(limb, w/8, w)
                const bool isr = id.i >= L;
                s0 = Src{rv.rec + a.off[isr ? H2R_PL_R_SUB : H2R_PL_Q_SUB] + (u64)(isr ? id.i - L : id.i) * 8, rv.rec, 2u};
            } else if (id.sect == 1) {                          // mul(): the column's constant 0, then its mul_add rows
                if (id.kind == ROWK_MUL_ADD) {
                    const u32 i = id.i, j = id.j, jmin = i >= L ? i - L + 1 : 0, im = i >= L ? i - L : i;
                    imm0 = id.qn ? sq[j] : sa[j]; imm1 = id.qn ? sn[i - j] : sb_[i - j];
                    if (j != jmin) s0 = s_acc(id.qn, j - 1, im);
                    s1 = s_acc(id.qn, j, im);
                }
            } else if (id.sect == 2) {                          // eq_b[i] = qn[i] + r[i]
                imm0 = sr[id.i];
                s0 = s_acc(true, id.i, id.i); s2 = s_wide(H2R_PL_EQB_LO, id.i);
            } else if (id.sect == 4) {                          // is_equal_muled column rows
                const u32 c = id.i;
                const u32 jmax = c < L ? c : L - 1, im = c < L ? c : c - L;
                if (id.kind >= ROWK_RANGE_CARRY) s0 = Src{rv.rec + a.off[H2R_PL_CARRY_SUB] + (u64)c * a.carry_sub_stride, rv.rec, 3u};
                else if (id.j == 0) { s0 = s_acc(false, jmax, im); s1 = c < L ? s_wide(H2R_PL_EQB_LO, c) : s_acc(true, jmax, im); s2 = s_wide(H2R_PL_AMB_LO, c); }
                else {
                    // rows 1..22: the lanes of a wave sit in ~22 different rows -- one table-driven plan for all of them instead
                    // of a switch whose cases ran one after the other under disjoint exec masks
                    auto tab_src = [&](u32 code) -> Src {
                        const u32 type = code & 7u, minus1 = (code >> 9) & 1u;
                        u32 pl = (code >> 3) & 63u;
                        if (type == 0u || (minus1 && c == 0)) return s_none();
                        if (type == 4u && c == C - 1) pl = H2R_PL_QACC;
                        const u32 idx = c - minus1;
                        const u32 esz = type == 1u ? 16u : (type == 3u ? LW / 8 : CB);
                        const u8 *lo = rv.rec + s_off[pl] + (u64)idx * esz;
                        if (type == 1u) {
                            if constexpr (LW == 64) return Src{lo, rv.rec + s_off[pl + 1] + (u64)idx * 8, 3u | 8u};
                            else return Src{lo, rv.rec, 3u | 4u};
                        }
                        return Src{lo, rv.rec, type == 3u ? (LW == 64 ? 2u : 1u) : (LW == 64 ? 3u : 2u)};
                    };
                    s0 = tab_src(s_col_src[id.j * 3]); s1 = tab_src(s_col_src[id.j * 3 + 1]); s2 = tab_src(s_col_src[id.j * 3 + 2]);
                }
                if (id.kind < ROWK_RANGE_CARRY) {
                    fp = rv.rec + a.off[H2R_PL_FLAGS] + (u64)c * 4;
                    fpp = c ? fp - 4 : fp; has_prev = c ? 1u : 0u;
                }
            }
            else if (id.sect == 5) fp = rv.rec + a.off[H2R_PL_FLAGS] + (u64)(C - 1) * 4;   // the final eq_bit: e2 of the last column
        }
        if (H2R_ADV_ABL == 5 && (id.sect == 4 || id.sect == 0)) { s0 = s_none(); s1 = s_none(); s2 = s_none(); fp = rv.rec; fpp = rv.rec; }
        fetch(s0, l0, h0, m0); fetch(s1, l1, h1, m1); fetch(s2, l2, h2, m2);
        fl = *reinterpret_cast<const u32 *>(fp);
        eprev = *reinterpret_cast<const u32 *>(fpp);
    };
    auto val = [&](const ulonglong2 &lo, u64 hi, u32 mode) -> U192 {
        const u32 wd = mode & 3u;
        if (!wd) return Z;
        return U192::make(wd == 1u ? (lo.x & 0xffffffffull) : lo.x, wd == 3u ? lo.y : 0, (mode & 4u) ? (u64)((i64)lo.y >> 63) : ((mode & 8u) ? hi : 0));
    };
    // Every branch below only CHOOSES the five cell values; the row is staged once, after them.  (With the staging inlined into each
    // case the lanes of a wave -- which sit in ~25 different row kinds in the is_equal_muled part -- ran ~15 copies of the ten
    // ds_write_b128 one after the other under disjoint exec masks.)
    auto build = [&](u32 r) {
        if (id.sect == 9) return;
        if (H2R_ADV_ABL == 4 && (id.sect == 4 || id.sect == 0)) { row(r, Z, Z, Z, Z, Z); return; }
        U192 v0 = Z, v1 = Z, v2 = Z, v3 = Z, v4 = Z;
        bool sg0 = false, sg1 = false, sg2 = false, need_inv = false;
        if (id.sect == 0) range_vals(l0.x, 0, 8, LW / 8, id.j, v0, v1, v2, v3, v4);   // eight sub-limbs, one byte each
        else if (id.sect == 4 && id.kind >= ROWK_RANGE_CARRY) range_vals(l0.x, l0.y, a.carry_nsub, a.carry_sub_bits, id.kind - ROWK_RANGE_CARRY, v0, v1, v2, v3, v4);
        else {
            const U192 c0 = val(l0, h0, m0), c1 = val(l1, h1, m1), c2 = val(l2, h2, m2);
            if (id.sect == 1) { if (id.kind == ROWK_MUL_ADD) { v0 = lim(imm0); v1 = lim(imm1); v2 = c0; v3 = c1; } }
            else if (id.sect == 2) { v0 = c0; v1 = lim(imm0); v2 = c2; }
            else if (id.sect == 3) { if (id.i == 0) v0 = B; else if (id.i == 3) { v0 = lim(1); v1 = lim(1); v2 = lim(1); } }
            else if (id.sect == 5) v0 = lim(fl >> 24);                                                 // assert_one [eq_bit]  :1062
            else {
                const u32 f1 = fl & 0xff, e1 = (fl >> 8) & 0xff, f2 = (fl >> 16) & 0xff, e2 = fl >> 24;
                const U192 d = c0 - c1;
                switch (id.j) {
                    case 4: case 10: v0 = B; v1 = c1; v2 = c2; break;
                    case 13: case 18: v0 = c0; v1 = c1; v2 = d; sg2 = true; break;                    // sub: d = x - y in the field
                    case 14: v0 = lim(f1); v1 = v0; v2 = v0; break;
                    case 19: v0 = lim(f2); v1 = v0; v2 = v0; break;
                    case 15: case 20: v0 = d; sg0 = true; v1 = lim(1); v2 = lim(id.j == 15 ? f1 : f2); need_inv = !(d == Z); break;   // [d, 1/d (1 when d = 0), r]
                    case 16: case 21: v0 = lim(id.j == 16 ? f1 : f2); v1 = d; sg1 = true; break;      // [r, d]
                    case 17: v0 = lim(has_prev ? (eprev >> 24) : 1u); v1 = lim(f1); v2 = lim(e1); break;   // and
                    case 22: v0 = lim(e1); v1 = lim(f2); v2 = lim(e2); break;                           // and
                    default: v0 = c0; v1 = c1; v2 = c2; sg2 = id.j == 0; sg0 = id.j == 1; break;
                }
            }
        }
        row(r, v0, v1, v2, v3, v4, sg2, sg0, sg1);
        if (need_inv && H2R_ADVICE_INV) inverse_cell(reinterpret_cast<u8 *>(stage) + (u64)(r % SR) * ADVICE_ROW_BYTES + 32, v0);
    };
    static_assert(SR <= 256, "one row per thread and stage");
    if (H2R_ADV_ABL != 2) plan_and_load(tid < SR ? tid : a.rows);
    for (u32 r0 = 0; r0 < a.rows; r0 += SR) {
      if (H2R_ADV_ABL != 1 && H2R_ADV_ABL != 2) build(rcur);
      __syncthreads();
      if (H2R_ADV_ABL != 2) if (r0 + SR < a.rows) plan_and_load(tid < SR ? r0 + SR + tid : a.rows);   // in flight while this stage leaves for HBM
      const u32 n_rows = a.rows - r0 < SR ? a.rows - r0 : SR;
      if (H2R_ADV_ABL != 3) advice_flush<256>(a.dst, a.mk, img, row_base + r0, n_rows, stage, tid);
      __syncthreads();
    }
}

// stand-alone RangeChip::assign decomposition of a value array (8- or 16-byte values)
struct DecompArgs {
    const u8 *values; u32 value_bytes; u64 count; u32 bit_len, sub_bits, nsub, has_ov;
    u8 *sub_out; u32 sub_stride; u32 *hist; u32 comp_len;
};
#ifdef H2R_TU_API   // a plain (non-template) kernel: defined in the one translation unit that launches it
__global__ __launch_bounds__(256) void decompose_kernel(DecompArgs a) {
    extern __shared__ u32 h[];
    const u32 hl = a.hist ? a.comp_len + (a.has_ov ? (1u << (a.bit_len % a.sub_bits)) : 0) : 0;
    for (u32 k = threadIdx.x; k < hl; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const u32 m = (1u << a.sub_bits) - 1;
    for (u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x; idx < a.count; idx += (u64)gridDim.x * blockDim.x) {
        u128 v = *reinterpret_cast<const u64 *>(a.values + idx * a.value_bytes);
        if (a.value_bytes == 16) v |= (u128)(*reinterpret_cast<const u64 *>(a.values + idx * 16 + 8)) << 64;
        for (u32 k = 0; k < a.nsub; ++k) {
            const u32 sv = (u32)v & m; v >>= a.sub_bits;
            if (a.sub_out) a.sub_out[idx * a.sub_stride + k] = (u8)sv;
            if (a.hist) atomicAdd(&h[(a.has_ov && k == a.nsub - 1) ? a.comp_len + sv : sv], 1u);
        }
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < hl; k += blockDim.x) if (h[k]) atomicAdd(&a.hist[k], h[k]);
}
#endif

}  // namespace h2r
