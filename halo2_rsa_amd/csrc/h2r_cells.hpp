// gfx950 kernel of libh2r: the advice-column image written DIRECTLY from a mul_mod's operands (a, b, q, r, n).
//
// The reference puts every value it computes straight into main-gate cells (main_gate.mul_add big_integer/chip.rs:408,
// range_chip.assign :590, :598, :880-885, the is_equal_muled ops :851-893): the prover-consumable form of the witness is the
// 5-column image of DESIGN.md section 2b, not the struct-of-planes record.  advice_kernel (h2r_kernels.hpp) converts a stored
// record into that image; it re-reads the 64 KB record through ~10 uncoalesced loads per row and is bound by the texture
// addresser, not by HBM.  cells_kernel needs no record: one WAVE per mul_mod keeps the operands (5 x L limbs) in LDS, walks the
// image's rows in order, 64 rows at a time, and recomputes everything on the way --
//   * mul(a, b) / mul(q, n) rows (chip.rs:400-412): lane = row, one limb product per lane, the running accumulators of a
//     column by a segmented wave scan (DPP row shifts / row broadcasts); a column that straddles two 64-row chunks takes the
//     previous chunk's last accumulator from lane 63 (v_readlane).  The last row of a column leaves its total in LDS.
//   * eq_b and is_equal_muled (chip.rs:617, :857-893): once the last mul row is built, the 2L - 1 un-carried columns get
//     a_b, the carries (three-level reduction, the last level a generate / propagate chain solved by ballot: the record
//     kernel's scheme) and the running eq_bit into LDS planes of ready-made cells; the 23 + nrc rows of a column then COPY their
//     cells from those planes through one table-driven fetch (no per-row-kind branches).  The input-independent
//     accumulated_extra chain (:869-875) reaches its fixed point at column 2, so its values are a 3-entry table.
//   * range rows (main_gate.decompose of a limb / carry) are cut from the value itself.
// 64 rows = 10,240 bytes are staged in LDS and leave as ten 1 KB store instructions (16 bytes per lane) that cover whole
// 128-byte lines (the first chunk of an item is cut so that every later one starts on a line), so the only HBM traffic is the
// image itself: 635,840 bytes written per RSA-2048 mul_mod against 1.3 KB read.  No workgroup barrier anywhere (a workgroup IS a
// wave) and no global load inside the row loop (vmcnt counts loads and stores alike: waiting for a load would wait for the
// previous chunk's stores).  Chunks that lie wholly inside the mul rows or the column rows of a VALID mul_mod take a branch-free
// fast path; chunks that straddle sections, and a mul_mod whose q, r are not its quotient and remainder (never produced by this
// library: is_zero's inverse witnesses appear), take the general path.
// Bound: HBM writes.  The image is byte-identical to advice_kernel's for every valid record (tests).
#pragma once

#include "h2r_kernels.hpp"

namespace h2r {

struct CellsArgs {
    const u64 *ktab;                     // CELLS_TAB_WORDS words (one table per ctx): [3][10] columns 0, 1, >= 2 of the accumulated_extra
                                         // constants (acc_extra + W [3 words], q_acc [2], mod_acc [1], nq [3], a - nq [1]); word_max
                                         // (chip.rs:838) at 32; the field modulus at 36; the field's FieldConsts at 40; the column rows'
                                         // packed fast-path sources (cells_pack_fast_src) at 56
    const void *opA, *opB, *opQ, *opR;   // limbs of item k at [k * op_stride, k * op_stride + L)
    u64 op_stride, qr_stride;            // (limbs) of opA / opB and of opQ / opR
    const void *n; u64 n_stride;         // [elem][L] limbs (stride 0 = shared)
    const u8 *status;                    // [elem] nullable; nonzero => the element's items are skipped
    u32 T; u64 n_items;                  // item = elem * T + t
    AdviceDst dst;                       // element e's image: pre_rows rows, then record t from row pre_rows + t * rows (+ the select rows) on
    const MontK *mk;                     // H2R_ADVICE_MONTGOMERY: the short Montgomery multipliers (the kernel keeps the ones it needs in VGPRs: as SGPRs they spilled)
    u32 rows, pre_rows;
    u32 sel_rows;                        // pow_mod (Var): rows left free behind every EVEN record for the bit's select rows
    u32 L, carry_sub_bits, carry_nsub;
    u32 per_col_magic;                   // ceil(2^32 / (23 + nrc)): row of the column part -> column index by one mul_hi
    u64 *dbg;                            // developer build (ABL & 256): s_memtime stamps of item dbg_item's chunks
    u32 dbg_item;
    u32 spread;                          // developer build (ABL & 1024): workgroup b -> item (b % spread) * (n / spread) + b / spread
};

constexpr u32 CELLS_KT_WORDS = 40;       // the table's words kept in LDS: 3 x 10 + padding (the general fetch reads three words), word_max, p
constexpr u32 CELLS_KT_WM = 32, CELLS_KT_P = 36, CELLS_KT_FC = 40, CELLS_KT_FSRC = 56;   // (FieldConsts: 13 words)
constexpr u32 CELLS_TAB_WORDS = CELLS_KT_FSRC + 72;   // + the fast path's packed source codes: 2 x 72 32-bit words
constexpr u32 CELLS_SRC_WORDS = 72;      // the column rows' source codes (23 x 3) + padding

// ---- general path: source of one cell of an is_equal_muled column row (rows 0..22 of the column, cells a, b, c) ----
//   bits 0-2 base (0 none, 1 AB, 2 EQB, 3 a_b = AB - EQB, 4 SUM, 5 the constants' table), bits 3-6 word offset in a table entry,
//   bits 7-8 words - 1 of a table entry, bit 9 column c - 1 (zero for c = 0), bits 10-11 transform (0 the value, 1 value >> w,
//   2 value mod 2^w, 3 value with its low limb cleared), bit 12 two's complement (field subtraction), bit 13 "the carry's
//   range-assigned duplicate": the last column compares with q_acc instead (chip.rs:888-892)
__host__ __device__ constexpr u32 cells_src(u32 base, u32 xf = 0, u32 m1 = 0, u32 sg = 0, u32 woff = 0, u32 nw = 3, u32 dup = 0) {
    return base | (woff << 3) | ((nw - 1) << 7) | (m1 << 9) | (xf << 10) | (sg << 12) | (dup << 13);
}
enum : u32 { CS_AB = 1, CS_EQB = 2, CS_AMB = 3, CS_SUM = 4, CS_KT = 5 };
__host__ __device__ constexpr u32 cells_col_src(u32 j, u32 k) {
    constexpr u32 COUT = cells_src(CS_SUM, 1), CMOD = cells_src(CS_SUM, 2), NQ1 = cells_src(CS_SUM, 3), SUM = cells_src(CS_SUM),
                  AMB = cells_src(CS_AMB, 0, 0, 1), CIN = cells_src(CS_SUM, 1, 1),
                  K_ACCX = cells_src(CS_KT, 0, 0, 0, 0, 3), K_QACC = cells_src(CS_KT, 0, 0, 0, 3, 2), K_MODACC = cells_src(CS_KT, 0, 0, 0, 5, 1),
                  K_NQ2 = cells_src(CS_KT, 0, 0, 0, 6, 3), K_AMNQ2 = cells_src(CS_KT, 0, 0, 0, 9, 1), K_QACC_M1 = cells_src(CS_KT, 0, 1, 0, 3, 2),
                  DUP = cells_src(CS_SUM, 1, 0, 0, 0, 3, 1);
    switch (j * 3 + k) {
        case 0 * 3 + 0: return cells_src(CS_AB); case 0 * 3 + 1: return cells_src(CS_EQB); case 0 * 3 + 2: return AMB;
        case 1 * 3 + 0: return AMB; case 1 * 3 + 1: return CIN; case 1 * 3 + 2: return SUM;
        case 2 * 3 + 0: return COUT;
        case 3 * 3 + 0: return CMOD;
        case 4 * 3 + 1: return COUT; case 4 * 3 + 2: return NQ1;
        case 5 * 3 + 0: return SUM; case 5 * 3 + 1: return NQ1; case 5 * 3 + 2: return CMOD;
        case 6 * 3 + 0: return CMOD; case 6 * 3 + 1: return CMOD;
        case 7 * 3 + 0: return K_QACC_M1; case 7 * 3 + 1: return K_ACCX;
        case 8 * 3 + 0: return K_QACC;
        case 9 * 3 + 0: return K_MODACC;
        case 10 * 3 + 1: return K_QACC; case 10 * 3 + 2: return K_NQ2;
        case 11 * 3 + 0: return K_ACCX; case 11 * 3 + 1: return K_NQ2; case 11 * 3 + 2: return K_AMNQ2;
        case 12 * 3 + 0: return K_MODACC; case 12 * 3 + 1: return K_AMNQ2;
        case 13 * 3 + 0: case 15 * 3 + 0: case 16 * 3 + 0: return CMOD;          // is_equal(c, mod_acc)
        case 13 * 3 + 1: case 15 * 3 + 1: case 16 * 3 + 1: return K_MODACC;
        case 18 * 3 + 0: case 20 * 3 + 0: case 21 * 3 + 0: return COUT;          // is_equal(carry, dup | acc_extra)
        case 18 * 3 + 1: case 20 * 3 + 1: case 21 * 3 + 1: return DUP;
        default: return 0;                                                       // 14, 17, 19, 22: flag bytes only
    }
}

// ---- fast path (a valid mul_mod): every cell of a column row is a COPY of a ready-made entry in LDS ----
// Constant entries (32 bytes): 0, 1, 2^w and, for columns 0, 1, >= 2, the five values of the accumulated_extra step.  Per-column
// entries (planes of 2L entries): AB, EQB, a_b (as its field element), SUM, NQ1 (32 bytes) and the carry, c (16 bytes: their
// high halves are zero).  A source code: bits 0-3 entry kind, bit 4 column c - 1 (the zero entry for c = 0).
enum : u32 { CE_ZERO = 0, CE_ONE, CE_BW, CE_K_ACCX, CE_K_QACC, CE_K_MODACC, CE_K_NQ2, CE_K_AMNQ2,     // constants (K_*: by min(c, 2))
             CE_AB, CE_EQB, CE_AMB, CE_SUM, CE_NQ1, CE_COUT, CE_CMOD, CE_M1 = 16 };
constexpr u32 CELLS_CONST_ENTRIES = 3 + 3 * 5;   // 0, 1, 2^w, then [3][ACCX, QACC, MODACC, NQ2, AMNQ2]
__host__ __device__ constexpr u32 cells_fast_src(u32 j, u32 k, bool last_col) {
    switch (j * 3 + k) {
        case 0 * 3 + 0: return CE_AB; case 0 * 3 + 1: return CE_EQB; case 0 * 3 + 2: return CE_AMB;
        case 1 * 3 + 0: return CE_AMB; case 1 * 3 + 1: return CE_COUT | CE_M1; case 1 * 3 + 2: return CE_SUM;
        case 2 * 3 + 0: return CE_COUT;
        case 3 * 3 + 0: return CE_CMOD;
        case 4 * 3 + 0: return CE_BW; case 4 * 3 + 1: return CE_COUT; case 4 * 3 + 2: return CE_NQ1;
        case 5 * 3 + 0: return CE_SUM; case 5 * 3 + 1: return CE_NQ1; case 5 * 3 + 2: return CE_CMOD;
        case 6 * 3 + 0: return CE_CMOD; case 6 * 3 + 1: return CE_CMOD;
        case 7 * 3 + 0: return CE_K_QACC | CE_M1; case 7 * 3 + 1: return CE_K_ACCX;
        case 8 * 3 + 0: return CE_K_QACC;
        case 9 * 3 + 0: return CE_K_MODACC;
        case 10 * 3 + 0: return CE_BW; case 10 * 3 + 1: return CE_K_QACC; case 10 * 3 + 2: return CE_K_NQ2;
        case 11 * 3 + 0: return CE_K_ACCX; case 11 * 3 + 1: return CE_K_NQ2; case 11 * 3 + 2: return CE_K_AMNQ2;
        case 12 * 3 + 0: return CE_K_MODACC; case 12 * 3 + 1: return CE_K_AMNQ2;
        // a valid mul_mod: every comparison holds -- d = 0, its "inverse" witness 1, every flag 1
        case 13 * 3 + 0: return CE_CMOD; case 13 * 3 + 1: return CE_K_MODACC;                  // sub [c, mod_acc, 0]
        case 14 * 3 + 0: case 14 * 3 + 1: case 14 * 3 + 2: return CE_ONE;                       // bit
        case 15 * 3 + 1: case 15 * 3 + 2: return CE_ONE;                                         // [0, 1, 1]
        case 16 * 3 + 0: return CE_ONE;                                                          // [1, 0]
        case 17 * 3 + 0: case 17 * 3 + 1: case 17 * 3 + 2: return CE_ONE;                       // and
        case 18 * 3 + 0: return CE_COUT; case 18 * 3 + 1: return last_col ? CE_K_QACC : CE_COUT;   // sub [carry, dup | acc_extra, 0]
        case 19 * 3 + 0: case 19 * 3 + 1: case 19 * 3 + 2: return CE_ONE;
        case 20 * 3 + 1: case 20 * 3 + 2: return CE_ONE;
        case 21 * 3 + 0: return CE_ONE;
        case 22 * 3 + 0: case 22 * 3 + 1: case 22 * 3 + 2: return CE_ONE;
        default: return CE_ZERO;
    }
}

// dynamic LDS of one wave (byte offsets): stage, operands, constants, column planes, flags, code tables
struct CellsLds { u32 ops, kt, ce, ab, eqb, sum, amb, nq1, cout, cmod, mab, meqb, msum, opsr, icout, fl, src, fsrc, stage, total; };
__host__ __device__ inline CellsLds cells_lds_plan(u32 limb_width, u32 L, bool mont = false, u32 nwv = 1) {
    // 64-bit limbs: 32-byte entries AB, EQB, SUM, a_b, NQ1 and 16-byte entries carry, c.  32-bit limbs (every value but a_b's field
    // element is below 2^128): 16-byte entries AB, EQB, SUM, a 32-byte a_b; NQ1, the carry and c are cut from the SUM entry on the way.
    // Montgomery cells are not the integers the column phase computes with: the integer planes AB, EQB, SUM stay what they are, and
    // every ready-made cell (AB, EQB, a_b, SUM, NQ1, carry, c) has a 32-byte entry of its own; so have the limbs of a, b, q, n (the
    // first two cells of every mul row), converted once per mul_mod.
    const bool w64 = limb_width == 64;
    const u32 es = w64 ? 32u : 16u, n = 2u * L + 1u;
    // one wave per workgroup: its 64-row stage in front; several: their stages BEHIND everything else (the fast path's source codes hold
    // LDS offsets / 16 in 13 bits: planes must stay below 128 KB)
    CellsLds p; u32 o = nwv > 1 ? 0u : 64u * ADVICE_ROW_BYTES;
    p.stage = 0;
    p.ops = o; o += 5u * L * 8u;
    p.kt = o; o += mont ? 0u : CELLS_KT_WORDS * 8u;    // (Montgomery: the table is read from global memory where it is needed)
    p.ce = o; o += CELLS_CONST_ENTRIES * 32u;
    if (mont) {
        // Montgomery cells: what bounds the kernel is how many waves a CU holds (it is VALU-issue bound and a wave alone on its SIMD
        // issues ~6 cycles per instruction: build-only 1.85 -> 1.57 -> 1.39 ms with 4 -> 5 -> 6 waves per CU), i.e. the LDS per wave:
        // <= 32,000 bytes for five, <= 26,880 for six (RSA-2048: 26,848).  Seven planes of 32-byte cells, one entry per column
        // (2L - 1); everything that is dead once the column phase starts LIVES IN THEM: the un-carried totals of mul(a, b) / mul(q, n)
        // (integers, 32-byte entries) in the AB / EQB planes' own slots -- lane c reads column c's integers and then writes column
        // c's cells --, the operand cells of the mul rows (4 L + 2 entries) in the NQ1 + COUT + CMOD planes.  The integer SUM plane
        // shrinks to the carries (16 bytes each, which the range rows need), and those lie over the limbs of a, b, q, n
        // ((2L - 1) x 16 <= 4L x 8 bytes: no mul row is built after the column phase; r stays, the eq_b rows read it).
        const u32 C = 2u * L - 1u, need = (4u * L + 2u + nwv + 2u) / 3u, ne = C > need ? C : need;   // (4L + 1 + nwv operand entries over three planes)
        p.mab = o; o += ne * 32u; p.meqb = o; o += ne * 32u; p.amb = o; o += ne * 32u; p.msum = o; o += ne * 32u;
        p.nq1 = o; o += ne * 32u; p.cout = o; o += ne * 32u; p.cmod = o; o += ne * 32u;
        p.ab = p.mab; p.eqb = p.meqb; p.sum = p.msum;   // (p.sum: no integer SUM plane -- unused)
        p.opsr = p.nq1;
        p.icout = p.ops;
    } else {
        p.ab = o; o += n * es; p.eqb = o; o += n * es; p.sum = o; o += n * es;
        p.amb = o; o += n * 32u; p.nq1 = o; o += w64 ? n * 32u : 0u;
        p.cout = o; o += w64 ? n * 16u : 0u; p.cmod = o; o += w64 ? n * 16u : 0u;
        p.mab = p.meqb = p.msum = p.opsr = p.icout = o;
    }
    p.fl = o; o += (2u * L + 3u) & ~3u;                 // one byte per column
    p.src = o; o += mont ? 0u : CELLS_SRC_WORDS * 4u;   // (Montgomery: only an inconsistent mul_mod reads the codes -- computed there)
    p.fsrc = o; o += 2u * CELLS_SRC_WORDS * 4u;
    p.total = (o + 15u) & ~15u;
    if (nwv > 1) { p.stage = p.total; p.total += nwv * 64u * ADVICE_ROW_BYTES; }
    return p;
}
__host__ __device__ inline u32 cells_lds_bytes(u32 limb_width, u32 L, bool mont = false, u32 nwv = 1) { return cells_lds_plan(limb_width, L, mont, nwv).total; }
// A fast-path source as the kernel wants it: bits 0-12 LDS offset / 16 of the plane's (or constant's) first entry, bits 13-16 entry
// stride / 16, bit 17 column c - 1, bit 18 indexed by min(column, 2) (the accumulated_extra constants), bit 19 the entry has a high
// half, bits 20-23 (32-bit limbs, canonical cells) the cut of the SUM entry (CE_NQ1 / CE_COUT / CE_CMOD)
__host__ __device__ inline u32 cells_pack_fast_src(const CellsLds &lp, u32 limb_width, u32 code, bool mont = false) {
    const u32 kind = code & 15u, m1 = (code & CE_M1) ? 1u : 0u;
    u32 base, stride = 0, byk = 0, hi = 1, xf = 0;
    if (kind < 3) base = lp.ce + kind * 32;
    else if (kind < CE_AB) { base = lp.ce + (3 + (kind - 3)) * 32; stride = 5 * 32; byk = 1; }
    else if (mont) {
        base = kind == CE_AB ? lp.mab : kind == CE_EQB ? lp.meqb : kind == CE_AMB ? lp.amb : kind == CE_SUM ? lp.msum : kind == CE_NQ1 ? lp.nq1 : kind == CE_COUT ? lp.cout : lp.cmod;
        stride = 32;
    } else if (limb_width == 64) {
        if (kind < CE_COUT) { base = kind == CE_AB ? lp.ab : kind == CE_EQB ? lp.eqb : kind == CE_AMB ? lp.amb : kind == CE_SUM ? lp.sum : lp.nq1; stride = 32; }
        else { base = kind == CE_COUT ? lp.cout : lp.cmod; stride = 16; hi = 0; }
    } else {
        if (kind == CE_AMB) { base = lp.amb; stride = 32; }
        else { base = kind == CE_AB ? lp.ab : kind == CE_EQB ? lp.eqb : lp.sum; stride = 16; hi = 0; xf = kind >= CE_NQ1 ? kind : 0; }
    }
    return (base / 16) | ((stride / 16) << 13) | (m1 << 17) | (byk << 18) | (hi << 19) | (xf << 20);
}

// Segmented inclusive scan of an NWD-dword unsigned value over the 64 lanes: lane l receives the sum of the values of lanes
// [h, l], h = the nearest lane <= l with head set.  Kogge-Stone inside the 16-lane DPP rows (row_shr 1/2/4/8), then the rows are
// chained with row_bcast 15 (rows 1, 3) and row_bcast 31 (rows 2, 3); the head flags travel with the values.
template <int NWD>
__device__ __forceinline__ void cells_seg_scan(u32 (&v)[NWD], bool head) {
    u32 f = head ? 1u : 0u;
    auto step = [&](auto ctrl_c, auto mask_c) {
        constexpr int ctrl = decltype(ctrl_c)::value, rmask = decltype(mask_c)::value;
        u32 in[NWD];
#pragma unroll
        for (int k = 0; k < NWD; ++k) in[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)v[k], ctrl, rmask, 0xf, false);
        const u32 fin = (u32)__builtin_amdgcn_update_dpp(0, (int)f, ctrl, rmask, 0xf, false);
        const u32 m = f ? 0u : ~0u;
        u32 c = 0;
#pragma unroll
        for (int k = 0; k < NWD; ++k) v[k] = __builtin_addc(v[k], in[k] & m, c, &c);
        f |= fin;
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1 and 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2 and 3
}

// Inclusive prefix sums of an NWD-dword value over the 64 lanes (no segments: the caller subtracts the sum in front of a lane's
// segment, fetched with ds_bpermute); the sums of 64 limb products cannot overflow NWD dwords.  One DPP add-with-carry per dword and
// step: row_shr 1/2/4/8 inside the 16-lane rows (lanes without a source add zero), then row_bcast 15 / 31 across them.  (Written as
// assembly: from the builtin the compiler makes a zero move, a DPP move and an add per dword.  The s_nop covers the two wait
// states between a VALU write of a register and a DPP read of it, which the assembler does not insert in inline code.)
#define H2R_DPP_STEP3(ctrl_) \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl_ "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl_ "\n\tv_addc_co_u32_dpp %2, vcc, %2, %2, vcc " ctrl_ "\n\ts_nop 1\n\t"
#define H2R_DPP_STEP5(ctrl_) \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl_ "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl_ "\n\tv_addc_co_u32_dpp %2, vcc, %2, %2, vcc " ctrl_ \
    "\n\tv_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl_ "\n\tv_addc_co_u32_dpp %4, vcc, %4, %4, vcc " ctrl_ "\n\ts_nop 1\n\t"
#define H2R_DPP_SHR(n_) "row_shr:" #n_ " row_mask:0xf bank_mask:0xf bound_ctrl:0"
template <int NWD>
__device__ __forceinline__ void cells_prefix_sum(u32 (&v)[NWD]) {
    static_assert(NWD == 3 || NWD == 5, "dwords of a column sum");
    if constexpr (NWD == 3) {
        asm volatile("s_nop 1\n\t" H2R_DPP_STEP3(H2R_DPP_SHR(1)) H2R_DPP_STEP3(H2R_DPP_SHR(2)) H2R_DPP_STEP3(H2R_DPP_SHR(4)) H2R_DPP_STEP3(H2R_DPP_SHR(8))
                     H2R_DPP_STEP3("row_bcast:15 row_mask:0xa bank_mask:0xf") H2R_DPP_STEP3("row_bcast:31 row_mask:0xc bank_mask:0xf")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) :: "vcc");
    } else {
        asm volatile("s_nop 1\n\t" H2R_DPP_STEP5(H2R_DPP_SHR(1)) H2R_DPP_STEP5(H2R_DPP_SHR(2)) H2R_DPP_STEP5(H2R_DPP_SHR(4)) H2R_DPP_STEP5(H2R_DPP_SHR(8))
                     H2R_DPP_STEP5("row_bcast:15 row_mask:0xa bank_mask:0xf") H2R_DPP_STEP5("row_bcast:31 row_mask:0xc bank_mask:0xf")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]) :: "vcc");
    }
}

// ABL (developer ablations, tools/cells_bench.hip; 0 in the library): 1 no row building, 2 no global stores, 4 plain instead of
// non-temporal stores, 8 no is_equal_muled rows (zero rows) and no column phase, 16 no mul rows, 64 no fast paths, 128 chunks not
// aligned to 128-byte lines, 512 workgroup = item (no XCD-contiguous mapping)
// MONT (H2R_ADVICE_MONTGOMERY): every cell is x * R mod p.  What that costs is one SHORT Montgomery product per value (mont_short,
// h2r_field.hpp: 17 multiply-adds per 32-bit digit of x) and the structure of the image keeps the number of values small:
//   * the limbs of a, b, q, n -- the first two cells of the 2 L^2 mul rows -- are converted once per mul_mod into an LDS plane;
//   * a mul row's accumulator is converted once (5 digits: 133 bits; 3 for 32-bit limbs) and the NEXT row's "previous accumulator"
//     cell is the same value, taken from the lane above (the chunk's first lane: from the previous chunk's last);
//   * the is_equal_muled planes hold ready-made cells, so their rows stay copies; constants are converted once per wave;
//   * sub-limbs are one-digit products.
// The planar form (dst.planar(): one contiguous vector per column) only changes how the staged chunk leaves: five 2 KB runs
// instead of one 10 KB run, each store instruction still a whole number of 128-byte lines.
// NWV > 1: a workgroup of NWV waves per mul_mod SHARING one set of planes (each wave has a stage of its own).  The planes are what
// limits the waves a CU holds for the long shapes in Montgomery form (7 x (2L - 1) x 32 bytes: 57 KB for 128 limbs -- two one-wave
// workgroups per CU); eight waves around one set are 137 KB.  The chunks before the one that runs the column phase are split into NWV
// contiguous runs (a wave entering the mul rows in the middle first recomputes the running sum of the column it enters: a silent pass over
// the chunks that column can reach back into), that chunk is wave 0's, the chunks behind it are split again; workgroup barriers in between.
template <int LW, int ABL = 0, bool MONT = false, int NWV = 1>
__global__ __launch_bounds__(64 * NWV) void cells_kernel(CellsArgs a) {
    using limb_t = typename LimbT<LW>::type;
    constexpr bool FAST = !(ABL & 64);
    constexpr int WW = LW == 64 ? 3 : 2;     // 64-bit words of a wide value
    constexpr int ESW = (LW == 64 || MONT) ? 4 : 2;    // 64-bit words of a plane entry
    constexpr int NWD = LW == 64 ? 5 : 3;    // dwords of a running column sum (133 / 71 bits)
    constexpr int WBITS = LW == 64 ? 150 : 90;   // MONT: bound of every wide value's magnitude (five / three 30-bit digits)
    constexpr u64 LMASK = LW == 64 ? ~0ull : 0xffffffffull;
    constexpr u32 NP = ADVICE_ROW_BYTES / 16;   // 16-byte pieces of a row
    extern __shared__ uint4 cells_smem[];
    const u32 lane = threadIdx.x & 63u, wv = NWV > 1 ? threadIdx.x >> 6 : 0u;
    const u32 L = a.L, L2 = 2 * L, C = 2 * L - 1;
    const CellsLds lp = cells_lds_plan(LW, L, MONT, NWV);
    u8 *smem = reinterpret_cast<u8 *>(cells_smem);
    uint4 *stage = reinterpret_cast<uint4 *>(smem + lp.stage) + (u64)wv * 64 * (ADVICE_ROW_BYTES / 16);   // this wave's 64 rows x 160 bytes
    auto wg_sync = [&]() { if constexpr (NWV > 1) __syncthreads(); else wave_sync(); };
    u64 *sa = reinterpret_cast<u64 *>(smem + lp.ops), *sb = sa + L, *sq = sb + L, *sn = sq + L, *sr = sn + L;
    u64 *kt_lds = reinterpret_cast<u64 *>(smem + lp.kt);
    const u64 *kt = MONT ? a.ktab : kt_lds;                           // (Montgomery: no LDS copy -- a handful of global reads per column)
    u64 *pAB = reinterpret_cast<u64 *>(smem + lp.ab), *pEQB = reinterpret_cast<u64 *>(smem + lp.eqb), *pSUM = reinterpret_cast<u64 *>(smem + lp.sum);
    u8 *pFL = smem + lp.fl;                                           // per column: bit 0 f1, 1 e1, 2 f2, 3 e2
    u32 *s_src = reinterpret_cast<u32 *>(smem + lp.src), *f_src = reinterpret_cast<u32 *>(smem + lp.fsrc);
    // the column phase's scratch lives in the stage (free between two chunks)
    u64 *xDH0 = reinterpret_cast<u64 *>(stage), *xDH1 = xDH0 + L2, *xSLO = xDH1 + L2;
    u32 *xSHI = reinterpret_cast<u32 *>(xSLO + L2);

    u32 item = (ABL & 512) ? blockIdx.x : xcd_contiguous_block(blockIdx.x, gridDim.x);
    if constexpr (ABL & 1024) {
        const u32 g = gridDim.x / a.spread;
        item = blockIdx.x < g * a.spread ? (blockIdx.x % a.spread) * g + blockIdx.x / a.spread : blockIdx.x;
    }
    const u32 elem = item / a.T, t = item - elem * a.T;
    if (a.status && a.status[elem]) return;
    {
        const u64 ib = (u64)item * a.op_stride, iq = (u64)item * a.qr_stride;
        for (u32 k = threadIdx.x; k < L; k += 64 * NWV) {
            sa[k] = reinterpret_cast<const limb_t *>(a.opA)[ib + k]; sb[k] = reinterpret_cast<const limb_t *>(a.opB)[ib + k];
            sq[k] = reinterpret_cast<const limb_t *>(a.opQ)[iq + k]; sr[k] = reinterpret_cast<const limb_t *>(a.opR)[iq + k];
            sn[k] = reinterpret_cast<const limb_t *>(a.n)[(u64)elem * a.n_stride + k];
        }
        if constexpr (!MONT) { if (threadIdx.x < CELLS_KT_WORDS) kt_lds[threadIdx.x] = a.ktab[threadIdx.x]; }
    }
    // MONT: the multipliers this shape needs, in registers for the whole kernel (read back from an LDS copy so that they ARE vector
    // registers: as kernel arguments they are SGPRs, the kernel has none to spare, and the spills sat in the middle of every product)
    constexpr int DA = (WBITS + 29) / 30, DC = LW == 64 ? 3 : 2;   // 30-bit digits of a wide value / of a limb, a carry
    struct { u32 p30[9], p32[8], n0, n0_32, bA[9], bC[9], b1[9]; } mv;
    if constexpr (MONT) {
        u32 *lmk = reinterpret_cast<u32 *>(stage);   // (the stage is free until the first chunk)
        const u32 *gmk = reinterpret_cast<const u32 *>(a.mk);
        for (u32 k = lane; k < sizeof(MontK) / 4; k += 64) lmk[k] = gmk[k];
        wave_sync();
        const MontK *m = reinterpret_cast<const MontK *>(lmk);
#pragma unroll
        for (int j = 0; j < 9; ++j) { mv.p30[j] = m->p30[j]; mv.bA[j] = m->bk30[DA][j]; mv.bC[j] = m->bk30[DC][j]; mv.b1[j] = m->bk30[1][j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) mv.p32[j] = m->p[j];
        mv.n0 = m->n0inv30; mv.n0_32 = m->n0inv;
    }
    const bool planar = a.dst.planar();
    u8 *img = a.dst.elem(elem);
    // `out`: the record's first row (its column-a cell); the other columns of a row lie col_pitch apart, the next row row_pitch further
    u8 *out = img + ((u64)a.pre_rows + (u64)t * a.rows + (u64)((t + 1) >> 1) * a.sel_rows) * a.dst.row_pitch;
    if (t == 0 && wv == 0 && lane < a.pre_rows * 5) {   // pow_mod_fixed_exp's acc = assign_constant(1, L): [1, 0, 0, 0, 0] then [0, ...]
        uint4 *pr = reinterpret_cast<uint4 *>(img + (u64)(lane / 5) * a.dst.row_pitch + (u64)(lane % 5) * a.dst.col_pitch);
        if (MONT && lane == 0) { const u32 *one = reinterpret_cast<const MontK *>(stage)->bk[0]; pr[0] = make_uint4(one[0], one[1], one[2], one[3]); pr[1] = make_uint4(one[4], one[5], one[6], one[7]); }
        else { pr[0] = make_uint4(lane == 0 ? 1u : 0u, 0, 0, 0); pr[1] = make_uint4(0, 0, 0, 0); }
    }
    const U192 Z = U192::make(0, 0, 0);
    const U192 Bw = LW == 64 ? U192::make(0, 1, 0) : U192::make(1ull << 32, 0, 0);   // 2^w
    // (word_max and the field modulus are read from the LDS copy of the table where they are needed: kernel arguments live in SGPRs)
    const uint4 Z4 = make_uint4(0, 0, 0, 0);
    auto lim = [&](u64 v) { return U192::make(v, 0, 0); };
    auto rdp = [&](const u64 *pl, u32 c) -> U192 { return U192::make(pl[(u64)c * ESW], pl[(u64)c * ESW + 1], WW == 3 ? pl[(u64)c * ESW + 2] : 0); };
    auto wrp = [&](u64 *pl, u32 c, const U192 &v) {   // (a whole entry: the fast path copies it as two 16-byte pieces)
        pl[(u64)c * ESW] = v.w[0]; pl[(u64)c * ESW + 1] = v.w[1];
        if constexpr (ESW == 4) { pl[(u64)c * ESW + 2] = v.w[2]; pl[(u64)c * ESW + 3] = 0; }
    };
    auto shr_limb = [&](const U192 &v) -> U192 { return v.shr(LW); };
    auto field4 = [&](const U192 &v, bool is_signed, u64 (&x)[4]) {   // canonical field element of a (two's complement) value
        x[0] = v.w[0]; x[1] = v.w[1]; x[2] = v.w[2]; x[3] = 0;
        if (is_signed && (v.w[2] >> 63)) {   // x < 0 -> p + x (mod 2^256)
            x[3] = ~0ull;
            u64 cy = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u64 s1 = x[k] + kt[CELLS_KT_P + k]; const u64 c1 = s1 < x[k]; const u64 s2 = s1 + cy; cy = c1 | (u64)(s2 < s1); x[k] = s2; }
        }
    };
    // Montgomery form of a (two's complement) value whose magnitude is below 2^(32 NWD): every value of this path
    auto mont_wide = [&](const U192 &v, bool is_signed, u32 (&tt)[8]) {
        const bool neg = is_signed && (v.w[2] >> 63) != 0;
        const U192 m = neg ? Z - v : v;
        u32 x[NWD];
        x[0] = (u32)m.w[0]; x[1] = (u32)(m.w[0] >> 32); x[2] = (u32)m.w[1];
        if constexpr (NWD == 5) { x[3] = (u32)(m.w[1] >> 32); x[4] = (u32)m.w[2]; }
        if constexpr (ABL & 16384) { for (int w_ = 0; w_ < 8; ++w_) tt[w_] = x[w_ % NWD]; }   // (developer ablation: every conversion free -- wrong cells, timing only)
        else { u32 d[DA]; digits30<DA, NWD>(x, d); mont30<DA>(d, mv.bA, mv.p30, mv.n0, mv.p32, tt); }
        if (neg) mont_neg(tt, mv.p32);
    };
    auto cell = [&](uint4 *p, const U192 &v, bool is_signed) {   // 32 bytes little-endian
        if constexpr (MONT) {
            u32 tt[8];
            mont_wide(v, is_signed, tt);
            p[0] = make_uint4(tt[0], tt[1], tt[2], tt[3]); p[1] = make_uint4(tt[4], tt[5], tt[6], tt[7]);
        } else {
            u64 x[4];
            field4(v, is_signed, x);
            p[0] = make_uint4((u32)x[0], (u32)(x[0] >> 32), (u32)x[1], (u32)(x[1] >> 32));
            p[1] = make_uint4((u32)x[2], (u32)(x[2] >> 32), (u32)x[3], (u32)(x[3] >> 32));
        }
    };
    // the cell of an unsigned value (lo, hi) below 2^BITS
    auto cell_k = [&](auto kc, u64 lo, u64 hi, uint4 &o0, uint4 &o1) {
        constexpr int BITS = decltype(kc)::value, K = (BITS + 31) / 32;
        if constexpr (MONT) {
            u32 x[K], tt[8];
            x[0] = (u32)lo;
            if constexpr (K > 1) x[1] = (u32)(lo >> 32);
            if constexpr (K > 2) x[2] = (u32)hi;
            if constexpr (K > 3) x[3] = (u32)(hi >> 32);
            constexpr int D = (BITS + 29) / 30;
            static_assert(D == 1 || D == DC, "a sub-limb, or a limb / carry");
            u32 d[D];
            digits30<D, K>(x, d);
            if constexpr (ABL & 16384) { for (int w_ = 0; w_ < 8; ++w_) tt[w_] = d[w_ % D]; }
            else mont30<D>(d, D == 1 ? mv.b1 : mv.bC, mv.p30, mv.n0, mv.p32, tt);
            o0 = make_uint4(tt[0], tt[1], tt[2], tt[3]); o1 = make_uint4(tt[4], tt[5], tt[6], tt[7]);
        } else {
            o0 = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)); o1 = make_uint4(0, 0, 0, 0);
        }
    };
    // ---- MONT: cells of the planes as operands (the integer planes are gone once the column phase has run) ----
    // x - y mod p on cells
    auto cell_sub = [&](const uint4 &x0, const uint4 &x1, const uint4 &y0, const uint4 &y1, uint4 &o0, uint4 &o1) {
        const u32 x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w}, y[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        u32 d[8], e[8]; u64 br = 0, cy = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u64 t_ = (u64)x[j] - y[j] - br; d[j] = (u32)t_; br = (t_ >> 32) & 1u; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u64 t_ = (u64)d[j] + mv.p32[j] + cy; e[j] = (u32)t_; cy = t_ >> 32; }
        o0 = br ? make_uint4(e[0], e[1], e[2], e[3]) : make_uint4(d[0], d[1], d[2], d[3]);
        o1 = br ? make_uint4(e[4], e[5], e[6], e[7]) : make_uint4(d[4], d[5], d[6], d[7]);
    };
    // the integer a cell stands for (x R -> x: one Montgomery product by 1).  Only a mul_mod whose q, r are NOT its quotient and remainder
    // comes here (never produced by this library): its column rows are rebuilt from the integers, is_zero's inverses and all.
    auto cell_int = [&](const uint4 *cp, bool is_signed) -> U192 {
        const uint4 c0 = cp[0], c1 = cp[1];
        const u32 x[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
        u32 tt[8];
        mont_short<8>(x, one, mv.p32, mv.n0_32, tt);
        if (is_signed && (tt[6] | tt[7])) {   // p - |x|: the negative difference
            u64 br = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const u64 t_ = (u64)tt[j] - mv.p32[j] - br; tt[j] = (u32)t_; br = (t_ >> 32) & 1u; }
        }
        return U192::make(((u64)tt[1] << 32) | tt[0], ((u64)tt[3] << 32) | tt[2], ((u64)tt[5] << 32) | tt[4]);
    };
    // the three cells of row k (0..22, the carry's range rows taken out) of column c of a VALID mul_mod: copies of plane entries
    auto col_row_copy = [&](u32 c, u32 k, uint4 (&o)[6]) {
        const bool lastc = c == C - 1;
        const u32 *codes = f_src + (lastc ? CELLS_SRC_WORDS : 0u) + k * 3;
        const u32 kc = c < 2 ? c : 2, kcp = c < 3 ? c - 1 : 2;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const u32 code = codes[q];
            const bool m1 = (code >> 17) & 1u, byk = (code >> 18) & 1u;
            const u32 idx = byk ? (m1 ? kcp : kc) : (m1 ? c - 1 : c);
            const bool zero = m1 && c == 0;
            const u32 addr = zero ? lp.ce : ((code & 0x1fffu) + idx * ((code >> 13) & 15u)) << 4;
            const u32 hi_off = (!zero && ((code >> 19) & 1u)) ? addr + 16 : lp.ce;
            o[2 * q] = *reinterpret_cast<const uint4 *>(smem + addr); o[2 * q + 1] = *reinterpret_cast<const uint4 *>(smem + hi_off);
        }
    };
    // row rr of RangeChip::assign of the value v (nsub sub-limbs of sub_bits bits, the last one possibly shorter): four sub-limbs in
    // columns a..d -- the LAST row reversed, so that the last (overflow) term is in column a, and zero-padded -- and in column e what
    // remains to be composed (main_gate.decompose)
    // (64-bit limbs: sub-limbs are bytes -- 8 of a limb, 8 + a shorter ninth of a carry; 32-bit limbs: the value fits 64 bits)
    auto range_vals = [&](u64 v_lo, u64 v_hi, u32 nsub, u32 sub_bits, u32 rr, u64 &c0, u64 &c1, u64 &c2, u64 &c3, u64 &rem_lo, u64 &rem_hi) {
        const u32 last = (nsub - 1) / 4, n_last = nsub - 4 * last;
        auto sub = [&](u32 k) -> u64 {
            if constexpr (LW == 64) return k < 8 ? (v_lo >> (8 * k)) & 0xff : (k == 8 ? v_hi & 0xff : 0);
            else return k < nsub ? (v_lo >> (k * sub_bits)) & ((1u << sub_bits) - 1) : 0;
        };
        const bool rev = rr >= last;
        const u32 k0 = rev ? nsub - 1 : 4 * rr;
        c0 = sub(k0);
        c1 = rev ? (n_last > 1 ? sub(k0 - 1) : 0) : sub(k0 + 1);
        c2 = rev ? (n_last > 2 ? sub(k0 - 2) : 0) : sub(k0 + 2);
        c3 = rev ? (n_last > 3 ? sub(k0 - 3) : 0) : sub(k0 + 3);
        if constexpr (LW == 64) {   // low bytes already composed: cleared (4 rr bytes)
            rem_lo = rr == 0 ? v_lo : (rr == 1 ? v_lo & ~0xffffffffull : 0);
            rem_hi = rr <= 2 ? v_hi : 0;
        } else {
            const u32 cl = 4 * rr * sub_bits;
            rem_lo = cl >= 64 ? 0 : (v_lo >> cl) << cl;
            rem_hi = 0;
        }
    };
    if constexpr (!MONT) { for (u32 k = threadIdx.x; k < ADVICE_COL_ROWS * 3; k += 64 * NWV) s_src[k] = cells_col_src(k / 3, k % 3); }
    if constexpr (FAST) {
        const u32 *packed = reinterpret_cast<const u32 *>(a.ktab + CELLS_KT_FSRC);
        for (u32 k = threadIdx.x; k < 2 * CELLS_SRC_WORDS; k += 64 * NWV) f_src[k] = packed[k];
    }
    wg_sync();
    if constexpr (MONT) {   // the limbs of a, b, q, n as cells: what the mul rows' first two columns hold
        uint4 *opsr = reinterpret_cast<uint4 *>(smem + lp.opsr);
        for (u32 k = threadIdx.x; k < 4 * L; k += 64 * NWV) {
            const u32 which = k / L, idx = k - which * L;
            const u64 v = (which == 0 ? sa : which == 1 ? sb : which == 2 ? sq : sn)[idx];
            cell_k(std::integral_constant<int, LW>{}, v, 0, opsr[2 * k], opsr[2 * k + 1]);
        }
        if (threadIdx.x < 4) opsr[2 * 4 * L + threadIdx.x] = Z4;
    }
    if constexpr (FAST) {   // the constant entries
        uint4 *ce = reinterpret_cast<uint4 *>(smem + lp.ce);
        if (wv == 0 && lane < CELLS_CONST_ENTRIES) {
            U192 v = Z;
            if (lane == CE_ONE) v = lim(1);
            else if (lane == CE_BW) v = Bw;
            else if (lane >= 3) {
                const u32 kc = (lane - 3) / 5, s = (lane - 3) % 5;
                const u64 *e = kt + kc * 10;
                v = s == 0 ? U192::make(e[0], e[1], e[2]) : s == 1 ? U192::make(e[3], e[4], 0) : s == 2 ? lim(e[5]) : s == 3 ? U192::make(e[6], e[7], e[8]) : lim(e[9]);
            }
            cell(ce + 2 * lane, v, false);
        }
    }

    // ---- the 2L - 1 un-carried columns: eq_b, a_b, the carries and the running eq_bit (chip.rs:614-623, 857-893) -> planes ----
    bool item_ok = true;
    // (NWV > 1: every wave comes here -- the chunk's owner from inside the chunk, the others from the phase loop.  The owner prepares the
    //  integers; then each wave walks ALL columns' integer arithmetic, which is cheap, and converts the cells of the planes that are its own:
    //  seven planes over the waves instead of seven conversions per column in one wave)
    auto column_phase = [&](const bool owner) {
        const U192 Wm = U192::make(kt[CELLS_KT_WM], kt[CELLS_KT_WM + 1], kt[CELLS_KT_WM + 2]);
        if constexpr (NWV > 1) __syncthreads();       // the totals the other waves' mul rows left
        u64 *const sDH0 = NWV > 1 ? reinterpret_cast<u64 *>(smem + lp.stage) : xDH0, *const sDH1 = sDH0 + L2, *const sSLO = sDH1 + L2;   // (the owner's stage)
        u32 *const sSHI = reinterpret_cast<u32 *>(sSLO + L2);
        if (owner) {
        for (u32 c = lane; c < C; c += 64) {
            const U192 A = rdp(pAB, c);
            U192 Q = rdp(pEQB, c);                 // (holds the q*n column until here)
            if (c < L) { Q = Q + lim(sr[c]); wrp(pEQB, c, Q); }                // eq_b[i] = qn[i] + r[i]  :617
            const U192 D = (A - Q) + Wm;          // a_b + word_max >= 0  :859-860
            const U192 dhi = shr_limb(D);
            sSLO[c] = D.w[0] & LMASK; sDH0[c] = dhi.w[0]; sDH1[c] = dhi.w[1];
        }
        wave_sync();
        for (u32 c = lane; c < C; c += 64) {
            const U192 S = lim(sSLO[c]) + (c ? U192::make(sDH0[c - 1], sDH1[c - 1], 0) : Z);
            sSLO[c] = S.w[0] & LMASK;
            sSHI[c] = (u32)shr_limb(S).w[0];
        }
        wave_sync();
        }
        if constexpr (NWV > 1) __syncthreads();
        auto mine = [&](u32 plane) -> bool { return NWV == 1 || plane % NWV == wv; };   // plane: 0 amb, 1 mab, 2 meqb, 3 msum, 4 nq1, 5 cout, 6 cmod
        bool cin = false, all_ok = true;
        for (u32 cb = 0; cb < C; cb += 64) {
            const u32 c = cb + lane;
            const bool col = c < C;
            const u64 slo = col ? sSLO[c] : 0;
            const u32 shp = (col && c) ? sSHI[c - 1] : 0;
            const u128 U = (u128)slo + shp;
            const bool gen = col && (U >> LW) != 0, prop = col && ((u64)U & LMASK) == LMASK;
            const CarryGroup cg = carry_group(__ballot(gen), __ballot(prop), cin, 64);
            const bool f = ((cg.cin_mask >> lane) & 1) != 0;
            cin = cg.cout;
            bool f1 = true, f2 = true;
            const U192 iA = col ? rdp(pAB, c) : Z, iQ = col ? rdp(pEQB, c) : Z;
            if constexpr (NWV > 1 && MONT) __syncthreads();   // (the integers of this block are read by every wave before any wave's cells replace them)
            if (col) {
                const U192 dhp = c ? U192::make(sDH0[c - 1], sDH1[c - 1], 0) : Z;
                const U192 carry_in = dhp + lim((u64)shp + (f ? 1u : 0u));
                const U192 amb = iA - iQ;
                const U192 sum = amb + Wm + carry_in;                          // :860-861
                const U192 cout = shr_limb(sum);
                if (owner) {
                    if constexpr (MONT) reinterpret_cast<uint4 *>(smem + lp.icout)[c] = make_uint4((u32)cout.w[0], (u32)(cout.w[0] >> 32), (u32)cout.w[1], (u32)(cout.w[1] >> 32));
                    else wrp(pSUM, c, sum);
                }
                const u32 kc = c < 2 ? c : 2;
                f1 = (sum.w[0] & LMASK) == kt[kc * 10 + 5];                      // cs_acc_eq  :873
                if (c == C - 1) f2 = cout.w[0] == kt[kc * 10 + 3] && cout.w[1] == kt[kc * 10 + 4];   // final_carry_eq  :890
                if constexpr (FAST) { if (mine(0)) cell(reinterpret_cast<uint4 *>(smem + lp.amb) + 2 * c, amb, true); }   // the ready-made cells of the column
                if constexpr (FAST && MONT) {
                    if (mine(1)) cell(reinterpret_cast<uint4 *>(smem + lp.mab) + 2 * c, iA, false);
                    if (mine(2)) cell(reinterpret_cast<uint4 *>(smem + lp.meqb) + 2 * c, iQ, false);
                    if (mine(3)) cell(reinterpret_cast<uint4 *>(smem + lp.msum) + 2 * c, sum, false);
                    if (mine(4)) cell(reinterpret_cast<uint4 *>(smem + lp.nq1) + 2 * c, U192::make(sum.w[0] & ~LMASK, sum.w[1], sum.w[2]), false);
                    if (mine(5)) cell(reinterpret_cast<uint4 *>(smem + lp.cout) + 2 * c, cout, false);
                    if (mine(6)) cell(reinterpret_cast<uint4 *>(smem + lp.cmod) + 2 * c, lim(sum.w[0] & LMASK), false);
                } else if constexpr (FAST && LW == 64) {
                    if (owner) {
                    cell(reinterpret_cast<uint4 *>(smem + lp.nq1) + 2 * c, U192::make(sum.w[0] & ~LMASK, sum.w[1], sum.w[2]), false);
                    reinterpret_cast<uint4 *>(smem + lp.cout)[c] = make_uint4((u32)cout.w[0], (u32)(cout.w[0] >> 32), (u32)cout.w[1], (u32)(cout.w[1] >> 32));
                    const u64 cm = sum.w[0] & LMASK;
                    reinterpret_cast<uint4 *>(smem + lp.cmod)[c] = make_uint4((u32)cm, (u32)(cm >> 32), 0, 0);
                    }
                }
            }
            const u64 bad = __ballot(col && !(f1 && f2));
            const bool prev_ok = all_ok && (bad & ((1ull << lane) - 1)) == 0;
            if (col && owner) {
                const u32 e1 = (prev_ok && f1) ? 1u : 0u, e2 = (e1 && f2) ? 1u : 0u;
                pFL[c] = (u8)((f1 ? 1u : 0u) | (e1 << 1) | ((f2 ? 1u : 0u) << 2) | (e2 << 3));
            }
            all_ok = all_ok && bad == 0;
        }
        item_ok = all_ok;
        if constexpr (NWV > 1) { if (owner && lane == 0) pFL[2 * L - 1] = all_ok ? 1 : 0; }   // (read again by the other waves behind the phase's barrier)
        if constexpr (NWV > 1) __syncthreads(); else wave_sync();
    };

    const u32 mul_rows = C + L * L, r_T3 = 4 * L, r_T5 = r_T3 + 2 * mul_rows, r_T6 = r_T5 + L + 4;
    const u32 nrc = (a.carry_nsub + 3) / 4, per_col = ADVICE_COL_ROWS + nrc;
    u32 carry[NWD];
#pragma unroll
    for (int k = 0; k < NWD; ++k) carry[k] = 0;

    bool columns_done = false;
    u32 dec_r0 = ~0u, dec_I = 0, dec_k = 0, dec_len = 1;   // the mul rows' fast path: row r0 + lane = position dec_k of column dec_I (both muls: 2C columns)
    // the first chunk ends where the image reaches a 128-byte line (160 = 128 + 32: at most three rows), every later chunk of 64
    // rows (80 lines) then starts on one
    u32 n_first = 64;
    if constexpr (!(ABL & 128)) {
        const u32 mis = (u32)(reinterpret_cast<u64>(out) & 127u);
        if (mis && (mis & 31u) == 0) n_first = (128u - mis) / 32u;
    }
    auto chunk = [&](const u32 r0, const u32 n_rows) {
        const u32 r = r0 + lane;
        const bool valid = lane < n_rows;
        uint4 *srow = stage + (u64)lane * NP;
        bool built = (ABL & 1) != 0;
        u64 t_0 = 0; u32 path = 0;
        if constexpr (ABL & 256) t_0 = __builtin_amdgcn_s_memtime();
        // ================= fast path: a chunk of mul(a, b) / mul(q, n) rows (not the one that holds the last of them) =================
        if (!built && !(ABL & (16 | 64)) && n_rows == 64 && r0 >= r_T3 && r0 + 64 < r_T5) {
            // row -> (mul, column i, position k in the column: 0 = the column's constant 0).  The first such chunk decodes its rows
            // in closed form; every later one moves each lane's (column, position) 64 rows on.
            if (dec_r0 != r0) {
                u32 e = r - r_T3;
                const u32 qn0 = e >= mul_rows ? 1u : 0u;
                e -= qn0 * mul_rows;
                // Columns 0 .. L-1 hold 2, 3, ..., L + 1 rows (head + accumulators): column i starts at i (i + 3) / 2.  Columns
                // L .. 2L-2 mirror columns L-2 .. 0, so counted from the END of the section the same closed form applies.
                const bool back = e >= L * (L + 3) / 2;
                const u32 ee = back ? mul_rows - 1 - e : e;
                u32 ii = (u32)((sqrtf(8.f * (float)ee + 9.f) - 3.f) * 0.5f);
                if ((ii + 1) * (ii + 4) / 2 <= ee) ++ii;          // the float estimate is off by at most one
                else if (ii * (ii + 3) / 2 > ee) --ii;
                const u32 kk = ee - ii * (ii + 3) / 2;
                const u32 i0 = back ? C - 1 - ii : ii;
                dec_k = back ? ii + 1 - kk : kk;
                dec_I = qn0 * C + i0;
                dec_len = (i0 < L ? i0 : C - 1 - i0) + 2;
                if constexpr (MONT) {   // (the general path left the integer: its cell goes where a chunk's last row leaves its accumulator cell)
                    u32 d[DA], cr[8];
                    digits30<DA, NWD>(carry, d);
                    if constexpr (ABL & 16384) { for (int w_ = 0; w_ < 8; ++w_) cr[w_] = d[w_ % DA]; } else mont30<DA>(d, mv.bA, mv.p30, mv.n0, mv.p32, cr);
                    uint4 *slot = reinterpret_cast<uint4 *>(smem + lp.opsr) + 2 * (4 * L + 1 + wv);   // (a slot per wave)
                    if (lane == 0) { slot[0] = make_uint4(cr[0], cr[1], cr[2], cr[3]); slot[1] = make_uint4(cr[4], cr[5], cr[6], cr[7]); }
                }
            } else {
                dec_k += 64;
                while (__ballot(dec_k >= dec_len) != 0) {
                    if (dec_k >= dec_len) {
                        dec_k -= dec_len; ++dec_I;
                        const u32 i1 = dec_I >= C ? dec_I - C : dec_I;
                        dec_len = (i1 < L ? i1 : C - 1 - i1) + 2;
                    }
                }
            }
            dec_r0 = r0 + 64;
            const u32 qn = dec_I >= C ? 1u : 0u, i = dec_I - qn * C, k = dec_k;
            const bool is_ma = k != 0;
            const u32 jmin = i >= L ? i - L + 1 : 0, j = jmin + k - 1;
            u64 x = 0, y = 0;
            if (is_ma) { x = (qn ? sq : sa)[j]; y = (qn ? sn : sb)[i - j]; }
            u32 p[NWD], own[NWD];
#pragma unroll
            for (int w = 0; w < NWD; ++w) p[w] = 0;
            if constexpr (LW == 64) {
                const u128 pr = (u128)x * y;
                p[0] = (u32)pr; p[1] = (u32)(pr >> 32); p[2] = (u32)(pr >> 64); p[3] = (u32)(pr >> 96);
            } else {
                const u64 pr = (u64)(u32)x * (u32)y;
                p[0] = (u32)pr; p[1] = (u32)(pr >> 32);
            }
#pragma unroll
            for (int w = 0; w < NWD; ++w) own[w] = p[w];
            // the column's running sum = (prefix sum over the lanes) - (prefix sum at the column's head row: lane - k), or + what
            // the column had gathered in the previous chunk when it began there
            cells_prefix_sum<NWD>(p);
            {
                const bool here = k <= lane;
                u32 base[NWD], c = 0;
#pragma unroll
                for (int w = 0; w < NWD; ++w) {
                    const u32 sh = (u32)__builtin_amdgcn_ds_bpermute((int)((lane - k) << 2), (int)p[w]);
                    base[w] = here ? ~sh : carry[w];                 // p - sh = p + ~sh + 1
                }
                c = here ? 1u : 0u;
#pragma unroll
                for (int w = 0; w < NWD; ++w) p[w] = __builtin_addc(p[w], base[w], c, &c);
            }
#pragma unroll
            for (int w = 0; w < NWD; ++w) carry[w] = (u32)__builtin_amdgcn_readlane((int)p[w], 63);
            u32 q[NWD], br = 0;
#pragma unroll
            for (int w = 0; w < NWD; ++w) q[w] = __builtin_subc(p[w], own[w], br, &br);
            if (is_ma && k == (i < L ? i + 1 : C - i)) {   // the column's last row: its total
                u64 *pl = qn ? pEQB : pAB;
                if constexpr (LW == 64) wrp(pl, i, U192::make(((u64)p[1] << 32) | p[0], ((u64)p[3] << 32) | p[2], p[4]));
                else wrp(pl, i, U192::make(((u64)p[1] << 32) | p[0], p[2], 0));
            }
            // [x_j, y_{i-j}, acc_prev, acc, 0]  :408 (a column's head row: all zero)
            if constexpr (MONT) {
                u32 acc_r[8];
                if constexpr (ABL & (2048 | 16384)) { for (int w = 0; w < 8; ++w) acc_r[w] = p[w % NWD]; }   // (developer: no conversion)
                else { u32 d[DA]; digits30<DA, NWD>(p, d); mont30<DA>(d, mv.bA, mv.p30, mv.n0, mv.p32, acc_r); }
                // the limb cells are copies of the operand plane (a column's head row: of its zero entry)
                uint4 *opsr = reinterpret_cast<uint4 *>(smem + lp.opsr);
                const u32 zi = 4 * L, xi = is_ma ? (qn ? 2 * L : 0) + j : zi, yi = is_ma ? (qn ? 3 * L : L) + (i - j) : zi;
                const uint4 x0 = opsr[2 * xi], x1 = opsr[2 * xi + 1], y0 = opsr[2 * yi], y1 = opsr[2 * yi + 1];
                srow[0] = x0; srow[1] = x1; srow[2] = y0; srow[3] = y1;
                srow[6] = make_uint4(acc_r[0], acc_r[1], acc_r[2], acc_r[3]); srow[7] = make_uint4(acc_r[4], acc_r[5], acc_r[6], acc_r[7]);
                // acc_prev IS the row above's acc (position k - 1 of the same column; the first multiply-add starts from the constant 0): a
                // copy of the cell the lane above has just staged -- for lane 0 of the cell the previous chunk's lane 63 left in the slot
                wave_sync();
                const uint4 *pa = k < 2 ? opsr + 2 * zi : (lane == 0 ? opsr + 2 * (zi + 1 + wv) : srow - NP + 6);
                const uint4 pv0 = pa[0], pv1 = pa[1];
                wave_sync();
                if (lane == 63) { opsr[2 * (zi + 1 + wv)] = srow[6]; opsr[2 * (zi + 1 + wv) + 1] = srow[7]; }
                srow[4] = pv0; srow[5] = pv1;
            } else {
            srow[0] = make_uint4((u32)x, (u32)(x >> 32), 0, 0); srow[1] = Z4;
            srow[2] = make_uint4((u32)y, (u32)(y >> 32), 0, 0); srow[3] = Z4;
            if constexpr (LW == 64) {
                srow[4] = make_uint4(q[0], q[1], q[2], q[3]); srow[5] = make_uint4(q[4], 0, 0, 0);
                srow[6] = make_uint4(p[0], p[1], p[2], p[3]); srow[7] = make_uint4(p[4], 0, 0, 0);
            } else {
                srow[4] = make_uint4(q[0], q[1], q[2], 0); srow[5] = Z4;
                srow[6] = make_uint4(p[0], p[1], p[2], 0); srow[7] = Z4;
            }
            }
            srow[8] = Z4; srow[9] = Z4;
            built = true; path = 1;
        }
        // ================= fast path: a chunk of is_equal_muled column rows of a valid mul_mod =================
        if constexpr (FAST) {
            if (!built && !(ABL & 8) && columns_done && item_ok && n_rows == 64 && r0 >= r_T6 && r0 + 64 < a.rows) {   // (not the chunk that holds the closing assert_one row)
                const u32 rr = r - r_T6;
                const u32 c = __umulhi(rr, a.per_col_magic);
                u32 k = rr - c * per_col;
                const bool lastc = c == C - 1;
                const bool is_range = !lastc && k >= 18 && k < 18 + nrc;
                if (!lastc && k >= 18 + nrc) k -= nrc;
                const u32 *codes = f_src + (lastc ? CELLS_SRC_WORDS : 0u) + (is_range ? 0u : k * 3);
                const u32 kc = c < 2 ? c : 2, kcp = c < 3 ? c - 1 : 2;   // (kcp is used for c >= 1 only)
                auto entry = [&](u32 code, u32 &hi_off, u32 &xf) -> u32 {   // LDS byte address of the cell's entry; hi_off: of its high half
                    const bool m1 = (code >> 17) & 1u, byk = (code >> 18) & 1u;
                    const u32 idx = byk ? (m1 ? kcp : kc) : (m1 ? c - 1 : c);
                    const bool zero = m1 && c == 0;
                    const u32 addr = zero ? lp.ce : ((code & 0x1fffu) + idx * ((code >> 13) & 15u)) << 4;
                    hi_off = (!zero && ((code >> 19) & 1u)) ? addr + 16 : lp.ce;   // (no high half: the zero entry)
                    xf = zero ? 0u : (code >> 20) & 15u;
                    return addr;
                };
                auto cut = [&](const uint4 &l, u32 xf) -> uint4 {   // 32-bit limbs: value >> 32, value mod 2^32, value with its low limb cleared
                    if constexpr (LW == 64 || MONT) return l;
                    else return xf == CE_COUT ? make_uint4(l.y, l.z, l.w, 0) : xf == CE_CMOD ? make_uint4(l.x, 0, 0, 0) : xf == CE_NQ1 ? make_uint4(0, l.y, l.z, l.w) : l;
                };
                u32 h0, h1, h2, x0, x1, x2;
                const u32 a0 = entry(codes[0], h0, x0), a1 = entry(codes[1], h1, x1), a2 = entry(codes[2], h2, x2);
                const uint4 l0 = cut(*reinterpret_cast<const uint4 *>(smem + a0), x0), g0 = *reinterpret_cast<const uint4 *>(smem + h0);
                const uint4 l1 = cut(*reinterpret_cast<const uint4 *>(smem + a1), x1), g1 = *reinterpret_cast<const uint4 *>(smem + h1);
                const uint4 l2 = cut(*reinterpret_cast<const uint4 *>(smem + a2), x2), g2 = *reinterpret_cast<const uint4 *>(smem + h2);
                uint4 o[NP] = {l0, g0, l1, g1, l2, g2, Z4, Z4, Z4, Z4};
                if constexpr (!MONT) {
                if (is_range) {   // RangeChip::assign(carry, ...)  :880-885
                    uint4 cv;
                    if constexpr (LW == 64) cv = reinterpret_cast<const uint4 *>(smem + lp.cout)[c];
                    else cv = cut(reinterpret_cast<const uint4 *>(smem + lp.sum)[c], CE_COUT);
                    u64 c0, c1, c2, c3, rl, rh;
                    range_vals(((u64)cv.y << 32) | cv.x, ((u64)cv.w << 32) | cv.z, a.carry_nsub, a.carry_sub_bits, k - 18, c0, c1, c2, c3, rl, rh);
                    o[0] = make_uint4((u32)c0, 0, 0, 0); o[1] = Z4; o[2] = make_uint4((u32)c1, 0, 0, 0); o[3] = Z4;
                    o[4] = make_uint4((u32)c2, 0, 0, 0); o[5] = Z4; o[6] = make_uint4((u32)c3, 0, 0, 0); o[7] = Z4;
                    o[8] = make_uint4((u32)rl, (u32)(rl >> 32), (u32)rh, (u32)(rh >> 32)); o[9] = Z4;
                }
                }
#pragma unroll
                for (u32 w = 0; w < NP; ++w) srow[w] = o[w];
                if constexpr (MONT) {
                    // The carries' range rows (:880-885) are the only cells of these rows that are not copies: at most nine rows of
                    // a chunk (nrc of every 23 + nrc), five cells each.  One lane per CELL converts its value -- one conversion per lane
                    // for the whole chunk instead of five in every lane -- and puts it where the row's copy pass left a placeholder.
                    const u32 rr0 = r0 - r_T6;
                    const u32 c0 = __umulhi(rr0, a.per_col_magic), k0 = rr0 - c0 * per_col;
                    const u32 G0 = k0 <= 18 ? c0 * nrc : (k0 < 18 + nrc ? c0 * nrc + (k0 - 18) : (c0 + 1) * nrc);   // the first range row at or behind rr0
                    wave_sync();
                    for (u32 tb = 0; ; tb += 64) {
                        const u32 t_ = tb + lane, G = G0 + t_ / 5, cellq = t_ % 5;
                        const u32 gc = nrc == 3 ? (G * 0xAAABu) >> 17 : (nrc == 2 ? G >> 1 : G / nrc), gj = G - gc * nrc, grr = gc * per_col + 18 + gj;   // (G < 2^15)
                        const bool task = grr < rr0 + 64 && gc < C - 1;
                        if (__ballot(task) == 0) break;
                        // the cell's value, straight from the carry (main_gate.decompose: four terms per row, the LAST row reversed and
                        // zero-padded; column e = the carry with the terms of the rows above cleared)
                        const uint4 ci = reinterpret_cast<const uint4 *>(smem + lp.icout)[task ? gc : 0u];
                        const U192 co = U192::make(((u64)ci.y << 32) | ci.x, ((u64)ci.w << 32) | ci.z, 0);
                        const u32 nsub = a.carry_nsub, sb = a.carry_sub_bits, lastr = nrc - 1, n_last = nsub - 4 * lastr;
                        const bool is_last = gj == lastr;
                        const u32 term = is_last ? nsub - 1 - cellq : 4 * gj + cellq;          // (cellq < 4)
                        const bool has_term = cellq < 4 && (!is_last || cellq < n_last);
                        const u32 sh = term * sb;                                               // < 128
                        const u64 shifted = sh >= 64 ? co.w[1] >> (sh - 64) : (sh ? (co.w[0] >> sh) | (co.w[1] << (64 - sh)) : co.w[0]);
                        const u32 clr = 4 * gj * sb;                                            // bits composed by the rows above
                        const u64 rl = clr >= 64 ? 0 : (clr ? (co.w[0] >> clr) << clr : co.w[0]);
                        const u64 rh = clr >= 64 ? (clr >= 128 ? 0 : (co.w[1] >> (clr - 64)) << (clr - 64)) : co.w[1];
                        const u64 vlo = cellq == 4 ? rl : (has_term ? shifted & ((1ull << sb) - 1) : 0), vhi = cellq == 4 ? rh : 0;
                        uint4 e0, e1;
                        cell_k(std::integral_constant<int, LW == 64 ? 90 : 60>{}, vlo, vhi, e0, e1);   // (a carry: 70 / 40 bits)
                        if (task) { uint4 *dst2 = stage + (u64)(grr - rr0) * NP + 2 * cellq; dst2[0] = e0; dst2[1] = e1; }
                    }
                }
                built = true; path = 2;
            }
        }
        // ================= general path: any chunk (sections may meet inside it) =================
        if (!built) {
        AdviceRowId id = advice_decode(valid ? r : 0u, L, nrc);
        if (!valid) { id.kind = ROWK_NOP; id.sect = 9; }
        U192 v0 = Z, v1 = Z, v2 = Z, v3 = Z, v4 = Z;
        bool sg0 = false, sg1 = false, sg2 = false, need_inv = false;
        uint4 cp[6] = {Z4, Z4, Z4, Z4, Z4, Z4};                        // MONT: cells taken from the planes as they are
        bool row_copied = false, row_eqb = false;
        // ---- mul(a, b), mul(q, n): one limb product per lane, the column's running sums by a segmented scan ----
        if (!(ABL & 16) && __ballot(id.sect == 1) != 0) {
            const bool is_ma = id.sect == 1 && id.kind == ROWK_MUL_ADD;
            u32 p[NWD], own[NWD];
#pragma unroll
            for (int k = 0; k < NWD; ++k) p[k] = 0;
            u64 x = 0, y = 0;
            if (is_ma) {
                x = (id.qn ? sq : sa)[id.j]; y = (id.qn ? sn : sb)[id.i - id.j];
                if constexpr (LW == 64) {
                    const u128 pr = (u128)x * y;
                    p[0] = (u32)pr; p[1] = (u32)(pr >> 32); p[2] = (u32)(pr >> 64); p[3] = (u32)(pr >> 96);
                } else {
                    const u64 pr = (u64)(u32)x * (u32)y;
                    p[0] = (u32)pr; p[1] = (u32)(pr >> 32);
                }
            }
#pragma unroll
            for (int k = 0; k < NWD; ++k) own[k] = p[k];
            if (lane == 0 && is_ma) {   // the column began in the previous chunk: its running sum so far
                u32 c = 0;
#pragma unroll
                for (int k = 0; k < NWD; ++k) p[k] = __builtin_addc(p[k], carry[k], c, &c);
            }
            cells_seg_scan<NWD>(p, !is_ma);
#pragma unroll
            for (int k = 0; k < NWD; ++k) carry[k] = (u32)__builtin_amdgcn_readlane((int)p[k], 63);
            if (is_ma) {
                U192 acc, prev;
                u32 q[NWD], br = 0;
#pragma unroll
                for (int k = 0; k < NWD; ++k) q[k] = __builtin_subc(p[k], own[k], br, &br);
                if constexpr (LW == 64) {
                    acc = U192::make(((u64)p[1] << 32) | p[0], ((u64)p[3] << 32) | p[2], p[4]);
                    prev = U192::make(((u64)q[1] << 32) | q[0], ((u64)q[3] << 32) | q[2], q[4]);
                } else {
                    acc = U192::make(((u64)p[1] << 32) | p[0], p[2], 0);
                    prev = U192::make(((u64)q[1] << 32) | q[0], q[2], 0);
                }
                v0 = lim(x); v1 = lim(y); v2 = prev; v3 = acc;                 // [x_j, y_{i-j}, acc_prev, acc]  :408
                if (id.j == (id.i < L ? id.i : L - 1)) wrp(id.qn ? pEQB : pAB, id.i, acc);   // the column's total
            }
        }
        // (read before the column phase: in a short record the q, r range rows share the chunk with it, and a Montgomery ctx's carries
        // then lie over the limbs of q)
        const u64 v_range = id.sect == 0 ? (id.i < L ? sq[id.i] : sr[id.i - L]) : 0;
        if (!(ABL & 8) && !columns_done && r0 + n_rows >= r_T5) {   // every mul row is built: the columns' carries before any row that needs them
            wave_sync();
            column_phase(true);
            columns_done = true;
        }
        // ---- the other sections: range rows, eq_b, the is_equal_muled preamble and its column rows ----
        if (id.sect == 0) {                                          // RangeChip::assign(q[k] / r[k], w / 8, w)  :590, :598
            u64 c0, c1, c2, c3, rl, rh;
            range_vals(v_range, 0, 8, LW / 8, id.j, c0, c1, c2, c3, rl, rh);
            v0 = lim(c0); v1 = lim(c1); v2 = lim(c2); v3 = lim(c3); v4 = U192::make(rl, rh, 0);
        } else if (id.sect == 2) {                                   // eq_b[i] = qn[i] + r[i]  :617
            if constexpr (MONT) { v1 = lim(sr[id.i]); row_eqb = true; const uint4 *e = reinterpret_cast<const uint4 *>(smem + lp.meqb) + 2 * id.i; cp[4] = e[0]; cp[5] = e[1]; }
            else { const U192 e = rdp(pEQB, id.i); v1 = lim(sr[id.i]); v0 = e - v1; v2 = e; }
        } else if (id.sect == 3) {                                   // :851-856
            if (id.i == 0) v0 = Bw; else if (id.i == 3) { v0 = lim(1); v1 = v0; v2 = v0; }
        } else if (id.sect == 5) {                                   // assert_equal_muled: main_gate.assert_one(eq_bit)  :1062
            v0 = lim((pFL[C - 1] >> 3) & 1u);
        } else if (!(ABL & 8) && id.sect == 4) {
            const u32 c = id.i;
            if (id.kind >= ROWK_RANGE_CARRY) {                       // RangeChip::assign(carry, ...)  :880-885
                U192 cout;
                if constexpr (MONT) { const uint4 ci = reinterpret_cast<const uint4 *>(smem + lp.icout)[c]; cout = U192::make(((u64)ci.y << 32) | ci.x, ((u64)ci.w << 32) | ci.z, 0); }
                else cout = shr_limb(rdp(pSUM, c));
                u64 c0, c1, c2, c3, rl, rh;
                range_vals(cout.w[0], cout.w[1], a.carry_nsub, a.carry_sub_bits, id.kind - ROWK_RANGE_CARRY, c0, c1, c2, c3, rl, rh);
                v0 = lim(c0); v1 = lim(c1); v2 = lim(c2); v3 = lim(c3); v4 = U192::make(rl, rh, 0);
            } else if (MONT && item_ok) {                            // a valid mul_mod: the row's cells are copies, as in the fast path
                if constexpr (MONT) { col_row_copy(c, id.j, cp); row_copied = true; }
            } else {
                const u32 j = id.j;
                // (MONT: an inconsistent mul_mod -- the planes hold cells, the integers come back out of them)
                auto gAB = [&](u32 ix) -> U192 { if constexpr (MONT) return cell_int(reinterpret_cast<const uint4 *>(smem + lp.mab) + 2 * ix, false); else return rdp(pAB, ix); };
                auto gEQB = [&](u32 ix) -> U192 { if constexpr (MONT) return cell_int(reinterpret_cast<const uint4 *>(smem + lp.meqb) + 2 * ix, false); else return rdp(pEQB, ix); };
                auto gSUM = [&](u32 ix) -> U192 { if constexpr (MONT) return cell_int(reinterpret_cast<const uint4 *>(smem + lp.msum) + 2 * ix, false); else return rdp(pSUM, ix); };
                auto fetch = [&](u32 code, bool &sg) -> U192 {
                    if (c == C - 1 && (code & (1u << 13))) code = cells_src(CS_KT, 0, 0, 0, 3, 2);   // the last column: acc_extra
                    const u32 base = code & 7u, m1 = (code >> 9) & 1u, xf = (code >> 10) & 3u;
                    sg = ((code >> 12) & 1u) != 0;
                    if (base == 0 || (m1 && c == 0)) return Z;
                    const u32 idx = c - m1;
                    if (base == CS_AMB) return gAB(idx) - gEQB(idx);   // a_b  :859 (two's complement)
                    U192 val;
                    if (base == CS_KT) {
                        const u64 *ptr = kt + (idx < 2 ? idx : 2) * 10 + ((code >> 3) & 15u);
                        const u32 nw = ((code >> 7) & 3u) + 1;
                        val = U192::make(ptr[0], nw > 1 ? ptr[1] : 0, nw > 2 ? ptr[2] : 0);
                    } else val = base == CS_AB ? gAB(idx) : base == CS_EQB ? gEQB(idx) : gSUM(idx);
                    if (xf == 0) return val;
                    if (xf == 1) return shr_limb(val);
                    if (xf == 2) return lim(val.w[0] & LMASK);
                    return U192::make(val.w[0] & ~LMASK, val.w[1], val.w[2]);
                };
                bool s0, s1, s2;
                auto src_code = [&](u32 q) -> u32 { if constexpr (MONT) return j < ADVICE_COL_ROWS ? cells_col_src(j, q) : 0u; else return s_src[j * 3 + q]; };
                const U192 c0 = fetch(src_code(0), s0), c1 = fetch(src_code(1), s1), c2 = fetch(src_code(2), s2);
                const u32 fl = pFL[c], eprev = c ? (pFL[c - 1] >> 3) & 1u : 1u;
                const u32 f1 = fl & 1u, e1 = (fl >> 1) & 1u, f2 = (fl >> 2) & 1u, e2 = (fl >> 3) & 1u;
                const U192 d = c0 - c1;
                switch (j) {
                    case 4: case 10: v0 = Bw; v1 = c1; v2 = c2; break;
                    case 13: case 18: v0 = c0; v1 = c1; v2 = d; sg2 = true; break;                      // sub: d = x - y in the field
                    case 14: v0 = lim(f1); v1 = v0; v2 = v0; break;
                    case 19: v0 = lim(f2); v1 = v0; v2 = v0; break;
                    case 15: case 20: v0 = d; sg0 = true; v1 = lim(1); v2 = lim(j == 15 ? f1 : f2); need_inv = !(d == Z); break;   // [d, 1/d (1 when d = 0), r]
                    case 16: case 21: v0 = lim(j == 16 ? f1 : f2); v1 = d; sg1 = true; break;         // [r, d]
                    case 17: v0 = lim(eprev); v1 = lim(f1); v2 = lim(e1); break;                      // and
                    case 22: v0 = lim(e1); v1 = lim(f2); v2 = lim(e2); break;                         // and
                    default: v0 = c0; v1 = c1; v2 = c2; sg0 = s0; sg1 = s1; sg2 = s2; break;
                }
            }
        }
        if constexpr (MONT) {
            // (every lane converts whatever any lane needs: a cell that is zero in the whole chunk is not converted, and a chunk of
            // nothing but range rows -- the q, r limbs' -- takes its sub-limbs as one-digit values)
            const bool is_rng = id.sect == 0 || (id.sect == 4 && id.kind >= ROWK_RANGE_CARRY);
            if (__ballot(valid && !is_rng) == 0) {
                using K1 = std::integral_constant<int, 8>;
                cell_k(K1{}, v0.w[0], 0, srow[0], srow[1]); cell_k(K1{}, v1.w[0], 0, srow[2], srow[3]);
                cell_k(K1{}, v2.w[0], 0, srow[4], srow[5]); cell_k(K1{}, v3.w[0], 0, srow[6], srow[7]);
                cell_k(std::integral_constant<int, LW == 64 ? 90 : 60>{}, v4.w[0], v4.w[1], srow[8], srow[9]);
            } else {
                auto put5 = [&](uint4 *p5, const U192 &v, bool sg) {
                    if (__ballot(valid && !(v == Z)) == 0) { p5[0] = Z4; p5[1] = Z4; }
                    else cell(p5, v, sg);
                };
                put5(srow, v0, sg0); put5(srow + 2, v1, sg1); put5(srow + 4, v2, sg2); put5(srow + 6, v3, false); put5(srow + 8, v4, false);
                if (row_copied) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) srow[q] = cp[q];
                }
                if (row_eqb) {   // [qn_i, r_i, eq_b_i]: eq_b_i is the plane's cell, r_i was converted above, qn_i = eq_b_i - r_i in the field
                    const uint4 r0_ = srow[2], r1_ = srow[3];
                    uint4 q0_, q1_;
                    cell_sub(cp[4], cp[5], r0_, r1_, q0_, q1_);
                    srow[0] = q0_; srow[1] = q1_; srow[4] = cp[4]; srow[5] = cp[5];
                }
            }
        }
        if (valid) {
            if constexpr (!MONT) { cell(srow, v0, sg0); cell(srow + 2, v1, sg1); cell(srow + 4, v2, sg2); cell(srow + 6, v3, false); cell(srow + 8, v4, false); }
            if (need_inv) {
                // 1 / d, main_gate.is_zero's witness: never taken for a valid mul_mod (every comparison is between equal values)
                Fe xe;
                field4(v0, true, xe.v);
                const FieldConsts fc = *reinterpret_cast<const FieldConsts *>(a.ktab + CELLS_KT_FC);
                Fe iv = fe_inv_fast(xe, fc);
                if constexpr (MONT) iv = fe_to_mont_k(iv, *a.mk);
                srow[2] = make_uint4((u32)iv.v[0], (u32)(iv.v[0] >> 32), (u32)iv.v[1], (u32)(iv.v[1] >> 32));
                srow[3] = make_uint4((u32)iv.v[2], (u32)(iv.v[2] >> 32), (u32)iv.v[3], (u32)(iv.v[3] >> 32));
            }
        }
        }   // general path
        // ---- the chunk leaves as whole 16-byte-per-lane lines ----
        wave_sync();
        u64 t_1 = 0;
        if constexpr (ABL & 256) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_1 = __builtin_amdgcn_s_memtime(); }
        u8 *dst = out + (u64)r0 * a.dst.row_pitch;
        auto put = [&](u8 *at, const uint4 &v) {
            if constexpr (ABL & 4) pst16(at, ((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z);
            else if constexpr (!(ABL & 2)) st16(at, ((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z);
            else if (v.x == 0x12345678u && v.w == 0x9abcdef0u) pst16(dst, 1, 2);   // (keeps the LDS reads alive)
        };
        if (!planar) {
            if (n_rows == 64) {   // ten LDS reads in flight, then ten 1 KB stores
                uint4 v[NP];
#pragma unroll
                for (u32 k = 0; k < NP; ++k) v[k] = stage[k * 64 + lane];
#pragma unroll
                for (u32 k = 0; k < NP; ++k) put(dst + (u64)(k * 64 + lane) * 16, v[k]);
            } else {
                for (u32 u = lane; u < n_rows * NP; u += 64) put(dst + (u64)u * 16, stage[u]);
            }
        } else {
            // planar columns: the chunk is five runs of n_rows x 32 bytes, column c's at dst + c * col_pitch (row pitch 32: dst is the
            // chunk's first row in column a).  Store k of the ten covers 16-byte pieces (k & 1) * 64 .. + 63 of column k / 2.
            if (n_rows == 64) {
                uint4 v[NP];
#pragma unroll
                for (u32 k = 0; k < NP; ++k) { const u32 piece = (k & 1) * 64 + lane; v[k] = stage[(piece >> 1) * NP + (k >> 1) * 2 + (piece & 1)]; }
#pragma unroll
                for (u32 k = 0; k < NP; ++k) put(dst + (u64)(k >> 1) * a.dst.col_pitch + (u64)((k & 1) * 64 + lane) * 16, v[k]);
            } else {
                for (u32 u = lane; u < n_rows * NP; u += 64) {
                    const u32 col = u / (2 * n_rows), piece = u - col * 2 * n_rows;
                    put(dst + (u64)col * a.dst.col_pitch + (u64)piece * 16, stage[(piece >> 1) * NP + col * 2 + (piece & 1)]);
                }
            }
        }
        wave_sync();
        if constexpr (ABL & 256) {
            if (item == a.dbg_item && a.dbg && lane == 0) {
                const u64 t_2 = __builtin_amdgcn_s_memtime();
                const u32 ci = r0 / 64 + (r0 % 64 ? 1 : 0);
                if (ci < 1000) { a.dbg[3 * ci] = path; a.dbg[3 * ci + 1] = t_1 - t_0; a.dbg[3 * ci + 2] = t_2 - t_1; }
            }
        }
    };
    if constexpr (NWV == 1) {
        for (u32 r0 = 0, n = n_first; r0 < a.rows; r0 += n, n = 64) {
            if (a.rows - r0 < n) n = a.rows - r0;
            chunk(r0, n);
        }
    } else {
        // chunk j: rows [cr0(j), cr0(j) + cn(j)); chunk 0 is the short one in front of the first 128-byte line
        const u32 n0 = n_first < a.rows ? n_first : a.rows, NC = 1 + (a.rows - n0 + 63) / 64;
        auto cr0 = [&](u32 j) -> u32 { return j ? n0 + 64 * (j - 1) : 0u; };
        auto cn = [&](u32 j) -> u32 { const u32 s0 = cr0(j), m = j ? 64u : n0; return a.rows - s0 < m ? a.rows - s0 : m; };
        const u32 cb = r_T5 <= n0 ? 0u : (r_T5 - n0 + 63) / 64;          // the chunk in which the last mul row is built: it runs the column phase
        const u32 WU = (L + 63) / 64;                                             // chunks a column reaches back over
        u32 last = ~0u;                                                           // the chunk whose running sum `carry` continues
        __syncthreads();                                                          // (operand cells, constants: written by all waves)
        for (u32 ph = 0; ph < 3; ++ph) {
            u32 lo, hi;
            if (ph == 0) { const u32 per = (cb + NWV - 1) / NWV; lo = wv * per; hi = lo + per < cb ? lo + per : cb; }
            else if (ph == 1) { lo = cb; hi = wv == 0 ? cb + 1 : cb; if (wv != 0) column_phase(false); }   // (the column phase is every wave's)
            else { const u32 m = NC - cb - 1, per = (m + NWV - 1) / NWV; lo = cb + 1 + wv * per; hi = lo + per < NC ? lo + per : NC; }
            if (ph == 2) { columns_done = true; item_ok = pFL[2 * L - 1] != 0; }
            for (u32 j = lo; j < hi; ++j) {
                if (ph < 2 && j && last != j - 1) {
                    // entering the mul rows in the middle: the running sum of the column that reaches into chunk j, from the chunks it can
                    // begin in (sums restart at every column's head row, so the value assumed in front of them does not matter)
#pragma unroll
                    for (int k = 0; k < NWD; ++k) carry[k] = 0;
                    for (u32 jj = j > WU ? j - WU : 0u; jj < j; ++jj) {
                        const u32 rs = cr0(jj) + lane;
                        const bool vs = lane < cn(jj);
                        const AdviceRowId id = advice_decode(vs ? rs : 0u, L, nrc);
                        const bool is_ma = vs && id.sect == 1 && id.kind == ROWK_MUL_ADD;
                        u32 pp[NWD];
#pragma unroll
                        for (int k = 0; k < NWD; ++k) pp[k] = 0;
                        if (is_ma) {
                            const u64 x = (id.qn ? sq : sa)[id.j], y = (id.qn ? sn : sb)[id.i - id.j];
                            if constexpr (LW == 64) { const u128 pr = (u128)x * y; pp[0] = (u32)pr; pp[1] = (u32)(pr >> 32); pp[2] = (u32)(pr >> 64); pp[3] = (u32)(pr >> 96); }
                            else { const u64 pr = (u64)(u32)x * (u32)y; pp[0] = (u32)pr; pp[1] = (u32)(pr >> 32); }
                        }
                        if (lane == 0 && is_ma) {
                            u32 c = 0;
#pragma unroll
                            for (int k = 0; k < NWD; ++k) pp[k] = __builtin_addc(pp[k], carry[k], c, &c);
                        }
                        cells_seg_scan<NWD>(pp, !is_ma);
#pragma unroll
                        for (int k = 0; k < NWD; ++k) carry[k] = (u32)__builtin_amdgcn_readlane((int)pp[k], 63);
                    }
                }
                chunk(cr0(j), cn(j));
                last = j;
            }
            __syncthreads();
        }
    }
}

}  // namespace h2r
