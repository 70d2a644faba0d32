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
//     kernel's scheme) and the running eq_bit into four LDS planes; the 23 + nrc rows of a column then read their cells from
//     those planes through one table-driven fetch (no per-row-kind branches around the loads).  The input-independent
//     accumulated_extra chain (:869-875) reaches its fixed point at column 2, so its values are a 3-entry table.
//   * range rows (main_gate.decompose of a limb / carry) are cut from the value itself.
// 64 rows = 10,240 bytes are staged in LDS and leave as ten 1 KB store instructions (16 bytes per lane, non-temporal), so the
// only HBM traffic is the image itself: 635,680 bytes written per RSA-2048 mul_mod against 1.3 KB read.  No workgroup barrier
// anywhere (a workgroup IS a wave); ~18 KB of LDS per wave for RSA-2048 = 8 waves per CU, a few store streams per CU.
// Bound: HBM writes.  The image is byte-identical to advice_kernel's for every valid record (tests).
#pragma once

#include "h2r_kernels.hpp"

namespace h2r {

struct CellsArgs {
    const u32 *desc;                     // [rows] advice_pack(advice_decode(r)) -- the ctx's row table (L2-resident)
    const u64 *ktab;                     // [3][10] + pad: columns 0, 1, >= 2 of the accumulated_extra constants:
                                         //   acc_extra + W [3 words], q_acc [2], mod_acc [1], nq [3], a - nq [1]
    const void *opA, *opB, *opQ, *opR;   // limbs of item k at [k * op_stride, k * op_stride + L)
    u64 op_stride, qr_stride;            // (limbs) of opA / opB and of opQ / opR
    const void *n; u64 n_stride;         // [elem][L] limbs (stride 0 = shared)
    const u8 *status;                    // [elem] nullable; nonzero => the element's items are skipped
    u32 T; u64 n_items;                  // item = elem * T + t
    u8 *out; u64 out_stride;             // element e's image at out + e * out_stride: pre_rows rows, then record t at + (pre_rows + t * rows) * 160
    u32 rows, pre_rows;
    u32 L, carry_sub_bits, carry_nsub;
    u64 wm[3];                           // word_max (chip.rs:838)
    FieldConsts f;
};

constexpr u32 CELLS_KT_WORDS = 32;       // 3 x 10 words + padding (the generic fetch reads three words)
constexpr u32 CELLS_SRC_WORDS = 72;      // the column rows' source codes (23 x 3) + padding

// Source of one cell of an is_equal_muled column row (rows 0..22 of the column, cells a, b, c):
//   bits 0-2 base (0 none, 1 AB, 2 EQB, 3 AMB, 4 SUM, 5 the constants' table), bits 3-6 word offset in a table entry,
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

// dynamic LDS of one wave (bytes): stage, operands, four column planes, flags, the constants' table
__host__ __device__ inline u32 cells_lds_bytes(u32 limb_width, u32 L) {
    const u32 ww = limb_width == 64 ? 3u : 2u;
    return 64u * ADVICE_ROW_BYTES + 5u * L * 8u + 4u * (2u * L * ww + 2u) * 8u + CELLS_KT_WORDS * 8u + 2u * L * 4u + CELLS_SRC_WORDS * 4u;
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

template <int LW>
__global__ __launch_bounds__(64) void cells_kernel(CellsArgs a) {
    using limb_t = typename LimbT<LW>::type;
    constexpr int WW = LW == 64 ? 3 : 2;     // 64-bit words of a wide value in the planes
    constexpr int NWD = LW == 64 ? 5 : 3;    // dwords of a running column sum (133 / 71 bits)
    constexpr u64 LMASK = LW == 64 ? ~0ull : 0xffffffffull;
    extern __shared__ uint4 cells_smem[];
    const u32 lane = threadIdx.x;
    const u32 L = a.L, L2 = 2 * L, C = 2 * L - 1;
    uint4 *stage = cells_smem;                                       // 64 rows x 160 bytes
    u64 *sa = reinterpret_cast<u64 *>(cells_smem + 64 * (ADVICE_ROW_BYTES / 16));
    u64 *sb = sa + L, *sq = sb + L, *sn = sq + L, *sr = sn + L;
    u64 *pAB = sr + L, *pEQB = pAB + (L2 * WW + 2), *pAMB = pEQB + (L2 * WW + 2), *pSUM = pAMB + (L2 * WW + 2);
    u64 *kt = pSUM + (L2 * WW + 2);
    u32 *pFL = reinterpret_cast<u32 *>(kt + CELLS_KT_WORDS);
    u32 *s_src = pFL + L2;
    // the column phase's scratch lives in the stage (free between two chunks)
    u64 *xDH0 = reinterpret_cast<u64 *>(stage), *xDH1 = xDH0 + L2, *xSLO = xDH1 + L2;
    u32 *xSHI = reinterpret_cast<u32 *>(xSLO + L2);

    const u32 item = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const u32 elem = item / a.T, t = item - elem * a.T;
    if (a.status && a.status[elem]) return;
    {
        const u64 ib = (u64)item * a.op_stride, iq = (u64)item * a.qr_stride;
        for (u32 k = lane; k < L; k += 64) {
            sa[k] = reinterpret_cast<const limb_t *>(a.opA)[ib + k]; sb[k] = reinterpret_cast<const limb_t *>(a.opB)[ib + k];
            sq[k] = reinterpret_cast<const limb_t *>(a.opQ)[iq + k]; sr[k] = reinterpret_cast<const limb_t *>(a.opR)[iq + k];
            sn[k] = reinterpret_cast<const limb_t *>(a.n)[(u64)elem * a.n_stride + k];
        }
        if (lane < CELLS_KT_WORDS) kt[lane] = a.ktab[lane];
    }
    u8 *out = a.out + (u64)elem * a.out_stride + ((u64)a.pre_rows + (u64)t * a.rows) * ADVICE_ROW_BYTES;
    if (t == 0 && lane < a.pre_rows * (ADVICE_ROW_BYTES / 16)) {   // pow_mod_fixed_exp's acc = assign_constant(1, L): [1, 0, 0, 0, 0] then [0, ...]
        uint4 *pr = reinterpret_cast<uint4 *>(a.out + (u64)elem * a.out_stride);
        pr[lane] = make_uint4(lane == 0 ? 1u : 0u, 0, 0, 0);
    }
    const U192 Z = U192::make(0, 0, 0);
    const U192 Bw = LW == 64 ? U192::make(0, 1, 0) : U192::make(1ull << 32, 0, 0);   // 2^w
    const U192 Wm = U192::make(a.wm[0], a.wm[1], a.wm[2]);
    auto lim = [&](u64 v) { return U192::make(v, 0, 0); };
    auto rdp = [&](const u64 *pl, u32 c) -> U192 { return U192::make(pl[(u64)c * WW], pl[(u64)c * WW + 1], WW == 3 ? pl[(u64)c * WW + 2] : 0); };
    auto rdp_s = [&](const u64 *pl, u32 c) -> U192 {   // two's complement
        const u64 w1 = pl[(u64)c * WW + 1];
        return U192::make(pl[(u64)c * WW], w1, WW == 3 ? pl[(u64)c * WW + 2] : (u64)((i64)w1 >> 63));
    };
    auto wrp = [&](u64 *pl, u32 c, const U192 &v) { pl[(u64)c * WW] = v.w[0]; pl[(u64)c * WW + 1] = v.w[1]; if constexpr (WW == 3) pl[(u64)c * WW + 2] = v.w[2]; };
    auto shr_limb = [&](const U192 &v) -> U192 { return v.shr(LW); };
    auto cell = [&](uint4 *p, const U192 &v, bool is_signed) {   // canonical field element, 32 bytes little-endian
        u64 x[4] = {v.w[0], v.w[1], v.w[2], 0};
        if (is_signed && (v.w[2] >> 63)) {   // x < 0 -> p + x (mod 2^256)
            x[3] = ~0ull;
            u64 cy = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u64 s1 = x[k] + a.f.p[k]; const u64 c1 = s1 < x[k]; const u64 s2 = s1 + cy; cy = c1 | (u64)(s2 < s1); x[k] = s2; }
        }
        p[0] = make_uint4((u32)x[0], (u32)(x[0] >> 32), (u32)x[1], (u32)(x[1] >> 32));
        p[1] = make_uint4((u32)x[2], (u32)(x[2] >> 32), (u32)x[3], (u32)(x[3] >> 32));
    };
    // row rr of RangeChip::assign of the value v (nsub sub-limbs of sub_bits bits, the last one possibly shorter): four sub-limbs in
    // columns a..d -- the LAST row reversed, so that the last (overflow) term is in column a, and zero-padded -- and in column e what
    // remains to be composed (main_gate.decompose)
    auto range_vals = [&](u64 v_lo, u64 v_hi, u32 nsub, u32 sub_bits, u32 rr, U192 &c0, U192 &c1, U192 &c2, U192 &c3, U192 &rem) {
        const u32 last = (nsub - 1) / 4, n_last = nsub - 4 * last;
        const u64 sm = (1ull << sub_bits) - 1;
        auto sub = [&](u32 k) -> u64 {
            const u32 sh = k * sub_bits;
            const u64 x = sh >= 64 ? v_hi >> (sh - 64) : ((v_lo >> sh) | (sh ? v_hi << (64 - sh) : 0));
            return k < nsub ? x & sm : 0;
        };
        const bool rev = rr >= last;
        const u32 k0 = rev ? nsub - 1 : 4 * rr;
        c0 = lim(sub(k0));
        c1 = lim(rev ? (n_last > 1 ? sub(k0 - 1) : 0) : sub(k0 + 1));
        c2 = lim(rev ? (n_last > 2 ? sub(k0 - 2) : 0) : sub(k0 + 2));
        c3 = lim(rev ? (n_last > 3 ? sub(k0 - 3) : 0) : sub(k0 + 3));
        const u32 cl = 4 * rr * sub_bits;   // low bits already composed: cleared
        rem = cl >= 64 ? U192::make(0, cl >= 128 ? 0 : (v_hi >> (cl - 64)) << (cl - 64), 0)
                       : U192::make(cl ? (v_lo >> cl) << cl : v_lo, v_hi, 0);
    };
    for (u32 k = lane; k < ADVICE_COL_ROWS * 3; k += 64) s_src[k] = cells_col_src(k / 3, k % 3);
    wave_sync();

    // ---- the 2L - 1 un-carried columns: eq_b, a_b, the carries and the running eq_bit (chip.rs:614-623, 857-893) -> planes ----
    auto column_phase = [&]() {
        for (u32 c = lane; c < C; c += 64) {
            const U192 A = rdp(pAB, c);
            U192 Q = rdp(pEQB, c);                 // (holds the q*n column until here)
            if (c < L) { Q = Q + lim(sr[c]); wrp(pEQB, c, Q); }                // eq_b[i] = qn[i] + r[i]  :617
            const U192 amb = A - Q;               // :859 (two's complement)
            wrp(pAMB, c, amb);
            const U192 D = amb + Wm;              // >= 0
            const U192 dhi = shr_limb(D);
            xSLO[c] = D.w[0] & LMASK; xDH0[c] = dhi.w[0]; xDH1[c] = dhi.w[1];
        }
        wave_sync();
        for (u32 c = lane; c < C; c += 64) {
            const U192 S = lim(xSLO[c]) + (c ? U192::make(xDH0[c - 1], xDH1[c - 1], 0) : Z);
            xSLO[c] = S.w[0] & LMASK;
            xSHI[c] = (u32)shr_limb(S).w[0];
        }
        wave_sync();
        bool cin = false, all_ok = true;
        for (u32 cb = 0; cb < C; cb += 64) {
            const u32 c = cb + lane;
            const bool col = c < C;
            const u64 slo = col ? xSLO[c] : 0;
            const u32 shp = (col && c) ? xSHI[c - 1] : 0;
            const u128 U = (u128)slo + shp;
            const bool gen = col && (U >> LW) != 0, prop = col && ((u64)U & LMASK) == LMASK;
            const CarryGroup cg = carry_group(__ballot(gen), __ballot(prop), cin, 64);
            const bool f = ((cg.cin_mask >> lane) & 1) != 0;
            cin = cg.cout;
            bool f1 = true, f2 = true;
            if (col) {
                const U192 dhp = c ? U192::make(xDH0[c - 1], xDH1[c - 1], 0) : Z;
                const U192 carry_in = dhp + lim((u64)shp + (f ? 1u : 0u));
                const U192 sum = rdp_s(pAMB, c) + Wm + carry_in;               // :860-861
                wrp(pSUM, c, sum);
                const U192 cout = shr_limb(sum);
                const u32 kc = c < 2 ? c : 2;
                f1 = (sum.w[0] & LMASK) == kt[kc * 10 + 5];                      // cs_acc_eq  :873
                if (c == C - 1) f2 = cout.w[0] == kt[kc * 10 + 3] && cout.w[1] == kt[kc * 10 + 4];   // final_carry_eq  :890
            }
            const u64 bad = __ballot(col && !(f1 && f2));
            const bool prev_ok = all_ok && (bad & ((1ull << lane) - 1)) == 0;
            if (col) {
                const u32 e1 = (prev_ok && f1) ? 1u : 0u, e2 = (e1 && f2) ? 1u : 0u;
                pFL[c] = (f1 ? 1u : 0u) | (e1 << 8) | ((f2 ? 1u : 0u) << 16) | (e2 << 24);
            }
            all_ok = all_ok && bad == 0;
        }
        wave_sync();
    };

    const u32 mul_rows = C + L * L, r_T5 = 4 * L + 2 * mul_rows;
    u32 carry[NWD];
#pragma unroll
    for (int k = 0; k < NWD; ++k) carry[k] = 0;
    u32 dnext = lane < a.rows ? a.desc[lane] : 0u;
    bool columns_done = false;
    for (u32 r0 = 0; r0 < a.rows; r0 += 64) {
        const u32 r = r0 + lane;
        const bool valid = r < a.rows;
        AdviceRowId id = advice_unpack(dnext);
        if (!valid) { id.kind = ROWK_NOP; id.sect = 9; }
        if (r0 + 64 < a.rows) dnext = r + 64 < a.rows ? a.desc[r + 64] : 0u;   // in flight while this chunk is built
        U192 v0 = Z, v1 = Z, v2 = Z, v3 = Z, v4 = Z;
        bool sg0 = false, sg1 = false, sg2 = false, need_inv = false;

        // ---- mul(a, b), mul(q, n): one limb product per lane, the column's running sums by a segmented scan ----
        if (__ballot(id.sect == 1) != 0) {
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
        if (!columns_done && r0 + 64 >= r_T5) {   // every mul row is built: the columns' carries before any row that needs them
            wave_sync();
            column_phase();
            columns_done = true;
        }
        // ---- the other sections: range rows, eq_b, the is_equal_muled preamble and its column rows ----
        if (id.sect == 0) {                                          // RangeChip::assign(q[k] / r[k], w / 8, w)  :590, :598
            const u64 v = id.i < L ? sq[id.i] : sr[id.i - L];
            range_vals(v, 0, 8, LW / 8, id.j, v0, v1, v2, v3, v4);
        } else if (id.sect == 2) {                                   // eq_b[i] = qn[i] + r[i]  :617
            const U192 e = rdp(pEQB, id.i);
            v1 = lim(sr[id.i]); v0 = e - v1; v2 = e;
        } else if (id.sect == 3) {                                   // :851-856
            if (id.i == 0) v0 = Bw; else if (id.i == 3) { v0 = lim(1); v1 = v0; v2 = v0; }
        } else if (id.sect == 4) {
            const u32 c = id.i;
            if (id.kind >= ROWK_RANGE_CARRY) {                       // RangeChip::assign(carry, ...)  :880-885
                const U192 cout = shr_limb(rdp(pSUM, c));
                range_vals(cout.w[0], cout.w[1], a.carry_nsub, a.carry_sub_bits, id.kind - ROWK_RANGE_CARRY, v0, v1, v2, v3, v4);
            } else {
                const u32 j = id.j;
                auto fetch = [&](u32 code, bool &sg) -> U192 {
                    if (c == C - 1 && (code & (1u << 13))) code = cells_src(CS_KT, 0, 0, 0, 3, 2);   // the last column: acc_extra
                    const u32 base = code & 7u, m1 = (code >> 9) & 1u, xf = (code >> 10) & 3u;
                    sg = ((code >> 12) & 1u) != 0;
                    if (base == 0 || (m1 && c == 0)) return Z;
                    const u32 idx = c - m1;
                    const u64 *ptr; u32 nw;
                    if (base == CS_KT) { ptr = kt + (idx < 2 ? idx : 2) * 10 + ((code >> 3) & 15u); nw = ((code >> 7) & 3u) + 1; }
                    else { ptr = (base == CS_AB ? pAB : base == CS_EQB ? pEQB : base == CS_AMB ? pAMB : pSUM) + (u64)idx * WW; nw = WW; }
                    const u64 w0 = ptr[0], w1r = ptr[1], w2r = ptr[2];
                    const u64 w1 = nw > 1 ? w1r : 0;
                    const u64 w2 = nw > 2 ? w2r : (sg ? (u64)((i64)w1 >> 63) : 0);
                    const U192 val = U192::make(w0, w1, w2);
                    if (xf == 0) return val;
                    if (xf == 1) return shr_limb(val);
                    if (xf == 2) return lim(w0 & LMASK);
                    return U192::make(w0 & ~LMASK, w1, w2);
                };
                bool s0, s1, s2;
                const U192 c0 = fetch(s_src[j * 3], s0), c1 = fetch(s_src[j * 3 + 1], s1), c2 = fetch(s_src[j * 3 + 2], s2);
                const u32 fl = pFL[c], eprev = c ? pFL[c - 1] >> 24 : 1u;
                const u32 f1 = fl & 0xff, e1 = (fl >> 8) & 0xff, f2 = (fl >> 16) & 0xff, e2 = fl >> 24;
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
        // ---- stage the row, then the chunk leaves as whole 16-byte-per-lane lines ----
        if (valid) {
            uint4 *p = stage + (u64)lane * (ADVICE_ROW_BYTES / 16);
            cell(p, v0, sg0); cell(p + 2, v1, sg1); cell(p + 4, v2, sg2); cell(p + 6, v3, false); cell(p + 8, v4, false);
            if (need_inv) {
                u64 x[4] = {v0.w[0], v0.w[1], v0.w[2], 0};
                if (v0.w[2] >> 63) {
                    x[3] = ~0ull;
                    u64 cy = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const u64 s1 = x[k] + a.f.p[k]; const u64 c1 = s1 < x[k]; const u64 s2 = s1 + cy; cy = c1 | (u64)(s2 < s1); x[k] = s2; }
                }
                // 1 / d, main_gate.is_zero's witness: never taken for a valid mul_mod (every comparison is between equal values)
                Fe xe; xe.v[0] = x[0]; xe.v[1] = x[1]; xe.v[2] = x[2]; xe.v[3] = x[3];
                const Fe iv = fe_inv_fast(xe, a.f);
                p[2] = make_uint4((u32)iv.v[0], (u32)(iv.v[0] >> 32), (u32)iv.v[1], (u32)(iv.v[1] >> 32));
                p[3] = make_uint4((u32)iv.v[2], (u32)(iv.v[2] >> 32), (u32)iv.v[3], (u32)(iv.v[3] >> 32));
            }
        }
        wave_sync();
        const u32 n_rows = a.rows - r0 < 64 ? a.rows - r0 : 64;
        u8 *dst = out + (u64)r0 * ADVICE_ROW_BYTES;
#pragma unroll
        for (u32 k = 0; k < ADVICE_ROW_BYTES / 16; ++k) {
            const u32 u = k * 64 + lane;
            if (u < n_rows * (ADVICE_ROW_BYTES / 16)) {
                const uint4 v = stage[u];
                st16(dst + (u64)u * 16, ((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z);
            }
        }
        wave_sync();
    }
}

}  // namespace h2r
