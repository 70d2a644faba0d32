// The copy constraints of the advice image and its layout as data: host-side tables of libh2r, plus the one small kernel that permutes
// an image's cells to a caller's layout.
//
// (1) Copy map.  maingate assigns a fresh row for every op and ties the INPUT cells of the row to the cells where their values
//     were first assigned with equality (copy) constraints -- `AssignedValue::from(a.limb(j))` feeding main_gate.mul_add
//     (big_integer/chip.rs:406-408), `carry` re-used across the steps of is_equal_muled (:861) and so on.  The image repeats the
//     VALUES; which cells are copies of which is a function of the row table alone.  copy_map_record() walks one mul_mod's rows
//     (advice_decode) and lists, for every input cell, its origin: another cell of the same record (row, column), or limb j of
//     one of the mul_mod's operands a, b, n (assigned outside the record: H2R_COPY_SRC_A / _B / _N).
// (2) Layout descriptor.  The placement of an op's cells in the five columns is third-party (maingate / halo2wrong rev 63bde545,
//     restated from recollection: DESIGN.md section 2b); h2r_advice_layout holds, per row kind, the physical column of each of the
//     restated shape's cells, so that a maintainer who re-pins a row shape against upstream changes DATA: the image is permuted by
//     advice_layout_kernel, the fixed rows by layout_permute_fixed().
#pragma once

#include <cstring>
#include <vector>

#include "h2r.h"
#include "h2r_kernels.hpp"

namespace h2r {

// rows of one mul_mod record that other rows copy from (functions of (L, nrc): the sections of advice_decode)
struct RecordRows {
    u32 L, C, nrc, mul_rows, r_T3, r_T5, r_T6p, r_T6, per_col;
    RecordRows(u32 L_, u32 nrc_) : L(L_), C(2 * L_ - 1), nrc(nrc_) {
        mul_rows = C + L * L; r_T3 = 4 * L; r_T5 = r_T3 + 2 * mul_rows; r_T6p = r_T5 + L; r_T6 = r_T6p + 4; per_col = ADVICE_COL_ROWS + nrc;
    }
    u32 limb_row(u32 k) const { return 2 * k; }                                        // RangeChip::assign(q[k] / r[k - L]): the value is column e of its first row
    u32 mul_row(u32 qn, u32 i, u32 k) const { return r_T3 + qn * mul_rows + advice_mul_colstart(i, L) + k; }   // k = 0: the column's constant 0
    u32 mul_last(u32 qn, u32 i) const { return mul_row(qn, i, i < L ? i + 1 : C - i); }   // the column's final accumulator: column d
    u32 col_row(u32 c, u32 j) const {   // row j (0..22) of column c of is_equal_muled; the carry's range rows sit between 17 and 18
        return r_T6 + c * per_col + j + ((c < C - 1 && j >= 18) ? nrc : 0);
    }
};

inline void copy_map_record(u32 L, u32 nrc, std::vector<h2r_copy> &out) {
    const RecordRows R(L, nrc);
    const u32 C = R.C;
    auto cp = [&](u32 row, u32 col, u32 src_row, u32 src_col) { out.push_back(h2r_copy{row, col, src_row, src_col}); };
    // mul(a, b), mul(q, n): [x_j, y_{i-j}, acc_prev, acc]  (chip.rs:400-412)
    for (u32 qn = 0; qn < 2; ++qn)
        for (u32 i = 0; i < C; ++i) {
            const u32 jmin = i >= L ? i - L + 1 : 0, terms = i < L ? i + 1 : C - i;
            for (u32 k = 1; k <= terms; ++k) {
                const u32 row = R.mul_row(qn, i, k), j = jmin + k - 1;
                if (qn) { cp(row, 0, R.limb_row(j), 4); cp(row, 1, H2R_COPY_SRC_N, i - j); }
                else { cp(row, 0, H2R_COPY_SRC_A, j); cp(row, 1, H2R_COPY_SRC_B, i - j); }
                cp(row, 2, row - 1, k == 1 ? 0 : 3);   // the column's constant 0, then the previous accumulator
            }
        }
    // eq_b[i] = qn[i] + r[i]  (:617)
    for (u32 i = 0; i < L; ++i) { cp(R.r_T5 + i, 0, R.mul_last(1, i), 3); cp(R.r_T5 + i, 1, R.limb_row(L + i), 4); }
    // is_equal_muled (:851-893); preamble rows: CONST_B [2^w], CONST0 (acc_extra), CONST0 (carry), BIT [1, 1, 1]
    const u32 pB = R.r_T6p, pX = R.r_T6p + 1, pC = R.r_T6p + 2, pE = R.r_T6p + 3;
    for (u32 c = 0; c < C; ++c) {
        auto r = [&](u32 j) { return R.col_row(c, j); };
        const bool last = c == C - 1;
        cp(r(0), 0, R.mul_last(0, c), 3);                                               // sub [ab_c, eqb_c, a_b]
        if (c < L) cp(r(0), 1, R.r_T5 + c, 2); else cp(r(0), 1, R.mul_last(1, c), 3);
        cp(r(1), 0, r(0), 2);                                                           // add_with_constant [a_b, carry_c, sum]
        if (c) cp(r(1), 1, R.col_row(c - 1, 2), 0); else cp(r(1), 1, pC, 0);
        cp(r(4), 0, pB, 0); cp(r(4), 1, r(2), 0);                                       // div_mod: mul [2^w, q, nq]
        cp(r(5), 0, r(1), 2); cp(r(5), 1, r(4), 2);                                     //          sub [sum, nq, sum - nq]
        cp(r(6), 0, r(3), 0); cp(r(6), 1, r(5), 2);                                     //          assert_equal [r, sum - nq]
        if (c) cp(r(7), 0, R.col_row(c - 1, 8), 0); else cp(r(7), 0, pX, 0);            // add_constant [acc_extra, acc_extra + W]
        cp(r(10), 0, pB, 0); cp(r(10), 1, r(8), 0);
        cp(r(11), 0, r(7), 1); cp(r(11), 1, r(10), 2);
        cp(r(12), 0, r(9), 0); cp(r(12), 1, r(11), 2);
        cp(r(13), 0, r(3), 0); cp(r(13), 1, r(9), 0);                                   // is_equal(c, mod_acc): sub, bit, [d, d', r], [r, d]
        cp(r(15), 0, r(13), 2); cp(r(15), 2, r(14), 0);
        cp(r(16), 0, r(14), 0); cp(r(16), 1, r(13), 2);
        if (c) cp(r(17), 0, R.col_row(c - 1, 22), 2); else cp(r(17), 0, pE, 0);         // and [eq_bit, r, eq_bit']
        cp(r(17), 1, r(14), 0);
        cp(r(18), 0, r(2), 0);                                                          // is_equal(carry, its range-assigned duplicate | acc_extra)
        if (!last) cp(r(18), 1, R.col_row(c, 17) + 1, 4); else cp(r(18), 1, r(8), 0);   // (the duplicate: column e of the assign's first row)
        cp(r(20), 0, r(18), 2); cp(r(20), 2, r(19), 0);
        cp(r(21), 0, r(19), 0); cp(r(21), 1, r(18), 2);
        cp(r(22), 0, r(17), 2); cp(r(22), 1, r(19), 0);
    }
    cp(R.col_row(C - 1, 22) + 1, 0, R.col_row(C - 1, 22), 2);                            // assert_one [eq_bit]  :1062 -- the record's last row
}

// ---- layout as data ----
// s'[column_of[k]] = s[k]: the selectors follow the cells; the two product terms a*b / c*d follow their pair of columns
inline bool layout_permute_fixed(const u8 (&col)[5], const h2r_fixed_row &in, h2r_fixed_row *out) {
    *out = in;
    const uint64_t (*src[5])[4] = {&in.sa, &in.sb, &in.sc, &in.sd, &in.se};
    uint64_t (*dst[5])[4] = {&out->sa, &out->sb, &out->sc, &out->sd, &out->se};
    for (int k = 0; k < 5; ++k) std::memcpy(*dst[col[k]], *src[k], sizeof in.sa);
    auto nz = [](const uint64_t (&v)[4]) { return (v[0] | v[1] | v[2] | v[3]) != 0; };
    const bool ab_to_cd = col[0] >= 2 && col[0] <= 3, cd_to_ab = col[2] <= 1;
    if (nz(in.s_mul_ab) && ab_to_cd) { std::memcpy(out->s_mul_cd, in.s_mul_ab, sizeof in.sa); if (!nz(in.s_mul_cd)) std::memset(out->s_mul_ab, 0, sizeof in.sa); }
    if (nz(in.s_mul_cd) && cd_to_ab) { std::memcpy(out->s_mul_ab, in.s_mul_cd, sizeof in.sa); if (!nz(in.s_mul_ab)) std::memset(out->s_mul_cd, 0, sizeof in.sa); }
    return true;
}
// what a permutation must respect for the gate / the lookups to keep meaning the same thing
inline bool layout_perm_valid(const u8 (&col)[5], const h2r_fixed_row &f, bool is_decompose_row) {
    bool seen[5] = {false, false, false, false, false};
    for (int k = 0; k < 5; ++k) { if (col[k] > 4 || seen[col[k]]) return false; seen[col[k]] = true; }
    auto nz = [](const uint64_t (&v)[4]) { return (v[0] | v[1] | v[2] | v[3]) != 0; };
    auto pair_ok = [&](int x, int y) { const int lo = col[x] < col[y] ? col[x] : col[y], hi = col[x] < col[y] ? col[y] : col[x]; return (lo == 0 && hi == 1) || (lo == 2 && hi == 3); };
    if (nz(f.s_mul_ab) && !pair_ok(0, 1)) return false;     // a * b must stay a product of one of the gate's two column pairs
    if (nz(f.s_mul_cd) && !pair_ok(2, 3)) return false;
    if (nz(f.s_mul_ab) && nz(f.s_mul_cd) && (col[0] >= 2) == (col[2] >= 2)) return false;
    if (is_decompose_row) {                                  // e carries "what remains" (se_next links it to the next row); the lookups read a..d, overflow reads a
        if (col[4] != 4 || col[0] != 0) return false;
    }
    return true;
}

struct LayoutArgs { const u8 *kinds; u64 rows; AdviceDst dst; u64 batch; const u8 *status; u8 perm[256][5]; };

// one thread per row: the row's five cells move to the columns the layout gives its kind (in place: a thread owns its row)
__global__ __launch_bounds__(256) void advice_layout_kernel(LayoutArgs a) {
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 elem = gid / a.rows;
    if (elem >= a.batch || (a.status && a.status[elem])) return;
    const u64 r = gid - elem * a.rows;
    const u8 *pm = a.perm[a.kinds[r]];
    if (pm[0] == 0 && pm[1] == 1 && pm[2] == 2 && pm[3] == 3 && pm[4] == 4) return;
    u8 *row = a.dst.elem(elem) + r * a.dst.row_pitch;   // (whole 32-byte cells move: the representation of their contents does not matter)
    uint4 v[10];
#pragma unroll
    for (int k = 0; k < 5; ++k) { const uint4 *c = reinterpret_cast<const uint4 *>(row + (u64)k * a.dst.col_pitch); v[2 * k] = c[0]; v[2 * k + 1] = c[1]; }
#pragma unroll
    for (int k = 0; k < 5; ++k) { uint4 *c = reinterpret_cast<uint4 *>(row + (u64)pm[k] * a.dst.col_pitch); c[0] = v[2 * k]; c[1] = v[2 * k + 1]; }
}

}  // namespace h2r
