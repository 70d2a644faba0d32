// gfx950 kernel of libh2r: the advice rows of BigIntChip::pow_mod (a VARIABLE exponent, big_integer/chip.rs:664-696) that are not
// mul_mod rows --
//   * main_gate.to_bits(limb, exp_limb_bits) of every exponent limb (:674-681).  maingate (halo2wrong rev 63bde545, not in the reference
//     tree: restated like the rest of DESIGN.md section 2b) assigns each bit with assign_bit, composes the bits with the same
//     `decompose` rows RangeChip::assign uses -- four terms per row in columns a..d, the LAST row reversed and zero-padded, column e =
//     what remains to be composed -- and asserts result == limb:
//         exp_limb_bits x  BIT [b, b, b]      ceil(exp_limb_bits / 4) x  BITS_COMPOSE [b_4r .. b_4r+3, remaining]      ASSERT_EQ [result, limb]
//   * per exponent bit, behind the rows of mul_mod(acc, squared): main_gate.select(muled.limb(j), acc.limb(j), e_bit) for every limb
//     (:688-691)      num_limbs x  SELECT [e_bit, muled_j, e_bit, acc_j, selected_j]
// The element's image is  [to_bits rows of every limb] [acc = assign_constant_fresh(1): CONST1, CONST0 (:682)]
//                         per bit: [mul_mod rows] [select rows] [square_mod rows]
// (the mul_mod rows are advice_kernel's / cells_kernel's, which leave `sel_rows` rows free behind every even record).
// One thread per row; the rows are 0.5 % of a Var element's bytes (plain stores: the 16-byte pieces of a row merge in L2).
#pragma once

#include "h2r_kernels.hpp"

namespace h2r {

enum : u32 { ROWK_SELECT_ = 15, ROWK_BITS_COMPOSE = 64 /* + row: a full row of four bits */,
             ROWK_BITS_COMPOSE_LAST = 80 /* + 4 * row + (terms - 1): the last (reversed, zero-padded) row */ };

struct VarRowsArgs {
    const u8 *trace; u64 elem_stride, off_e_bits, off_selected, selected_stride;   // the pow trace: e_bits[], selected[bit][limb]
    const void *opA, *opR; u64 op_stride;          // operands of item elem * T + t: acc = a of mul_mod 2 bit, muled = r of it
    const u8 *status; u64 batch;
    u32 L, T, nbits, exp_limb_bits, e_num_limbs;
    u32 rows;                                      // rows of one mul_mod
    AdviceDst dst; const MontK *mk;                // row 0 of dst = the element's first to_bits row
};

__host__ __device__ inline u32 var_to_bits_rows(u32 exp_limb_bits) { return exp_limb_bits + (exp_limb_bits + 3) / 4 + 1; }
// kind of row i of to_bits(limb, nb)
__host__ __device__ inline u32 var_to_bits_kind(u32 i, u32 nb) {
    const u32 nc = (nb + 3) / 4;
    if (i < nb) return ROWK_BIT;
    if (i < nb + nc) { const u32 rr = i - nb; return rr + 1 < nc ? ROWK_BITS_COMPOSE + rr : ROWK_BITS_COMPOSE_LAST + 4 * rr + (nb - 4 * rr - 1); }
    return ROWK_ASSERT_EQ;
}

template <int LW>
__global__ __launch_bounds__(256) void var_rows_kernel(VarRowsArgs a) {
    using limb_t = typename LimbT<LW>::type;
    const u32 nb = a.exp_limb_bits, per_limb = var_to_bits_rows(nb), rows_a = a.e_num_limbs * per_limb, rows_c = a.nbits * a.L;
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 elem = gid / (rows_a + rows_c);
    if (elem >= a.batch || (a.status && a.status[elem])) return;
    const u32 k = (u32)(gid - elem * (rows_a + rows_c));
    const u8 *et = a.trace + elem * a.elem_stride;
    u64 c[5] = {0, 0, 0, 0, 0};   // the five cells (every value of these rows fits 64 bits)
    u64 out_row;
    if (k < rows_a) {
        const u32 l = k / per_limb, i = k - l * per_limb, nc = (nb + 3) / 4;
        const u8 *bits = et + a.off_e_bits + (u64)l * nb;
        out_row = k;
        if (i < nb) { c[0] = c[1] = c[2] = bits[i]; }
        else {
            u64 limb = 0;
            for (u32 t = 0; t < nb; ++t) limb |= (u64)(bits[t] & 1u) << t;
            if (i < nb + nc) {
                const u32 rr = i - nb, lo = 4 * rr, n_terms = nb - lo < 4 ? nb - lo : 4;
                const bool last = rr + 1 == nc;
                for (u32 q = 0; q < n_terms; ++q) c[q] = bits[last ? lo + n_terms - 1 - q : lo + q];
                c[4] = lo >= 64 ? 0 : (limb >> lo) << lo;
            } else { c[0] = limb; c[1] = limb; }
        }
    } else {
        const u32 s = k - rows_a, t = s / a.L, j = s - t * a.L;
        const u64 item = elem * a.T + 2ull * t;
        const u64 bit = et[a.off_e_bits + t];
        const u64 muled = reinterpret_cast<const limb_t *>(a.opR)[item * a.op_stride + j];
        const u64 acc = reinterpret_cast<const limb_t *>(a.opA)[item * a.op_stride + j];
        const u64 sel = reinterpret_cast<const limb_t *>(et + a.off_selected + (u64)t * a.selected_stride)[j];
        c[0] = bit; c[1] = muled; c[2] = bit; c[3] = acc; c[4] = sel;
        out_row = (u64)rows_a + 2 + (u64)t * (2ull * a.rows + a.L) + a.rows + j;
    }
    u8 *img = a.dst.elem(elem);
#pragma unroll
    for (int q = 0; q < 5; ++q) advice_put_cell(a.dst, a.mk, img, out_row, q, make_uint4((u32)c[q], (u32)(c[q] >> 32), 0, 0), make_uint4(0, 0, 0, 0));
}

}  // namespace h2r
