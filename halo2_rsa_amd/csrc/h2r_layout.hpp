// Host-side parameter and trace-layout derivation for libh2r (no device code here).
//
// Mirrors the parameter functions of the reference:
//   BigIntChip::new               src/big_integer/chip.rs:1174-1185
//   BigIntChip::compute_range_lens src/big_integer/chip.rs:1220-1249
//   sublimb_bit_len / compute_mul_word_max / bits_size   src/big_integer/chip.rs:1352-1372
//   is_equal_muled's word_max / carry_bits               src/big_integer/chip.rs:838-842
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "h2r.h"

namespace h2r {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;
using u128 = unsigned __int128;

constexpr u32 kNumLookupLimbs = 8;  // big_integer/chip.rs:1163

// Small fixed 256-bit unsigned integer for host-side constants (word_max and the acc_extra chain).
struct U256 {
    u64 v[4] = {0, 0, 0, 0};
    static U256 from64(u64 x) { U256 r; r.v[0] = x; return r; }
    U256 operator+(const U256 &o) const {
        U256 r; u128 c = 0;
        for (int i = 0; i < 4; ++i) { c += (u128)v[i] + o.v[i]; r.v[i] = (u64)c; c >>= 64; }
        return r;
    }
    U256 operator-(const U256 &o) const {
        U256 r; u64 br = 0;
        for (int i = 0; i < 4; ++i) {
            u64 t = v[i] - o.v[i]; u64 b1 = v[i] < o.v[i];
            u64 t2 = t - br; u64 b2 = t < br; r.v[i] = t2; br = b1 | b2;
        }
        return r;
    }
    U256 shr(unsigned s) const {  // 0 < s <= 64
        U256 r;
        if (s == 64) { r.v[0] = v[1]; r.v[1] = v[2]; r.v[2] = v[3]; r.v[3] = 0; return r; }
        for (int i = 0; i < 4; ++i) r.v[i] = (v[i] >> s) | (i < 3 ? v[i + 1] << (64 - s) : 0);
        return r;
    }
    U256 shl(unsigned s) const {  // 0 < s <= 64
        U256 r;
        if (s == 64) { r.v[3] = v[2]; r.v[2] = v[1]; r.v[1] = v[0]; r.v[0] = 0; return r; }
        for (int i = 3; i >= 0; --i) r.v[i] = (v[i] << s) | (i > 0 ? v[i - 1] >> (64 - s) : 0);
        return r;
    }
    u64 low(unsigned w) const { return w == 64 ? v[0] : (v[0] & ((1ull << w) - 1)); }
    unsigned bits() const {
        for (int i = 3; i >= 0; --i) if (v[i]) return 64u * i + (64u - (unsigned)__builtin_clzll(v[i]));
        return 0;
    }
};

inline u32 sublimb_bit_len(u32 bit_len_limb) {  // big_integer/chip.rs:1357-1365
    u32 val = bit_len_limb / kNumLookupLimbs;
    return val == 0 ? 1 : val;
}
inline u32 n_sublimbs(u32 bit_len) {
    u32 s = sublimb_bit_len(bit_len);
    return bit_len / s + (bit_len % s ? 1 : 0);
}
inline U256 compute_mul_word_max(u32 w, u32 min_n) {  // big_integer/chip.rs:1368-1372
    u64 bm1 = w == 64 ? ~0ull : ((1ull << w) - 1);
    u128 sq = (u128)bm1 * bm1;
    U256 s; s.v[0] = (u64)sq; s.v[1] = (u64)(sq >> 64);
    U256 acc;
    for (u32 i = 0; i < min_n; ++i) acc = acc + s;
    return acc + U256::from64(bm1);
}
inline void compute_range_lens(u32 w, u32 L, u32 comp[3], u32 over[3]) {  // big_integer/chip.rs:1220-1249
    u32 out_comp = w / kNumLookupLimbs;
    u32 out_over = w % out_comp;
    u32 fresh_carry_bits = (w + 2) - w;  // bits(2 * 2^w) - w
    u32 fresh_comp = sublimb_bit_len(fresh_carry_bits);
    u32 fresh_over = fresh_carry_bits % fresh_comp;
    U256 wm = compute_mul_word_max(w, L);
    u32 mul_carry_bits = (wm + wm).bits() - w;
    u32 mul_comp = sublimb_bit_len(mul_carry_bits);
    u32 mul_over = mul_carry_bits % mul_comp;
    comp[0] = out_comp; comp[1] = fresh_comp; comp[2] = mul_comp;
    over[0] = out_over; over[1] = fresh_over; over[2] = mul_over;
}
inline u32 field_num_bits(u32 field) {
    switch (field) {
        case H2R_FIELD_BN254_FR: case H2R_FIELD_BN254_FQ: return 254;
        case H2R_FIELD_PASTA_FP: case H2R_FIELD_PASTA_FQ: return 255;
        default: return 0;
    }
}

// The field modulus p (little-endian 64-bit words): a_b = a[i] - b[i] (big_integer/chip.rs:859) is a FIELD subtraction,
// so a negative difference is the element p - |x|.
inline void field_modulus(u32 field, u64 p[4]) {
    static const u64 kP[4][4] = {
        {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},   // bn256::Fr
        {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},   // bn256::Fq
        {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull, 0x4000000000000000ull},   // pasta::Fp
        {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0000000000000000ull, 0x4000000000000000ull},   // pasta::Fq
    };
    for (int i = 0; i < 4; ++i) p[i] = kP[field < 4 ? field : 0][i];
}

inline u64 round_up(u64 x, u64 a) { return (x + a - 1) / a * a; }

// Fills every field of h2r_layout for (w, L).  Plane sizes are rounded up to 256 bytes so that every
// plane (and every record) starts on a 256-byte boundary; the padding is never written.
inline void layout_compute(u32 w, u32 L, h2r_layout *o) {
    std::memset(o, 0, sizeof *o);
    const u32 C = 2 * L - 1;
    U256 wm = compute_mul_word_max(w, L);
    o->limb_width = w; o->num_limbs = L; o->num_cols = C;
    o->word_max_bits = wm.bits();
    o->carry_bits = (wm + wm).bits() - w;  // big_integer/chip.rs:841-842
    o->limb_bytes = w / 8;
    o->wide_bytes = 8 * ((o->word_max_bits + 2 + 63) / 64);
    o->carry_bytes = 8 * ((o->carry_bits + 63) / 64);
    o->limb_sub_bits = sublimb_bit_len(w);
    o->limb_nsub = n_sublimbs(w);
    o->carry_sub_bits = sublimb_bit_len(o->carry_bits);
    o->carry_nsub = n_sublimbs(o->carry_bits);
    o->carry_sub_stride = 16;  // one 16-byte store per column (carry_nsub <= 16)
    const u32 LB = o->limb_bytes, WB = o->wide_bytes, CB = o->carry_bytes;
    const u32 HI = WB > 16 ? WB - 16 : 0;
    auto set = [&](int p, u32 elem, u32 count) { o->plane_elem[p] = elem; o->plane_count[p] = elem ? count : 0; };
    set(H2R_PL_Q, LB, L); set(H2R_PL_R, LB, L);
    set(H2R_PL_Q_SUB, (u32)round_up(o->limb_nsub, 8), L); set(H2R_PL_R_SUB, (u32)round_up(o->limb_nsub, 8), L);
    set(H2R_PL_AB_LO, 16, L * L); set(H2R_PL_AB_HI, HI, L * L);
    set(H2R_PL_QN_LO, 16, L * L); set(H2R_PL_QN_HI, HI, L * L);
    set(H2R_PL_EQB_LO, 16, L); set(H2R_PL_EQB_HI, HI, L);
    set(H2R_PL_AMB_LO, 16, C); set(H2R_PL_AMB_HI, HI, C);
    set(H2R_PL_SUM_LO, 16, C); set(H2R_PL_SUM_HI, HI, C);
    set(H2R_PL_CARRY, CB, C); set(H2R_PL_CMOD, LB, C);
    set(H2R_PL_NQ1_LO, 16, C); set(H2R_PL_NQ1_HI, HI, C); set(H2R_PL_AMNQ1, LB, C);
    set(H2R_PL_ACCX_LO, 16, C); set(H2R_PL_ACCX_HI, HI, C);
    set(H2R_PL_QACC, CB, C); set(H2R_PL_MODACC, LB, C);
    set(H2R_PL_NQ2_LO, 16, C); set(H2R_PL_NQ2_HI, HI, C); set(H2R_PL_AMNQ2, LB, C);
    set(H2R_PL_FLAGS, 4, C); set(H2R_PL_CARRY_DUP, CB, C - 1); set(H2R_PL_CARRY_SUB, o->carry_sub_stride, C - 1);
    // Accumulator planes: planar (four separate regions, rows of L entries) or interleaved row by row -- group g of
    // acc_steps_per_group steps holds the LO rows [ab half | qn half] of its steps and then one HI row [ab | qn]
    // (limb_width 64: the third words of two steps share 16-byte slots), so that one step of one mul_mod writes ONE
    // contiguous run and its product phase is one sequential stream.  Measured choice per shape below.
    // (same-box A/B, profiles/r01_layout_ab.txt: interleaved +0.5..1.3 % for 64-bit limbs, no difference for 32-bit)
    const bool interleaved = (w == 64);
    u64 off = 0;
    for (int p = 0; p < H2R_PL_COUNT; ++p) {
        if (interleaved && p == H2R_PL_AB_LO) {
            const u64 acc = off;
            o->acc_steps_per_group = HI ? 2 : 1;
            o->acc_lo_row_bytes = 2 * L * 16;
            o->acc_lo_group_bytes = (u64)o->acc_steps_per_group * o->acc_lo_row_bytes + (HI ? 2ull * L * 16 : 0);
            o->acc_hi_group_bytes = o->acc_lo_group_bytes;
            o->plane_off[H2R_PL_AB_LO] = acc;
            o->plane_off[H2R_PL_QN_LO] = acc + (u64)L * 16;
            o->plane_off[H2R_PL_AB_HI] = acc + (u64)o->acc_steps_per_group * o->acc_lo_row_bytes;
            o->plane_off[H2R_PL_QN_HI] = o->plane_off[H2R_PL_AB_HI] + (u64)L * 16;
            off += round_up((u64)(L / o->acc_steps_per_group) * o->acc_lo_group_bytes, 256);
            continue;
        }
        if (interleaved && (p == H2R_PL_AB_HI || p == H2R_PL_QN_LO || p == H2R_PL_QN_HI)) continue;
        o->plane_off[p] = off;
        // per-column planes reserve C+1 entries so that a full 2L-thread group may address them
        u64 cnt = o->plane_count[p];
        if (cnt == C || cnt == C - 1) cnt = 2 * L;
        off += round_up((u64)o->plane_elem[p] * cnt, 256);
    }
    if (!interleaved) {
        o->acc_steps_per_group = 1; o->acc_lo_row_bytes = 0;
        o->acc_lo_group_bytes = (u64)L * 16; o->acc_hi_group_bytes = (u64)L * 16;
    }
    o->record_stride = off;
    // HBM channel interleaving: the record kernel is sensitive to the record stride (one 256-byte unit less or more
    // than this choice costs 2..10 %, profiles/r01_stride_sweep.txt; re-swept after every layout change).
#ifndef H2R_RECORD_PAD_UNITS
#define H2R_RECORD_PAD_UNITS 2   // (developer variants: tools/record_pad_ab.sh)
#endif
    if (w == 64 && L == 32) o->record_stride += 256ull * H2R_RECORD_PAD_UNITS;
    const u64 per_col = 5ull * WB + 2ull * CB + 4ull * LB + 4;
    o->stream_bytes = 2ull * L * (LB + o->limb_nsub) + 2ull * L * L * WB + (u64)L * WB + (u64)C * per_col +
                      (u64)(C - 1) * (CB + o->carry_nsub);
}

// RefreshAux::new(limb_width, n_l, n_r).increased_limbs_vec (big_integer/mod.rs:428-482): how many extra limbs the
// i-th Muled limb spills into when it is cut into limb_width-bit chunks.  Returns the vector length.
inline u32 refresh_aux_increased_limbs(u32 w, u32 n_l, u32 n_r, u8 *inc /* >= n_l + n_r + 2 entries */) {
    const u32 d = n_l + n_r - 1;
    U256 muled[2 * 128 + 8];
    u32 len = d;
    const u64 bm1 = w == 64 ? ~0ull : ((1ull << w) - 1);
    const u128 sq = (u128)bm1 * bm1;
    U256 sqv; sqv.v[0] = (u64)sq; sqv.v[1] = (u64)(sq >> 64);
    for (u32 i = 0; i < d; ++i) {   // products a[j] * b[i-j] in column i: j from max(0, i+1-n_r) to min(i, n_l-1)  (mod.rs:438-447)
        const u32 j0 = n_r >= i + 1 ? 0 : i + 1 - n_r, j1 = i < n_l - 1 ? i : n_l - 1;
        for (u32 k = j0; k <= j1; ++k) muled[i] = muled[i] + sqv;
    }
    u32 n = 0;
    for (u32 cur = 0; cur <= d; ++cur) {
        if (cur >= len) muled[len++] = U256();
        const u32 nb = muled[cur].bits();
        const u32 chunks = nb % w == 0 ? nb / w : nb / w + 1;
        inc[n++] = (u8)(chunks - 1);
        U256 t = muled[cur]; u64 ch[8];
        for (u32 j = 0; j < chunks; ++j) { ch[j] = t.low(w); t = t.shr(w); }
        muled[cur] = U256();
        for (u32 j = 0; j < chunks; ++j) { while (len <= cur + j) muled[len++] = U256(); muled[cur + j] = muled[cur + j] + U256::from64(ch[j]); }
    }
    return n;
}

// Exponent helpers: e.to_bytes_le() + Self::bits_size(e) (big_integer/chip.rs:717-728).
inline u32 exp_num_bits(const u8 *e_le, size_t e_len) {
    while (e_len > 0 && e_le[e_len - 1] == 0) --e_len;
    if (e_len == 0) return 0;
    return (u32)(8 * (e_len - 1) + (32 - __builtin_clz((unsigned)e_le[e_len - 1])));
}
inline u32 exp_bit(const u8 *e_le, u32 i) { return (e_le[i / 8] >> (i % 8)) & 1u; }

}  // namespace h2r
