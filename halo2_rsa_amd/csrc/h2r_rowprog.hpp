// Advice rows of the Fresh-integer family (add, sub, add_mod, sub_mod, is_zero, the comparisons, assert_in_field):
// the ops' flat-stream values (AuxGeom regions written by aux_kernel / fresh_kernel / the step kernel's aux role) laid out
// as rows of the main gate's five advice columns, every cell the ops assign.
//
// These ops are a tree of small main-gate calls whose shape depends only on (op, limb counts), never on the data.  The
// host therefore walks the reference's control flow ONCE per (ctx, op) SYMBOLICALLY -- each value is "where it lives"
// (a byte range of the element's region, a limb of an operand, a constant) instead of a number -- and records one
// descriptor per row (RowProgBuilder, below: same walk as big_integer/chip.rs:245-373, 452-528, 754-805, 908-1006,
// 1286-1318).  The kernel is an interpreter of that table: one thread per row, 5 cells fetched, 160 bytes staged in
// LDS and written out as whole 16-byte lines.  ~1,500 rows (240 KB) per assert_in_field of RSA-2048.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "h2r_kernels.hpp"
#include "h2r_field.hpp"

namespace h2r {

// row kinds beyond the mul_mod image's (h2r.h H2R_ROW_*)
enum { ROWK_SELECT = 15, ROWK_NOT = 16, /* ROWK_ASSERT_ONE = 17: h2r_kernels.hpp (the mul_mod image ends with one) */ ROWK_CONST_BM1 = 18, ROWK_ASSERT_ZERO = 19,
       ROWK_CONST_EM = 20,    // + j: assign_constant of the j-th constant of the encoded-message check (em_const)
       ROWK_RANGE_U32 = 48,   // + row of RangeChip::assign(value, 4, 32): eight 4-bit sub-limbs (src/chip.rs:170-171)
       ROWK_CONST_COEFF8 = 56 }; // + j: assign_constant(2^(8j)), the byte coefficients of the hashed-message limbs (src/lib.rs:228-229)

// the constants RSAChip::verify_pkcs1v15_signature assigns (src/chip.rs:149-152, 169, 174, 179, 190)
constexpr u32 EM_CONSTS = 6;
__host__ __device__ inline u64 em_const(u32 j) {
    switch (j) {
        case 0: return 217300885422736416ull;    // prefix_64_1
        case 1: return 938447882527703397ull;    // prefix_64_2
        case 2: return 1ull << 32;               // u32_v
        case 3: return 3158320ull;               // prefix_32
        case 4: return 4294967295ull;            // ff_32
        default: return 562949953421311ull;      // last_em
    }
}

enum : u8 {
    RP_ZERO = 0, RP_ONE, RP_B /* 2^w */, RP_BM1 /* 2^w - 1 */,
    RP_STREAM,              // little-endian unsigned integer of `width` bytes at region + off
    RP_IN_A, RP_IN_B, RP_IN_N,   // limb `off` of the element's operand
    RP_DIFF01,              // cell 0 - cell 1 (field subtraction): main_gate.sub's result
    RP_DIFF34,              // operand 3 - operand 4: the value is_zero is asked about
    RP_INV34,               // its inverse, or 1 when it is zero: is_zero's witness
    RP_CONST64,             // em_const(off)
    RP_POW2,                // 2^off, off < 64
    RP_HIDDEN = 0x80        // operand only: the cell itself is unassigned (zero)
};
struct RpCell { u32 off; u8 type, width; uint16_t pad; };
struct RpRow { RpCell c[5]; u32 kind, pad; };   // 48 bytes
static_assert(sizeof(RpRow) == 48, "three 16-byte loads per row");

// ---- host: the symbolic walk ---------------------------------------------------------------------------------------
struct RowProgBuilder {
    const AuxGeom g;
    std::vector<RpRow> rows;
    u64 off = 0;   // section cursor inside the element's region, as fresh_sections / the kernels advance it
    using V = RpCell;
    using Int = std::vector<V>;

    explicit RowProgBuilder(const AuxGeom &g_) : g(g_) {}
    static V mk(u8 type, u32 off = 0, u8 width = 0) { V v; v.off = off; v.type = type; v.width = width; v.pad = 0; return v; }
    static V zero() { return mk(RP_ZERO); }
    static V one() { return mk(RP_ONE); }
    V st(u64 o, u32 width) const { return mk(RP_STREAM, (u32)o, (u8)width); }
    static V hidden(V v) { v.type |= RP_HIDDEN; return v; }
    static Int operand(u8 type, u32 n) { Int r; for (u32 i = 0; i < n; ++i) r.push_back(mk(type, i)); return r; }

    void row(u32 kind, V c0 = zero(), V c1 = zero(), V c2 = zero(), V c3 = zero(), V c4 = zero()) {
        RpRow r; r.c[0] = c0; r.c[1] = c1; r.c[2] = c2; r.c[3] = c3; r.c[4] = c4; r.kind = kind; r.pad = 0;
        rows.push_back(r);
    }
    void range_limb(u64 ra_off) {   // RangeChip::assign(limb, w/8, w): value at ra_off, its eight sub-limb bytes behind it
        row(ROWK_RANGE_LIMB + 0, st(ra_off, g.LB));
        row(ROWK_RANGE_LIMB + 1, st(ra_off, g.LB));
    }
    void is_zero_rows(V x, V y, V flag) {   // main_gate.is_zero(d = x - y): assign_bit(r), [d, d', r], [r, d]
        row(ROWK_BIT, flag, flag, flag);
        row(ROWK_ISZERO_INV, mk(RP_DIFF34), mk(RP_INV34), flag, hidden(x), hidden(y));
        row(ROWK_ISZERO_RA, flag, mk(RP_DIFF34), zero(), hidden(x), hidden(y));
    }
    void is_equal(V x, V y, V flag) {       // main_gate.is_equal = sub + is_zero
        row(ROWK_SUB, x, y, mk(RP_DIFF01));
        is_zero_rows(x, y, flag);
    }

    Int add(const Int &a, const Int &b) {   // chip.rs:245-297; section layout of aux_add
        const u32 n = (u32)std::max(a.size(), b.size());
        const u64 sec = off;
        off += g.add_sz(n);
        row(ROWK_CONST0);                   // zero_value :254
        row(ROWK_CONST_B, mk(RP_B));        // limb_max_val :267
        Int out;
        V carry = zero();
        for (u32 i = 0; i < n; ++i) {
            const u64 q = sec + (u64)i * g.STEP;
            const V ai = i < a.size() ? a[i] : zero(), bi = i < b.size() ? b[i] : zero();
            const V a_b = st(q, g.SB), sum = st(q + g.SB, g.SB), c = st(q + 2 * g.SB, g.LB), cy = st(q + 2 * g.SB + g.RA, g.LB),
                    cac = st(q + 2 * g.SB + 2 * g.RA, g.SB);
            row(ROWK_ADD, ai, bi, a_b);                 // :272
            row(ROWK_ADD, a_b, carry, sum);             // :273
            range_limb(q + 2 * g.SB);                   // c :279-280
            range_limb(q + 2 * g.SB + g.RA);            // carry :281-282
            row(ROWK_MUL_ADD, cy, mk(RP_B), c, cac);    // :283
            row(ROWK_ASSERT_EQ, sum, cac);              // :285
            out.push_back(c);
            carry = cy;
        }
        out.push_back(carry);                           // :290
        return out;
    }
    V is_equal_fresh(const Int &a, const Int &b) {   // chip.rs:780-805; (flag, running AND) byte pairs of aux_eq
        const u32 n1 = (u32)a.size(), n2 = (u32)b.size();
        const bool larger = n1 > n2;
        const u32 n = larger ? n1 : n2;
        const u64 sec = off;
        off += g.eq_sz(n);
        row(ROWK_BIT, one(), one(), one());
        V eq = one();
        for (u32 i = 0; i < n; ++i) {
            V flag = st(sec + 2ull * i, 1), run = st(sec + 2ull * i + 1, 1);
            if (larger && i >= n2) is_zero_rows(a[i], zero(), flag);
            else if (!larger && i >= n1) is_zero_rows(b[i], zero(), flag);
            else is_equal(a[i], b[i], flag);
            row(ROWK_MUL, eq, flag, run);               // and
            eq = run;
        }
        return eq;
    }
    V is_zero(const Int &a) {                        // chip.rs:754-767 (the kernel writes it as aux_eq against zero)
        const u64 sec = off;
        off += g.eq_sz((u32)a.size());
        row(ROWK_BIT, one(), one(), one());
        V bit = one();
        for (u32 i = 0; i < a.size(); ++i) {
            V flag = st(sec + 2ull * i, 1), run = st(sec + 2ull * i + 1, 1);
            is_zero_rows(a[i], zero(), flag);
            row(ROWK_MUL, bit, flag, run);
            bit = run;
        }
        return bit;
    }
    Int sub_unchecked(const Int &a, const Int &b) {  // chip.rs:1286-1318; aux_subu
        const u32 n1 = (u32)a.size();
        const u64 sec = off;
        off += g.cl_sz(n1);
        Int c;
        for (u32 i = 0; i < n1; ++i) { range_limb(sec + (u64)i * g.RA); c.push_back(st(sec + (u64)i * g.RA, g.LB)); }
        const Int added = add(b, c);
        const V ok = is_equal_fresh(a, added);
        row(ROWK_ASSERT_ONE, ok);                       // assert_equal_fresh -> assert_one
        return c;
    }
    std::pair<Int, V> sub(const Int &a, const Int &b) {   // chip.rs:310-373; aux_sub
        const u32 nA = (u32)a.size(), n2 = (u32)b.size(), m = std::max(nA, n2), n1 = m + 1;
        Int max_int;
        for (u32 i = 0; i < n2; ++i) { row(ROWK_CONST_BM1, mk(RP_BM1)); max_int.push_back(mk(RP_BM1)); }   // max_value :138-154
        const Int inflated_a = add(a, max_int);
        const Int is_ = sub_unchecked(inflated_a, b);
        const V not_ov = st(off, 1), ov = st(off + 1, 1);
        off += 16;
        row(ROWK_BIT, one(), one(), one());             // one :326
        is_equal(is_[n2], one(), not_ov);               // :330
        row(ROWK_NOT, not_ov, ov);                      // :331
        row(ROWK_CONST0);                               // zero_value :343
        Int sel_l, sel_r;
        const u64 sl = off, sr = off + AuxGeom::a16((u64)n1 * g.LB);
        off = sr + AuxGeom::a16((u64)m * g.LB);
        for (u32 i = 0; i < n1; ++i) {                  // :345-357
            const V v = st(sl + (u64)i * g.LB, g.LB);
            row(ROWK_SELECT, not_ov, is_[i], not_ov, i >= n2 ? zero() : b[i], v);
            sel_l.push_back(v);
        }
        for (u32 i = 0; i < m; ++i) {                   // :358-367
            const V v = st(sr + (u64)i * g.LB, g.LB);
            if (i >= nA) row(ROWK_SELECT, not_ov, max_int[i], not_ov, zero(), v);
            else if (i >= n2) row(ROWK_SELECT, not_ov, zero(), not_ov, a[i], v);
            else row(ROWK_SELECT, not_ov, max_int[i], not_ov, a[i], v);
            sel_r.push_back(v);
        }
        Int real = sub_unchecked(sel_l, sel_r);
        return {real, ov};
    }
    V is_less_than(const Int &a, const Int &b) {     // chip.rs:908-919; aux_less_than
        const V ov = sub(a, b).second;
        const V is_eq = is_equal_fresh(a, b);
        const V is_not_eq = st(off, 1), lt = st(off + 1, 1);
        off += 16;
        row(ROWK_NOT, is_eq, is_not_eq);                // :917
        row(ROWK_MUL, ov, is_not_eq, lt);               // :918
        return lt;
    }
    void select_mod(const Int &x, const Int &y, V cond, u32 n_limbs) {   // tail of add_mod :466-478 / sub_mod :512-525
        row(ROWK_CONST0);
        const u32 num = (u32)std::max(x.size(), y.size());
        for (u32 i = 0; i < num; ++i)
            row(ROWK_SELECT, cond, i < x.size() ? x[i] : zero(), cond, i < y.size() ? y[i] : zero(), st(off + (u64)i * g.LB, g.LB));
        for (u32 i = n_limbs; i < num; ++i) row(ROWK_ASSERT_ZERO, st(off + (u64)i * g.LB, g.LB));
    }

    void range_u32(u64 ra_off) { row(ROWK_RANGE_U32 + 0, st(ra_off, 4)); row(ROWK_RANGE_U32 + 1, st(ra_off, 4)); }
    // RSAChip::verify_pkcs1v15_signature after the modpow (src/chip.rs:138-198; 64-bit limbs): operand A = powed, B = hashed (4 limbs);
    // region = the encoded-message region the aux role wrote (flags, the two 32-bit range assigns, the recomposed limb)
    void build_em() {
        const u32 L = g.L;
        const Int A = operand(RP_IN_A, L), H = operand(RP_IN_B, 4);
        auto K = [&](u32 j) { return mk(RP_CONST64, j); };
        V is_eq = one();                                    // the cell of the preamble's assign_constant(1), :137
        auto and_ = [&](V flag, V run) { row(ROWK_MUL, is_eq, flag, run); is_eq = run; };
        for (u32 i = 0; i < 4; ++i) { is_equal(A[i], H[i], st(2 * i, 1)); and_(st(2 * i, 1), st(2 * i + 1, 1)); }   // :141-144
        row(ROWK_CONST_EM + 0, K(0)); row(ROWK_CONST_EM + 1, K(1));                                                  // :149-152
        is_equal(A[4], K(0), st(8, 1)); is_equal(A[5], K(1), st(9, 1));                                              // :153-154
        and_(st(8, 1), st(10, 1)); and_(st(9, 1), st(11, 1));                                                        // :155-156
        range_u32(12); range_u32(24);                                                                                // :170-171
        const V low = st(12, 4), high = st(24, 4), concat = st(36, 8);
        row(ROWK_CONST_EM + 2, K(2));                                                                                // :172
        row(ROWK_MUL_ADD, high, K(2), low, concat);                                                                  // :173
        row(ROWK_ASSERT_EQ, A[6], concat);                                                                           // :174
        row(ROWK_CONST_EM + 3, K(3)); is_equal(low, K(3), st(44, 1)); and_(st(44, 1), st(45, 1));                    // :175-177
        row(ROWK_CONST_EM + 4, K(4)); is_equal(high, K(4), st(46, 1)); and_(st(46, 1), st(47, 1));                   // :180-182
        row(ROWK_CONST_BM1, mk(RP_BM1));                                                                             // ff_64 :183-184
        for (u32 i = 7; i + 1 < L; ++i) { const u64 f = 48 + 2ull * (i - 7); is_equal(A[i], mk(RP_BM1), st(f, 1)); and_(st(f, 1), st(f + 1, 1)); }   // :185-188
        const u64 f = 48 + 2ull * (L - 8);
        row(ROWK_CONST_EM + 5, K(5)); is_equal(A[L - 1], K(5), st(f, 1)); and_(st(f, 1), st(f + 1, 1));              // :190-197
    }
    // RSASignatureVerifier::verify_pkcs1v15_signature, src/lib.rs:225-239: the four hashed-message limbs composed from the 32
    // reversed digest bytes; region = [32 byte cells][32 running limb values, 8 bytes each] (sha256_kernel, h2r_sha256.hpp)
    void build_hashed_msg() {
        for (u32 i = 0; i < 4; ++i) {
            row(ROWK_CONST0);                                                        // limb_val = assign_constant(0) :226
            V limb = zero();
            for (u32 j = 0; j < 8; ++j) {
                const V coeff = mk(RP_POW2, 8 * j), nv = st(32 + 8ull * (8 * i + j), 8);
                row(ROWK_CONST_COEFF8 + j, coeff);                                   // :228-229
                row(ROWK_MUL_ADD, coeff, st(8 * i + j, 1), limb, nv);                // :230-235
                limb = nv;
            }
        }
    }
    void build_verify_preamble() { row(ROWK_CONST1, one()); }   // is_eq = assign_constant(1), src/chip.rs:137 (before the modpow)

    // returns false for an unknown op
    bool build(u32 op, bool assert_one) {
        const u32 L = g.L;
        const Int A = operand(RP_IN_A, L), B = operand(RP_IN_B, L), N = operand(RP_IN_N, L);
        V bit = zero();
        bool has_bit = true;
        switch (op) {
            case FRESH_ADD: add(A, B); has_bit = false; break;
            case FRESH_SUB: sub(A, B); has_bit = false; break;
            case FRESH_ADD_MOD: { const Int added = add(A, B); auto s = sub(added, N); select_mod(added, s.first, s.second, L); has_bit = false; break; }
            case FRESH_SUB_MOD: {
                auto s1 = sub(A, B); auto s2 = sub(N, s1.first);
                row(ROWK_ASSERT_ZERO, s2.second);       // :510
                select_mod(s2.first, s1.first, s1.second, L); has_bit = false; break;
            }
            case FRESH_IS_ZERO: bit = is_zero(A); break;
            case FRESH_IS_EQUAL_FRESH: bit = is_equal_fresh(A, B); break;
            case FRESH_IS_LESS_THAN: case FRESH_IS_IN_FIELD: bit = is_less_than(A, B); break;
            case FRESH_IS_LESS_THAN_OR_EQUAL: bit = sub(A, B).second; break;
            case FRESH_IS_GREATER_THAN: { const V le = sub(A, B).second; bit = st(off, 1); row(ROWK_NOT, le, bit); break; }              // :954-963
            case FRESH_IS_GREATER_THAN_OR_EQUAL: { const V lt = is_less_than(A, B); bit = st(off, 1); row(ROWK_NOT, lt, bit); break; }   // :976-985
            default: return false;
        }
        if (assert_one) { if (!has_bit) return false; row(ROWK_ASSERT_ONE, bit); }
        return true;
    }
};

// ---- device: the interpreter ---------------------------------------------------------------------------------------
struct RowProgArgs {
    const RpRow *prog; u32 rows;
    const void *a, *b, *n; u64 a_stride, b_stride, n_stride;   // limbs between the elements' operands (0: one shared integer)
    const u8 *trace; u64 elem_stride, first_off;               // element e's region at trace + e * elem_stride + first_off
    const u8 *status; u64 batch;
    AdviceDst dst; const MontK *mk;     // the program's first row = row 0 of dst
    FieldConsts f;
    const u32 *inv_rows; u32 n_inv;     // the rows with an RP_INV34 cell (rowprog_inv_kernel fills those cells)
};

// the value a cell descriptor names, for element `elem` whose region starts at `reg`
template <int LW>
__device__ __forceinline__ Fe rp_fetch(const RowProgArgs &a, const u8 *reg, u32 elem, const RpCell &c) {
    using limb_t = typename LimbT<LW>::type;
    Fe v = fe_zero();
    switch (c.type & 0x7f) {
        case RP_ONE: v.v[0] = 1; break;
        case RP_B: if (LW == 64) v.v[1] = 1; else v.v[0] = 1ull << 32; break;
        case RP_BM1: v.v[0] = LW == 64 ? ~0ull : 0xffffffffull; break;
        case RP_STREAM: {
            const u8 *p = reg + c.off;
            if (c.width == 1) v.v[0] = p[0];
            else {   // 4, 8, 12 or 16 bytes on a 4-byte boundary
                const u32 *p4 = reinterpret_cast<const u32 *>(p);
                const u32 nw = c.width / 4;
                const u32 w0 = p4[0], w1 = nw > 1 ? p4[1] : 0, w2 = nw > 2 ? p4[2] : 0, w3 = nw > 3 ? p4[3] : 0;
                v.v[0] = ((u64)w1 << 32) | w0; v.v[1] = ((u64)w3 << 32) | w2;
            }
            break;
        }
        case RP_CONST64: v.v[0] = em_const(c.off); break;
        case RP_POW2: v.v[0] = 1ull << c.off; break;
        case RP_IN_A: v.v[0] = reinterpret_cast<const limb_t *>(a.a)[(u64)elem * a.a_stride + c.off]; break;
        case RP_IN_B: v.v[0] = reinterpret_cast<const limb_t *>(a.b)[(u64)elem * a.b_stride + c.off]; break;
        case RP_IN_N: v.v[0] = reinterpret_cast<const limb_t *>(a.n)[(u64)elem * a.n_stride + c.off]; break;
        default: break;
    }
    return v;
}

// SR rows per workgroup (= its threads), staged in SR x 160 bytes of LDS.  Next to a cells kernel that fills every CU's LDS a workgroup
// only starts where retiring cells workgroups have left SR x 160 bytes free: the smaller stage is the one that finds room.
template <int LW, u32 SR = 256>
__global__ __launch_bounds__(SR) void rowprog_kernel(RowProgArgs a) {
    using limb_t = typename LimbT<LW>::type;
    __shared__ uint4 stage[SR * (ADVICE_ROW_BYTES / 16)];
    const u32 tid = threadIdx.x;
    const u32 nchunks = (a.rows + SR - 1) / SR;
    const u32 item = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const u32 elem = item / nchunks, r0 = (item - elem * nchunks) * SR;
    if (a.status && a.status[elem]) return;
    const u8 *reg = a.trace + (u64)elem * a.elem_stride + a.first_off;
    const u32 r = r0 + tid;
    if (r < a.rows) {
        const uint4 *pr = reinterpret_cast<const uint4 *>(a.prog + r);
        union { uint4 q[3]; RpRow row; } u;
        u.q[0] = pr[0]; u.q[1] = pr[1]; u.q[2] = pr[2];
        const RpRow &rw = u.row;
        auto fetch = [&](const RpCell &c) -> Fe { return rp_fetch<LW>(a, reg, elem, c); };
        Fe c[5];
        const bool u32r = rw.kind >= ROWK_RANGE_U32 && rw.kind < ROWK_RANGE_U32 + 2;
        if ((rw.kind >= ROWK_RANGE_LIMB && rw.kind < ROWK_RANGE_LIMB + 2) || u32r) {
            // main_gate.decompose of a limb (eight w/8-bit sub-limbs) or of a 32-bit half (eight 4-bit ones): four sub-limb
            // bytes per row (the last row reversed), column e = what remains
            const u8 *p = reg + rw.c[0].off;
            const u32 *ps = reinterpret_cast<const u32 *>(p + (u32r ? 4 : LW / 8));
            const u64 subs = ((u64)ps[1] << 32) | ps[0];
            const u32 sb = u32r ? 4 : LW / 8;
            u64 rem = 0;
            const u32 k0 = (rw.kind & 1) ? 4u : 0u;
#pragma unroll
            for (u32 k = 0; k < 8; ++k) if (k >= k0) rem += ((subs >> (8 * k)) & 0xff) << (k * sb);
            for (int q = 0; q < 5; ++q) c[q] = fe_zero();
#pragma unroll
            for (u32 q = 0; q < 4; ++q) c[q].v[0] = (subs >> (8 * (k0 ? 7 - q : q))) & 0xff;
            c[4].v[0] = rem;
        } else {
#pragma unroll
            for (int q = 0; q < 5; ++q) c[q] = fetch(rw.c[q]);
            bool need34 = false, need_inv = false;
#pragma unroll
            for (int q = 0; q < 3; ++q) { const u32 t = rw.c[q].type; need34 = need34 || t == RP_DIFF34 || t == RP_INV34; need_inv = need_inv || t == RP_INV34; }
            Fe d34 = fe_zero(), inv = fe_zero();
            if (need34) {
                d34 = fe_sub(c[3], c[4], a.f.p);
                if (need_inv) inv.v[0] = 1;   // d = 0: is_zero's witness is 1; d != 0: rowprog_inv_kernel overwrites the cell with 1/d
            }
            const Fe d01 = fe_sub(c[0], c[1], a.f.p);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const u32 t = rw.c[q].type;
                if (t & RP_HIDDEN) c[q] = fe_zero();
                else if (t == RP_DIFF01) c[q] = d01;
                else if (t == RP_DIFF34) c[q] = d34;
                else if (t == RP_INV34) c[q] = inv;
            }
        }
        uint4 *sp = stage + (u64)tid * (ADVICE_ROW_BYTES / 16);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            sp[2 * q] = make_uint4((u32)c[q].v[0], (u32)(c[q].v[0] >> 32), (u32)c[q].v[1], (u32)(c[q].v[1] >> 32));
            sp[2 * q + 1] = make_uint4((u32)c[q].v[2], (u32)(c[q].v[2] >> 32), (u32)c[q].v[3], (u32)(c[q].v[3] >> 32));
        }
    }
    __syncthreads();
    const u32 n_rows = a.rows - r0 < SR ? a.rows - r0 : SR;
    advice_flush<SR>(a.dst, a.mk, a.dst.elem(elem), r0, n_rows, stage, tid);
}


// main_gate.is_zero's inverse witnesses, in their own launch: a Fermat inversion is ~380 Montgomery products (~0.5 ms for a
// wave), and inside rowprog_kernel the few rows that need one (d != 0: e.g. the 32 limb comparisons of is_equal_fresh(x, n) in an
// assert_in_field) are spread thinly over waves whose other lanes wait -- 5.1 ms for 1,024 RSA-2048 elements.  Here one thread
// looks at one (element, is_zero row); the rows with d != 0 are packed into a dense list in LDS, so the waves that invert are
// full, and every cell 1 = 1/d goes out as two 16-byte stores behind the image rowprog_kernel wrote.
template <int LW, u32 NT = 256>
__global__ __launch_bounds__(NT) void rowprog_inv_kernel(RowProgArgs a) {
    __shared__ u32 cnt;
    __shared__ u32 l_elem[NT], l_row[NT];
    __shared__ Fe l_d[NT];
    const u32 tid = threadIdx.x;
    if (tid == 0) cnt = 0;
    __syncthreads();
    const u64 g = (u64)blockIdx.x * NT + tid;
    if (g < a.batch * a.n_inv) {
        const u32 elem = (u32)(g / a.n_inv), r = a.inv_rows[g - (u64)elem * a.n_inv];
        if (!(a.status && a.status[elem])) {
            const RpRow *rw = a.prog + r;
            const u8 *reg = a.trace + (u64)elem * a.elem_stride + a.first_off;
            const Fe d = fe_sub(rp_fetch<LW>(a, reg, elem, rw->c[3]), rp_fetch<LW>(a, reg, elem, rw->c[4]), a.f.p);
            if (!fe_is_zero(d)) { const u32 slot = atomicAdd(&cnt, 1u); l_elem[slot] = elem; l_row[slot] = r; l_d[slot] = d; }
        }
    }
    __syncthreads();
    if (tid < cnt) {
        const Fe iv = fe_inv_fast(l_d[tid], a.f);
        advice_put_cell(a.dst, a.mk, a.dst.elem(l_elem[tid]), l_row[tid], 1, make_uint4((u32)iv.v[0], (u32)(iv.v[0] >> 32), (u32)iv.v[1], (u32)(iv.v[1] >> 32)),
                        make_uint4((u32)iv.v[2], (u32)(iv.v[2] >> 32), (u32)iv.v[3], (u32)(iv.v[3] >> 32)));
    }
}

}  // namespace h2r
