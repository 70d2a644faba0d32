// The chip shim, compiled: integration/gpu_chip.rs replays the reference's control flow over the flat witness stream the GPU wrote
// (h2r_pow_trace_emit_stream) and hands every value to the main gate / range chip in the order the reference's code issues the calls.
// Rust cannot be compiled in this image, so this is its twin in C++ against a MOCK RegionCtx / MainGate / RangeChip:
//   * every mock op checks the relation its gate enforces on the values it is given (mul_add: a b + c = out, ...), as MockProver would;
//   * every call is counted: the totals must be the reference's (19 mul_mods, 38,912 mul_add, 2,394 range assigns,
//     2,394 div_mod_main_gate per RSA-2048 e = 65537 element);
//   * the stream must be consumed to the last byte -- for pow_mod_fixed_exp (RSAPubE::Fix) and pow_mod (RSAPubE::Var).
// Function names, argument order and statement order follow big_integer/chip.rs:386-419 (mul), 542-629 (mul_mod), 642-649 (square_mod),
// 664-696 (pow_mod), 710-742 (pow_mod_fixed_exp), 822-895 (is_equal_muled), 1053-1063 (assert_equal_muled), 1323-1349
// (div_mod_main_gate) and src/chip.rs:99-114 (modpow_public_key).  TEST CODE.  Build: tests/test_chip_replay.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "h2r.h"

#define REQUIRE(cond)                                                                                       \
    do {                                                                                                    \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

// ---- a field element / an assigned cell -------------------------------------------------------------------------------------
struct Fe { uint64_t v[4]; };
static bool operator==(const Fe &a, const Fe &b) { return !std::memcmp(a.v, b.v, 32); }
static Fe small(uint64_t x) { return Fe{{x, 0, 0, 0}}; }
struct AssignedValue { Fe value; };                         // maingate's AssignedValue<F>: a cell and the value in it
using AssignedLimb = AssignedValue;                         // big_integer/mod.rs:237-266
using AssignedInteger = std::vector<AssignedLimb>;          // big_integer/mod.rs:306-382 (Fresh and Muled alike here)

struct Field {                                              // the ctx's field, through the library's own host arithmetic
    const h2r_ctx *ctx;
    Fe op(uint32_t o, const Fe &a, const Fe &b) const { Fe r; REQUIRE(h2r_field_eval(ctx, o, a.v, b.v, r.v) == H2R_OK); return r; }
    Fe add(const Fe &a, const Fe &b) const { return op(0, a, b); }
    Fe sub(const Fe &a, const Fe &b) const { return op(1, a, b); }
    Fe mul(const Fe &a, const Fe &b) const { return op(2, a, b); }
};

// ---- the mock region: counts what the shim asks of it ------------------------------------------------------------------------
struct MockRegionCtx {
    uint64_t assign_constant = 0, assign_value = 0, assign_bit = 0, mul_add = 0, add = 0, sub = 0, add_with_constant = 0, add_constant = 0, mul = 0,
             assert_equal = 0, is_equal = 0, and_ = 0, select = 0, to_bits = 0, range_assign = 0, range_sublimbs = 0, assert_one = 0;
};

// maingate::MainGate as the shim uses it: the OUTPUT value of every op comes from the GPU's stream (`out`), the mock checks the gate
struct MockMainGate {
    Field f;
    AssignedValue assign_constant(MockRegionCtx &c, const Fe &k) const { ++c.assign_constant; return {k}; }
    AssignedValue assign_value(MockRegionCtx &c, const Fe &v) const { ++c.assign_value; return {v}; }
    AssignedValue assign_bit(MockRegionCtx &c, const Fe &v) const { ++c.assign_bit; REQUIRE(v == small(0) || v == small(1)); return {v}; }
    AssignedValue mul_add(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const AssignedValue &acc, const Fe &out) const {
        ++c.mul_add; REQUIRE(f.add(f.mul(a.value, b.value), acc.value) == out); return {out};
    }
    AssignedValue add(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const Fe &out) const { ++c.add; REQUIRE(f.add(a.value, b.value) == out); return {out}; }
    AssignedValue sub(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const Fe &out) const { ++c.sub; REQUIRE(f.sub(a.value, b.value) == out); return {out}; }
    AssignedValue add_with_constant(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const Fe &k, const Fe &out) const {
        ++c.add_with_constant; REQUIRE(f.add(f.add(a.value, b.value), k) == out); return {out};
    }
    AssignedValue add_constant(MockRegionCtx &c, const AssignedValue &a, const Fe &k, const Fe &out) const { ++c.add_constant; REQUIRE(f.add(a.value, k) == out); return {out}; }
    AssignedValue mul(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const Fe &out) const { ++c.mul; REQUIRE(f.mul(a.value, b.value) == out); return {out}; }
    void assert_equal(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b) const { ++c.assert_equal; REQUIRE(a.value == b.value); }
    AssignedValue is_equal(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const Fe &out) const {
        ++c.is_equal; REQUIRE(out == small(a.value == b.value ? 1 : 0)); return {out};
    }
    AssignedValue and_(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const Fe &out) const { ++c.and_; REQUIRE(f.mul(a.value, b.value) == out); return {out}; }
    AssignedValue select(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &b, const AssignedValue &cond, const Fe &out) const {
        ++c.select; REQUIRE(out == (cond.value == small(1) ? a.value : b.value)); return {out};
    }
    std::vector<AssignedValue> to_bits(MockRegionCtx &c, const AssignedValue &x, const std::vector<uint8_t> &bits) const {
        ++c.to_bits;
        REQUIRE(bits.size() <= 64);
        uint64_t v = 0; std::vector<AssignedValue> out;
        for (size_t i = 0; i < bits.size(); ++i) { REQUIRE(bits[i] <= 1); v |= (uint64_t)bits[i] << i; out.push_back({small(bits[i])}); }
        REQUIRE(x.value == small(v));
        return out;
    }
    void assert_one(MockRegionCtx &c, const AssignedValue &a) const { ++c.assert_one; REQUIRE(a.value == small(1)); }
};
// maingate::RangeChip::assign(ctx, value, limb_bit_len, bit_len): the sub-limbs come from the stream too
struct MockRangeChip {
    AssignedValue assign(MockRegionCtx &c, const Fe &value, uint32_t limb_bit_len, uint32_t bit_len, const uint8_t *sub, uint32_t nsub) const {
        ++c.range_assign; c.range_sublimbs += nsub;
        REQUIRE(nsub == bit_len / limb_bit_len + (bit_len % limb_bit_len ? 1 : 0));
        unsigned __int128 acc = 0;
        for (uint32_t k = 0; k < nsub; ++k) {
            const uint32_t width = (k == nsub - 1 && bit_len % limb_bit_len) ? bit_len % limb_bit_len : limb_bit_len;
            REQUIRE(sub[k] < (1u << width));                                         // the lookup
            acc |= (unsigned __int128)sub[k] << (k * limb_bit_len);
        }
        REQUIRE(value.v[0] == (uint64_t)acc && value.v[1] == (uint64_t)(acc >> 64) && value.v[2] == 0 && value.v[3] == 0);   // the composition
        return {value};
    }
};

// ---- the witness stream of one element (h2r_pow_trace_emit_stream, H2R_STREAM_FIELD_AB) ---------------------------------------
struct WitnessStream {
    const uint8_t *p, *end; const h2r_layout *lo;
    Fe take(uint32_t bytes) { REQUIRE(p + bytes <= end && bytes <= 32); Fe r{{0, 0, 0, 0}}; std::memcpy(r.v, p, bytes); p += bytes; return r; }
    Fe limb() { return take(lo->limb_bytes); }
    Fe wide() { return take(lo->wide_bytes); }
    Fe carry() { return take(lo->carry_bytes); }
    Fe field() { return take(32); }
    Fe flag() { return take(1); }
    const uint8_t *bytes(uint32_t n) { REQUIRE(p + n <= end); const uint8_t *r = p; p += n; return r; }
};

// ---- the shim: reference control flow, values from the stream -----------------------------------------------------------------
struct GpuBigIntChip {
    const h2r_ctx *ctx; h2r_layout lo; MockMainGate main_gate; MockRangeChip range_chip; Fe word_max, limb_max_v;
    uint32_t limb_width, num_limbs;
    uint64_t n_mul_mod = 0, n_div_mod = 0, n_mul = 0;

    // big_integer/chip.rs:386-419
    AssignedInteger mul(MockRegionCtx &c, WitnessStream &s, const AssignedInteger &a, const AssignedInteger &b) {
        ++n_mul;
        const size_t d0 = a.size(), d1 = b.size(), d = d0 + d1 - 1;
        AssignedInteger c_vals;
        for (size_t i = 0; i < d; ++i) {
            AssignedValue acc = main_gate.assign_constant(c, small(0));
            size_t j = d1 >= i + 1 ? 0 : i + 1 - d1;
            while (j < d0 && j <= i) {
                const size_t k = i - j;
                acc = main_gate.mul_add(c, a[j], b[k], acc, s.wide());
                ++j;
            }
            c_vals.push_back(acc);
        }
        return c_vals;
    }
    // big_integer/chip.rs:1323-1349
    std::pair<AssignedValue, AssignedValue> div_mod_main_gate(MockRegionCtx &c, const AssignedValue &a, const AssignedValue &n, const Fe &q_v, const Fe &r_v,
                                                              const Fe &nq_v, const Fe &a_sub_nq_v) {
        ++n_div_mod;
        const AssignedValue q = main_gate.assign_value(c, q_v), a_mod_n = main_gate.assign_value(c, r_v);
        const AssignedValue nq = main_gate.mul(c, n, q, nq_v);
        const AssignedValue a_sub_nq = main_gate.sub(c, a, nq, a_sub_nq_v);
        main_gate.assert_equal(c, a_mod_n, a_sub_nq);
        return {q, a_mod_n};
    }
    // big_integer/chip.rs:822-895
    AssignedValue is_equal_muled(MockRegionCtx &c, WitnessStream &s, const AssignedInteger &a, const AssignedInteger &b, size_t num_limbs_l, size_t num_limbs_r) {
        const size_t nl = num_limbs_l + num_limbs_r - 1;
        const AssignedValue limb_max = main_gate.assign_constant(c, limb_max_v);
        AssignedValue accumulated_extra = main_gate.assign_constant(c, small(0));
        std::vector<AssignedValue> carry{main_gate.assign_constant(c, small(0))}, cs;
        AssignedValue eq_bit = main_gate.assign_bit(c, small(1));
        for (size_t i = 0; i < nl; ++i) {
            const AssignedValue a_b = main_gate.sub(c, a[i], b[i], s.field());                               // :859 (a FIELD subtraction)
            const AssignedValue sum = main_gate.add_with_constant(c, a_b, carry[i], word_max, s.wide());      // :860-861
            const Fe q1 = s.carry(), r1 = s.limb(), nq1 = s.wide(), amnq1 = s.limb();
            auto dm = div_mod_main_gate(c, sum, limb_max, q1, r1, nq1, amnq1);                                // :864
            carry.push_back(dm.first); cs.push_back(dm.second);
            accumulated_extra = main_gate.add_constant(c, accumulated_extra, word_max, s.wide());            // :869-870
            const Fe q2 = s.carry(), r2 = s.limb(), nq2 = s.wide(), amnq2 = s.limb();
            auto dm2 = div_mod_main_gate(c, accumulated_extra, limb_max, q2, r2, nq2, amnq2);                 // :871
            const AssignedValue cs_acc_eq = main_gate.is_equal(c, cs[i], dm2.second, s.flag());               // :873
            eq_bit = main_gate.and_(c, eq_bit, cs_acc_eq, s.flag());                                          // :874
            accumulated_extra = dm2.first;
            if (i < nl - 1) {
                const Fe dup = s.carry();
                const uint8_t *sub = s.bytes(lo.carry_nsub);
                REQUIRE(dup == carry[i + 1].value);
                const AssignedValue range_assigned = range_chip.assign(c, dup, lo.carry_sub_bits, lo.carry_bits, sub, lo.carry_nsub);   // :879-885
                const AssignedValue range_eq = main_gate.is_equal(c, carry[i + 1], range_assigned, s.flag());
                eq_bit = main_gate.and_(c, eq_bit, range_eq, s.flag());
            } else {
                const AssignedValue final_carry_eq = main_gate.is_equal(c, carry[i + 1], accumulated_extra, s.flag());   // :890
                eq_bit = main_gate.and_(c, eq_bit, final_carry_eq, s.flag());
            }
        }
        return eq_bit;
    }
    // big_integer/chip.rs:542-629
    AssignedInteger mul_mod(MockRegionCtx &c, WitnessStream &s, const AssignedInteger &a, const AssignedInteger &b, const AssignedInteger &n) {
        ++n_mul_mod;
        const size_t n1 = a.size(), n2 = b.size();
        REQUIRE(n1 == n.size());                                                                              // :555
        AssignedInteger quotient_int, prod_int;
        for (size_t k = 0; k < n2; ++k) { const Fe q = s.limb(); quotient_int.push_back(range_chip.assign(c, q, lo.limb_sub_bits, limb_width, s.bytes(lo.limb_nsub), lo.limb_nsub)); }   // :588-591
        for (size_t k = 0; k < n1; ++k) { const Fe r = s.limb(); prod_int.push_back(range_chip.assign(c, r, lo.limb_sub_bits, limb_width, s.bytes(lo.limb_nsub), lo.limb_nsub)); }       // :596-599
        const AssignedInteger ab = mul(c, s, a, b), qn = mul(c, s, quotient_int, n);                          // :608-609
        AssignedInteger eq_a, eq_b;
        for (size_t i = 0; i < n1 + n2 - 1; ++i) {
            eq_a.push_back(ab[i]);
            if (i < n1) eq_b.push_back(main_gate.add(c, qn[i], prod_int[i], s.wide()));                      // :617
            else eq_b.push_back(qn[i]);
        }
        main_gate.assert_one(c, is_equal_muled(c, s, eq_a, eq_b, n1, n2));                                    // assert_equal_muled :1053-1063
        return prod_int;
    }
    AssignedInteger square_mod(MockRegionCtx &c, WitnessStream &s, const AssignedInteger &a, const AssignedInteger &n) { return mul_mod(c, s, a, a, n); }   // :642-649
    // assign_constant(1, num_limbs): the real limb, then ONE shared zero cell (:1252-1281)
    AssignedInteger assign_constant_one(MockRegionCtx &c, size_t nlimbs) {
        AssignedInteger r{main_gate.assign_constant(c, small(1))};
        const AssignedValue zero = main_gate.assign_constant(c, small(0));
        for (size_t k = 1; k < nlimbs; ++k) r.push_back(zero);
        return r;
    }
    // big_integer/chip.rs:710-742
    AssignedInteger pow_mod_fixed_exp(MockRegionCtx &c, WitnessStream &s, const AssignedInteger &a, const std::vector<uint8_t> &e_le, const AssignedInteger &n) {
        size_t num_e_bits = 0;
        for (size_t i = 0; i < 8 * e_le.size(); ++i) if ((e_le[i / 8] >> (i % 8)) & 1) num_e_bits = i + 1;
        AssignedInteger acc = assign_constant_one(c, a.size()), squared = a;
        for (size_t bit = 0; bit < num_e_bits; ++bit) {
            const AssignedInteger cur_sq = squared;
            squared = square_mod(c, s, cur_sq, n);
            if (!((e_le[bit / 8] >> (bit % 8)) & 1)) continue;
            acc = mul_mod(c, s, acc, cur_sq, n);
        }
        return acc;
    }
    // big_integer/chip.rs:664-696; the stream starts with the exponent's bits (h2r_pow_trace_flatten_ex)
    AssignedInteger pow_mod(MockRegionCtx &c, WitnessStream &s, const AssignedInteger &a, const AssignedInteger &e, const AssignedInteger &n, uint32_t exp_limb_bits) {
        std::vector<AssignedValue> e_bits;
        for (const AssignedLimb &e_limb : e) {
            const uint8_t *b = s.bytes(exp_limb_bits);
            for (const AssignedValue &v : main_gate.to_bits(c, e_limb, std::vector<uint8_t>(b, b + exp_limb_bits))) e_bits.push_back(v);   // :677
        }
        AssignedInteger acc = assign_constant_one(c, num_limbs), squared = a;
        for (const AssignedValue &e_bit : e_bits) {
            const AssignedInteger muled = mul_mod(c, s, acc, squared, n);                                     // :686
            for (size_t j = 0; j < num_limbs; ++j) acc[j] = main_gate.select(c, muled[j], acc[j], e_bit, s.limb());   // :688-691
            squared = square_mod(c, s, squared, n);                                                           // :693
        }
        return acc;
    }
};

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd() { uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
#define HIPCK(x) REQUIRE((x) == hipSuccess)

static AssignedInteger assigned(const uint64_t *limbs, size_t n) { AssignedInteger r; for (size_t k = 0; k < n; ++k) r.push_back({small(limbs[k])}); return r; }

int main() {
    const uint32_t w = 64, bits = 2048, L = bits / w, B = 3;
    h2r_params pr{w, bits, H2R_FIELD_BN254_FR, 0};
    h2r_ctx *ctx = nullptr;
    REQUIRE(h2r_ctx_create(&pr, &ctx) == H2R_OK);
    GpuBigIntChip chip{};
    chip.ctx = ctx; chip.limb_width = w; chip.num_limbs = L; chip.main_gate.f.ctx = ctx;
    REQUIRE(h2r_trace_layout(ctx, &chip.lo) == H2R_OK);
    chip.limb_max_v = Fe{{0, 1, 0, 0}};                                        // 2^64
    {   // word_max = L (2^w - 1)^2 + (2^w - 1)  (chip.rs:1368-1372)
        const Fe m = Fe{{~0ull, 0, 0, 0}};
        chip.word_max = chip.main_gate.f.add(chip.main_gate.f.mul(small(L), chip.main_gate.f.mul(m, m)), m);
    }
    std::vector<uint64_t> hx(B * L), hn(B * L);
    for (uint32_t e = 0; e < B; ++e) {
        for (uint32_t k = 0; k < L; ++k) { hn[e * L + k] = rnd(); hx[e * L + k] = rnd(); }
        hn[e * L] |= 1; hn[e * L + L - 1] |= 1ull << 63; hx[e * L + L - 1] &= ~(1ull << 63);   // x < n
    }
    void *dx, *dn, *dout, *dtrace, *dws, *dstream, *dst;
    HIPCK(hipMalloc(&dx, hx.size() * 8)); HIPCK(hipMalloc(&dn, hn.size() * 8)); HIPCK(hipMalloc(&dout, hx.size() * 8)); HIPCK(hipMalloc(&dst, B));
    HIPCK(hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice)); HIPCK(hipMemcpy(dn, hn.data(), hn.size() * 8, hipMemcpyHostToDevice));

    // ---------------- RSAPubE::Fix(65537): pow_mod_fixed_exp ----------------
    {
        const uint8_t e_le[3] = {1, 0, 1};
        h2r_pow_layout pl;
        REQUIRE(h2r_pow_fixed_layout(ctx, e_le, 3, &pl) == H2R_OK);
        HIPCK(hipMalloc(&dtrace, B * pl.elem_stride)); HIPCK(hipMalloc(&dws, h2r_workspace_bytes(ctx, B, pl.num_mul_mods)));
        REQUIRE(h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, B, 0, dtrace, dout, static_cast<uint8_t *>(dst), dws, nullptr) == H2R_OK);
        const uint64_t sb = h2r_pow_stream_bytes(ctx, &pl, H2R_STREAM_FIELD_AB);
        HIPCK(hipMalloc(&dstream, B * sb));
        REQUIRE(h2r_pow_trace_emit_stream(ctx, &pl, dtrace, 0, B, H2R_STREAM_FIELD_AB, dstream, sb, 0, nullptr) == H2R_OK);
        HIPCK(hipDeviceSynchronize());
        std::vector<uint8_t> hs(B * sb), st(B); std::vector<uint64_t> hout(B * L);
        HIPCK(hipMemcpy(hs.data(), dstream, hs.size(), hipMemcpyDeviceToHost)); HIPCK(hipMemcpy(st.data(), dst, B, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(hout.data(), dout, hout.size() * 8, hipMemcpyDeviceToHost));
        for (uint32_t e = 0; e < B; ++e) {
            REQUIRE(st[e] == H2R_OK);
            MockRegionCtx rc{};
            chip.n_mul_mod = chip.n_div_mod = chip.n_mul = 0;
            WitnessStream s{hs.data() + e * sb, hs.data() + (e + 1) * sb, &chip.lo};
            const AssignedInteger acc = chip.pow_mod_fixed_exp(rc, s, assigned(&hx[e * L], L), std::vector<uint8_t>(e_le, e_le + 3), assigned(&hn[e * L], L));
            for (uint32_t k = 0; k < L; ++k) { REQUIRE(s.limb() == acc[k].value); REQUIRE(acc[k].value == small(hout[e * L + k])); }   // the stream ends with the result limbs
            REQUIRE(s.p == s.end);                                                     // consumed to the last byte
            // the reference's call counts
            REQUIRE(chip.n_mul_mod == 19 && chip.n_mul == 38 && rc.mul_add == 38912 && chip.n_div_mod == 2394 && rc.range_assign == 2394);
            REQUIRE(rc.range_sublimbs == 20330);
            REQUIRE(rc.sub == 19 * 63 + 2394 && rc.add_with_constant == 19 * 63 && rc.add_constant == 19 * 63 && rc.add == 19 * 32);
            REQUIRE(rc.is_equal == 19 * 126 && rc.and_ == 19 * 126 && rc.assert_one == 19 && rc.assign_bit == 19);
            REQUIRE(rc.assign_constant == 2 + 19 * (2 * 63 + 3) && rc.assign_value == 2 * 2394 && rc.mul == 2394 && rc.assert_equal == 2394);
        }
        std::printf("Fix: %u elements x %llu stream bytes replayed, every gate relation holds, counts = the reference's\n", B, (unsigned long long)sb);
        HIPCK(hipFree(dtrace)); HIPCK(hipFree(dws)); HIPCK(hipFree(dstream));
    }
    // ---------------- RSAPubE::Var: pow_mod with a 5-bit exponent limb (src/chip.rs:283, 327) ----------------
    {
        const uint32_t exp_limb_bits = 5;
        std::vector<uint64_t> he = {19, 31, 1};
        void *de; HIPCK(hipMalloc(&de, B * 8)); HIPCK(hipMemcpy(de, he.data(), B * 8, hipMemcpyHostToDevice));
        h2r_pow_layout pl;
        REQUIRE(h2r_pow_var_layout(ctx, 1, exp_limb_bits, &pl) == H2R_OK);
        HIPCK(hipMalloc(&dtrace, B * pl.elem_stride)); HIPCK(hipMalloc(&dws, h2r_workspace_bytes(ctx, B, pl.num_mul_mods)));
        REQUIRE(h2r_pow_mod_batch(ctx, dx, de, 1, exp_limb_bits, dn, B, 0, dtrace, dout, static_cast<uint8_t *>(dst), dws, nullptr) == H2R_OK);
        const uint64_t sb = h2r_pow_stream_bytes(ctx, &pl, H2R_STREAM_FIELD_AB);
        HIPCK(hipMalloc(&dstream, B * sb));
        REQUIRE(h2r_pow_trace_emit_stream(ctx, &pl, dtrace, 0, B, H2R_STREAM_FIELD_AB, dstream, sb, 0, nullptr) == H2R_OK);
        HIPCK(hipDeviceSynchronize());
        std::vector<uint8_t> hs(B * sb); std::vector<uint64_t> hout(B * L);
        HIPCK(hipMemcpy(hs.data(), dstream, hs.size(), hipMemcpyDeviceToHost)); HIPCK(hipMemcpy(hout.data(), dout, hout.size() * 8, hipMemcpyDeviceToHost));
        for (uint32_t e = 0; e < B; ++e) {
            MockRegionCtx rc{};
            chip.n_mul_mod = chip.n_div_mod = chip.n_mul = 0;
            WitnessStream s{hs.data() + e * sb, hs.data() + (e + 1) * sb, &chip.lo};
            const AssignedInteger acc = chip.pow_mod(rc, s, assigned(&hx[e * L], L), assigned(&he[e], 1), assigned(&hn[e * L], L), exp_limb_bits);
            for (uint32_t k = 0; k < L; ++k) { REQUIRE(s.limb() == acc[k].value); REQUIRE(acc[k].value == small(hout[e * L + k])); }
            REQUIRE(s.p == s.end);
            REQUIRE(chip.n_mul_mod == 2 * exp_limb_bits && rc.select == exp_limb_bits * L && rc.to_bits == 1 && rc.mul_add == 2ull * exp_limb_bits * 2048);
        }
        std::printf("Var: %u elements x %llu stream bytes replayed\n", B, (unsigned long long)sb);
        HIPCK(hipFree(dtrace)); HIPCK(hipFree(dws)); HIPCK(hipFree(dstream)); HIPCK(hipFree(de));
    }
    h2r_ctx_destroy(ctx);
    std::printf("CHIP_REPLAY_OK\n");
    return 0;
}
