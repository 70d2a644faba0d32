// Prime-field arithmetic of the four fields the reference instantiates its chips over (bn256 Fr / Fq, pasta Fp / Fq:
// examples/rsa_example.rs:148, big_integer/chip.rs:1461-1463), for the parts of the witness that are FIELD values rather
// than integers: the theta-compressed lookup inputs of halo2's lookup argument (tag * theta + value) and the inverse
// witness of main_gate.is_zero.  Elements are four little-endian 64-bit words, canonical (< p) at every interface; the
// Montgomery form (R = 2^256) is internal to mul / inverse.  Host and device share the code (hipcc compiles both).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace h2r {

struct Fe { uint64_t v[4]; };

struct FieldConsts {
    uint64_t p[4];      // modulus
    uint64_t r2[4];     // R^2 mod p
    uint64_t one[4];    // R mod p (Montgomery form of 1)
    uint64_t n0inv;     // -p^{-1} mod 2^64
};

#define H2R_FD __host__ __device__ __forceinline__

H2R_FD bool fe_is_zero(const Fe &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
H2R_FD bool fe_eq(const Fe &a, const Fe &b) { return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2] && a.v[3] == b.v[3]; }
// order of the canonical integers = the fields' `Ord` (halo2curves / pasta_curves compare to_repr() from the top byte)
H2R_FD bool fe_lt(const Fe &a, const Fe &b) {
#pragma unroll
    for (int k = 3; k >= 0; --k) { if (a.v[k] != b.v[k]) return a.v[k] < b.v[k]; }
    return false;
}
H2R_FD bool ge_p(const uint64_t (&x)[4], const uint64_t (&p)[4]) {
#pragma unroll
    for (int k = 3; k >= 0; --k) { if (x[k] != p[k]) return x[k] > p[k]; }
    return true;
}
H2R_FD Fe fe_zero() { Fe r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0; return r; }
H2R_FD Fe fe_small(uint64_t x) { Fe r; r.v[0] = x; r.v[1] = r.v[2] = r.v[3] = 0; return r; }

// a + b mod p (a, b < p < 2^255: no carry out of 256 bits)
H2R_FD Fe fe_add(const Fe &a, const Fe &b, const uint64_t (&p)[4]) {
    uint64_t s[4]; uint64_t cy = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint64_t t = a.v[k] + b.v[k]; const uint64_t c1 = t < a.v[k]; const uint64_t u = t + cy; cy = c1 | (uint64_t)(u < t); s[k] = u; }
    Fe r;
    if (ge_p(s, p)) {
        uint64_t br = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t t = s[k] - p[k]; const uint64_t b1 = s[k] < p[k]; const uint64_t u = t - br; br = b1 | (uint64_t)(t < br); r.v[k] = u; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = s[k];
    }
    return r;
}
// a - b mod p
H2R_FD Fe fe_sub(const Fe &a, const Fe &b, const uint64_t (&p)[4]) {
    Fe r; uint64_t br = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint64_t t = a.v[k] - b.v[k]; const uint64_t b1 = a.v[k] < b.v[k]; const uint64_t u = t - br; br = b1 | (uint64_t)(t < br); r.v[k] = u; }
    if (br) {
        uint64_t cy = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t t = r.v[k] + p[k]; const uint64_t c1 = t < r.v[k]; const uint64_t u = t + cy; cy = c1 | (uint64_t)(u < t); r.v[k] = u; }
    }
    return r;
}
// small * a mod p by double-and-add (tags are small integers; 32 conditional additions at most)
H2R_FD Fe fe_mul_small(const Fe &a, uint32_t m, const uint64_t (&p)[4]) {
    Fe acc = fe_zero(), cur = a;
    while (m) {
        if (m & 1u) acc = fe_add(acc, cur, p);
        cur = fe_add(cur, cur, p);
        m >>= 1;
    }
    return acc;
}

H2R_FD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// t += a * b + c; returns the high word (the carry)
H2R_FD uint64_t mac(uint64_t &t, uint64_t a, uint64_t b, uint64_t c) {
    const uint64_t lo = a * b, hi = mulhi64(a, b);
    const uint64_t s1 = t + lo; const uint64_t c1 = s1 < t;
    const uint64_t s2 = s1 + c; const uint64_t c2 = s2 < s1;
    t = s2;
    return hi + c1 + c2;   // cannot overflow: (2^64-1)^2 + 2 (2^64-1) < 2^128
}
// Montgomery product a * b * R^-1 mod p (CIOS, 4 x 64-bit words); inputs < p, output < p
H2R_FD Fe fe_mont_mul(const Fe &a, const Fe &b, const FieldConsts &f) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) c = mac(t[j], a.v[j], b.v[i], c);
        const uint64_t s = t[4] + c; t[5] = s < t[4]; t[4] = s;
        const uint64_t m = t[0] * f.n0inv;
        uint64_t z = t[0];
        c = mac(z, m, f.p[0], 0);
#pragma unroll
        for (int j = 1; j < 4; ++j) { c = mac(t[j], m, f.p[j], c); t[j - 1] = t[j]; }
        const uint64_t s2 = t[4] + c; const uint64_t c2 = s2 < t[4];
        t[3] = s2; t[4] = t[5] + c2; t[5] = 0;
    }
    uint64_t x[4] = {t[0], t[1], t[2], t[3]};
    Fe r;
    if (t[4] || ge_p(x, f.p)) {
        uint64_t br = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t u = x[k] - f.p[k]; const uint64_t b1 = x[k] < f.p[k]; const uint64_t w = u - br; br = b1 | (uint64_t)(u < br); r.v[k] = w; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = x[k];
    }
    return r;
}
H2R_FD Fe fe_to_mont(const Fe &a, const FieldConsts &f) { Fe r2; for (int k = 0; k < 4; ++k) r2.v[k] = f.r2[k]; return fe_mont_mul(a, r2, f); }
H2R_FD Fe fe_from_mont(const Fe &a, const FieldConsts &f) { return fe_mont_mul(a, fe_small(1), f); }
// canonical a * b mod p
H2R_FD Fe fe_mul(const Fe &a, const Fe &b, const FieldConsts &f) { return fe_mont_mul(fe_to_mont(a, f), b, f); }
// a^(p-2) by square-and-multiply: ~380 Montgomery products (kept as the cross-check of fe_inv: h2r_field_eval op 4)
H2R_FD Fe fe_inv_fermat(const Fe &a, const FieldConsts &f) {
    uint64_t e[4] = {f.p[0] - 2, f.p[1], f.p[2], f.p[3]};   // p is odd and > 2: no borrow
    const Fe am = fe_to_mont(a, f);
    Fe acc; for (int k = 0; k < 4; ++k) acc.v[k] = f.one[k];
    for (int bit = 255; bit >= 0; --bit) {
        acc = fe_mont_mul(acc, acc, f);
        if ((e[bit >> 6] >> (bit & 63)) & 1ull) acc = fe_mont_mul(acc, am, f);
    }
    return fe_from_mont(acc, f);
}
// a^-1 mod p for a in [1, p), p an odd prime; canonical in and out.  main_gate.is_zero's inverse witness.
// Binary extended Euclid (HAC 14.61) with the invariants x1 * a == u and x2 * a == v (mod p): every round makes the larger of
// two odd values even by a subtraction, then halves the even one -- ~510 rounds of a few 256-bit add / sub / shifts, no
// multiplier at all.  On the GPU a wave of these takes ~0.1 ms where the 380 Montgomery products of a^(p-2) (quarter-rate
// 32-bit multiplies) took 0.9 ms.  The round cap only matters for an input outside [1, p) (no inverse: garbage out, no hang).
H2R_FD Fe fe_inv(const Fe &a, const FieldConsts &f) {
    Fe u = a, v, x1 = fe_small(1), x2 = fe_zero();
    for (int k = 0; k < 4; ++k) v.v[k] = f.p[k];
    auto is_one = [](const Fe &x) { return x.v[0] == 1 && (x.v[1] | x.v[2] | x.v[3]) == 0; };
    auto shr1 = [](Fe &x, uint64_t top) {
        x.v[0] = (x.v[0] >> 1) | (x.v[1] << 63); x.v[1] = (x.v[1] >> 1) | (x.v[2] << 63);
        x.v[2] = (x.v[2] >> 1) | (x.v[3] << 63); x.v[3] = (x.v[3] >> 1) | (top << 63);
    };
    auto half_mod = [&](Fe &x) {   // x / 2 mod p: (x or x + p) >> 1
        uint64_t cy = 0;
        if (x.v[0] & 1ull) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint64_t t = x.v[k] + f.p[k]; const uint64_t c1 = t < x.v[k]; const uint64_t w = t + cy; cy = c1 | (uint64_t)(w < t); x.v[k] = w; }
        }
        shr1(x, cy);
    };
    auto sub256 = [](Fe &x, const Fe &y) {   // x -= y, x >= y
        uint64_t br = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t t = x.v[k] - y.v[k]; const uint64_t b1 = x.v[k] < y.v[k]; const uint64_t w = t - br; br = b1 | (uint64_t)(t < br); x.v[k] = w; }
    };
    for (int round = 0; round < 1024 && !is_one(u) && !is_one(v); ++round) {
        if ((u.v[0] & 1ull) && (v.v[0] & 1ull)) {
            if (fe_lt(u, v)) { sub256(v, u); x2 = fe_sub(x2, x1, f.p); }
            else { sub256(u, v); x1 = fe_sub(x1, x2, f.p); }
        }
        if (!(u.v[0] & 1ull)) { shr1(u, 0); half_mod(x1); }
        else { shr1(v, 0); half_mod(x2); }
    }
    return is_one(u) ? x1 : x2;
}

// s^-1 mod p for a one-word s, 2 <= s < 2^64 < p: classical Euclid on (p, s).  After the first step (p = q0 * s + r: a restoring
// division, one bit per round) every remainder fits one word, so a step is one 64-bit division and one 64 x 256-bit
// multiply-accumulate on the cofactor's magnitude (|t[i+1]| = |t[i-1]| + q[i] * |t[i]|, signs alternate, |t| < p throughout):
// <= 93 steps, ~35 on average -- an order of magnitude less work than the 510 rounds of 256-bit shifts of fe_inv.  The difference
// main_gate.is_zero is asked about in this path is always of that kind: +-(limb - limb), +-(limb - constant), a flag.
H2R_FD Fe fe_inv_word(uint64_t s, const FieldConsts &f) {
    uint64_t tc[4] = {0, 0, 0, 0}, tp[4] = {1, 0, 0, 0};   // |t_cur| (starts as q0), |t_prev|
    uint64_t rc = 0, rp = s;
    for (int bit = 255; bit >= 0; --bit) {
        const uint64_t top = rc >> 63;
        rc = (rc << 1) | ((f.p[bit >> 6] >> (bit & 63)) & 1ull);
        if (top || rc >= s) { rc -= s; tc[bit >> 6] |= 1ull << (bit & 63); }
    }
    bool neg = true;                                        // t_cur = -q0
    while (rc > 1) {
        const uint64_t q = rp / rc, rn = rp - q * rc;
        uint64_t c = 0, tn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { tn[k] = tp[k]; c = mac(tn[k], q, tc[k], c); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { tp[k] = tc[k]; tc[k] = tn[k]; }
        rp = rc; rc = rn; neg = !neg;
    }
    Fe t; for (int k = 0; k < 4; ++k) t.v[k] = tc[k];
    return neg ? fe_sub(fe_zero(), t, f.p) : t;
}
// a^-1 mod p, a in [1, p): the one-word Euclid when a or p - a fits a word, the binary algorithm otherwise
H2R_FD Fe fe_inv_fast(const Fe &a, const FieldConsts &f) {
    if ((a.v[1] | a.v[2] | a.v[3]) == 0) return a.v[0] == 1 ? a : fe_inv_word(a.v[0], f);
    const Fe na = fe_sub(fe_zero(), a, f.p);
    if ((na.v[1] | na.v[2] | na.v[3]) == 0) return na.v[0] == 1 ? a : fe_sub(fe_zero(), fe_inv_word(na.v[0], f), f.p);
    return fe_inv(a, f);
}

// ---- the prover's in-memory form: x * R mod p, R = 2^256 (halo2curves bn256 / pasta field elements are four 64-bit words in
// Montgomery form) -------------------------------------------------------------------------------------------------------
// Almost every cell of the witness is a SHORT integer (a sub-limb, a limb, a 70-bit carry, a 133-bit accumulator), and for a
// K-dword x the product x * R is one Montgomery multiplication by 2^(32 K) * R mod p with only K reduction steps:
//     x * (2^(32 K) R) * 2^(-32 K) = x R (mod p),   result < 2 p before the final conditional subtraction,
// i.e. 17 K 32-bit multiply-adds instead of the 136 of the generic R^2 multiplication.  MontK holds those multipliers.
struct MontK {
    uint32_t p[8];         // modulus, 32-bit digits
    uint32_t n0inv;        // -p^-1 mod 2^32
    uint32_t pad_[3];
    uint32_t bk[9][8];     // bk[K] = 2^(32 K) * R mod p (bk[0] = R mod p: the Montgomery form of 1; bk[8] = R^2 mod p)
    // the same in radix 2^30 (mont30 below: the form the GPU's 32 x 32 + 64 multiply-add runs without a single carry instruction)
    uint32_t p30[9];       // modulus, 30-bit digits
    uint32_t n0inv30;      // -p^-1 mod 2^30
    uint32_t bk30[10][9];  // bk30[D] = 2^(30 D) * R mod p, 30-bit digits
};

template <int K>
H2R_FD void mont_short(const uint32_t (&x)[K], const uint32_t *b, const uint32_t *p, uint32_t n0inv, uint32_t (&t)[8]) {
    static_assert(K >= 1 && K <= 8, "digits of the short operand");
    uint32_t t8 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint64_t acc = (uint64_t)x[i] * b[j] + t[j] + c; t[j] = (uint32_t)acc; c = acc >> 32; }
        t8 = (uint32_t)c;                                  // (t < 2 p < 2^255 after every step: nothing is pending in t8)
        const uint32_t m = t[0] * n0inv;
        c = ((uint64_t)m * p[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) { const uint64_t acc = (uint64_t)m * p[j] + t[j] + c; t[j - 1] = (uint32_t)acc; c = acc >> 32; }
        t[7] = t8 + (uint32_t)c;                           // < 2^31
    }
    // t in [0, 2 p): one conditional subtraction
    uint32_t d[8]; uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint64_t s = (uint64_t)t[j] - p[j] - br; d[j] = (uint32_t)s; br = (s >> 32) & 1u; }
    if (!br) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = d[j];
    }
}
// The same product in radix 2^30.  v_mad_u64_u32 adds a 32 x 32-bit product to a 64-bit accumulator; with 30-bit digits a product is
// below 2^60, so a column accumulator takes the two products of each of up to seven steps without overflowing and NO carry is
// handled inside the loop: one instruction per digit product, all products of a step independent of each other (a wave that is
// alone on its SIMD has nothing else to hide a carry chain's latency behind).  x: D digits of 30 bits (the value is below 2^(30 D));
// b = 2^(30 D) R mod p and p as nine 30-bit digits; result x R mod p as eight 32-bit words.
template <int D>
H2R_FD void mont30(const uint32_t (&x)[D], const uint32_t *b, const uint32_t *p30, uint32_t n0inv30, const uint32_t *p32, uint32_t (&out)[8]) {
    static_assert(D >= 1 && D <= 9, "30-bit digits of the short operand");
    constexpr uint32_t M = (1u << 30) - 1;
    uint64_t T[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) T[j] = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) T[j] += (uint64_t)x[i] * b[j];
        const uint32_t m = ((uint32_t)T[0] * n0inv30) & M;
#pragma unroll
        for (int j = 0; j < 9; ++j) T[j] += (uint64_t)m * p30[j];
        const uint64_t cy = T[0] >> 30;                       // (T[0] is a multiple of 2^30 now)
#pragma unroll
        for (int j = 0; j < 8; ++j) T[j] = T[j + 1];
        T[0] += cy; T[8] = 0;
        if (D >= 8 && i == 3) {                                // eight and nine steps: the accumulators are brought back below 2^31 once
#pragma unroll
            for (int j = 0; j < 8; ++j) { T[j + 1] += T[j] >> 30; T[j] &= M; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { T[j + 1] += T[j] >> 30; T[j] &= M; }
    // nine 30-bit digits (the value is below 2 p < 2^255) -> eight words
    uint32_t t[8];
    t[0] = (uint32_t)T[0] | ((uint32_t)T[1] << 30);
#pragma unroll
    for (int k = 1; k < 8; ++k) t[k] = ((uint32_t)T[k] >> (2 * k)) | ((uint32_t)T[k + 1] << (30 - 2 * k));
    uint32_t d[8]; uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint64_t s = (uint64_t)t[j] - p32[j] - br; d[j] = (uint32_t)s; br = (s >> 32) & 1u; }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = br ? t[j] : d[j];
}
// the 30-bit digits of a value given as W 32-bit words (D <= ceil(32 W / 30))
template <int D, int W>
H2R_FD void digits30(const uint32_t (&w)[W], uint32_t (&d)[D]) {
    constexpr uint32_t M = (1u << 30) - 1;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int bit = 30 * i, k = bit / 32, sh = bit % 32;
        uint32_t v = k < W ? w[k] >> sh : 0u;
        if (sh > 2 && k + 1 < W) v |= w[k + 1] << (32 - sh);
        d[i] = v & M;
    }
}
// x R mod p of a value below 2^BITS given as W 32-bit words
template <int BITS, int W>
H2R_FD void mont_bits(const uint32_t (&w)[W], const MontK &mk, uint32_t (&out)[8]) {
    constexpr int D = (BITS + 29) / 30;
    static_assert(D <= 9 && 32 * W + 29 >= 30 * D, "digits of the value");
    uint32_t d[D];
    digits30<D, W>(w, d);
    mont30<D>(d, mk.bk30[D], mk.p30, mk.n0inv30, mk.p, out);
}
// p - t for t in [0, p) (0 stays 0): the Montgomery form of -x from that of x
H2R_FD void mont_neg(uint32_t (&t)[8], const uint32_t *p) {
    uint32_t nz = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) nz |= t[j];
    if (!nz) return;
    uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint64_t s = (uint64_t)p[j] - t[j] - br; t[j] = (uint32_t)s; br = (s >> 32) & 1u; }
}
// canonical element (< p) -> Montgomery form, whatever its size: the generic K = 8 case
H2R_FD Fe fe_to_mont_k(const Fe &a, const MontK &m) {
    uint32_t x[8], t[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[2 * k] = (uint32_t)a.v[k]; x[2 * k + 1] = (uint32_t)(a.v[k] >> 32); }
    mont_short<8>(x, m.bk[8], m.p, m.n0inv, t);
    Fe r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = ((uint64_t)t[2 * k + 1] << 32) | t[2 * k];
    return r;
}

// Host: derive the Montgomery constants of modulus p.
inline void field_consts_init(const uint64_t p[4], FieldConsts *f);
inline void montk_init(const uint64_t p[4], MontK *m) {
    FieldConsts f;
    field_consts_init(p, &f);
    for (int k = 0; k < 4; ++k) { m->p[2 * k] = (uint32_t)p[k]; m->p[2 * k + 1] = (uint32_t)(p[k] >> 32); }
    m->n0inv = (uint32_t)f.n0inv;
    m->pad_[0] = m->pad_[1] = m->pad_[2] = 0;
    Fe x; for (int k = 0; k < 4; ++k) x.v[k] = f.one[k];   // R mod p, then doubled 32 times per entry
    for (int K = 0; K <= 8; ++K) {
        for (int k = 0; k < 4; ++k) { m->bk[K][2 * k] = (uint32_t)x.v[k]; m->bk[K][2 * k + 1] = (uint32_t)(x.v[k] >> 32); }
        for (int i = 0; i < 32; ++i) x = fe_add(x, x, f.p);
    }
    auto to30 = [](const uint64_t (&v)[4], uint32_t *d) {
        uint32_t w[8];
        for (int k = 0; k < 4; ++k) { w[2 * k] = (uint32_t)v[k]; w[2 * k + 1] = (uint32_t)(v[k] >> 32); }
        uint32_t t[9];
        digits30<9, 8>(w, t);
        for (int k = 0; k < 9; ++k) d[k] = t[k];
    };
    to30(f.p, m->p30);
    m->n0inv30 = m->n0inv & ((1u << 30) - 1);
    for (int k = 0; k < 4; ++k) x.v[k] = f.one[k];
    for (int D = 0; D <= 9; ++D) {
        to30(x.v, m->bk30[D]);
        for (int i = 0; i < 30; ++i) x = fe_add(x, x, f.p);
    }
}
inline void field_consts_init(const uint64_t p[4], FieldConsts *f) {
    for (int k = 0; k < 4; ++k) f->p[k] = p[k];
    uint64_t inv = 1;                                   // Newton: inv = p^-1 mod 2^64
    for (int i = 0; i < 6; ++i) inv *= 2 - p[0] * inv;
    f->n0inv = (uint64_t)0 - inv;
    Fe x = fe_small(1);                                 // 2^k mod p by doubling
    for (int i = 0; i < 512; ++i) {
        x = fe_add(x, x, f->p);
        if (i == 255) for (int k = 0; k < 4; ++k) f->one[k] = x.v[k];
    }
    for (int k = 0; k < 4; ++k) f->r2[k] = x.v[k];
}

}  // namespace h2r
