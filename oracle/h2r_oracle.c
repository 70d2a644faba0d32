/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the halo2-rsa hot path in plain C.
 *
 * Restates, value for value and in the reference's own call order, what
 *   RSAChip::modpow_public_key      (reference src/chip.rs:99-114)
 *   BigIntChip::pow_mod_fixed_exp   (src/big_integer/chip.rs:710-742)
 *   BigIntChip::pow_mod             (src/big_integer/chip.rs:664-696)
 *   BigIntChip::mul_mod/square_mod  (src/big_integer/chip.rs:542-629, 642-649)
 *   BigIntChip::mul                 (src/big_integer/chip.rs:386-419)
 *   BigIntChip::is_equal_muled      (src/big_integer/chip.rs:822-895)
 *   BigIntChip::div_mod_main_gate   (src/big_integer/chip.rs:1323-1349)
 * assign, as the flat op-trace stream described in oracle/pyref.py.  The big-integer arithmetic the
 * reference delegates to num-bigint 0.4 (Cargo.toml:16; `*`, `/`, `%`, `>>` at chip.rs:562-584) is
 * restated here as schoolbook multiplication and Knuth algorithm D on 32-bit digits; the results are
 * mathematically unique (floor quotient / remainder), so any correct algorithm pins the same values.
 *
 * Pinning: tests/test_oracle_golden.py checks this file against the tests/golden fixtures -- the
 * reference's own known-answer vectors (src/chip.rs:703-713, 748-758, 798;
 * src/big_integer/chip.rs:2844-3098, 3123-3246; src/big_integer/mod.rs:509) -- and against the
 * independent Python big-int restatement oracle/pyref.py on random inputs.
 *
 * The reference itself (Rust, un-vendored git dependencies, no cargo in this image) cannot be
 * built here, so there is no oracle/_ref; see DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#include "h2r_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define NUM_LOOKUP_LIMBS 8 /* big_integer/chip.rs:1163 */
#define MAXL 256           /* max limbs supported by the oracle */
#define MAXD (2 * MAXL)    /* 32-bit digits of one operand at w = 64 */

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } u256;

/* ------------------------------------------------------------------------------------------------
 * 256-bit helpers (two's complement when signed)
 * ---------------------------------------------------------------------------------------------- */
static u256 u256_zero(void) { u256 r = {{0, 0, 0, 0}}; return r; }
static u256 u256_from64(uint64_t x) { u256 r = {{x, 0, 0, 0}}; return r; }
static u256 u256_add(u256 a, u256 b) {
    u256 r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    return r;
}
static u256 u256_sub(u256 a, u256 b) {
    u256 r; uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t t = a.v[i] - b.v[i]; uint64_t b1 = a.v[i] < b.v[i];
        uint64_t t2 = t - br; uint64_t b2 = t < br;
        r.v[i] = t2; br = b1 | b2;
    }
    return r;
}
static u256 u256_mul64(uint64_t a, uint64_t b) {
    u128 p = (u128)a * b; u256 r = {{(uint64_t)p, (uint64_t)(p >> 64), 0, 0}}; return r;
}
static u256 u256_shr(u256 a, unsigned s) { /* 0 < s <= 64 */
    u256 r;
    if (s == 64) { r.v[0] = a.v[1]; r.v[1] = a.v[2]; r.v[2] = a.v[3]; r.v[3] = 0; return r; }
    for (int i = 0; i < 4; ++i) r.v[i] = (a.v[i] >> s) | (i < 3 ? a.v[i + 1] << (64 - s) : 0);
    return r;
}
static u256 u256_shl(u256 a, unsigned s) { /* 0 < s <= 64 */
    u256 r;
    if (s == 64) { r.v[3] = a.v[2]; r.v[2] = a.v[1]; r.v[1] = a.v[0]; r.v[0] = 0; return r; }
    for (int i = 3; i >= 0; --i) r.v[i] = (a.v[i] << s) | (i > 0 ? a.v[i - 1] >> (64 - s) : 0);
    return r;
}
static uint64_t u256_low(u256 a, unsigned w) { return w == 64 ? a.v[0] : (a.v[0] & ((1ull << w) - 1)); }
static int u256_eq(u256 a, u256 b) { return memcmp(&a, &b, sizeof a) == 0; }
static unsigned u256_bits(u256 a) {
    for (int i = 3; i >= 0; --i) if (a.v[i]) return 64u * i + (64u - (unsigned)__builtin_clzll(a.v[i]));
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * flat-stream writer
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t *p; } wr;
static void put256(wr *s, u256 x, unsigned nbytes) {
    if (!s->p) return;
    memcpy(s->p, &x, nbytes); /* host is little-endian (x86-64) */
    s->p += nbytes;
}
static void put64(wr *s, uint64_t x, unsigned nbytes) {
    if (!s->p) return;
    uint64_t t[2] = {x, 0};
    memcpy(s->p, t, nbytes);
    s->p += nbytes;
}

static unsigned sublimb_bit_len(unsigned bit_len_limb) { /* big_integer/chip.rs:1357-1365 */
    unsigned val = bit_len_limb / NUM_LOOKUP_LIMBS;
    return val == 0 ? 1 : val;
}
static unsigned n_sublimbs(unsigned bit_len) {
    unsigned s = sublimb_bit_len(bit_len);
    return bit_len / s + (bit_len % s ? 1 : 0);
}
/* One RangeChip::assign(value, sub_bits, bit_len) [3P maingate]: the value, then its sub-limbs. */
static void emit_range_assign(wr *s, u256 v, unsigned sub_bits, unsigned bit_len, unsigned nbytes) {
    put256(s, v, nbytes);
    unsigned n = bit_len / sub_bits + (bit_len % sub_bits ? 1 : 0);
    u256 t = v;
    for (unsigned k = 0; k < n; ++k) {
        put64(s, t.v[0] & ((1ull << sub_bits) - 1), 1);
        t = u256_shr(t, sub_bits);
    }
}

/* ------------------------------------------------------------------------------------------------
 * parameters
 * ---------------------------------------------------------------------------------------------- */
static u256 compute_mul_word_max(unsigned w, unsigned min_n) { /* big_integer/chip.rs:1368-1372 */
    /* min_n*(B-1)^2 + (B-1) */
    uint64_t bm1 = w == 64 ? ~0ull : ((1ull << w) - 1);
    u256 sq = u256_mul64(bm1, bm1);
    u256 acc = u256_zero();
    for (unsigned i = 0; i < min_n; ++i) acc = u256_add(acc, sq);
    return u256_add(acc, u256_from64(bm1));
}

void h2ro_compute_range_lens(uint32_t w, uint32_t L, uint32_t comp[3], uint32_t over[3]) {
    /* big_integer/chip.rs:1220-1249 */
    unsigned out_comp = w / NUM_LOOKUP_LIMBS;
    unsigned out_over = w % out_comp;
    unsigned fresh_word_max_width = w + 2; /* bits(2 * 2^w) */
    unsigned fresh_carry_bits = fresh_word_max_width - w;
    unsigned fresh_comp = sublimb_bit_len(fresh_carry_bits);
    unsigned fresh_over = fresh_carry_bits % fresh_comp;
    u256 wm = compute_mul_word_max(w, L);
    unsigned mul_width = u256_bits(u256_add(wm, wm));
    unsigned mul_carry_bits = mul_width - w;
    unsigned mul_comp = sublimb_bit_len(mul_carry_bits);
    unsigned mul_over = mul_carry_bits % mul_comp;
    comp[0] = out_comp; comp[1] = fresh_comp; comp[2] = mul_comp;
    over[0] = out_over; over[1] = fresh_over; over[2] = mul_over;
}

int h2ro_params_init(h2ro_params *p, uint32_t w, uint32_t L) {
    if (!p || (w != 32 && w != 64) || L == 0 || L > MAXL) return H2RO_E_SHAPE;
    memset(p, 0, sizeof *p);
    p->w = w; p->L = L;
    u256 wm = compute_mul_word_max(w, L);
    memcpy(p->word_max, &wm, sizeof wm);
    p->word_max_bits = u256_bits(wm);
    p->carry_bits = u256_bits(u256_add(wm, wm)) - w; /* big_integer/chip.rs:841-842 */
    p->LB = w / 8;
    p->WB = 8 * ((p->word_max_bits + 2 + 63) / 64);
    p->CB = 8 * ((p->carry_bits + 63) / 64);
    p->limb_sub_bits = sublimb_bit_len(w);
    p->limb_nsub = n_sublimbs(w);
    p->carry_sub_bits = sublimb_bit_len(p->carry_bits);
    p->carry_nsub = n_sublimbs(p->carry_bits);
    uint64_t C = 2ull * L - 1;
    uint64_t per_col = 5ull * p->WB + 2ull * p->CB + 4ull * p->LB + 4;
    p->mul_mod_stream_bytes = 2ull * L * (p->LB + p->limb_nsub) + 2ull * L * L * p->WB + (uint64_t)L * p->WB
                              + C * per_col + (C - 1) * (p->CB + p->carry_nsub);
    return H2RO_OK;
}

static uint64_t get_limb(const void *a, unsigned i, unsigned w) {
    return w == 64 ? ((const uint64_t *)a)[i] : ((const uint32_t *)a)[i];
}
static void set_limb(void *a, unsigned i, unsigned w, uint64_t x) {
    if (w == 64) ((uint64_t *)a)[i] = x; else ((uint32_t *)a)[i] = (uint32_t)x;
}

/* ------------------------------------------------------------------------------------------------
 * big integers on 32-bit digits (restates num-bigint's `*`, `/`, `%`)
 * ---------------------------------------------------------------------------------------------- */
static void bn_mul(uint32_t *out, const uint32_t *a, int na, const uint32_t *b, int nb) {
    memset(out, 0, sizeof(uint32_t) * (size_t)(na + nb));
    for (int i = 0; i < na; ++i) {
        uint64_t carry = 0, ai = a[i];
        for (int j = 0; j < nb; ++j) {
            uint64_t t = ai * b[j] + out[i + j] + carry;
            out[i + j] = (uint32_t)t; carry = t >> 32;
        }
        out[i + nb] = (uint32_t)carry;
    }
}
static int bn_len(const uint32_t *a, int n) { while (n > 0 && a[n - 1] == 0) --n; return n; }

/* Knuth TAOCP vol.2 4.3.1 algorithm D: q[m-n+1], r[n] from u[m] / v[n]; v[n-1] != 0, m >= n. */
static void bn_divmod(uint32_t *q, uint32_t *r, const uint32_t *u, int m, const uint32_t *v, int n) {
    if (n == 1) {
        uint64_t k = 0;
        for (int j = m - 1; j >= 0; --j) { uint64_t t = (k << 32) | u[j]; q[j] = (uint32_t)(t / v[0]); k = t % v[0]; }
        r[0] = (uint32_t)k;
        return;
    }
    static __thread uint32_t un[2 * MAXD + 2], vn[MAXD + 1];
    int s = __builtin_clz(v[n - 1]);
    for (int i = n - 1; i > 0; --i) vn[i] = (v[i] << s) | (s ? (uint32_t)((uint64_t)v[i - 1] >> (32 - s)) : 0);
    vn[0] = v[0] << s;
    un[m] = s ? (uint32_t)((uint64_t)u[m - 1] >> (32 - s)) : 0;
    for (int i = m - 1; i > 0; --i) un[i] = (u[i] << s) | (s ? (uint32_t)((uint64_t)u[i - 1] >> (32 - s)) : 0);
    un[0] = u[0] << s;
    for (int j = m - n; j >= 0; --j) {
        uint64_t num = ((uint64_t)un[j + n] << 32) | un[j + n - 1];
        uint64_t qhat = num / vn[n - 1], rhat = num % vn[n - 1];
        while (qhat >= (1ull << 32) || qhat * vn[n - 2] > ((rhat << 32) | un[j + n - 2])) {
            --qhat; rhat += vn[n - 1];
            if (rhat >= (1ull << 32)) break;
        }
        int64_t borrow = 0; uint64_t carry = 0;
        for (int i = 0; i < n; ++i) {
            uint64_t pr = qhat * vn[i] + carry; carry = pr >> 32;
            int64_t t = (int64_t)un[i + j] - borrow - (int64_t)(pr & 0xffffffffu);
            un[i + j] = (uint32_t)t; borrow = (t < 0) ? 1 : 0;
        }
        int64_t t = (int64_t)un[j + n] - borrow - (int64_t)carry;
        un[j + n] = (uint32_t)t;
        if (t < 0) { /* add back */
            --qhat; uint64_t c = 0;
            for (int i = 0; i < n; ++i) { c += (uint64_t)un[i + j] + vn[i]; un[i + j] = (uint32_t)c; c >>= 32; }
            un[j + n] += (uint32_t)c;
        }
        q[j] = (uint32_t)qhat;
    }
    for (int i = 0; i < n - 1; ++i) r[i] = (un[i] >> s) | (s ? (uint32_t)((uint64_t)un[i + 1] << (32 - s)) : 0);
    r[n - 1] = un[n - 1] >> s;
}

/* AssignedInteger::to_big_uint (big_integer/mod.rs:348-359): limbs -> 32-bit digits. */
static int limbs_to_digits(uint32_t *d, const void *a, unsigned L, unsigned w) {
    if (w == 64) { for (unsigned i = 0; i < L; ++i) { uint64_t x = ((const uint64_t *)a)[i]; d[2 * i] = (uint32_t)x; d[2 * i + 1] = (uint32_t)(x >> 32); } return (int)(2 * L); }
    for (unsigned i = 0; i < L; ++i) d[i] = ((const uint32_t *)a)[i];
    return (int)L;
}
static uint64_t digits_limb(const uint32_t *d, unsigned i, unsigned w) {
    return w == 64 ? ((uint64_t)d[2 * i] | ((uint64_t)d[2 * i + 1] << 32)) : d[i];
}

/* ------------------------------------------------------------------------------------------------
 * BigIntChip::mul -- big_integer/chip.rs:386-419
 * ---------------------------------------------------------------------------------------------- */
static void mul_columns(const h2ro_params *p, const uint64_t *a, unsigned d0, const uint64_t *b, unsigned d1, wr *s, u256 *cols) {
    unsigned d = d0 + d1 - 1;
    for (unsigned i = 0; i < d; ++i) {
        u256 acc = u256_zero();                                /* assign_constant(0), :402 */
        unsigned j = (d1 >= i + 1) ? 0 : i + 1 - d1;          /* :403 */
        while (j < d0 && j <= i) {                             /* :404 */
            unsigned k = i - j;
            acc = u256_add(u256_mul64(a[j], b[k]), acc);       /* main_gate.mul_add, :408 */
            put256(s, acc, p->WB);
            ++j;
        }
        cols[i] = acc;                                         /* :411 */
    }
}

int h2ro_mul_columns(const h2ro_params *p, const void *a, const void *b, uint8_t *stream, uint64_t *cols_out) {
    uint64_t A[MAXL], B[MAXL]; static __thread u256 cols[2 * MAXL];
    for (unsigned i = 0; i < p->L; ++i) { A[i] = get_limb(a, i, p->w); B[i] = get_limb(b, i, p->w); }
    wr s = {stream};
    mul_columns(p, A, p->L, B, p->L, &s, cols);
    if (cols_out) memcpy(cols_out, cols, sizeof(u256) * (2 * p->L - 1));
    return H2RO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * BigIntChip::is_equal_muled -- big_integer/chip.rs:822-895 (num_limbs_l = num_limbs_r = L)
 * div_mod_main_gate (:1323-1349) is inlined at its two call sites with n = 2^w.
 * ---------------------------------------------------------------------------------------------- */
static int is_equal_muled(const h2ro_params *p, const u256 *a, const u256 *b, wr *s) {
    unsigned w = p->w, num_limbs = 2 * p->L - 1;
    u256 word_max; memcpy(&word_max, p->word_max, sizeof word_max);   /* :838 */
    unsigned carry_bits = p->carry_bits;                                /* :841-842 */
    u256 acc_extra = u256_zero();                                       /* :852 */
    u256 carry = u256_zero();                                           /* :855 */
    int eq_bit = 1;                                                     /* :856 */
    for (unsigned i = 0; i < num_limbs; ++i) {
        u256 a_b = u256_sub(a[i], b[i]);                                /* :859 (two's complement) */
        put256(s, a_b, p->WB);
        u256 sum = u256_add(u256_add(a_b, carry), word_max);            /* :860-861 */
        put256(s, sum, p->WB);
        u256 new_carry = u256_shr(sum, w);                              /* :864 -> :1336 */
        uint64_t c = u256_low(sum, w);                                  /*        :1337 */
        put256(s, new_carry, p->CB);
        put64(s, c, p->LB);
        u256 nq = u256_shl(new_carry, w);                               /* :1345 */
        put256(s, nq, p->WB);
        put256(s, u256_sub(sum, nq), p->LB);                            /* :1346 */
        carry = new_carry;                                              /* :865 */
        acc_extra = u256_add(acc_extra, word_max);                      /* :869-870 */
        put256(s, acc_extra, p->WB);
        u256 q_acc = u256_shr(acc_extra, w);                            /* :871 */
        uint64_t mod_acc = u256_low(acc_extra, w);
        put256(s, q_acc, p->CB);
        put64(s, mod_acc, p->LB);
        u256 nq2 = u256_shl(q_acc, w);
        put256(s, nq2, p->WB);
        put256(s, u256_sub(acc_extra, nq2), p->LB);
        int cs_acc_eq = (c == mod_acc);                                 /* :873 */
        put64(s, (uint64_t)cs_acc_eq, 1);
        eq_bit &= cs_acc_eq;                                            /* :874 */
        put64(s, (uint64_t)eq_bit, 1);
        acc_extra = q_acc;                                              /* :875 */
        if (i < num_limbs - 1) {
            emit_range_assign(s, new_carry, p->carry_sub_bits, carry_bits, p->CB); /* :879-885 */
            int range_eq = 1;                                           /* :886 (same value) */
            put64(s, (uint64_t)range_eq, 1);
            eq_bit &= range_eq;                                         /* :887 */
            put64(s, (uint64_t)eq_bit, 1);
        } else {
            int final_carry_eq = u256_eq(new_carry, acc_extra);         /* :890 */
            put64(s, (uint64_t)final_carry_eq, 1);
            eq_bit &= final_carry_eq;                                   /* :891 */
            put64(s, (uint64_t)eq_bit, 1);
        }
    }
    return eq_bit;
}

/* ------------------------------------------------------------------------------------------------
 * BigIntChip::mul_mod -- big_integer/chip.rs:542-629
 * ---------------------------------------------------------------------------------------------- */
static int mul_mod_limbs(const h2ro_params *p, const uint64_t *a, const uint64_t *b, const uint64_t *n, wr *s, uint64_t *r_out) {
    unsigned w = p->w, L = p->L;
    static __thread uint32_t ad[MAXD], bd[MAXD], nd[MAXD], full[2 * MAXD], qd[2 * MAXD + 1], rd[MAXD];
    static __thread u256 ab[2 * MAXL], qn[2 * MAXL], eq_b[2 * MAXL];
    uint64_t q[MAXL] = {0}, r[MAXL] = {0};
    int D = (w == 64) ? (int)(2 * L) : (int)L;
    for (unsigned i = 0; i < L; ++i) {
        if (w == 64) { ad[2 * i] = (uint32_t)a[i]; ad[2 * i + 1] = (uint32_t)(a[i] >> 32); bd[2 * i] = (uint32_t)b[i]; bd[2 * i + 1] = (uint32_t)(b[i] >> 32); nd[2 * i] = (uint32_t)n[i]; nd[2 * i + 1] = (uint32_t)(n[i] >> 32); }
        else { ad[i] = (uint32_t)a[i]; bd[i] = (uint32_t)b[i]; nd[i] = (uint32_t)n[i]; }
    }
    int nl = bn_len(nd, D);
    if (nl == 0) return H2RO_E_ZERO_MODULUS;                            /* :566 divides by zero */
    bn_mul(full, ad, D, bd, D);                                         /* :562 */
    memset(qd, 0, sizeof qd); memset(rd, 0, sizeof rd);
    bn_divmod(qd, rd, full, 2 * D, nd, nl);                             /* :564-567 */
    for (int i = D; i < 2 * D - nl + 1; ++i) if (qd[i]) return H2RO_E_NOT_REDUCED; /* :584 */
    for (unsigned i = 0; i < L; ++i) { q[i] = digits_limb(qd, i, w); r[i] = digits_limb(rd, i, w); } /* :570-582 */
    for (unsigned i = 0; i < L; ++i) emit_range_assign(s, u256_from64(q[i]), p->limb_sub_bits, w, p->LB); /* :588-591 */
    for (unsigned i = 0; i < L; ++i) emit_range_assign(s, u256_from64(r[i]), p->limb_sub_bits, w, p->LB); /* :596-599 */
    mul_columns(p, a, L, b, L, s, ab);                                  /* :608 */
    mul_columns(p, q, L, n, L, s, qn);                                  /* :609 */
    for (unsigned i = 0; i < 2 * L - 1; ++i) {                          /* :614-623 */
        if (i < L) { eq_b[i] = u256_add(qn[i], u256_from64(r[i])); put256(s, eq_b[i], p->WB); } /* :617 */
        else eq_b[i] = qn[i];
    }
    int ok = is_equal_muled(p, ab, eq_b, s);                            /* :626 */
    if (!ok) return H2RO_E_SHAPE;                                       /* assert_one :1062 -- unreachable */
    memcpy(r_out, r, sizeof(uint64_t) * L);                             /* :628 */
    return H2RO_OK;
}

static void load_limbs(uint64_t *dst, const void *src, unsigned L, unsigned w) { for (unsigned i = 0; i < L; ++i) dst[i] = get_limb(src, i, w); }
static void store_limbs(void *dst, const uint64_t *src, unsigned L, unsigned w) { for (unsigned i = 0; i < L; ++i) set_limb(dst, i, w, src[i]); }

int h2ro_mul_mod(const h2ro_params *p, const void *a, const void *b, const void *n, uint8_t *stream, void *r_out) {
    uint64_t A[MAXL], B[MAXL], N[MAXL], R[MAXL];
    load_limbs(A, a, p->L, p->w); load_limbs(B, b, p->L, p->w); load_limbs(N, n, p->L, p->w);
    wr s = {stream};
    int st = mul_mod_limbs(p, A, B, N, &s, R);
    if (st == H2RO_OK && r_out) store_limbs(r_out, R, p->L, p->w);
    return st;
}

/* ------------------------------------------------------------------------------------------------
 * BigIntChip::pow_mod_fixed_exp -- big_integer/chip.rs:710-742
 * ---------------------------------------------------------------------------------------------- */
static size_t e_num_bits(const uint8_t *e_le, size_t e_len) { /* Self::bits_size(e), :717 */
    while (e_len > 0 && e_le[e_len - 1] == 0) --e_len;
    if (e_len == 0) return 0;
    return 8 * (e_len - 1) + (size_t)(32 - __builtin_clz((unsigned)e_le[e_len - 1]));
}
uint64_t h2ro_pow_fixed_stream_bytes(const h2ro_params *p, const uint8_t *e_le, size_t e_len) {
    size_t nb = e_num_bits(e_le, e_len); uint64_t T = 0;
    for (size_t i = 0; i < nb; ++i) T += 1 + ((e_le[i / 8] >> (i % 8)) & 1);
    return T * p->mul_mod_stream_bytes + (uint64_t)p->L * p->LB;
}
int h2ro_pow_mod_fixed_exp(const h2ro_params *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len, uint8_t *stream, void *out) {
    unsigned L = p->L, w = p->w;
    uint64_t N[MAXL], acc[MAXL], squared[MAXL], cur_sq[MAXL];
    load_limbs(N, n, L, w); load_limbs(squared, x, L, w);               /* :730 */
    memset(acc, 0, sizeof acc); acc[0] = 1;                             /* :729 const 1 padded to L limbs */
    wr s = {stream};
    size_t nb = e_num_bits(e_le, e_len);                                /* :717-728 */
    for (size_t i = 0; i < nb; ++i) {                                   /* :731 */
        int bit = (e_le[i / 8] >> (i % 8)) & 1;                         /* :723-724 */
        memcpy(cur_sq, squared, sizeof(uint64_t) * L);                  /* :732 */
        int st = mul_mod_limbs(p, cur_sq, cur_sq, N, &s, squared);     /* :734 square_mod -> :648 */
        if (st) return st;
        if (!bit) continue;                                             /* :735-737 */
        st = mul_mod_limbs(p, acc, cur_sq, N, &s, acc);                 /* :739 */
        if (st) return st;
    }
    for (unsigned i = 0; i < L; ++i) put64(&s, acc[i], p->LB);
    if (out) store_limbs(out, acc, L, w);                               /* :741 */
    return H2RO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * BigIntChip::pow_mod (variable exponent) -- big_integer/chip.rs:664-696
 * ---------------------------------------------------------------------------------------------- */
uint64_t h2ro_pow_var_stream_bytes(const h2ro_params *p, uint32_t e_num_limbs, uint32_t exp_limb_bits) {
    uint64_t nb = (uint64_t)e_num_limbs * exp_limb_bits;
    return nb + nb * (2 * p->mul_mod_stream_bytes + (uint64_t)p->L * p->LB) + (uint64_t)p->L * p->LB;
}
int h2ro_pow_mod(const h2ro_params *p, const void *x, const void *e_limbs, uint32_t e_num_limbs, uint32_t exp_limb_bits, const void *n, uint8_t *stream, void *out) {
    unsigned L = p->L, w = p->w;
    uint64_t N[MAXL], acc[MAXL], squared[MAXL], muled[MAXL];
    load_limbs(N, n, L, w); load_limbs(squared, x, L, w);               /* :683 */
    memset(acc, 0, sizeof acc); acc[0] = 1;                             /* :682 */
    wr s = {stream};
    /* main_gate.to_bits(limb, exp_limb_bits) (:677) constrains limb == sum of its exp_limb_bits bits: a wider limb
     * makes the reference's circuit unsatisfiable */
    if (exp_limb_bits == 0 || exp_limb_bits > w) return H2RO_E_SHAPE;
    for (uint32_t l = 0; l < e_num_limbs; ++l)
        if (exp_limb_bits < 64 && (get_limb(e_limbs, l, w) >> exp_limb_bits) != 0) return H2RO_E_SHAPE;
    for (uint32_t l = 0; l < e_num_limbs; ++l)                          /* :674-681 to_bits, LSB first */
        for (uint32_t t = 0; t < exp_limb_bits; ++t) put64(&s, (get_limb(e_limbs, l, w) >> t) & 1, 1);
    for (uint32_t l = 0; l < e_num_limbs; ++l)
        for (uint32_t t = 0; t < exp_limb_bits; ++t) {                  /* :684 */
            int bit = (int)((get_limb(e_limbs, l, w) >> t) & 1);
            int st = mul_mod_limbs(p, acc, squared, N, &s, muled);      /* :686 */
            if (st) return st;
            for (unsigned j = 0; j < L; ++j) { if (bit) acc[j] = muled[j]; put64(&s, acc[j], p->LB); } /* :688-691 select */
            st = mul_mod_limbs(p, squared, squared, N, &s, squared);    /* :693 */
            if (st) return st;
        }
    for (unsigned i = 0; i < L; ++i) put64(&s, acc[i], p->LB);
    if (out) store_limbs(out, acc, L, w);                               /* :695 */
    return H2RO_OK;
}

/* big_integer/utils.rs:2-17 -- the reference tests' expected answer (iterative form of the same
 * recursion: left-to-right square and multiply; b == 0 returns 1 without reducing). */
int h2ro_big_pow_mod(const h2ro_params *p, const void *a, const uint8_t *e_le, size_t e_len, const void *n, void *out) {
    unsigned L = p->L, w = p->w;
    static __thread uint32_t ad[MAXD], nd[MAXD], xd[MAXD], t[2 * MAXD], qd[2 * MAXD + 1], rd[MAXD];
    int D = limbs_to_digits(ad, a, L, w); limbs_to_digits(nd, n, L, w);
    int nl = bn_len(nd, D);
    if (nl == 0) return H2RO_E_ZERO_MODULUS;
    memset(xd, 0, sizeof xd); xd[0] = 1;
    size_t nb = e_num_bits(e_le, e_len);
    for (size_t i = nb; i-- > 0;) {
        bn_mul(t, xd, D, xd, D); memset(qd, 0, sizeof qd); memset(rd, 0, sizeof rd);
        bn_divmod(qd, rd, t, 2 * D, nd, nl); memcpy(xd, rd, sizeof(uint32_t) * (size_t)D);
        if ((e_le[i / 8] >> (i % 8)) & 1) {
            bn_mul(t, ad, D, xd, D); memset(qd, 0, sizeof qd); memset(rd, 0, sizeof rd);
            bn_divmod(qd, rd, t, 2 * D, nd, nl); memcpy(xd, rd, sizeof(uint32_t) * (size_t)D);
        }
    }
    for (unsigned i = 0; i < L; ++i) set_limb(out, i, w, digits_limb(xd, i, w));
    return H2RO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * assert_in_field -- big_integer/chip.rs:1150-1158 -> 998-1006 -> 908-919
 * ---------------------------------------------------------------------------------------------- */
/* BigIntChip::add, big_integer/chip.rs:245-297; returns max_n + 1 limbs. */
static unsigned add_fresh(const h2ro_params *p, const uint64_t *a, unsigned n1, const uint64_t *b, unsigned n2, wr *s, uint64_t *c_vals) {
    unsigned w = p->w, max_n = n1 < n2 ? n2 : n1, SB = p->LB + 8;
    uint64_t carry = 0;                                                 /* :268 */
    for (unsigned i = 0; i < max_n; ++i) {
        uint64_t ai = i < n1 ? a[i] : 0, bi = i < n2 ? b[i] : 0;        /* :258-263 zero padding */
        u256 a_b = u256_add(u256_from64(ai), u256_from64(bi));          /* :272 */
        put256(s, a_b, SB);
        u256 sum = u256_add(a_b, u256_from64(carry));                   /* :273 */
        put256(s, sum, SB);
        uint64_t c = u256_low(sum, w);                                  /* :276 */
        uint64_t cy = u256_shr(sum, w).v[0];                            /* :277 */
        emit_range_assign(s, u256_from64(c), p->limb_sub_bits, w, p->LB);  /* :279-280 */
        emit_range_assign(s, u256_from64(cy), p->limb_sub_bits, w, p->LB); /* :281-282 */
        put256(s, u256_add(u256_shl(u256_from64(cy), w), u256_from64(c)), SB); /* :283 */
        c_vals[i] = c; carry = cy;                                      /* :286-287 */
    }
    c_vals[max_n] = carry;                                              /* :290 */
    return max_n + 1;
}
/* BigIntChip::is_equal_fresh, big_integer/chip.rs:780-805 */
static int is_equal_fresh(const uint64_t *a, unsigned n1, const uint64_t *b, unsigned n2, wr *s) {
    int is_a_larger = n1 > n2; unsigned max_n = is_a_larger ? n1 : n2; int eq_bit = 1;
    for (unsigned i = 0; i < max_n; ++i) {
        int flag;
        if (is_a_larger && i >= n2) flag = a[i] == 0;
        else if (!is_a_larger && i >= n1) flag = b[i] == 0;
        else flag = a[i] == b[i];
        put64(s, (uint64_t)flag, 1); eq_bit &= flag; put64(s, (uint64_t)eq_bit, 1);
    }
    return eq_bit;
}
/* BigIntChip::sub_unchecked, big_integer/chip.rs:1286-1318; returns -1 where the reference panics. */
static int sub_unchecked(const h2ro_params *p, const uint64_t *a, unsigned n1, const uint64_t *b, unsigned n2, wr *s, uint64_t *c) {
    unsigned w = p->w;
    if (n1 < n2) return -1;                                             /* :1294 */
    uint64_t borrow = 0, mask = w == 64 ? ~0ull : ((1ull << w) - 1);
    for (unsigned i = 0; i < n1; ++i) {                                 /* :1300 a_big - b_big, then :1304-1311 */
        uint64_t bi = i < n2 ? b[i] : 0;
        uint64_t t = (a[i] - bi - borrow) & mask;
        borrow = (a[i] < bi + borrow) || (bi + borrow < bi);
        c[i] = t;
    }
    if (borrow) return -1;                                              /* :1300 underflow panics */
    for (unsigned i = 0; i < n1; ++i) emit_range_assign(s, u256_from64(c[i]), p->limb_sub_bits, w, p->LB); /* :1307-1308 */
    uint64_t added[MAXL + 4];
    unsigned na = add_fresh(p, b, n2, c, n1, s, added);                 /* :1315 */
    if (!is_equal_fresh(a, n1, added, na, s)) return -1;                /* :1316 */
    return 0;
}
/* BigIntChip::sub, big_integer/chip.rs:310-373 */
static int sub_fresh(const h2ro_params *p, const uint64_t *a, unsigned n1, const uint64_t *b, unsigned n2, wr *s, int *is_overflowed, uint64_t *real) {
    uint64_t max_int[MAXL + 4], inflated_a[MAXL + 4], inflated_subed[MAXL + 4], sel_l[MAXL + 4], sel_r[MAXL + 4];
    uint64_t mask = p->w == 64 ? ~0ull : ((1ull << p->w) - 1);
    for (unsigned i = 0; i < n2; ++i) max_int[i] = mask;               /* :319 max_value :138-154 */
    unsigned nia = add_fresh(p, a, n1, max_int, n2, s, inflated_a);     /* :321 */
    if (sub_unchecked(p, inflated_a, nia, b, n2, s, inflated_subed)) return -1; /* :323 */
    int is_not_overflowed = inflated_subed[n2] == 1;                    /* :330 */
    put64(s, (uint64_t)is_not_overflowed, 1);
    *is_overflowed = !is_not_overflowed;                                /* :331 */
    put64(s, (uint64_t)*is_overflowed, 1);
    unsigned num_limbs_l = nia, num_limbs_r = n1 > n2 ? n1 : n2;
    for (unsigned i = 0; i < num_limbs_l; ++i) {                        /* :345-357 */
        uint64_t v = (i >= n2) ? (is_not_overflowed ? inflated_subed[i] : 0) : (is_not_overflowed ? inflated_subed[i] : b[i]);
        put64(s, v, p->LB); sel_l[i] = v;
    }
    for (unsigned i = 0; i < num_limbs_r; ++i) {                        /* :358-367 */
        uint64_t v;
        if (i >= n1) v = is_not_overflowed ? max_int[i] : 0;
        else if (i >= n2) v = is_not_overflowed ? 0 : a[i];
        else v = is_not_overflowed ? max_int[i] : a[i];
        put64(s, v, p->LB); sel_r[i] = v;
    }
    return sub_unchecked(p, sel_l, num_limbs_l, sel_r, num_limbs_r, s, real); /* :371 */
}
uint64_t h2ro_in_field_stream_bytes(const h2ro_params *p) {
    uint64_t L = p->L, LB = p->LB, ns = p->limb_nsub, SB = LB + 8;
    uint64_t add_step = 3 * SB + 2 * (LB + ns);
    uint64_t add_L = L * add_step, add_L1 = (L + 1) * add_step;
    uint64_t su1 = (L + 1) * (LB + ns) + add_L1 + 2 * (L + 2);          /* sub_unchecked on L+1 limbs */
    return add_L + su1 + 2 + (L + 1) * LB + L * LB + su1 + 2 * L + 2;
}
int h2ro_assert_in_field(const h2ro_params *p, const void *a, const void *n, uint8_t *stream, int *is_less) {
    uint64_t A[MAXL], N[MAXL]; load_limbs(A, a, p->L, p->w); load_limbs(N, n, p->L, p->w);
    wr s = {stream};
    int is_overflowed = 0;
    uint64_t real[MAXL + 4];
    if (sub_fresh(p, A, p->L, N, p->L, &s, &is_overflowed, real)) return H2RO_E_SHAPE; /* :939 */
    int is_eq = is_equal_fresh(A, p->L, N, p->L, &s);                   /* :916 */
    int is_not_eq = !is_eq;                                             /* :917 */
    put64(&s, (uint64_t)is_not_eq, 1);
    int lt = is_overflowed & is_not_eq;                                 /* :918 */
    put64(&s, (uint64_t)lt, 1);
    if (is_less) *is_less = lt;
    return lt ? H2RO_OK : H2RO_E_NOT_IN_FIELD;                          /* assert_one, :1157 */
}

/* ------------------------------------------------------------------------------------------------
 * the Fresh-integer family of BigIntInstructions
 *   add :245-297, sub :310-373, add_mod :452-481, sub_mod :495-528, is_zero :754-767,
 *   is_equal_fresh :780-805, is_less_than :908-919, is_less_than_or_equal :932-941,
 *   is_greater_than :954-963, is_greater_than_or_equal :976-985, is_in_field :998-1006
 * ---------------------------------------------------------------------------------------------- */
static uint64_t add_bytes(const h2ro_params *p, uint64_t n) { return n * (3ull * (p->LB + 8) + 2ull * (p->LB + p->limb_nsub)); }
static uint64_t subu_bytes(const h2ro_params *p, uint64_t n1) { return n1 * (p->LB + p->limb_nsub) + add_bytes(p, n1) + 2 * (n1 + 1); }
static uint64_t sub_bytes(const h2ro_params *p, uint64_t nA, uint64_t nB) {
    uint64_t m = nA > nB ? nA : nB, n1 = m + 1;
    return add_bytes(p, m) + subu_bytes(p, n1) + 2 + n1 * p->LB + m * p->LB + subu_bytes(p, n1);
}
uint64_t h2ro_fresh_op_stream_bytes(const h2ro_params *p, int op) {
    uint64_t L = p->L;
    switch (op) {
        case H2RO_OP_ADD: return add_bytes(p, L);
        case H2RO_OP_SUB: return sub_bytes(p, L, L);
        case H2RO_OP_ADD_MOD: return add_bytes(p, L) + sub_bytes(p, L + 1, L) + (L + 2) * p->LB;
        case H2RO_OP_SUB_MOD: return sub_bytes(p, L, L) + sub_bytes(p, L, L + 1) + (L + 2) * p->LB;
        case H2RO_OP_IS_ZERO: return 2 * L;
        case H2RO_OP_IS_EQUAL_FRESH: return 2 * L;
        case H2RO_OP_IS_LESS_THAN: case H2RO_OP_IS_IN_FIELD: return sub_bytes(p, L, L) + 2 * L + 2;
        case H2RO_OP_IS_LESS_THAN_OR_EQUAL: return sub_bytes(p, L, L);
        case H2RO_OP_IS_GREATER_THAN: return sub_bytes(p, L, L) + 1;
        case H2RO_OP_IS_GREATER_THAN_OR_EQUAL: return sub_bytes(p, L, L) + 2 * L + 2 + 1;
        default: return 0;
    }
}
static int is_less_than(const h2ro_params *p, const uint64_t *A, const uint64_t *B, wr *s, int *lt) {
    int ov = 0; uint64_t real[MAXL + 4];
    if (sub_fresh(p, A, p->L, B, p->L, s, &ov, real)) return -1;       /* :915 -> :939 */
    int is_eq = is_equal_fresh(A, p->L, B, p->L, s);                    /* :916 */
    put64(s, (uint64_t)!is_eq, 1);                                      /* :917 */
    *lt = ov & !is_eq; put64(s, (uint64_t)*lt, 1);                      /* :918 */
    return 0;
}
/* value_out receives *nvalue limbs (may be NULL for predicate ops); flag_out the predicate / overflow bit. */
int h2ro_fresh_op(const h2ro_params *p, int op, const void *a, const void *b, const void *n, uint8_t *stream,
                  void *value_out, uint32_t *nvalue, int *flag_out) {
    unsigned L = p->L, w = p->w;
    uint64_t A[MAXL], B[MAXL], N[MAXL], v1[MAXL + 4], v2[MAXL + 4], res[MAXL + 4];
    load_limbs(A, a, L, w);
    if (b) load_limbs(B, b, L, w); else memset(B, 0, sizeof B);
    if (n) load_limbs(N, n, L, w); else memset(N, 0, sizeof N);
    wr s = {stream};
    unsigned nv = 0; int flag = -1, ov = 0, ov2 = 0;
    switch (op) {
        case H2RO_OP_ADD: nv = add_fresh(p, A, L, B, L, &s, res); break;
        case H2RO_OP_SUB: if (sub_fresh(p, A, L, B, L, &s, &ov, res)) return H2RO_E_SHAPE; nv = L + 1; flag = ov; break;
        case H2RO_OP_ADD_MOD: {                                            /* chip.rs:452-481 */
            unsigned na = add_fresh(p, A, L, B, L, &s, v1);                 /* :462 */
            if (sub_fresh(p, v1, na, N, L, &s, &ov, v2)) return H2RO_E_SHAPE; /* :464 */
            unsigned num = na + 1;                                          /* subed.num_limbs() */
            for (unsigned i = 0; i < num; ++i) {                            /* :469-474 */
                uint64_t ad = i < na ? v1[i] : 0;
                res[i] = ov ? ad : v2[i]; put64(&s, res[i], p->LB);
            }
            for (unsigned i = L; i < num; ++i) if (res[i]) return H2RO_E_SHAPE;   /* :475-478 assert_zero */
            nv = L; break;
        }
        case H2RO_OP_SUB_MOD: {                                            /* chip.rs:495-528 */
            if (sub_fresh(p, A, L, B, L, &s, &ov, v1)) return H2RO_E_SHAPE;       /* :506, L+1 limbs */
            if (sub_fresh(p, N, L, v1, L + 1, &s, &ov2, v2)) return H2RO_E_SHAPE; /* :509, L+2 limbs */
            if (ov2) return H2RO_E_NOT_IN_FIELD;                            /* :510 assert_zero(is_overflowed2) */
            unsigned num = L + 2;
            for (unsigned i = 0; i < num; ++i) {                            /* :516-521 */
                uint64_t s1 = i < L + 1 ? v1[i] : 0;
                res[i] = ov ? v2[i] : s1; put64(&s, res[i], p->LB);
            }
            for (unsigned i = L; i < num; ++i) if (res[i]) return H2RO_E_SHAPE;
            nv = L; break;
        }
        case H2RO_OP_IS_ZERO: {                                            /* chip.rs:754-767 */
            int bit = 1;
            for (unsigned i = 0; i < L; ++i) { int z = A[i] == 0; put64(&s, (uint64_t)z, 1); bit &= z; put64(&s, (uint64_t)bit, 1); }
            flag = bit; break;
        }
        case H2RO_OP_IS_EQUAL_FRESH: flag = is_equal_fresh(A, L, B, L, &s); break;
        case H2RO_OP_IS_LESS_THAN: case H2RO_OP_IS_IN_FIELD: if (is_less_than(p, A, B, &s, &flag)) return H2RO_E_SHAPE; break;
        case H2RO_OP_IS_LESS_THAN_OR_EQUAL: if (sub_fresh(p, A, L, B, L, &s, &ov, res)) return H2RO_E_SHAPE; flag = ov; break;
        case H2RO_OP_IS_GREATER_THAN:                                      /* chip.rs:954-963 */
            if (sub_fresh(p, A, L, B, L, &s, &ov, res)) return H2RO_E_SHAPE;
            flag = !ov; put64(&s, (uint64_t)flag, 1); break;
        case H2RO_OP_IS_GREATER_THAN_OR_EQUAL:                             /* chip.rs:976-985 */
            if (is_less_than(p, A, B, &s, &flag)) return H2RO_E_SHAPE;
            flag = !flag; put64(&s, (uint64_t)flag, 1); break;
        default: return H2RO_E_SHAPE;
    }
    if (nvalue) *nvalue = nv;
    if (value_out && nv) store_limbs(value_out, res, nv, w);
    if (flag_out) *flag_out = flag;
    return H2RO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * refresh (big_integer/chip.rs:168-233) with RefreshAux::new(w, L, L)
 * (big_integer/mod.rs:428-482), and stand-alone is_equal_muled (chip.rs:822-895).
 * Muled integers are passed as 2L-1 values of 4 x u64 (256-bit little-endian).
 * ---------------------------------------------------------------------------------------------- */
static unsigned refresh_aux(unsigned w, unsigned L, unsigned *inc) { /* mod.rs:428-482, num_limbs_l = num_limbs_r = L */
    static __thread u256 muled[2 * MAXL + 4];
    unsigned d = 2 * L - 1, len = d;
    uint64_t bm1 = w == 64 ? ~0ull : ((1ull << w) - 1);
    u256 sq = u256_mul64(bm1, bm1);
    for (unsigned i = 0; i < d; ++i) {
        unsigned cnt = (i < L) ? i + 1 : 2 * L - 1 - i;
        muled[i] = u256_zero();
        for (unsigned k = 0; k < cnt; ++k) muled[i] = u256_add(muled[i], sq);
    }
    unsigned n = 0;
    for (unsigned cur = 0; cur <= d; ++cur) {                          /* while cur_d <= max_d */
        if (cur >= len) { muled[len++] = u256_zero(); }
        unsigned nb = u256_bits(muled[cur]);
        unsigned chunks = nb % w == 0 ? nb / w : nb / w + 1;
        inc[n++] = chunks - 1;
        u256 t = muled[cur]; uint64_t ch[8];
        for (unsigned j = 0; j < chunks; ++j) { ch[j] = u256_low(t, w); t = u256_shr(t, w); }
        muled[cur] = u256_zero();
        for (unsigned j = 0; j < chunks; ++j) {
            while (len <= cur + j) muled[len++] = u256_zero();
            muled[cur + j] = u256_add(muled[cur + j], u256_from64(ch[j]));
        }
    }
    return n;
}
uint64_t h2ro_refresh_stream_bytes(const h2ro_params *p) {
    unsigned inc[2 * MAXL + 4]; unsigned n = refresh_aux(p->w, p->L, inc);
    uint64_t b = 0;
    for (unsigned i = 0; i < n; ++i) b += (uint64_t)(inc[i] + 1) * (p->CB + p->LB + p->WB + p->LB) + (uint64_t)inc[i] * p->WB;
    return b + (uint64_t)n * (p->LB + p->limb_nsub);
}
int h2ro_refresh(const h2ro_params *p, const uint64_t *muled /* (2L-1) x 4 u64 */, uint8_t *stream, void *fresh_out /* 2L limbs */) {
    unsigned w = p->w, L = p->L;
    unsigned inc[2 * MAXL + 4]; unsigned nf = refresh_aux(w, L, inc);
    static __thread u256 r[2 * MAXL + 8];
    wr s = {stream};
    for (unsigned i = 0; i < nf + 4; ++i) r[i] = u256_zero();
    for (unsigned i = 0; i < 2 * L - 1; ++i) memcpy(&r[i], muled + 4 * i, sizeof(u256));   /* :186-192 */
    for (unsigned i = 0; i < nf; ++i) {                                  /* :195 */
        u256 limb = r[i];
        for (unsigned j = 0; j < inc[i] + 1; ++j) {                      /* :198 */
            u256 q = u256_shr(limb, w); uint64_t n = u256_low(limb, w);  /* :201 */
            put256(&s, q, p->CB); put64(&s, n, p->LB);
            u256 nq = u256_shl(q, w);
            put256(&s, nq, p->WB); put256(&s, u256_sub(limb, nq), p->LB);
            if (j == 0) r[i] = u256_from64(n);                           /* :204 */
            else { r[i + j] = u256_add(r[i + j], u256_from64(n)); put256(&s, r[i + j], p->WB); }   /* :207 */
            limb = q;
        }
        if (u256_bits(limb)) return H2RO_E_NOT_REDUCED;                  /* :213 assert_zero */
    }
    for (unsigned i = 0; i < nf; ++i) {                                  /* :217-226 */
        emit_range_assign(&s, r[i], p->limb_sub_bits, w, p->LB);
        if (fresh_out) set_limb(fresh_out, i, w, r[i].v[0]);
    }
    return H2RO_OK;
}
uint64_t h2ro_is_equal_muled_stream_bytes(const h2ro_params *p) {
    uint64_t C = 2ull * p->L - 1;
    return C * (5ull * p->WB + 2ull * p->CB + 4ull * p->LB + 4) + (C - 1) * (p->CB + p->carry_nsub);
}
int h2ro_is_equal_muled(const h2ro_params *p, const uint64_t *a, const uint64_t *b, uint8_t *stream, int *eq_bit) {
    static __thread u256 A[2 * MAXL], B[2 * MAXL];
    for (unsigned i = 0; i < 2 * p->L - 1; ++i) { memcpy(&A[i], a + 4 * i, sizeof(u256)); memcpy(&B[i], b + 4 * i, sizeof(u256)); }
    wr s = {stream};
    int e = is_equal_muled(p, A, B, &s);
    if (eq_bit) *eq_bit = e;
    return H2RO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * the encoded-message check of RSAChip::verify_pkcs1v15_signature,
 * src/chip.rs:136-198 (LIMB_WIDTH = 64 only, src/chip.rs:203)
 * ---------------------------------------------------------------------------------------------- */
uint64_t h2ro_pkcs1v15_stream_bytes(const h2ro_params *p) {
    return 8 + 4 + 2 * (4 + 8) + 8 + 4 + 2ull * (p->L - 8) + 2;
}
int h2ro_pkcs1v15_em_check(const h2ro_params *p, const void *powed_, const uint64_t hashed[4], uint8_t *stream, int *is_valid) {
    if (p->w != 64 || p->L < 9) return H2RO_E_SHAPE;
    const uint64_t *powed = (const uint64_t *)powed_;
    wr s = {stream}; int is_eq = 1, f;
    for (unsigned i = 0; i < 4; ++i) { f = powed[i] == hashed[i]; put64(&s, (uint64_t)f, 1); is_eq &= f; put64(&s, (uint64_t)is_eq, 1); } /* :141-144 */
    int f1 = powed[4] == 217300885422736416ull;                         /* :150, :153 */
    int f2 = powed[5] == 938447882527703397ull;                         /* :152, :154 */
    put64(&s, (uint64_t)f1, 1); put64(&s, (uint64_t)f2, 1);
    is_eq &= f1; put64(&s, (uint64_t)is_eq, 1);                         /* :155 */
    is_eq &= f2; put64(&s, (uint64_t)is_eq, 1);                         /* :156 */
    uint64_t low = powed[6] & 0xffffffffull, high = powed[6] >> 32;     /* :159-168 */
    emit_range_assign(&s, u256_from64(low), 4, 32, 4);                  /* :170 */
    emit_range_assign(&s, u256_from64(high), 4, 32, 4);                 /* :171 */
    put64(&s, (high << 32) + low, 8);                                   /* :173 */
    f = low == 3158320ull; put64(&s, (uint64_t)f, 1); is_eq &= f; put64(&s, (uint64_t)is_eq, 1);   /* :175-177 */
    f = high == 4294967295ull; put64(&s, (uint64_t)f, 1); is_eq &= f; put64(&s, (uint64_t)is_eq, 1); /* :180-182 */
    for (unsigned i = 7; i < p->L - 1; ++i) { f = powed[i] == 18446744073709551615ull; put64(&s, (uint64_t)f, 1); is_eq &= f; put64(&s, (uint64_t)is_eq, 1); } /* :185-188 */
    f = powed[p->L - 1] == 562949953421311ull;                          /* :191-196 */
    put64(&s, (uint64_t)f, 1); is_eq &= f; put64(&s, (uint64_t)is_eq, 1);
    if (is_valid) *is_valid = is_eq;
    return H2RO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * batch driver for the CPU-baseline timing leg (pthreads, one element per task)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const h2ro_params *p; const uint8_t *x, *n; const uint8_t *e_le; size_t e_len;
    uint64_t batch; uint8_t *stream; uint8_t *out; uint8_t *status; uint64_t stream_bytes;
    volatile uint64_t *next; pthread_mutex_t *mu;
} batch_job;
static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    size_t eb = (size_t)j->p->L * j->p->LB;
    for (;;) {
        pthread_mutex_lock(j->mu); uint64_t i = (*j->next)++; pthread_mutex_unlock(j->mu);
        if (i >= j->batch) break;
        int st = h2ro_pow_mod_fixed_exp(j->p, j->x + i * eb, j->n + i * eb, j->e_le, j->e_len,
                                        j->stream ? j->stream + i * j->stream_bytes : NULL, j->out ? j->out + i * eb : NULL);
        if (j->status) j->status[i] = (uint8_t)st;
    }
    return NULL;
}
int h2ro_pow_mod_fixed_exp_batch(const h2ro_params *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len,
                                 uint64_t batch, uint8_t *stream, void *out, uint8_t *status, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile uint64_t next = 0; pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    batch_job job = {p, (const uint8_t *)x, (const uint8_t *)n, e_le, e_len, batch, stream, (uint8_t *)out, status,
                     h2ro_pow_fixed_stream_bytes(p, e_le, e_len), &next, &mu};
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, batch_worker, &job);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    return H2RO_OK;
}

/* Timing driver for the cpu_baseline leg of bench.py: `nthreads` persistent pthreads, thread t runs elements
 * t, t + nthreads, ... of the batch, `passes` times over, each writing the element's full op-trace stream into its own
 * reusable buffer (allocated and touched before the clock starts).  Returns the wall seconds between the start
 * barrier and the last thread's finish through *seconds, and the number of elements whose status was not OK. */
#include <time.h>
typedef struct {
    const h2ro_params *p; const uint8_t *x, *n; const uint8_t *e_le; size_t e_len;
    uint64_t batch, passes, stream_bytes; int tid, nthreads; pthread_barrier_t *start; uint64_t failed;
} time_job;
static void *time_worker(void *arg) {
    time_job *j = (time_job *)arg;
    size_t eb = (size_t)j->p->L * j->p->LB;
    uint8_t *buf = (uint8_t *)malloc(j->stream_bytes ? j->stream_bytes : 1);
    uint8_t res[MAXL * 8];
    if (buf) memset(buf, 0, j->stream_bytes);
    pthread_barrier_wait(j->start);
    if (buf)
        for (uint64_t r = 0; r < j->passes; ++r)
            for (uint64_t i = (uint64_t)j->tid; i < j->batch; i += (uint64_t)j->nthreads)
                if (h2ro_pow_mod_fixed_exp(j->p, j->x + i * eb, j->n + i * eb, j->e_le, j->e_len, buf, res) != H2RO_OK) j->failed++;
    free(buf);
    return NULL;
}
int h2ro_pow_mod_fixed_exp_timed(const h2ro_params *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len,
                                 uint64_t batch, uint64_t passes, int nthreads, double *seconds, uint64_t *failed) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    time_job *jobs = (time_job *)malloc(sizeof(time_job) * (size_t)nthreads);
    pthread_barrier_t start;
    if (!th || !jobs || pthread_barrier_init(&start, NULL, (unsigned)nthreads + 1)) { free(th); free(jobs); return H2RO_E_SHAPE; }
    uint64_t sb = h2ro_pow_fixed_stream_bytes(p, e_le, e_len);
    for (int t = 0; t < nthreads; ++t) {
        time_job j = {p, (const uint8_t *)x, (const uint8_t *)n, e_le, e_len, batch, passes, sb, t, nthreads, &start, 0};
        jobs[t] = j;
        pthread_create(&th[t], NULL, time_worker, &jobs[t]);
    }
    struct timespec t0, t1;
    pthread_barrier_wait(&start);            /* every thread has its buffer and is ready */
    clock_gettime(CLOCK_MONOTONIC, &t0);
    uint64_t bad = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); bad += jobs[t].failed; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (failed) *failed = bad;
    pthread_barrier_destroy(&start); free(th); free(jobs);
    return H2RO_OK;
}

/* -----------------------------------------------------------------------------------------------------------------
 * The caller of the path: RSASignatureVerifier::verify_pkcs1v15_signature, src/lib.rs:183-246.
 * Step 1 (:205-209) is SHA-256 of the message bytes.  The reference gets the digest VALUES from the third-party crates
 * sha2 0.10.6 (tests, benches: Sha256::digest) and halo2-dynamic-sha256 (the chip's digest cells); both compute
 * FIPS 180-4 SHA-256, restated here from the standard (section 4.1.2 functions, 4.2.2 constants, 5.1.1 padding, 5.3.3 initial
 * hash value, 6.2.2 computation) and pinned by the standard's own example digests in tests/golden/sha256_kat.json.
 */
static uint32_t sha_rotr(uint32_t x, unsigned n) { return (x >> n) | (x << (32 - n)); }
static const uint32_t SHA_K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
    0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
    0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
    0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
    0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
    0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
static void sha_block(uint32_t H[8], const uint8_t blk[64]) {
    uint32_t W[64];
    for (int t = 0; t < 16; ++t) W[t] = ((uint32_t)blk[4 * t] << 24) | ((uint32_t)blk[4 * t + 1] << 16) | ((uint32_t)blk[4 * t + 2] << 8) | blk[4 * t + 3];
    for (int t = 16; t < 64; ++t) {
        uint32_t s0 = sha_rotr(W[t - 15], 7) ^ sha_rotr(W[t - 15], 18) ^ (W[t - 15] >> 3);
        uint32_t s1 = sha_rotr(W[t - 2], 17) ^ sha_rotr(W[t - 2], 19) ^ (W[t - 2] >> 10);
        W[t] = s1 + W[t - 7] + s0 + W[t - 16];
    }
    uint32_t v[8];
    memcpy(v, H, sizeof v);
    for (int t = 0; t < 64; ++t) {
        uint32_t S1 = sha_rotr(v[4], 6) ^ sha_rotr(v[4], 11) ^ sha_rotr(v[4], 25);
        uint32_t ch = (v[4] & v[5]) ^ (~v[4] & v[6]);
        uint32_t T1 = v[7] + S1 + ch + SHA_K[t] + W[t];
        uint32_t S0 = sha_rotr(v[0], 2) ^ sha_rotr(v[0], 13) ^ sha_rotr(v[0], 22);
        uint32_t mj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
        uint32_t T2 = S0 + mj;
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + T1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = T1 + T2;
    }
    for (int k = 0; k < 8; ++k) H[k] += v[k];
}
void h2ro_sha256(const uint8_t *msg, uint64_t len, uint8_t digest[32]) {
    uint32_t H[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint64_t full = len / 64;
    for (uint64_t b = 0; b < full; ++b) sha_block(H, msg + 64 * b);
    uint8_t tail[128];
    uint64_t rem = len - 64 * full;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    uint64_t tl = rem + 9 <= 64 ? 64 : 128, bits = len * 8;
    for (int k = 0; k < 8; ++k) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha_block(H, tail);
    if (tl == 128) sha_block(H, tail + 64);
    for (int k = 0; k < 8; ++k) { digest[4 * k] = (uint8_t)(H[k] >> 24); digest[4 * k + 1] = (uint8_t)(H[k] >> 16); digest[4 * k + 2] = (uint8_t)(H[k] >> 8); digest[4 * k + 3] = (uint8_t)H[k]; }
}

/* src/lib.rs:210-239: hashed_bytes.reverse(); for each of the 32 / 8 limbs: limb_val = assign_constant(0), then for j in 0..8
 * limb_val = mul_add(assign_constant(2^(8j)), hashed_bytes[8i + j], limb_val).  Stream (288 bytes): the 32 reversed byte
 * cells, one byte each, then every limb_val the mul_add chain assigns, 8 bytes little-endian each; constants are not streamed. */
uint64_t h2ro_hashed_msg_stream_bytes(void) { return 32 + 32 * 8; }
void h2ro_hashed_msg(const uint8_t digest[32], uint64_t hashed[4], uint8_t *stream) {
    uint8_t hb[32];
    for (int k = 0; k < 32; ++k) hb[k] = digest[31 - k];                   /* :213 */
    if (stream) memcpy(stream, hb, 32);
    for (int i = 0; i < 4; ++i) {                                           /* :225 */
        uint64_t limb_val = 0;                                              /* :226 */
        for (int j = 0; j < 8; ++j) {
            uint64_t coeff = 1ull << (8 * j);                               /* :228-229 */
            limb_val = coeff * hb[8 * i + j] + limb_val;                    /* :230-235 */
            if (stream) for (int k = 0; k < 8; ++k) stream[32 + 8 * (8 * i + j) + k] = (uint8_t)(limb_val >> (8 * k));
        }
        hashed[i] = limb_val;                                               /* :237 */
    }
}
