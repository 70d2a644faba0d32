// One wavefront per element: the dependent mul_mod chain for integers of K <= 32 digits (RSA-1024 and below), register-resident.
//
// 2K <= 64 product columns = exactly the lanes of one wavefront, so a whole Barrett mul_mod (three K x K products, the quotient
// estimate, the remainder and its corrections) runs inside ONE wave: no workgroup barrier, no partial sums through LDS, no operand
// reads from LDS.  Column c of A * B lives in lane c; step j adds A[j] * B[c - j] to it: A[j] is a scalar (v_readlane), B is held
// one digit per lane and moves up one lane per step (DPP wave_shr:1, zero entering lane 0).  The 96-bit column sums become digits
// with three DPP shifts and one ballot carry resolution.  Numbers of 2K digits sit in lanes 0..2K-1; Barrett's x1 = floor(x / B^K)
// and the quotient live in lanes K..2K-1 and are read as the scalar operand straight from there (v_readlane with base K).
//
// The four-wave kernel (h2r_kernels.hpp) spends a K = 32 mul_mod mostly on what this form does not have: six workgroup barriers, the
// partial-sum round trips and wave 0's serial reduce (3.8 us per dependent RSA-1024 mul_mod at 1,024 elements per call,
// profiles/r05_chain_accounting.txt).  Workgroups of this kernel are FOUR INDEPENDENT waves (four elements): the step launch keeps
// its 256-thread workgroups for the record role.
#pragma once

namespace h2r {

__device__ __forceinline__ u32 wv_shr1(u32 v) {   // lane c <- lane c - 1; lane 0 <- 0 (bound_ctrl: a lane without a source reads zero)
    return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
template <int K> __device__ __forceinline__ constexpr u64 wv_mask() { return K >= 64 ? ~0ull : ((1ull << K) - 1); }

// lane c (c < 2K) returns digit c of A * B.  A[j] = lane (ABASE + j) of `a` (any lanes: read as scalars); B[j] = lane j of `b`,
// whose lanes >= K must be zero.  Lanes >= 2K return 0.
template <int K, int ABASE>
__device__ __forceinline__ u32 wave_product(u32 a, u32 b, int lane) {
    static_assert(K >= 2 && K <= 32 && ABASE + K <= 64, "one wavefront holds the 2K product columns");
    u64 acc = 0;
    u32 ov = 0, bs = b;
    u32 aj = (u32)__builtin_amdgcn_readlane((int)a, ABASE);
    // Step j: acc(64) += A[j] * bs with the multiplier's carry-out feeding the third word, then A[j + 1] and bs one lane up for the next step.
    // ONE asm block per step: the two independent instructions sit between the multiply and the v_addc that reads its carry (gfx950 wants two
    // wait states there), and between the v_readlane and the next multiply that reads its SGPR -- no s_nop, four VALU instructions per step.
#pragma unroll
    for (int j = 0; j < K; ++j) {
        u64 carry;
        u32 an, bn;
        if (j + 1 < K) {
            asm volatile("v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
                         "v_readlane_b32 %3, %7, %8\n\t"
                         "v_mov_b32_dpp %4, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_addc_co_u32_e64 %1, %2, 0, %1, %2"
                         : "+v"(acc), "+v"(ov), "=&s"(carry), "=&s"(an), "=&v"(bn) : "s"(aj), "v"(bs), "v"(a), "n"(ABASE + j + 1));
            aj = an; bs = bn;
        } else {
            asm volatile("v_mad_u64_u32 %0, %2, %3, %4, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
                         : "+v"(acc), "+v"(ov), "=&s"(carry) : "s"(aj), "v"(bs));
        }
    }
    // x0 | x1 | x2 (c) = the 96-bit sum of column c;  t(c) = x0(c) + x1(c-1) + x2(c-2);  digit(c) = lo32 t(c) + hi32 t(c-1) + carry
    const u32 x0 = (u32)acc, x1 = (u32)(acc >> 32), x2 = ov;
    const u64 t = (u64)x0 + wv_shr1(x1) + wv_shr1(wv_shr1(x2));
    const u64 d = (u64)(u32)t + wv_shr1((u32)(t >> 32));
    const CarryGroup cg = carry_group(__ballot((d >> 32) != 0), __ballot((u32)d == 0xffffffffu), false, 64);
    return (u32)d + (u32)((cg.cin_mask >> lane) & 1);
}

// (2K-digit number, digit c in lane c) << sh, sh < 32 K.  Returns true when non-zero bits leave the 2K digits.
template <int K>
__device__ __forceinline__ bool wave_shl2k(u32 &x, u32 sh, int lane) {
    const int ws = (int)(sh >> 5), bs = (int)(sh & 31);
    const bool in = lane < 2 * K;
    const bool lost = in && ((lane + ws >= 2 * K && x != 0) || (lane + ws == 2 * K - 1 && bs && (x >> (32 - bs)) != 0));
    const u32 a0 = (u32)__shfl((int)x, lane - ws), a1 = (u32)__shfl((int)x, lane - ws - 1);
    const u32 v0 = lane - ws >= 0 ? a0 : 0u, v1 = lane - ws - 1 >= 0 ? a1 : 0u;
    x = in ? (bs ? ((v0 << bs) | (v1 >> (32 - bs))) : v0) : 0u;
    return __ballot(lost) != 0;
}
// (K-digit number in lanes 0..K-1, zero above) >> sh
template <int K>
__device__ __forceinline__ u32 wave_shr(u32 x, u32 sh, int lane) {
    const int ws = (int)(sh >> 5), bs = (int)(sh & 31);
    const u32 a0 = (u32)__shfl((int)x, (lane + ws) & 63), a1 = (u32)__shfl((int)x, (lane + ws + 1) & 63);
    const u32 v0 = lane + ws < K ? a0 : 0u, v1 = lane + ws + 1 < K ? a1 : 0u;
    return lane < K ? (bs ? ((v0 >> bs) | (v1 << (32 - bs))) : v0) : 0u;
}

// One BigIntChip::mul_mod's off-circuit arithmetic (big_integer/chip.rs:562-584): (q, r) = divmod(a * b, n) by Barrett reduction with the
// per-modulus n' = n << shift and mu' (chain_modulus_setup's definitions).  a, b, nn, mu: K digits in lanes 0..K-1, zero above.
// q, r: lanes 0..K-1.  The same arithmetic, estimate and correction loop as block_mulmod: identical q, r and statuses.
#ifndef H2R_WAVE_MULMOD_NOINLINE
#define H2R_WAVE_MULMOD_INLINE __forceinline__
#else   // (developer variant: ONE copy of the three unrolled products per kernel instead of one per call site -- four in chain_element)
#define H2R_WAVE_MULMOD_INLINE __attribute__((noinline))
#endif
template <int K>
__device__ H2R_WAVE_MULMOD_INLINE int wave_mulmod(u32 shift, int lane, u32 a, u32 b, u32 nn, u32 mu, u32 &q, u32 &r) {
    int status = H2R_OK;
    const bool lo = lane < K, hi = lane >= K && lane < 2 * K;
    u32 x = wave_product<K, 0>(a, b, lane);                                   // x = a * b
    if (shift) { if (wave_shl2k<K>(x, shift, lane)) status = H2R_E_NOT_REDUCED; }   // x' = x << s (wave-uniform branch)
    // q^ = x1 + floor(x1 * mu' / B^K),  x1 = floor(x' / B^K): lanes K..2K-1 of x
    const u32 y = wave_product<K, K>(x, mu, lane);
    u32 qh;
    {
        const u64 d = hi ? (u64)x + y : 0;
        const CarryGroup cg = carry_group(__ballot((d >> 32) != 0) >> K, __ballot(hi && (u32)d == 0xffffffffu) >> K, false, K);
        qh = hi ? (u32)d + (u32)((cg.cin_mask >> (lane - K)) & 1) : 0u;
        if (cg.cout) status = H2R_E_NOT_REDUCED;
    }
    // R = x' - q^ * n'   (0 <= R < 7 n': q^ may be up to 6 short of the true quotient); only digits 0..K of the product matter
    const u32 z = wave_product<K, K>(qh, nn, lane);
    const u32 dk = (u32)__builtin_amdgcn_readlane((int)z, K);
    u32 rl;
    bool b0;
    {
        const CarryGroup cg = carry_group(__ballot(lo && x < z), __ballot(lo && x == z), false, K);
        rl = lo ? x - z - (u32)((cg.cin_mask >> lane) & 1) : 0u;
        b0 = cg.cout;
    }
    u32 rtop = (u32)__builtin_amdgcn_readlane((int)x, K) - dk - (b0 ? 1u : 0u);
    for (int it = 0; it < 8; ++it) {
        bool ge = true;   // rl >= nn over lanes 0..K-1
        {
            const u64 ne = __ballot(lo && rl != nn);
            if (ne) ge = ((__ballot(lo && rl > nn) >> (63 - __builtin_clzll(ne))) & 1) != 0;
        }
        if (rtop == 0 && !ge) break;
        const CarryGroup cs = carry_group(__ballot(lo && rl < nn), __ballot(lo && rl == nn), false, K);
        rl = lo ? rl - nn - (u32)((cs.cin_mask >> lane) & 1) : 0u;
        rtop -= cs.cout ? 1u : 0u;
        const CarryGroup ci = carry_group(0, __ballot(hi && qh == 0xffffffffu) >> K, true, K);   // q^ += 1
        if (hi) qh += (u32)((ci.cin_mask >> (lane - K)) & 1);
        if (ci.cout) status = H2R_E_NOT_REDUCED;
    }
    r = shift ? wave_shr<K>(rl, shift, lane) : rl;                            // r = R >> s
    const u32 ql = (u32)__shfl((int)qh, (lane + K) & 63);
    q = lo ? ql : 0u;
    return status;
}

// mu' = floor((B^(2K) - 1) / n') - B^K for the normalised modulus n' (lanes 0..K-1), by NEWTON'S ITERATION on the wave's own products instead of
// Knuth's digit-by-digit division (wave_reciprocal: K steps of ~0.45 us each, every one a chain of VALU -> ballot -> SALU -> VALU round trips that a
// lone wave cannot hide: 16 of the 56 us of an RSA-1024 e = 65537 chain).  y = B^K + m approximates R = floor(T / n'), T = B^(2K) - 1, FROM BELOW:
//     e = T - n' y  (= the bitwise complement of n' y: T is all ones),   y <- y + floor(y e / B^(2K))   (the m e_lo / B^K term dropped: still from below)
// never overshoots (y n' <= B^(2K)) and squares the relative error: from the 31 good bits of the top digit's reciprocal, 4 / 5 / 6 steps of two
// products each reach K = 8 / 16 / 32 digits to within one unit (checked against big integers over adversarial and random moduli); the exact
// floor then follows from one more residual: while e >= n': y += 1, e -= n'.
template <int K>
__device__ __forceinline__ u32 wave_reciprocal_newton(u32 nn, int lane) {
    constexpr int ITERS = K <= 8 ? 4 : (K <= 16 ? 5 : 6);
    const bool lo = lane < K, hi = lane >= K && lane < 2 * K;
    const u32 ntop = (u32)__builtin_amdgcn_readlane((int)nn, K - 1);
    const u64 c = ntop == 0xffffffffu ? 0xffffffffull : 0xffffffffffffffffull / ((u64)ntop + 1);   // < 2^33 (n' normalised: ntop >= 2^31)
    u32 m = (lane == K - 1 && c > 0xffffffffull) ? (u32)(c - 0x100000000ull) : 0u;              // y0 = max(B^K, c B^(K-1)) <= R
    const u32 nsh_ = (u32)__shfl((int)nn, (lane - K) & 63), nsh = hi ? nsh_ : 0u;                 // n' B^K
    auto residual = [&](u32 mm) -> u32 {   // T - n' (B^K + mm), digit c in lane c (n' y <= T: the sum below has no carry out of digit 2K - 1)
        const u32 p = wave_product<K, 0>(nn, mm, lane);
        const u64 d = (u64)p + nsh;
        const CarryGroup cg = carry_group(__ballot((d >> 32) != 0), __ballot((u32)d == 0xffffffffu), false, 64);
        return lane < 2 * K ? ~((u32)d + (u32)((cg.cin_mask >> lane) & 1)) : 0u;
    };
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        const u32 e = residual(m);
        const u32 q = wave_product<K, K>(e, m, lane);                       // e_hi * m
        const u64 d0 = lo ? (u64)q + e : 0;                                  // the carry of (q_lo + e_lo) into digit K
        const CarryGroup c0 = carry_group(__ballot((d0 >> 32) != 0), __ballot(lo && (u32)d0 == 0xffffffffu), false, K);
        const u64 d1 = hi ? (u64)q + e : 0;                                  // inc = e_hi + q_hi + that carry (lanes K..2K-1)
        const CarryGroup c1 = carry_group(__ballot((d1 >> 32) != 0) >> K, __ballot(hi && (u32)d1 == 0xffffffffu) >> K, c0.cout, K);
        const u32 inc = hi ? (u32)d1 + (u32)((c1.cin_mask >> (lane - K)) & 1) : 0u;
        const u32 incl = (u32)__shfl((int)inc, (lane + K) & 63);
        const u64 d2 = lo ? (u64)m + incl : 0;                               // m += inc  (y + inc <= R < 2 B^K: no carry out)
        const CarryGroup c2 = carry_group(__ballot((d2 >> 32) != 0), __ballot(lo && (u32)d2 == 0xffffffffu), false, K);
        m = lo ? (u32)d2 + (u32)((c2.cin_mask >> lane) & 1) : 0u;
    }
    u32 e = residual(m);
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {                                         // the exact floor: at most one or two steps
        bool ge = __ballot(hi && e != 0) != 0;
        if (!ge) {
            ge = true;
            const u64 ne = __ballot(lo && e != nn);
            if (ne) ge = ((__ballot(lo && e > nn) >> (63 - __builtin_clzll(ne))) & 1) != 0;
        }
        if (!ge) break;
        const u32 sub = lo ? nn : 0u;
        const CarryGroup cs = carry_group(__ballot(lane < 2 * K && e < sub), __ballot(lane < 2 * K && e == sub), false, 64);
        e = lane < 2 * K ? e - sub - (u32)((cs.cin_mask >> lane) & 1) : 0u;
        const CarryGroup ci = carry_group(0, __ballot(lo && m == 0xffffffffu), true, K);
        if (lo) m += (u32)((ci.cin_mask >> lane) & 1);
    }
    return m;
}

// The per-modulus constants, wave-local: returns H2R_E_ZERO_MODULUS for n = 0 (reference divides by zero, chip.rs:566).
template <int K>
__device__ __forceinline__ int wave_modulus_setup(ChainLds<K, 1> &s, u32 nraw, int lane, u32 &shift, u32 &nn, u32 &mu) {
    shift = 0; nn = nraw; mu = 0;
    const u64 nz = __ballot(lane < K && nraw != 0);
    if (!nz) return H2R_E_ZERO_MODULUS;
    const int top_digit = 63 - __builtin_clzll(nz);
    const u32 topv = (u32)__builtin_amdgcn_readlane((int)nraw, top_digit);
    shift = 32u * (u32)(K - 1 - top_digit) + (u32)__builtin_clz(topv);
    if (shift) (void)wave_shl2k<K>(nn, shift, lane);   // (n < B^K: nothing leaves, the lanes above K-1 stay zero)
#ifndef H2R_WAVE_NEWTON   // shipped: the digit-by-digit reciprocal the four-wave form uses (wave-local, on registers)
    u32 nn1[1] = {nn}, mu1[1] = {0};
    wave_reciprocal<K, 1>(s, nn1, lane, 0, mu1);
    mu = lane < K ? mu1[0] : 0u;
#else
    // (developer variant -DH2R_WAVE_NEWTON: exact -- digit for digit Knuth's on 157 parity tests and 15,000 adversarial moduli with -DH2R_WAVE_RECIP_CHECK --
    //  and 4.4 us shorter per element alone (15.2 -> 10.8 us), but the step launch does not wait for it: 14.1 M either way at 1,024 per call, 15.6 against
    //  15.8 M at 2,048, where its twelve extra products cost issue slots: profiles/r06_chain_wave.txt)
    (void)s;
    mu = wave_reciprocal_newton<K>(nn, lane);
#ifdef H2R_WAVE_RECIP_CHECK
    {
        u32 nn1[1] = {nn}, mu1[1] = {0};
        wave_reciprocal<K, 1>(s, nn1, lane, 0, mu1);
        if (__ballot(lane < K && mu1[0] != mu) != 0) return H2R_E_INTERNAL;
    }
#endif
#endif
    return H2R_OK;
}

}  // namespace h2r
