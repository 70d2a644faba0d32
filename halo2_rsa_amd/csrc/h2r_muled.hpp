// Muled integers off the RSA path, for ANY operand shape (the reference takes d0 != d1 in mul, any
// RefreshAux::new(w, n_l, n_r) in refresh and n_l != n_r in is_equal_muled: big_integer/chip.rs:168-233, 386-399, 822-895):
//   refresh_kernel      BigIntChip::refresh.  The reference walks the limbs in order, cutting each into limb_width-bit chunks and
//                       adding chunk j into limb i+j BEFORE that limb is cut -- a second-order carry recurrence
//                           v[i] = a[i] + chunk1(v[i-1]) + chunk2(v[i-2]).
//                       It is solved in PARALLEL, one thread per limb, by fixed-point iteration from v = a: a sweep recomputes every
//                       v[i] from its neighbours' current values; a perturbation of v[i-1] changes chunk1 only when the low part of
//                       v[i] overflows, so the sweeps stop after 2-3 rounds (limb i is exact after i sweeps at the latest, and a fixed
//                       point IS the sequential solution, by induction on i).  Every streamed value is then a function of v[i-2..i].
//   is_equal_muled_kernel  BigIntChip::is_equal_muled for n_l + n_r - 1 columns with word_max = f(min(n_l, n_r)): the carry chain
//                       carry[i+1] = (a[i] - b[i] + W + carry[i]) >> w and the input-independent accumulated_extra chain are
//                       first-order instances of the same fixed-point sweep; the running AND is a block-wide first-failure index.
// Both write the element's FLAT STREAM (the reference's assignment order) into LDS and copy it out as 16-byte lines.
#pragma once

#include "h2r_kernels.hpp"

namespace h2r {

constexpr int MULED_MAX = 2 * 128 + 4;   // limbs / columns per element

__device__ __forceinline__ U192 u192_shr_w(const U192 &v, u32 w) { return v.shr(w); }
__device__ __forceinline__ u64 u192_low_w(const U192 &v, u32 w) { return w == 64 ? v.w[0] : (v.w[0] & 0xffffffffull); }
__device__ __forceinline__ bool u192_nz(const U192 &v) { return (v.w[0] | v.w[1] | v.w[2]) != 0; }

// little-endian value of nbytes (4, 8, 16, 24 or 32) at a 4-byte aligned LDS address
__device__ __forceinline__ void lds_put(u8 *o, const U192 &v, u32 nbytes) {
    u32 *d = reinterpret_cast<u32 *>(o);
    d[0] = (u32)v.w[0];
    if (nbytes >= 8) d[1] = (u32)(v.w[0] >> 32);
    if (nbytes >= 16) { d[2] = (u32)v.w[1]; d[3] = (u32)(v.w[1] >> 32); }
    if (nbytes >= 24) { d[4] = (u32)v.w[2]; d[5] = (u32)(v.w[2] >> 32); }
}
// byte-granular variant (the is_equal_muled stream packs flag bytes between the values)
__device__ __forceinline__ void lds_put_bytes(u8 *o, const U192 &v, u32 nbytes) {
    for (u32 k = 0; k < nbytes; ++k) o[k] = (u8)(v.w[k >> 3] >> (8 * (k & 7)));
}

struct RefreshArgs {
    const u64 *muled; u64 muled_stride;   // [elem][muled_stride] x 4 u64; columns 0 .. d-1 are read
    u64 batch; u32 d, nf, w;
    u8 inc[MULED_MAX];                    // RefreshAux::increased_limbs_vec (mod.rs:428-482)
    u32 off[MULED_MAX];                   // stream offset of limb i's div_mod section
    u32 range_off;                        // stream offset of the range-assign section (nf x (LB + 8))
    u8 *trace; u64 elem_stride;
    void *fresh_out;                      // [elem][nf] limbs (nullable)
    u8 *status;
    u32 LB, WB, CB, stream_bytes;
};

__global__ __launch_bounds__(256) void refresh_kernel(RefreshArgs a) {
    __shared__ u64 v0[MULED_MAX], v1[MULED_MAX]; __shared__ u32 v2[MULED_MAX];
    extern __shared__ uint4 refresh_stage[];
    const u32 tid = threadIdx.x, w = a.w, nf = a.nf;
    const u64 elem = blockIdx.x;
    const u64 *m = a.muled + elem * a.muled_stride * 4;
    // thread t owns limbs t and t + 256 (nf <= 258)
    U192 av[2], cur[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 i = tid + 256 * k;
        av[k] = (i < a.d) ? U192::make(m[4ull * i], m[4ull * i + 1], m[4ull * i + 2]) : U192::make(0, 0, 0);
        cur[k] = av[k];
        if (i < (u32)MULED_MAX) { v0[i] = cur[k].w[0]; v1[i] = cur[k].w[1]; v2[i] = (u32)cur[k].w[2]; }
    }
    __syncthreads();
    auto load = [&](u32 i) { return U192::make(v0[i], v1[i], v2[i]); };
    auto chunk = [&](const U192 &v, u32 j) -> u64 {   // chunk j (0, 1, 2) of a value of at most three chunks; the LAST chunk keeps what remains
        U192 t = v;
        for (u32 k = 0; k < j; ++k) t = u192_shr_w(t, w);
        return u192_low_w(t, w);
    };
    // fixed-point sweeps of v[i] = a[i] + chunk1(v[i-1]) [inc[i-1] >= 1] + chunk2(v[i-2]) [inc[i-2] >= 2]
    for (u32 it = 0; it <= nf; ++it) {
        U192 nv[2]; bool changed = false;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32 i = tid + 256 * k;
            nv[k] = av[k];
            if (i < nf) {
                if (i >= 1 && a.inc[i - 1] >= 1) nv[k] = nv[k] + U192::make(chunk(load(i - 1), 1), 0, 0);
                if (i >= 2 && a.inc[i - 2] >= 2) nv[k] = nv[k] + U192::make(chunk(load(i - 2), 2), 0, 0);
                changed = changed || !(nv[k] == cur[k]);
            }
        }
        const int any = __syncthreads_or(changed ? 1 : 0);   // (also orders the reads above before the writes below)
        if (!any) break;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32 i = tid + 256 * k;
            cur[k] = nv[k];
            if (i < nf) { v0[i] = nv[k].w[0]; v1[i] = nv[k].w[1]; v2[i] = (u32)nv[k].w[2]; }
        }
        __syncthreads();
    }
    // emission: limb i's section holds, for j = 0 .. inc[i]: q, n, 2^w * q, limb - 2^w * q  [, refreshed[i+j] after adding n  (j >= 1)]
    u8 *stage = reinterpret_cast<u8 *>(refresh_stage);
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 i = tid + 256 * k;
        if (i >= nf) continue;
        u8 *o = stage + a.off[i];
        U192 limb = cur[k];                                        // refreshed_limbs[i] when its turn comes (:197)
        const u32 reps = (u32)a.inc[i] + 1;
        for (u32 j = 0; j < reps; ++j) {                           // :198
            const U192 q = u192_shr_w(limb, w);                    // div_mod_main_gate(limb, 2^w)  (:201 -> :1323-1349)
            const u64 n = u192_low_w(limb, w);
            lds_put(o, q, a.CB); o += a.CB;
            lds_put(o, U192::make(n, 0, 0), a.LB); o += a.LB;
            lds_put(o, w == 64 ? U192::make(0, q.w[0], q.w[1]) : q.shl(32), a.WB); o += a.WB;   // nq = 2^w * q
            lds_put(o, U192::make(n, 0, 0), a.LB); o += a.LB;       // a - nq
            if (j >= 1) {                                          // refreshed_limbs[i + j] += n  (:207): what that limb holds right then
                const u32 tg = i + j;
                U192 t = (tg < a.d) ? U192::make(m[4ull * tg], m[4ull * tg + 1], m[4ull * tg + 2]) : U192::make(0, 0, 0);
                if (j == 1) {   // limb i-1's chunk 2 arrived earlier, limb i's chunk 1 now: the complete v[i+1]
                    if (i >= 1 && a.inc[i - 1] >= 2) t = t + U192::make(chunk(load(i - 1), 2), 0, 0);
                }
                t = t + U192::make(n, 0, 0);
                lds_put(o, t, a.WB); o += a.WB;
            }
            limb = q;
        }
        if (u192_nz(limb)) bad = true;                             // assert_zero(limb), :213
        // range-assign the refreshed limb (:217-226): value + its eight sub-limb bytes
        const u64 fv = u192_low_w(cur[k], w);
        u8 *r = stage + a.range_off + (u64)i * (a.LB + 8);
        lds_put(r, U192::make(fv, 0, 0), a.LB);
        const u64 sb = w == 64 ? limb_sub_bytes<64>(fv) : limb_sub_bytes<32>(fv);
        lds_put(r + a.LB, U192::make(sb, 0, 0), 8);
        if (a.fresh_out) {
            if (w == 64) reinterpret_cast<u64 *>(a.fresh_out)[elem * nf + i] = fv;
            else reinterpret_cast<u32 *>(a.fresh_out)[elem * nf + i] = (u32)fv;
        }
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (tid == 0) a.status[elem] = (u8)(any_bad ? H2R_E_NOT_REDUCED : H2R_OK);
    u8 *dst = a.trace + elem * a.elem_stride;
    for (u32 k = tid; k < (a.stream_bytes + 15) / 16; k += 256) { const uint4 x = refresh_stage[k]; pst16(dst + 16ull * k, ((u64)x.y << 32) | x.x, ((u64)x.w << 32) | x.z); }
}

struct EqMuledArgs {
    const u64 *ma, *mb; u64 muled_stride;   // [elem][muled_stride] x 4 u64
    u64 batch; u32 C, w;                     // C = n_l + n_r - 1 columns
    u64 wm[3];                               // word_max = compute_mul_word_max(w, min(n_l, n_r))  (chip.rs:838)
    u32 carry_bits, carry_sub_bits, carry_nsub;
    u32 LB, WB, CB, AB;                      // AB: bytes of a_b in the stream (WB two's complement, or 32 = field element)
    u64 p[4];                                // field modulus (AB == 32)
    u32 per_col, stream_bytes;               // bytes of a column's step (with its range assign), of the element's stream
    u8 *trace; u64 elem_stride;
    u8 *eq_out;                              // [elem] (nullable)
};

__global__ __launch_bounds__(256) void is_equal_muled_kernel(EqMuledArgs a) {
    __shared__ u64 c0[MULED_MAX], c1[MULED_MAX];   // carry[i] (carry into column i), < 2^(carry_bits)
    __shared__ u64 x0[MULED_MAX], x1[MULED_MAX];   // accumulated_extra before column i
    __shared__ u32 first_bad;                       // first column whose flags are not all 1
    extern __shared__ uint4 eq_stage[];
    const u32 tid = threadIdx.x, w = a.w, C = a.C;
    const u64 elem = blockIdx.x;
    const U192 W = U192::make(a.wm[0], a.wm[1], a.wm[2]);
    const bool act = tid < C;                       // one thread per column (C <= 255)
    U192 ab = U192::make(0, 0, 0), base = W;
    if (act) {
        const u64 *pa = a.ma + (elem * a.muled_stride + tid) * 4, *pb = a.mb + (elem * a.muled_stride + tid) * 4;
        ab = U192::make(pa[0], pa[1], pa[2]) - U192::make(pb[0], pb[1], pb[2]);   // a_b, two's complement (:859)
        base = ab + W;                                                              // a_b + word_max >= 0 for limbs <= word_max
    }
    if (tid == 0) first_bad = 0xffffffffu;
    for (u32 i = tid; i <= C; i += 256) { c0[i] = 0; c1[i] = 0; x0[i] = 0; x1[i] = 0; }
    __syncthreads();
    // carry[i+1] = (base[i] + carry[i]) >> w  and  X[i+1] = (W + X[i]) >> w: fixed-point sweeps, both chains at once
    U192 mc = U192::make(0, 0, 0), mx = U192::make(0, 0, 0);   // this column's outgoing carry / accumulated_extra
    for (u32 it = 0; it <= C; ++it) {
        bool changed = false;
        U192 nc = mc, nx = mx;
        if (act) {
            nc = u192_shr_w(base + U192::make(c0[tid], c1[tid], 0), w);
            nx = u192_shr_w(W + U192::make(x0[tid], x1[tid], 0), w);
            changed = !(nc == mc) || !(nx == mx);
        }
        const int any = __syncthreads_or(changed ? 1 : 0);
        if (!any) break;
        mc = nc; mx = nx;
        if (act) { c0[tid + 1] = nc.w[0]; c1[tid + 1] = nc.w[1]; x0[tid + 1] = nx.w[0]; x1[tid + 1] = nx.w[1]; }
        __syncthreads();
    }
    // this column's step (chip.rs:857-893)
    const U192 cin = U192::make(c0[act ? tid : 0], c1[act ? tid : 0], 0), xin = U192::make(x0[act ? tid : 0], x1[act ? tid : 0], 0);
    const U192 sum = base + cin;                       // a_b + carry[i] + word_max  (:860-861)
    const U192 ncar = u192_shr_w(sum, w);              // div_mod_main_gate(sum, 2^w) (:864)
    const u64 cm = u192_low_w(sum, w);
    const U192 accx = xin + W;                         // :869-870
    const U192 qacc = u192_shr_w(accx, w);             // :871
    const u64 modacc = u192_low_w(accx, w);
    const bool last = tid == C - 1;
    const u32 f1 = cm == modacc ? 1u : 0u;             // cs_acc_eq (:873)
    const u32 f2 = last ? ((ncar == qacc) ? 1u : 0u) : 1u;   // final_carry_eq (:890) | range_eq (:886: the range-assigned copy equals the carry)
    if (act && !(f1 && f2)) atomicMin(&first_bad, tid);
    __syncthreads();
    const u32 fb = first_bad;
    if (act) {
        u8 *o = reinterpret_cast<u8 *>(eq_stage) + (u64)tid * a.per_col;
        if (a.AB == 32) {   // canonical element of the field: negative -> p - |a_b|
            u64 x[4] = {ab.w[0], ab.w[1], ab.w[2], 0};
            if (ab.w[2] >> 63) {
                x[3] = ~0ull;
                u64 cy = 0;
                for (int k = 0; k < 4; ++k) { const u64 s1 = x[k] + a.p[k]; const u64 k1 = s1 < x[k]; const u64 s2 = s1 + cy; cy = k1 | (u64)(s2 < s1); x[k] = s2; }
            }
            for (u32 k = 0; k < 32; ++k) o[k] = (u8)(x[k >> 3] >> (8 * (k & 7)));
        } else lds_put_bytes(o, ab, a.AB);
        o += a.AB;
        lds_put_bytes(o, sum, a.WB); o += a.WB;
        lds_put_bytes(o, ncar, a.CB); o += a.CB;
        lds_put_bytes(o, U192::make(cm, 0, 0), a.LB); o += a.LB;
        lds_put_bytes(o, w == 64 ? U192::make(0, ncar.w[0], ncar.w[1]) : ncar.shl(32), a.WB); o += a.WB;   // nq
        lds_put_bytes(o, U192::make(cm, 0, 0), a.LB); o += a.LB;                                              // sum - nq
        lds_put_bytes(o, accx, a.WB); o += a.WB;
        lds_put_bytes(o, qacc, a.CB); o += a.CB;
        lds_put_bytes(o, U192::make(modacc, 0, 0), a.LB); o += a.LB;
        lds_put_bytes(o, w == 64 ? U192::make(0, qacc.w[0], qacc.w[1]) : qacc.shl(32), a.WB); o += a.WB;
        lds_put_bytes(o, U192::make(modacc, 0, 0), a.LB); o += a.LB;
        const u32 e1 = (tid < fb || (tid == fb && f1)) ? 1u : 0u;   // eq_bit after AND cs_acc_eq (:874): all earlier flags and f1
        const u32 e2 = tid < fb ? 1u : 0u;                          // ... after AND range_eq / final_carry_eq (:887 / :891)
        o[0] = (u8)f1; o[1] = (u8)e1; o += 2;
        if (!last) {   // RangeChip::assign(carry, sublimb_bit_len(carry_bits), carry_bits): value + its sub-limbs (:879-885)
            lds_put_bytes(o, ncar, a.CB); o += a.CB;
            const u32 msk = (1u << a.carry_sub_bits) - 1;
            for (u32 k = 0; k < a.carry_nsub; ++k) {
                const u32 sh = k * a.carry_sub_bits;
                const u64 lo = sh < 64 ? (ncar.w[0] >> sh) | (sh ? ncar.w[1] << (64 - sh) : 0) : (ncar.w[1] >> (sh - 64));
                o[k] = (u8)((u32)lo & msk);
            }
            o += a.carry_nsub;
        }
        o[0] = (u8)f2; o[1] = (u8)e2;
        if (last && a.eq_out) a.eq_out[elem] = (u8)e2;
    }
    __syncthreads();
    u8 *dst = a.trace + elem * a.elem_stride;
    for (u32 k = tid; k < (a.stream_bytes + 15) / 16; k += 256) { const uint4 x = eq_stage[k]; pst16(dst + 16ull * k, ((u64)x.y << 32) | x.x, ((u64)x.w << 32) | x.z); }
}

// zero-padded copy of [batch][d] limbs into [batch][L] (mul with operands shorter than the ctx's num_limbs)
struct PadArgs { const u8 *src; u8 *dst; u64 batch; u32 d, L, LB; };
__global__ __launch_bounds__(256) void pad_limbs_kernel(PadArgs a) {
    const u64 n = a.batch * a.L;
    for (u64 idx = (u64)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (u64)gridDim.x * 256) {
        const u64 e = idx / a.L; const u32 k = (u32)(idx - e * a.L);
        if (a.LB == 8) reinterpret_cast<u64 *>(a.dst)[idx] = k < a.d ? reinterpret_cast<const u64 *>(a.src)[e * a.d + k] : 0;
        else reinterpret_cast<u32 *>(a.dst)[idx] = k < a.d ? reinterpret_cast<const u32 *>(a.src)[e * a.d + k] : 0;
    }
}

}  // namespace h2r
